// Direct (im2col-free) convolution on CDNA4 matrix cores.
//
// Replaces the cuDNN calls behind nn.Conv2d + BatchNorm2d(eval) + (Leaky)ReLU / residual add /
// sigmoid range casts of the reference networks:
//   conv3x3 / conv1x1 / BasicBlock      tpm/lib/network/rtpose_light3d.py:24-72
//   make_stages Conv2d+BN+LeakyReLU     tpm/lib/network/rtpose_light3d.py:222-246
//   forward() sigmoid casts             tpm/lib/network/rtpose_light3d.py:335-337,348-350
//   YoloPoseNet neck/head + slice casts tpm/lib/network/yolo_posenet.py:101-126,146-156
//   resnet.BasicBlock (stride-2 + 1x1)  tpm/lib/network/resnet.py:27-56,134-148
//
// Formulation: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (cin-chunk, tap, cin).
//   * MFMA "A" operand = weights, streamed global -> VGPR.  They are pre-packed on the host in
//     exactly the per-lane fragment order, so one wave-load is 1 KiB (bf16) of contiguous memory.
//   * MFMA "B" operand = activations.  A block owns R full output rows of one image; the input
//     halo tile ((R-1)*stride+KS rows x Wo*stride+KS-1 cols x 64 channels) is staged once per
//     64-channel chunk into LDS and re-read by all KS*KS taps and all cout tiles.
//   * LDS image: [halo pixel][64 ch], 128 B (bf16) / 256 B (f32) per pixel, row pitch a multiple
//     of 8 pixels, 16-B (bf16) / 32-B (f32) slots XOR-swizzled with (pixel & 7) so that the 16
//     lanes ds_read_b128 services together hit 16 different slots of the 256-B bank row.
//   * bf16: v_mfma_f32_16x16x32_bf16, fp32 accumulate.  f32 ("parity" mode):
//     v_mfma_f32_16x16x4_f32, bit-exact fp32 FMA chains (no xf32 on gfx950).
//   * C/D layout (col = lane&15 -> pixel, row = 4*(lane>>4)+reg -> cout): each lane ends up
//     with 4 consecutive output channels of one pixel = one 8-B / 16-B NHWC store.
//   * Epilogue fuses folded-BN bias, residual add, ReLU / LeakyReLU(0.1) / sigmoid casts, and
//     can emit NHWC (next layer) and/or NCHW f32 (API boundary) in the same pass.
#pragma once
#include "pn_internal.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int PREC> struct Elem;
template <> struct Elem<PN_PREC_BF16> {
    typedef __bf16 T;
    static constexpr int PIXB = 128;   // LDS bytes per halo pixel (64 channels)
    static constexpr int FRAGB = 1024; // bytes of one packed A fragment (16 couts x 32 k)
    static constexpr int SUBX = 64;    // address XOR selecting the second 32-channel half
    struct Frag { bf16x8 v; };
};
template <> struct Elem<PN_PREC_F32> {
    typedef float T;
    static constexpr int PIXB = 256;
    static constexpr int FRAGB = 2048;
    static constexpr int SUBX = 128;
    struct Frag { f32x4 lo, hi; };
};

template <int CFG> struct TileCfg;
template <> struct TileCfg<PN_CFG_C128> { static constexpr int WC = 4, WP = 1, CT = 2, PT = 7; };
template <> struct TileCfg<PN_CFG_C64>  { static constexpr int WC = 2, WP = 2, CT = 2, PT = 4; };
template <> struct TileCfg<PN_CFG_C32>  { static constexpr int WC = 1, WP = 4, CT = 2, PT = 2; };
template <> struct TileCfg<PN_CFG_C16>  { static constexpr int WC = 1, WP = 4, CT = 1, PT = 2; };

__device__ __forceinline__ float pn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float pn_activate(float v, int act, int co, int naf) {
    switch (act) {
        case PN_ACT_RELU: return v > 0.f ? v : 0.f;
        case PN_ACT_LEAKY: return v > 0.f ? v : v * 0.1f;
        case PN_ACT_SIG_PM2: return (pn_sigmoid(v) - 0.5f) * 4.f;
        case PN_ACT_SIG: return pn_sigmoid(v);
        case PN_ACT_YOLO: {
            int f = co % naf;
            float s = pn_sigmoid(v);
            if (f < 2) return (s - 0.5f) * 2.f;
            if (f < 4) return s * 2.f;
            if (f == 4) return s;
            return (s - 0.5f) * 4.f;
        }
        default: return v;
    }
}

template <int PREC> __device__ __forceinline__ typename Elem<PREC>::Frag load_a_frag(const char *p);
template <> __device__ __forceinline__ Elem<PN_PREC_BF16>::Frag load_a_frag<PN_PREC_BF16>(const char *p) {
    Elem<PN_PREC_BF16>::Frag f;
    f.v = *reinterpret_cast<const bf16x8 *>(p);
    return f;
}
template <> __device__ __forceinline__ Elem<PN_PREC_F32>::Frag load_a_frag<PN_PREC_F32>(const char *p) {
    Elem<PN_PREC_F32>::Frag f;   // packed as [half][lane][4 floats]: both halves lane-contiguous
    f.lo = *reinterpret_cast<const f32x4 *>(p);
    f.hi = *reinterpret_cast<const f32x4 *>(p + 1024);
    return f;
}

template <int PREC> __device__ __forceinline__ typename Elem<PREC>::Frag read_b_frag(const char *smem, int addr);
template <> __device__ __forceinline__ Elem<PN_PREC_BF16>::Frag read_b_frag<PN_PREC_BF16>(const char *smem, int addr) {
    Elem<PN_PREC_BF16>::Frag f;
    f.v = *reinterpret_cast<const bf16x8 *>(smem + addr);
    return f;
}
template <> __device__ __forceinline__ Elem<PN_PREC_F32>::Frag read_b_frag<PN_PREC_F32>(const char *smem, int addr) {
    Elem<PN_PREC_F32>::Frag f;
    f.lo = *reinterpret_cast<const f32x4 *>(smem + addr);
    f.hi = *reinterpret_cast<const f32x4 *>(smem + addr + 16);
    return f;
}

__device__ __forceinline__ f32x4 mma(const Elem<PN_PREC_BF16>::Frag &a, const Elem<PN_PREC_BF16>::Frag &b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma(const Elem<PN_PREC_F32>::Frag &a, const Elem<PN_PREC_F32>::Frag &b, f32x4 c) {
    // k-slot s of lane-quarter q is channel 8q+s for BOTH operands (any consistent k order is valid)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[0], b.lo[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[1], b.lo[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[2], b.lo[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[3], b.lo[3], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[0], b.hi[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[1], b.hi[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[2], b.hi[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[3], b.hi[3], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ void store4(__bf16 *p, const float v[4]) {
    bf16x4 o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4 *>(p) = o;
}
__device__ __forceinline__ void store4(float *p, const float v[4]) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4 *>(p) = o;
}

template <int PREC, int KS, int STRIDE, int PITCH, int CFG>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvProblem *__restrict__ probs) {
    typedef Elem<PREC> E;
    typedef typename E::T T;
    typedef typename E::Frag Frag;
    constexpr int PIXB = E::PIXB, FRAGB = E::FRAGB, SUBX = E::SUBX;
    constexpr int WC = TileCfg<CFG>::WC, WP = TileCfg<CFG>::WP, CT = TileCfg<CFG>::CT, PT = TileCfg<CFG>::PT;
    constexpr int KK = KS * KS, PAD = KS / 2;
    constexpr int NCH = PIXB / 16;      // 16-B pieces per halo pixel
    constexpr int PPI = 256 / NCH;      // halo pixels staged per block pass
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvProblem &P = probs[blockIdx.y];
    const int bx = blockIdx.x;
    if (bx >= P.nblocks) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WP, wp = wave % WP;
    const int c = lane & 15, q = lane >> 4;

    const int cb = bx % P.cout_blocks;
    const int tt = bx / P.cout_blocks;
    const int tile = tt % P.tiles_per_img;
    const int b = tt / P.tiles_per_img;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int oy0 = ty * P.R, ox0 = tx * P.Wt;
    const int R = min(P.R, P.Ho - oy0);
    const int Wo = P.Wo;
    const int Wc = min(P.Wt, Wo - ox0);          // output columns of this tile
    const int npix = R * Wc;
    const int HRa = (R - 1) * STRIDE + KS;       // halo rows actually needed
    const int HC = (Wc - 1) * STRIDE + KS;       // halo columns
    const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;
    const float inv_wc = 1.0f / (float)Wc;

    // ---- per-lane LDS addresses of tap (ky=0,kx) for each pixel tile, first 32-channel half ----
    int baddr[PT][KS];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int slot = (wp * PT + pt) * 16 + c;
        int s = slot < npix ? slot : 0;
        int ry = (int)(((float)s + 0.5f) * inv_wc);
        int rx = s - ry * Wc;
        int hp0 = ry * STRIDE * PITCH + rx * STRIDE;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            int hp = hp0 + kx;
            if (PREC == PN_PREC_BF16)
                baddr[pt][kx] = hp * PIXB + ((q ^ (hp & 7)) << 4);
            else
                baddr[pt][kx] = hp * PIXB + ((q ^ (hp & 7)) << 5);
        }
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ctile0 = (cb * WC + wc) * CT;
    const char *wptr[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
        wptr[ct] = (const char *)P.wpack + (size_t)(ctile0 + ct) * P.ksteps * FRAGB + lane * 16;

    Frag a_cur[CT], a_nxt[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) a_nxt[ct] = load_a_frag<PREC>(wptr[ct]);

    // staging geometry (thread -> 16-B piece `ch` of halo pixels p0, p0+PPI, ...)
    const int ch = tid % NCH;
    const int p0 = tid / NCH;
    const int npx = HRa * HC;
    const float inv_hc = 1.0f / (float)HC;
    const char *in_base = (const char *)P.in + ((size_t)P.in_coff * sizeof(T)) + ch * 16;

    for (int chunk = 0; chunk < P.cin_chunks; ++chunk) {
        if (chunk) __syncthreads();
        // ---- stage the halo tile of this 64-channel chunk ----
        for (int p = p0; p < npx; p += PPI) {
            int hy = (int)(((float)p + 0.5f) * inv_hc);
            int hx = p - hy * HC;
            int iy = iy0 + hy, ix = ix0 + hx;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if ((unsigned)iy < (unsigned)P.H && (unsigned)ix < (unsigned)P.W) {
                size_t pix = (size_t)(b * P.H + iy) * P.W + ix;
                v = *reinterpret_cast<const uint4 *>(in_base + (pix * P.in_cs + (size_t)chunk * 64) * sizeof(T));
            }
            int hp = hy * PITCH + hx;
            int dst;
            if (PREC == PN_PREC_BF16)
                dst = hp * PIXB + ((ch ^ (hp & 7)) << 4);
            else
                dst = hp * PIXB + ((((ch >> 1) ^ (hp & 7)) << 5) | ((ch & 1) << 4));
            *reinterpret_cast<uint4 *>(smem + dst) = v;
        }
        __syncthreads();

#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    a_cur[ct] = a_nxt[ct];
                    wptr[ct] += FRAGB;
                    a_nxt[ct] = load_a_frag<PREC>(wptr[ct]);   // wpack has one spare fragment at its end
                }
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    if ((wp * PT + pt) * 16 < npix) {      // wave-uniform
                        Frag bf = read_b_frag<PREC>(smem, (baddr[pt][kx] ^ (sub * SUBX)) + ky * PITCH * PIXB);
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) acc[ct][pt] = mma(a_cur[ct], bf, acc[ct][pt]);
                    }
                }
            }
        }
    }

    // ---- epilogue ----
    const int act = P.act;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int co0 = (ctile0 + ct) * 16 + 4 * q;
        if (co0 >= P.cout) continue;
        const f32x4 bias4 = *reinterpret_cast<const f32x4 *>(P.bias + co0);
        const bool full = (co0 + 3) < P.cout;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            int slot = (wp * PT + pt) * 16 + c;
            if (slot >= npix) continue;
            int ry = (int)(((float)slot + 0.5f) * inv_wc);
            int rx = ox0 + (slot - ry * Wc);
            size_t opix = (size_t)(b * P.Ho + oy0 + ry) * Wo + rx;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[ct][pt][r] + bias4[r];
            if (P.res) {
                const T *rp = (const T *)P.res + opix * P.res_cs + P.res_coff + co0;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (full || co0 + r < P.cout) v[r] += (float)rp[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = pn_activate(v[r], act, co0 + r, P.yolo_naf);
            if (P.out) {
                T *op = (T *)P.out + opix * P.out_cs + P.out_coff + co0;
                if (full) store4(op, v);
                else
                    for (int r = 0; r < 4; ++r)
                        if (co0 + r < P.cout) op[r] = (T)v[r];
            }
            if (P.out_nchw) {
                size_t hw = (size_t)P.Ho * Wo;
                float *np = P.out_nchw + ((size_t)b * P.cout + co0) * hw + (size_t)(oy0 + ry) * Wo + rx;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (full || co0 + r < P.cout) np[r * hw] = v[r];
            }
        }
    }
}


template <int PREC, int KS, int STRIDE, int PITCH, int CFG>
static int conv_launch_one(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    auto kern = conv_mfma_kernel<PREC, KS, STRIDE, PITCH, CFG>;
    if (L.lds_bytes > 160 * 1024)
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "conv halo tile needs %zu B of LDS", L.lds_bytes);
    if (L.lds_bytes > 48 * 1024) {
        static size_t configured = 0;   // per instantiation
        if (configured < L.lds_bytes) {
            PN_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_bytes));
            configured = L.lds_bytes;
        }
    }
    dim3 grid(L.max_blocks, L.nprob), block(256);
    hipLaunchKernelGGL(kern, grid, block, L.lds_bytes, stream, L.probs_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

#define PN_CASE(PREC, KS, ST, PITCH, CFG)                                                       \
    if (L.prec == PREC && L.ks == KS && L.stride == ST && L.pitch == PITCH && L.cfg == CFG)     \
        return conv_launch_one<PREC, KS, ST, PITCH, CFG>(ctx, L, stream);
#define PN_CASES_PREC(KS, ST, PITCH, CFG) \
    PN_CASE(PN_PREC_BF16, KS, ST, PITCH, CFG) PN_CASE(PN_PREC_F32, KS, ST, PITCH, CFG)
#define PN_CASES_ALLCFG(KS, ST, PITCH)                                                 \
    PN_CASES_PREC(KS, ST, PITCH, PN_CFG_C128) PN_CASES_PREC(KS, ST, PITCH, PN_CFG_C64) \
    PN_CASES_PREC(KS, ST, PITCH, PN_CFG_C32) PN_CASES_PREC(KS, ST, PITCH, PN_CFG_C16)

// each conv_inst_*.hip implements one of these for its share of the instantiations; returns
// 1 when the launch description is not one of its cases
int pn_launch_conv_part0(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv_part1(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv_part2(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv_part3(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
