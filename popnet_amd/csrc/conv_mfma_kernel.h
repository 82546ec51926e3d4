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
#include <type_traits>
#include "pn_internal.h"
#ifndef PN_STAMP_AT
#define PN_STAMP_AT(i) do {} while (0)     // scripts/convlab.hip: in-kernel s_memtime timeline
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// Pointers read out of the ConvProblem record are "generic" to the compiler, which then emits flat_*
// memory ops (they tick vmcnt AND lgkmcnt and can only be waited for with 0/0, serialising against
// the LDS pipeline).  Everything that touches HBM goes through explicit global-address-space pointers.
#define PN_GLOBAL __attribute__((address_space(1)))
typedef const PN_GLOBAL char *gcptr;
typedef PN_GLOBAL char *gptr;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native vector: loadable through address-space pointers

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
template <> struct TileCfg<PN_CFG_C64W> { static constexpr int WC = 2, WP = 2, CT = 2, PT = 7; };

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

template <int PREC> __device__ __forceinline__ typename Elem<PREC>::Frag load_a_frag(gcptr p);
template <> __device__ __forceinline__ Elem<PN_PREC_BF16>::Frag load_a_frag<PN_PREC_BF16>(gcptr p) {
    Elem<PN_PREC_BF16>::Frag f;
    f.v = *reinterpret_cast<const PN_GLOBAL bf16x8 *>(p);
    return f;
}
template <> __device__ __forceinline__ Elem<PN_PREC_F32>::Frag load_a_frag<PN_PREC_F32>(gcptr p) {
    Elem<PN_PREC_F32>::Frag f;   // packed as [half][lane][4 floats]: both halves lane-contiguous
    f.lo = *reinterpret_cast<const PN_GLOBAL f32x4 *>(p);
    f.hi = *reinterpret_cast<const PN_GLOBAL f32x4 *>(p + 1024);
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

__device__ __forceinline__ void store4(PN_GLOBAL __bf16 *p, const float v[4]) {
    bf16x4 o;
    o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
    *reinterpret_cast<PN_GLOBAL bf16x4 *>(p) = o;
}
__device__ __forceinline__ void store4(PN_GLOBAL float *p, const float v[4]) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<PN_GLOBAL f32x4 *>(p) = o;
}

// Register-prefetched halo staging: MAXST = 16-B pieces per thread needed to hold the largest halo
// tile of this geometry (pn_conv_stage_maxpx is the host-side mirror used when R is chosen);
// 0 = too many registers, stage with the plain load->store loop instead.
constexpr int pn_stage_maxpx_c(int ks, int stride, int pitch) {
    return stride != 1 ? 0 : (ks == 1 ? 128 : (pitch <= 32 ? 192 : (pitch <= 64 ? 288 : 360)));
}
template <int PREC, int KS, int STRIDE, int PITCH, int CFG> struct StageCfg {
    static constexpr int NCH = Elem<PREC>::PIXB / 16;
    static constexpr int RAW = (pn_stage_maxpx_c(KS, STRIDE, PITCH) * NCH + 255) / 256;
    static constexpr int MAXST = (RAW <= 12 && CFG != PN_CFG_C64W) ? RAW : 0;
};

// Weight fragments are addressed as (wave-uniform 64-bit base in SGPRs) + (32-bit lane offset): the
// compiler then emits the saddr form of global_load and bumps the base with scalar adds -- no vector
// ALU work per k-step.
template <int PREC> __device__ __forceinline__ typename Elem<PREC>::Frag load_a_frag2(gcptr base, unsigned voff) {
    return load_a_frag<PREC>(base + voff);
}
// Same stream through a buffer resource: fragment offset in an SGPR, 32-bit lane offset -- hipcc otherwise widens
// the lane offset to 64 bits and rebuilds a VGPR address pair per load (2 VALU + 6 VGPRs per k-step, measured:
// profiles/README.md v23).
template <int PREC> __device__ __forceinline__ typename Elem<PREC>::Frag load_a_frag_buf(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ Elem<PN_PREC_BF16>::Frag load_a_frag_buf<PN_PREC_BF16>(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(Elem<PN_PREC_BF16>::Frag, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}
template <> __device__ __forceinline__ Elem<PN_PREC_F32>::Frag load_a_frag_buf<PN_PREC_F32>(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    Elem<PN_PREC_F32>::Frag f;
    f.lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
    f.hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + 1024u, 0));
    return f;
}

template <int PREC, int KS, int STRIDE, int PITCH, int CFG>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvProblem *__restrict__ probs) {
    typedef Elem<PREC> E;
    typedef typename E::T T;
    typedef typename E::Frag Frag;
    constexpr int PIXB = E::PIXB, FRAGB = E::FRAGB, SUBX = E::SUBX;
    constexpr int WC = TileCfg<CFG>::WC, WP = TileCfg<CFG>::WP, CT = TileCfg<CFG>::CT, PT = TileCfg<CFG>::PT;
    constexpr int KK = KS * KS, PAD = KS / 2;
    constexpr int NCH = PIXB / 16;      // 16-B pieces per halo pixel
    constexpr int PPI = 256 / NCH;      // halo pixels staged per block pass
    constexpr int ES = (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const ConvProblem &P = probs[blockIdx.y];
    if ((int)blockIdx.x >= P.nblocks) return;
    // XCD-aware remap (speed only): hardware deals consecutive block ids round-robin over the 8
    // XCDs; give each XCD a contiguous range of logical blocks so that blocks sharing an input
    // halo / a weight slice hit the same L2.  Bijective for any nblocks.
    int bx;
    {
        const int nb = P.nblocks, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int qq = nb >> 3, rr = nb & 7;
        bx = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }

    const int tid = threadIdx.x;
    PN_STAMP_AT(0);
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

    // ---- per-lane LDS addresses of tap (ky=0,kx) for each pixel tile, current 32-channel half ----
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
            baddr[pt][kx] = hp * PIXB + ((q ^ (hp & 7)) << (PREC == PN_PREC_BF16 ? 4 : 5));
        }
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- software pipeline -------------------------------------------------------------------
    //  * k order inside a 64-channel chunk: (32-channel half, tap): the swizzled LDS address of the
    //    second half is the first half's XOR SUBX, so the XOR is applied once per half (not per read);
    //  * weights (A): NA-slot register queue, loads run NA-1 k-steps ahead of their MFMAs;
    //  * activations (B): DB fragments in flight, read between the MFMAs of earlier pixel tiles;
    //  * halo staging: the NEXT chunk is fetched global -> registers before this chunk's MFMAs and
    //    written to the other LDS image after them (one barrier per chunk).
    // No branch inside the K loop: pixel tiles beyond the tile's valid pixels compute on clamped
    // addresses and are masked in the epilogue only.
    constexpr int NSTEP = KK * 2;                       // k32 steps per 64-channel chunk
    constexpr int NA = (PREC == PN_PREC_BF16) ? ((NSTEP % 6 == 0) ? 6 : 2) : ((NSTEP % 3 == 0) ? 3 : 2);
    constexpr int NITEM = NSTEP * PT;                   // (k-step, pixel tile) items per chunk
    constexpr int DB = (PREC == PN_PREC_BF16) ? ((NITEM % 3 == 0) ? 3 : 2) : 1;   // B fragments in flight
    constexpr int MAXST = StageCfg<PREC, KS, STRIDE, PITCH, CFG>::MAXST;   // 0: stage without register prefetch
    const int nchunks = P.cin_chunks;
    const int in_wrap = P.in_wrap;

    // weight stream: scalar base per cout tile + lane offset
    const int ctile0 = (cb * WC + wc) * CT;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(P.wpack), 0, 0x7fffffff, 0x00020000);
    unsigned wbase[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wbase[ct] = (unsigned)__builtin_amdgcn_readfirstlane((ctile0 + ct) * P.ksteps * FRAGB);
    const unsigned wlane = (unsigned)lane * 16u;
    // bf16: buffer loads (fewer VGPRs); fp32 parity mode: the flat form measured 2 % faster (two loads per fragment)
#define PN_LOAD_W(off) load_w(off)
    auto load_w = [&](unsigned off) {
        if constexpr (PREC == PN_PREC_F32) return load_a_frag2<PREC>((gcptr)P.wpack + off, wlane);
        else return load_a_frag_buf<PREC>(wrsrc, wlane, off);
    };

    // halo staging: thread -> 16-B piece `ch` of halo pixels p0, p0+PPI, ...; coordinates advance
    // incrementally (no division per piece), global offsets are 32-bit from a scalar image base.
    const int ch = tid % NCH;
    const int p0 = tid / NCH;
    const int npx = HRa * HC;
    const int hy_first = (int)(((float)p0 + 0.5f) / (float)HC);
    const int hx_first = p0 - hy_first * HC;
    const int wrap1 = PPI / HC, wrap_rem = PPI - wrap1 * HC;         // one staging step = wrap1 rows + wrap_rem columns
    gcptr img = (gcptr)P.in + ((size_t)b * P.H * P.W * P.in_cs + P.in_coff) * ES + ch * 16;
    const int row_b = P.W * P.in_cs * ES, col_b = P.in_cs * ES;
    auto piece = [&](int hy, int hx, int &soff, int &dst, bool &inb) {
        const int iy = iy0 + hy, ix = ix0 + hx;
        inb = (unsigned)iy < (unsigned)P.H && (unsigned)ix < (unsigned)P.W;
        const int iyc = min(max(iy, 0), P.H - 1), ixc = min(max(ix, 0), P.W - 1);   // always a valid address
        soff = iyc * row_b + ixc * col_b;
        const int hp = hy * PITCH + hx;
        dst = hp * PIXB + (PREC == PN_PREC_BF16 ? ((ch ^ (hp & 7)) << 4) : ((((ch >> 1) ^ (hp & 7)) << 5) | ((ch & 1) << 4)));
    };
    auto advance = [&](int &hy, int &hx) {
        hx += wrap_rem;
        hy += wrap1;
        if (hx >= HC) { hx -= HC; ++hy; }
    };
    u32x4 st[MAXST > 0 ? MAXST : 1];
    auto stage_load = [&](int chunk) {                   // all loads issued back to back, no waits
        int hy = hy_first, hx = hx_first;
        gcptr src = img + (size_t)(chunk >= in_wrap ? chunk - in_wrap : chunk) * 64 * ES;      // bf16x3: the third plane pair reads the hi plane again
#pragma unroll
        for (int it = 0; it < MAXST; ++it) {
            int soff, dst; bool inb;
            piece(hy, hx, soff, dst, inb);
            st[it] = *reinterpret_cast<const PN_GLOBAL u32x4 *>(src + (unsigned)soff);
            advance(hy, hx);
        }
    };
    auto stage_store = [&](char *buf) {                  // zero padding is applied here, not at load time
        int hy = hy_first, hx = hx_first;
#pragma unroll
        for (int it = 0; it < MAXST; ++it) {
            int soff, dst; bool inb;
            piece(hy, hx, soff, dst, inb);
            if (p0 + it * PPI < npx) *reinterpret_cast<u32x4 *>(buf + dst) = inb ? st[it] : u32x4{0u, 0u, 0u, 0u};
            advance(hy, hx);
        }
    };
    auto stage_direct = [&](int chunk, char *buf) {      // large halos: batches of 4 loads, then 4 stores
        int hy = hy_first, hx = hx_first;
        gcptr src = img + (size_t)(chunk >= in_wrap ? chunk - in_wrap : chunk) * 64 * ES;      // bf16x3: the third plane pair reads the hi plane again
        for (int pb = p0; pb < npx; pb += 4 * PPI) {
            u32x4 v[4];
            int dst[4]; bool inb[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int soff;
                piece(hy, hx, soff, dst[k], inb[k]);
                v[k] = *reinterpret_cast<const PN_GLOBAL u32x4 *>(src + (unsigned)soff);
                advance(hy, hx);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (pb + k * PPI < npx) *reinterpret_cast<u32x4 *>(buf + dst[k]) = inb[k] ? v[k] : u32x4{0u, 0u, 0u, 0u};
        }
    };

    char *buf0 = smem, *buf1 = P.lds_two ? smem + P.lds_buf_bytes : smem;
    Frag aq[NA][CT];
#pragma unroll
    for (int d = 0; d < NA - 1; ++d)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) aq[d][ct] = PN_LOAD_W(wbase[ct] + d * FRAGB);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wbase[ct] += (NA - 1) * FRAGB;
    if (MAXST > 0) { stage_load(0); stage_store(buf0); }
    else stage_direct(0, buf0);
    PN_STAMP_AT(1);
    __syncthreads();
    PN_STAMP_AT(2);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const char *sm = (chunk & 1) ? buf1 : buf0;
        char *nbuf = (chunk & 1) ? buf0 : buf1;
        const bool more = chunk + 1 < nchunks;
        if (MAXST > 0 && more) stage_load(chunk + 1);

        // item j = (k-step s = half * KK + tap, pixel tile pt); the half is selected by the XOR state of baddr
#define PN_BADDR(j) (baddr[(j) % PT][(((j) / PT) % KK) % KS] + ((((j) / PT) % KK) / KS) * PITCH * PIXB)
        Frag bq[DB];
        if (DB > 1) {
#pragma unroll
            for (int j = 0; j < DB - 1; ++j) bq[j] = read_b_frag<PREC>(sm, PN_BADDR(j));
            __builtin_amdgcn_sched_barrier(0);          // keep the primed reads out of the pinned sequence below
        }
#define PN_BADDRX(j) ((baddr[(j) % PT][(((j) / PT) % KK) % KS] ^ SUBX) + ((((j) / PT) % KK) / KS) * PITCH * PIXB)
#pragma unroll
        for (int h = 0; h < 2; ++h) {                    // the two 32-channel halves of the chunk
#pragma unroll
            for (int i = 0; i < KK * PT; ++i) {
                const int j = h * KK * PT + i, s = j / PT, pt = j % PT;
                if (pt == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {    // wpack ends with NA-1 spare fragments
                        aq[(s + NA - 1) % NA][ct] = PN_LOAD_W(wbase[ct]);
                        wbase[ct] += FRAGB;
                    }
                }
                const int jr = (DB > 1) ? j + DB - 1 : j;                // item whose B fragment is read now
                if (jr < NITEM) {
                    // the few prefetches that already belong to the other half take the XOR explicitly
                    if (jr / (KK * PT) != h) bq[jr % DB] = read_b_frag<PREC>(sm, PN_BADDRX(jr));
                    else bq[jr % DB] = read_b_frag<PREC>(sm, PN_BADDR(jr));
                }
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[ct][pt] = mma(aq[s % NA][ct], bq[j % DB], acc[ct][pt]);
                if (DB > 1) {
                    // pin the issue order (the scheduler otherwise sinks every prefetch down to its use)
                    if (pt == 0) __builtin_amdgcn_sched_group_barrier(0x020, CT, 0);             // weight loads
                    if (jr < NITEM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);           // one LDS read
                    __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);                          // CT MFMAs
                }
            }
#pragma unroll
            for (int p2 = 0; p2 < PT; ++p2)              // switch half (and back to the first one after the second)
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) baddr[p2][kx] ^= SUBX;
        }
#undef PN_BADDRX
#undef PN_BADDR
        PN_STAMP_AT(3 + 2 * (chunk & 3));
        if (more) {
            if (!P.lds_two) __syncthreads();             // single LDS image: wait for every wave's reads
            if (MAXST > 0) stage_store(nbuf);
            else stage_direct(chunk + 1, nbuf);
            __syncthreads();
        }
        PN_STAMP_AT(4 + 2 * (chunk & 3));
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // The MFMA C layout gives a lane 4 consecutive rows (couts) of one pixel per accumulator.  The
    // packed weight rows are permuted on the host (net.hip::prepare_conv, pn_conv_row_channel) so
    // that the CT accumulators of a lane together are LC = 4*CT CONSECUTIVE output channels: each
    // lane adds the folded bias, the residual (one 16-B load), applies the activation and writes
    // one 16-B (bf16, CT = 2) NHWC piece per pixel tile straight from registers -- no LDS transpose,
    // no barrier; the four lane quarters of a wave cover 64 contiguous bytes of a pixel's line.
    PN_STAMP_AT(11);
    constexpr int BC = WC * CT * 16;
    constexpr int LC = CT * 4;
    const int cw = (cb * WC + wc) * (CT * 16) + LC * q;          // first output channel of this lane
    const int cout = P.cout, act = P.act;
    float bias[LC];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 b4 = *reinterpret_cast<const PN_GLOBAL f32x4 *>((const PN_GLOBAL float *)P.bias + cw + 4 * ct);
        bias[4 * ct + 0] = b4[0]; bias[4 * ct + 1] = b4[1]; bias[4 * ct + 2] = b4[2]; bias[4 * ct + 3] = b4[3];
    }
    (void)BC;
    const bool full = cw + LC <= cout;
    const PN_GLOBAL T *res_base = P.res ? (const PN_GLOBAL T *)P.res + P.res_coff + cw : nullptr;
    PN_GLOBAL T *out_base = P.out ? (PN_GLOBAL T *)P.out + P.out_coff + cw : nullptr;
    PN_GLOBAL float *nchw = (PN_GLOBAL float *)P.out_nchw;
    const int res_cs = P.res_cs, out_cs = P.out_cs, Ho = P.Ho, naf = P.yolo_naf;
    const int split = P.split, res_split = P.res_split;
    const int pix0 = (b * Ho + oy0) * Wo + ox0;
    auto finish = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;            // -1: dispatch at run time (sigmoid casts: last layers only)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int slot = (wp * PT + pt) * 16 + c;
            if (slot >= npix || cw >= cout) continue;
            const int ry = (int)(((float)slot + 0.5f) * inv_wc);
            const int rx = slot - ry * Wc;
            const int opix = pix0 + ry * Wo + rx;
            float v[LC];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * ct + i] = acc[ct][pt][i] + bias[4 * ct + i];
            for (int pl = 0; pl < (res_base ? (res_split ? 2 : 1) : 0); ++pl) {      // bf16x3: residual = hi plane + lo plane
                const PN_GLOBAL T *rp = res_base + (unsigned)(opix * res_cs + pl * res_split);
                if (full) {
                    T rv[LC];
                    if (LC * sizeof(T) == 16) {
                        *reinterpret_cast<u32x4 *>(rv) = *reinterpret_cast<const PN_GLOBAL u32x4 *>(rp);
                    } else if (LC * sizeof(T) == 32) {
                        reinterpret_cast<u32x4 *>(rv)[0] = reinterpret_cast<const PN_GLOBAL u32x4 *>(rp)[0];
                        reinterpret_cast<u32x4 *>(rv)[1] = reinterpret_cast<const PN_GLOBAL u32x4 *>(rp)[1];
                    } else {
#pragma unroll
                        for (int k = 0; k < LC; ++k) rv[k] = rp[k];
                    }
#pragma unroll
                    for (int k = 0; k < LC; ++k) v[k] += (float)rv[k];
                } else {
                    for (int k = 0; k < LC; ++k)
                        if (cw + k < cout) v[k] += (float)rp[k];
                }
            }
#pragma unroll
            for (int k = 0; k < LC; ++k) {
                if (ACT == PN_ACT_NONE) {}
                else if (ACT == PN_ACT_RELU) v[k] = v[k] > 0.f ? v[k] : 0.f;
                else if (ACT == PN_ACT_LEAKY) v[k] = v[k] > 0.f ? v[k] : v[k] * 0.1f;
                else v[k] = pn_activate(v[k], act, cw + k, naf);
            }
            if (out_base) {
                // bf16x3: two planes [hi | lo] `split` channels apart, hi = bf16(v), lo = bf16(v - hi)
                for (int pl = 0; pl < (split ? 2 : 1); ++pl) {
                    PN_GLOBAL T *op = out_base + (unsigned)(opix * out_cs + pl * split);
                    T ov[LC];
#pragma unroll
                    for (int k = 0; k < LC; ++k) {
                        const T hi = (T)v[k];
                        ov[k] = pl == 1 ? (T)(v[k] - (float)hi) : hi;
                    }
                    if (full) {
                        if (LC * sizeof(T) == 16) {
                            *reinterpret_cast<PN_GLOBAL u32x4 *>(op) = *reinterpret_cast<u32x4 *>(ov);
                        } else if (LC * sizeof(T) == 32) {
                            reinterpret_cast<PN_GLOBAL u32x4 *>(op)[0] = reinterpret_cast<u32x4 *>(ov)[0];
                            reinterpret_cast<PN_GLOBAL u32x4 *>(op)[1] = reinterpret_cast<u32x4 *>(ov)[1];
                        } else {
#pragma unroll
                            for (int k = 0; k < LC; ++k) op[k] = ov[k];
                        }
                    } else {
                        for (int k = 0; k < LC; ++k)
                            if (cw + k < cout) op[k] = ov[k];
                    }
                }
            }
            if (nchw) {                                       // API-boundary layout: 16 lanes = 16 consecutive pixels of a plane
                const size_t hw = (size_t)Ho * Wo;
                PN_GLOBAL float *np = nchw + ((size_t)b * cout + cw) * hw + (size_t)(oy0 + ry) * Wo + (ox0 + rx);
                for (int k = 0; k < LC; ++k)
                    if (cw + k < cout) np[(size_t)k * hw] = v[k];
            }
        }
    };
    if (act == PN_ACT_RELU) finish(std::integral_constant<int, PN_ACT_RELU>{});
    else if (act == PN_ACT_LEAKY) finish(std::integral_constant<int, PN_ACT_LEAKY>{});
    else if (act == PN_ACT_NONE) finish(std::integral_constant<int, PN_ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
    PN_STAMP_AT(12);
}


template <int PREC, int KS, int STRIDE, int PITCH, int CFG>
static int conv_launch_one(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    auto kern = conv_mfma_kernel<PREC, KS, STRIDE, PITCH, CFG>;
    if (L.lds_bytes > 160 * 1024)
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "conv halo tile needs %zu B of LDS", L.lds_bytes);
    if (L.lds_bytes > 48 * 1024) {
        static PnLdsAttr attr;          // per instantiation, per device
        if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(kern), L.lds_bytes)) return rc;
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
