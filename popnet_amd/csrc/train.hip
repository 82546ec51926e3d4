// Training step kernels (BASELINE configs[4]): fp32, NCHW (the layout of the reference's tensors), one
// C-ABI entry per differentiable primitive of rtpose_light3d in train mode.  The host side (popnet_amd/train.py) strings
// them together in the order autograd would.
//   conv forward / data gradient / weight gradient   implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains)
//       nn.Conv2d of tpm/lib/network/rtpose_light3d.py:31-40,144-146,232-246 and its autograd
//   train-mode BatchNorm2d (+ residual add + ReLU / LeakyReLU(0.1)) forward and backward      rtpose_light3d.py:52-70,147,244
//   AvgPool2d(3, 2, 1) forward / backward                                                     rtpose_light3d.py:152,158
//   sigmoid heads + rtpose_light3d_loss_fgweight forward and gradient                         rtpose_light3d.py:335-337, losses.py:65-106
//   SGD with Nesterov momentum                                                                train_rtpose_light3d_kdh3d_mpaug.py:313-316
// Roofline: the three GEMM-shaped kernels are MFMA-bound (fp32-input matrix peak 157 TFLOP/s), everything else HBM-bound.
#include <cmath>
#include <cstdlib>
#include "pn_internal.h"

typedef float t_f32x4 __attribute__((ext_vector_type(4)));

struct TConv {
    const float *x;      // [N, Cin, H, W]
    const float *w;      // [Cout, Cin * KS * KS]
    const float *bias;   // [Cout] or nullptr
    float *y;            // [N, Cout, Ho, Wo]
    int N, Cin, H, W, Cout, Ho, Wo, stride, pad, accumulate;
    int Kdim;            // Cin * KS * KS
    int P;               // N * Ho * Wo
    // round 6 (the planes training engine's stem, trainx.hip): y / dY as a channel-minor planes tensor [pixel][pl_cs] instead of NCHW f32 --
    // PL = 1: two bf16 planes [hi | lo] `pl_split` elements apart (value = hi + lo), PL = 2: one f32 plane.  Same values, same MFMA order as the
    // NCHW form followed / preceded by trainx_kernels.h's layout pass: bit-identical, one 103 MB round trip less each way.
    void *pl = nullptr;
    int pl_cs = 0, pl_split = 0;
};
typedef __attribute__((ext_vector_type(8))) __bf16 t_bf8;
typedef __attribute__((ext_vector_type(4))) __bf16 t_bf4;
typedef __attribute__((ext_vector_type(2))) float t_f32x2;

// ---------------------------------------------------------------------------------------------------------------------
// Forward (and, with flipped weights, the stride-1 data gradient): D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],
// k = (ci, ky, kx).  Block = 64 couts x 128 pixels, 4 waves (each 64 couts x 32 pixels = 4 x 2 MFMA tiles), K chunks of 16
// staged through LDS (weights k-major, the gathered input k-major; lanes run along the pixels, so every global gather is a
// run of consecutive addresses), next chunk's global loads in flight during the MFMAs of the current one.
// ---------------------------------------------------------------------------------------------------------------------
#define TC_KC 16
#define TC_AP 80      // LDS pitches: 4 k rows x 16 lanes of an MFMA operand read fall on 64 different banks
#define TC_BP 144

// ---------------------------------------------------------------------------------------------------------------------
// Reuse-aware block order (round 5).  The dispatcher deals the workgroups of a launch round-robin over the 8 XCDs in linear-id order
// (x fastest) and every XCD has its own 4 MB L2.  With the plain (tile, cout block) / (ci block, cout block, slice) grids the blocks
// that read the SAME operand tile -- the cout blocks of one input tile in the forward / data-gradient kernels, the (ci, cout) blocks of
// one pixel slice in the weight gradient -- sat on different XCDs or ran a whole grid apart in time: every operand tile crossed the
// fabric once per sharer (a 256 -> 256 weight gradient staged 444 MB for 51 MB of tensors).  t_logical_block() maps the hardware id to
// a LOGICAL id such that each XCD owns one contiguous range of logical ids (a bijection, the conv kernels' remap of net.hip); the
// kernels decode the logical id with the sharing dimension fastest, so sharers run at the same time behind the same L2.  Which block
// computes which tile changes, nothing else: results are bit-identical.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int t_logical_block() {
    const int nb = (int)(gridDim.x * gridDim.y * gridDim.z);
    const int L = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    const int xcd = L & 7, idx = L >> 3, qq = nb >> 3, rr = nb & 7;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
}
// Sixteen loaded values pinned in registers at this point of the program: every one of their loads has been issued before the first is
// waited for.  Without it the compiler sinks each load to its single use (legal: the addresses are provably distinct from the stores in
// between) and waits for it there -- a dependent memory round trip per element.
#define T_PIN16(a) do { _Pragma("unroll") for (int _k = 0; _k < 16; ++_k) asm volatile("" : "+v"((a)[_k])); } while (0)
// forward-type grids (tiles, cout blocks): the cout blocks of a tile are neighbours in logical order
#define T_DECODE_TILE_CB(tile, cb) const int _lg = t_logical_block(), cb = _lg % (int)gridDim.y, tile = _lg / (int)gridDim.y
// weight-gradient grids (column blocks, cout blocks, slices): a slice's blocks are one contiguous logical range
#define T_DECODE_XYZ(bx, by, bz) const int _lg = t_logical_block(), bx = _lg % (int)gridDim.x, by = (_lg / (int)gridDim.x) % (int)gridDim.y, bz = _lg / (int)(gridDim.x * gridDim.y)

template <int KS, int PL = 0>
__global__ __launch_bounds__(256) void tconv_fwd_kernel(TConv c) {
    __shared__ float As[TC_KC][TC_AP];
    __shared__ float Bs[TC_KC][TC_BP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_TILE_CB(pblk, cblk);
    const int p0 = pblk * 128, co0 = cblk * 64;
    const int HoWo = c.Ho * c.Wo;
    // staging roles
    const int b_pn = t & 127, b_k0 = t >> 7;            // B: pixel column, k rows b_k0 + 2j
    const int a_co = t & 63, a_k0 = (t >> 6) * 4;       // A: cout row, k rows a_k0 + j
    const int bp = p0 + b_pn;
    const bool bp_ok = bp < c.P;
    int bn = 0, iy0 = 0, ix0 = 0;
    if (bp_ok) {
        bn = bp / HoWo;
        const int rem = bp - bn * HoWo, oy = rem / c.Wo, ox = rem - oy * c.Wo;
        iy0 = oy * c.stride - c.pad;
        ix0 = ox * c.stride - c.pad;
    }
    // every load is unconditional (an out-of-range element reads index 0; it is zeroed when the value goes to LDS, one chunk
    // later -- a select right after the load would put an s_waitcnt vmcnt(0) in front of the MFMAs the load is meant to
    // overlap with, and a conditional load becomes a branch of its own: 73 of them in the first version of this loop)
    const float *xb = c.x + (size_t)bn * c.Cin * c.H * c.W;
    const bool a_ok = co0 + a_co < c.Cout;
    const float *wa = c.w + (a_ok ? (size_t)(co0 + a_co) * c.Kdim : 0);

    float ra[4], rb[8];
    unsigned okm = 0;          // bit j: rb[j] valid, bit 8 + j: ra[j] valid
    auto load = [&](int k0) {
        okm = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + a_k0 + j;
            const int ok = (int)(a_ok & (k < c.Kdim));
            ra[j] = wa[k & -ok];
            okm |= (unsigned)ok << (8 + j);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + b_k0 + 2 * j;
            const int ci = k / (KS * KS), rr = k - ci * (KS * KS), ky = rr / KS, kx = rr - ky * KS;
            const int iy = iy0 + ky, ix = ix0 + kx;
            const int ok = (int)(bp_ok & (k < c.Kdim) & (iy >= 0) & (iy < c.H) & (ix >= 0) & (ix < c.W));
            rb[j] = xb[((ci * c.H + iy) * c.W + ix) & -ok];
            okm |= (unsigned)ok << j;
        }
    };
    t_f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = t_f32x4{0.f, 0.f, 0.f, 0.f};

    load(0);
    for (int k0 = 0; k0 < c.Kdim; k0 += TC_KC) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) As[a_k0 + j][a_co] = (okm >> (8 + j)) & 1u ? ra[j] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) Bs[b_k0 + 2 * j][b_pn] = (okm >> j) & 1u ? rb[j] : 0.f;
        __syncthreads();
        if (k0 + TC_KC < c.Kdim) load(k0 + TC_KC);
#pragma unroll
        for (int ks = 0; ks < TC_KC / 4; ++ks) {
            float a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = As[4 * ks + q][16 * m + r];
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = Bs[4 * ks + q][32 * wave + 16 * n + r];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
    }
    // bias of this lane's sixteen couts, gathered once (round 5: `v += c.bias[co]` inside the store loop compiled to load, s_waitcnt vmcnt(0), add --
    // sixteen dependent round trips per pixel tile, as did the accumulate form's old values, which the compiler had sunk to their uses)
    float bv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int co = co0 + 16 * (k >> 2) + 4 * q + (k & 3);
        bv[k] = c.bias ? c.bias[co < c.Cout ? co : 0] : 0.f;
    }
    T_PIN16(bv);
    // epilogue: lane holds couts 16m + 4q + i of pixel 32 wave + 16 n + r; the accumulate form first gathers all old values
    // (32 independent loads in flight), then adds and stores
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int p = p0 + 32 * wave + 16 * n + r;
        const int pok = (int)(p < c.P);
        const int pc = p & -pok;
        if (PL) {       // planes: this lane's four consecutive couts of a 16-cout tile are one 8-byte (bf16 hi, lo) / 16-byte (f32) store
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int co = co0 + 16 * m + 4 * q;
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = acc[m][n][i] + bv[4 * m + i];
                if (!pok || co >= c.Cout) continue;
                if (PL == 2) {
                    *reinterpret_cast<t_f32x4 *>((float *)c.pl + (size_t)pc * c.pl_cs + co) = t_f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    t_bf4 h, l;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const __bf16 hi = (__bf16)v[i];
                        h[i] = hi;
                        l[i] = (__bf16)(v[i] - (float)hi);
                    }
                    __bf16 *o = (__bf16 *)c.pl + (size_t)pc * c.pl_cs + co;
                    *reinterpret_cast<t_bf4 *>(o) = h;
                    *reinterpret_cast<t_bf4 *>(o + c.pl_split) = l;
                }
            }
            continue;
        }
        const int img = pc / HoWo, rem = pc - img * HoWo;
        float *yb = c.y + (size_t)img * c.Cout * HoWo + rem;
        float old[16];
        if (c.accumulate) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + 16 * m + 4 * q + i;
                    old[4 * m + i] = yb[(co * HoWo) & -(int)(co < c.Cout)];
                }
            T_PIN16(old);                               // all sixteen loads issued, THEN used (see T_PIN16)
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * m + 4 * q + i;
                float v = acc[m][n][i];
                v += bv[4 * m + i];
                if (c.accumulate) v += old[4 * m + i];
                if (pok && co < c.Cout) yb[(size_t)co * HoWo] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolutions (95 % of the step's FLOPs), second generation.  PMC on the gather kernel above: the texture
// addresser is as busy per CU as the matrix pipe per SIMD (TA/MFMA 1.01, MfmaUtil 28 %) -- every input element is fetched
// nine times, once per tap, four bytes per lane.  Here a block stages the HALO TILE of 16 input channels once (rows of the
// image, coalesced) plus the 9 x 16 x 64 weight slice (pre-transposed to [tap][ci][cout] by wpack_kernel, coalesced along
// cout) and runs 288 MFMAs per wave between two barriers; the nine taps are LDS address offsets.  Output tile = R rows x TW
// columns of one image, R * TW <= 128 (28-wide maps: 4 x 28, 56: 2 x 56, 112: 1 x 112).  Two blocks per CU overlap one
// block's staging with the other's MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
struct TTile {
    int TW, R, tiles_x, tiles_y;   // output tile, tiles per image
    int HC, HR;                    // halo columns / rows = TW + 2, R + 2
    int CHP;                       // LDS floats per halo channel (HR * HC rounded up to = 16 mod 64: forward, = 4 mod 64: wgrad)
    int NI;                        // ceil(HR * HC / 256)
};
#define TT_AP 80
#define TT_MAXNI 2

// wp[tap][ci][co] = flip ? W[co_w = ci][ci_w = co][8 - tap] : W[co][ci][tap]   (flip: the data gradient's transposed, rotated weights)
__global__ void wpack3_kernel(const float *__restrict__ w, float *__restrict__ wp, int Cout, int Cin, int flip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * Cin * Cout) return;
    const int co = i % Cout, ci = (i / Cout) % Cin, tap = i / (Cout * Cin);
    wp[i] = flip ? w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : w[((size_t)co * Cin + ci) * 9 + tap];
}

__global__ __launch_bounds__(256, 2) void tconv3_tile_kernel(TConv c, TTile g, const float *__restrict__ wp) {
    extern __shared__ float t_smem[];
    float *As = t_smem;                       // [9 * 16][TT_AP]   weights  (tap, channel) x cout
    float *Hs = t_smem + 144 * TT_AP;         // [16][CHP]         halo tile of 16 input channels
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_TILE_CB(b, cblk);
    const int tx = b % g.tiles_x, ty = (b / g.tiles_x) % g.tiles_y, img = b / (g.tiles_x * g.tiles_y);
    const int y0 = ty * g.R, x0 = tx * g.TW, co0 = cblk * 64;
    const int HW = c.H * c.W, HoWo = c.Ho * c.Wo;
    // this lane's two pixel slots (MFMA columns)
    int hb[2], opix[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int sl = 32 * wave + 16 * n + r;
        const int ry = sl / g.TW, rx = sl - ry * g.TW;
        const bool ok = ry < g.R && y0 + ry < c.Ho && x0 + rx < c.Wo;
        hb[n] = ok ? ry * g.HC + rx : 0;
        opix[n] = ok ? (y0 + ry) * c.Wo + x0 + rx : -1;
    }
    // halo elements this thread stages for every channel: offset inside the image plane (or -1 = zero padding)
    int hoff[TT_MAXNI], hdst[TT_MAXNI];
#pragma unroll
    for (int i = 0; i < TT_MAXNI; ++i) {
        const int e = t + 256 * i;
        const int hy = e / g.HC, hx = e - hy * g.HC;
        const int iy = y0 - c.pad + hy, ix = x0 - c.pad + hx;
        const bool in = e < g.HR * g.HC;
        hdst[i] = in ? e : -1;
        hoff[i] = (in && iy >= 0 && iy < c.H && ix >= 0 && ix < c.W) ? iy * c.W + ix : -1;
    }
    const float *xb = c.x + (size_t)img * c.Cin * HW;
    const int a_co = t & 63, a_r0 = t >> 6;
    const bool a_ok = co0 + a_co < c.Cout;
    const float *wa = wp + (a_ok ? co0 + a_co : 0);

    t_f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = t_f32x4{0.f, 0.f, 0.f, 0.f};

    // software pipeline: the global loads of chunk c0 + 16 stay in flight (in registers) under the 288 MFMAs of chunk c0, so a
    // block does not depend on its CU neighbour being in the opposite phase (co-resident blocks start together and stay in
    // lockstep: both stage, then both compute at half rate each)
    float rh[16 * TT_MAXNI];
    float rw[36];
    auto load = [&](int c0) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const int cok = (int)(c0 + kk < c.Cin);
#pragma unroll
            for (int i = 0; i < TT_MAXNI; ++i) {
                const int ok = cok & (int)(hoff[i] >= 0);
                rh[kk * TT_MAXNI + i] = xb[((c0 + kk) * HW + hoff[i]) & -ok];
            }
        }
#pragma unroll
        for (int j = 0; j < 36; ++j) {            // row = tap * 16 + kk
            const int row = a_r0 + 4 * j, tap = row >> 4, kk = row & 15;
            const int ok = (int)(c0 + kk < c.Cin);
            rw[j] = wa[((tap * c.Cin + c0 + kk) * c.Cout) & -ok];
        }
    };
    load(0);
    for (int c0 = 0; c0 < c.Cin; c0 += 16) {
        __syncthreads();                          // the previous chunk's MFMAs have read their fragments
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < TT_MAXNI; ++i)
                if (hdst[i] >= 0) Hs[kk * g.CHP + hdst[i]] = (c0 + kk < c.Cin && hoff[i] >= 0) ? rh[kk * TT_MAXNI + i] : 0.f;
#pragma unroll
        for (int j = 0; j < 36; ++j) {
            const int row = a_r0 + 4 * j;
            As[row * TT_AP + a_co] = (a_ok && c0 + (row & 15) < c.Cin) ? rw[j] : 0.f;
        }
        __syncthreads();
        if (c0 + 16 < c.Cin) load(c0 + 16);
        // 9 taps x 4 k-quads; the operands of the next step are read from LDS before the 8 MFMAs of the current one issue (the
        // compiler's own schedule was read -> s_waitcnt lgkmcnt(0) -> 4 MFMAs, every LDS latency exposed).  The tap loop is
        // NOT unrolled: fully unrolled the hoisted LDS addresses of 36 steps push the kernel past 256 VGPRs.
        float a[2][4], bb[2][2];
        auto frag = [&](int tap, int ks, float (&fa)[4], float (&fb)[2]) {
            const int ty3 = tap / 3;
            const int toff = ty3 * g.HC + (tap - 3 * ty3);
#pragma unroll
            for (int m = 0; m < 4; ++m) fa[m] = As[(tap * 16 + 4 * ks + q) * TT_AP + 16 * m + r];
#pragma unroll
            for (int n = 0; n < 2; ++n) fb[n] = Hs[(4 * ks + q) * g.CHP + hb[n] + toff];
        };
        frag(0, 0, a[0], bb[0]);
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) frag(tap, ks + 1, a[(ks + 1) & 1], bb[(ks + 1) & 1]);
                else frag(tap < 8 ? tap + 1 : 8, 0, a[0], bb[0]);           // (the last one re-reads tap 8: harmless)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks & 1][m], bb[ks & 1][n], acc[m][n], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);      // next step's 6 LDS reads ...
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // ... then this step's 8 MFMAs
            }
        }
    }
    // bias of this lane's sixteen couts, gathered once (round 5: `v += c.bias[co]` inside the store loop compiled to load, s_waitcnt vmcnt(0), add --
    // sixteen dependent round trips per pixel tile, as did the accumulate form's old values, which the compiler had sunk to their uses)
    float bv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int co = co0 + 16 * (k >> 2) + 4 * q + (k & 3);
        bv[k] = c.bias ? c.bias[co < c.Cout ? co : 0] : 0.f;
    }
    T_PIN16(bv);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int pok = (int)(opix[n] >= 0);
        float *yb = c.y + (size_t)img * c.Cout * HoWo + (opix[n] & -pok);
        float old[16];
        if (c.accumulate) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + 16 * m + 4 * q + i;
                    old[4 * m + i] = yb[(co * HoWo) & -(int)(co < c.Cout)];
                }
            T_PIN16(old);                               // all sixteen loads issued, THEN used (see T_PIN16)
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * m + 4 * q + i;
                float v = acc[m][n][i];
                v += bv[4 * m + i];
                if (c.accumulate) v += old[4 * m + i];
                if (pok && co < c.Cout) yb[(size_t)co * HoWo] = v;
            }
    }
}

// Weight gradient of the 3x3 stride-1 convolutions on the same tiles: a block owns 64 couts x (16 input channels x 9 taps) and
// walks a slice of the output tiles; per tile it stages dY [128 pixel slots][64 couts] and the halo tile of its 16 input
// channels, then 32 pixel groups x 9 taps = 288 MFMAs per wave (A = dY: row = cout, k = pixel; B = X: k = pixel, column =
// channel, one 16-column MFMA tile per tap, the tap again an LDS address offset).  Partials per slice, reduced in order.
#define TT_YP 81      // = 17 (mod 64): conflict-free pixel-major writes, near conflict-free (q * 17 + r) MFMA operand reads
#define TT_HP 17      // halo tile of the weight gradient is PIXEL-major: [halo pixel][16 channels + 1]
__global__ __launch_bounds__(256, 2) void tconv3_wgrad_tile_kernel(TConv c, TTile g, float *__restrict__ partial, int tiles_per_slice, int ntiles) {
    extern __shared__ float t_smem[];
    float *Ys = t_smem;                         // [128][TT_YP]   dY tile, pixel-major
    float *Hs = t_smem + 128 * TT_YP;           // [HR * HC][TT_HP]  halo tile of this block's 16 input channels, pixel-major
    int *hbt = (int *)(Hs + g.HR * g.HC * TT_HP);   // [128]       pixel slot -> halo pixel of its top-left tap
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_XYZ(bx_, by_, slice);
    const int c0 = bx_ * 16, co0 = by_ * 64;
    const int HW = c.H * c.W, HoWo = c.Ho * c.Wo;
    // per-thread constants of the staging: the dY slot and the halo elements (same for every tile up to the tile origin)
    const int sl = t & 127, ry = sl / g.TW, rx = sl - ry * g.TW, y_c0 = t >> 7;
    int hy[TT_MAXNI], hx[TT_MAXNI], hdst[TT_MAXNI];
#pragma unroll
    for (int i = 0; i < TT_MAXNI; ++i) {
        const int e = t + 256 * i;
        hy[i] = e / g.HC;
        hx[i] = e - hy[i] * g.HC;
        hdst[i] = e < g.HR * g.HC ? e : -1;
    }
    t_f32x4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    const int tbeg = slice * tiles_per_slice, tend = min(tbeg + tiles_per_slice, ntiles);
    for (int tile = tbeg; tile < tend; ++tile) {
        const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, img = tile / (g.tiles_x * g.tiles_y);
        const int y0 = ty * g.R, x0 = tx * g.TW;
        const bool sok = ry < g.R && y0 + ry < c.Ho && x0 + rx < c.Wo;
        const float *dyb = c.y + (size_t)img * c.Cout * HoWo + (sok ? (y0 + ry) * c.Wo + x0 + rx : 0);
        const float *xb = c.x + (size_t)img * c.Cin * HW;
        float rd[32], rh[16 * TT_MAXNI];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int co = co0 + y_c0 + 2 * j;
            rd[j] = dyb[(co * HoWo) & -(int)(sok & (co < c.Cout))];
        }
        int hoff[TT_MAXNI];
#pragma unroll
        for (int i = 0; i < TT_MAXNI; ++i) {
            const int iy = y0 - c.pad + hy[i], ix = x0 - c.pad + hx[i];
            hoff[i] = (hdst[i] >= 0 && iy >= 0 && iy < c.H && ix >= 0 && ix < c.W) ? iy * c.W + ix : -1;
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < TT_MAXNI; ++i) {
                const int ok = (int)(c0 + kk < c.Cin) & (int)(hoff[i] >= 0);
                rh[kk * TT_MAXNI + i] = xb[((c0 + kk) * HW + hoff[i]) & -ok];
            }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int co = co0 + y_c0 + 2 * j;
            Ys[sl * TT_YP + y_c0 + 2 * j] = (sok && co < c.Cout) ? rd[j] : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int i = 0; i < TT_MAXNI; ++i)
                if (hdst[i] >= 0) Hs[hdst[i] * TT_HP + kk] = (c0 + kk < c.Cin && hoff[i] >= 0) ? rh[kk * TT_MAXNI + i] : 0.f;
        if (t < 128) hbt[t] = sok ? ry * g.HC + rx : 0;
        __syncthreads();
#pragma unroll 4
        for (int pg = 0; pg < 32; ++pg) {
            const float a = Ys[(4 * pg + q) * TT_YP + 16 * wave + r];
            const int hb = hbt[4 * pg + q] * TT_HP + r;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Hs[hb + ((tap / 3) * g.HC + (tap % 3)) * TT_HP], acc[tap], 0, 0, 0);
        }
    }
    float *pb = partial + (size_t)slice * c.Cout * c.Kdim;
    if (c0 + r < c.Cin) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * wave + 4 * q + i;
                if (co < c.Cout) pb[((size_t)co * c.Cin + c0 + r) * 9 + tap] = acc[tap][i];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same 3x3 tile convolution on the bf16 matrix instruction with SPLIT operands ("bf16x3", opt-in through
// pn_train_set_precision): every fp32 value v is staged as hi = bf16(v), lo = bf16(v - hi) (16 mantissa bits together) and
// a product sum is hi*hi + hi*lo + lo*hi in fp32 accumulators (the dropped lo*lo term is 2^-16 relative) -- three
// v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) do the work of eight v_mfma_f32_16x16x4_f32 (32 cycles each): 5.3x less
// matrix-pipe time for fp32-class results.  Chunks of 32 input channels; LDS images are [row][32 channels] bf16 with an
// 80-byte pitch (16-byte fragment reads and 16-byte staging writes both conflict-free); the weight slice is staged one
// kernel row (3 taps) at a time.
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 t_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned t_u32x4 __attribute__((ext_vector_type(4)));
#define TX_PITCH 80                     // bytes per [32 x bf16] row
#define TX_A_BYTES (3 * 64 * TX_PITCH)  // one plane of the weight slice of one kernel row

__device__ __forceinline__ void t_split8(const float (&v)[8], unsigned okmask, t_bf16x8 &hi, t_bf16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = (okmask >> j) & 1u ? v[j] : 0.f;
        const __bf16 h = (__bf16)x;
        hi[j] = h;
        lo[j] = (__bf16)(x - (float)h);
    }
}

// Weights of the split-bf16 kernel, packed once per launch in the LDS image's own order: wpx[plane][chunk of 32 ci][tap][cout][32 ci]
// bf16 (plane 0 = hi, 1 = lo; channels beyond Cin are zero), so that staging a kernel row is six 16-byte copies per thread instead
// of 24 dword loads + the split arithmetic per thread.  flip as in wpack3_kernel (the data gradient's rotated, transposed weights).
__global__ void wpack3_x3_kernel(const float *__restrict__ w, __bf16 *__restrict__ wpx, int Cout, int Cin, int flip) {
    const int chunks = (Cin + 31) / 32;
    const size_t plane = (size_t)chunks * 9 * Cout * 32;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const int ch = (int)(i & 31), co = (int)((i >> 5) % Cout), tap = (int)(((i >> 5) / Cout) % 9), chunk = (int)((i >> 5) / Cout / 9);
    const int ci = chunk * 32 + ch;
    float v = 0.f;
    if (ci < Cin) v = flip ? w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : w[((size_t)co * Cin + ci) * 9 + tap];
    const __bf16 h = (__bf16)v;
    wpx[i] = h;
    wpx[plane + i] = (__bf16)(v - (float)h);
}

// Every cached pack of a context refreshed by ONE launch (pn_train_pack_refresh): the block finds its descriptor (<= ~70 entries, scanned by
// every thread: uniform) and runs wpack3_kernel's / wpack3_x3_kernel's element arithmetic on it -- the same values as the per-call packs.
struct TPackDesc { const float *w; void *dst; int Cout, Cin, flip, x3; unsigned first_block, pad; };
__global__ void wpack_all_kernel(const TPackDesc *__restrict__ tab, int n) {
    int e = 0;
    while (e + 1 < n && blockIdx.x >= tab[e + 1].first_block) ++e;
    const TPackDesc d = tab[e];
    const size_t i = (size_t)(blockIdx.x - d.first_block) * blockDim.x + threadIdx.x;
    const int Cout = d.Cout, Cin = d.Cin;
    if (d.x3) {
        const int chunks = (Cin + 31) / 32;
        const size_t plane = (size_t)chunks * 9 * Cout * 32;
        if (i >= plane) return;
        const int ch = (int)(i & 31), co = (int)((i >> 5) % Cout), tap = (int)(((i >> 5) / Cout) % 9), chunk = (int)((i >> 5) / Cout / 9);
        const int ci = chunk * 32 + ch;
        float v = 0.f;
        if (ci < Cin) v = d.flip ? d.w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : d.w[((size_t)co * Cin + ci) * 9 + tap];
        const __bf16 h = (__bf16)v;
        __bf16 *wpx = (__bf16 *)d.dst;
        wpx[i] = h;
        wpx[plane + i] = (__bf16)(v - (float)h);
    } else {
        if (i >= (size_t)9 * Cin * Cout) return;
        const int co = (int)(i % Cout), ci = (int)((i / Cout) % Cin), tap = (int)(i / ((size_t)Cout * Cin));
        ((float *)d.dst)[i] = d.flip ? d.w[((size_t)ci * Cout + co) * 9 + (8 - tap)] : d.w[((size_t)co * Cin + ci) * 9 + tap];
    }
}

static int t_ws(pn_ctx *ctx, size_t bytes, void **out);
// The pack buffer of (w, shape, flip, precision): the context's scratch when the cache is off (packed by the caller on every call), else
// the cached entry -- *fresh tells the caller whether this step's refresh has already filled it.
static int t_pack_get(pn_ctx *ctx, const float *w, int Cout, int Cin, int flip, int x3, size_t bytes, hipStream_t s, void **buf, bool *fresh) {
    *fresh = false;
    if (!ctx->train_pack_cache) return t_ws(ctx, bytes, buf);
    for (auto &e : ctx->train_packs)
        if (e.w == w && e.Cout == Cout && e.Cin == Cin && e.flip == flip && e.x3 == x3) {
            *buf = e.buf; *fresh = e.fresh; e.fresh = true;           // (a stale entry is packed by the caller right now)
            return PN_OK;
        }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    if (cap != hipStreamCaptureStatusNone)
        return pn_set_error(ctx, PN_ERR_STATE, "pn_train_pack_cache: a convolution met weights that were never packed while the stream is capturing (run one eager step first)");
    pn_ctx::PackEntry e;
    e.w = w; e.Cout = Cout; e.Cin = Cin; e.flip = flip; e.x3 = x3; e.bytes = bytes; e.fresh = true; e.buf = nullptr;
    PN_HIP_CHECK(ctx, hipMalloc(&e.buf, bytes));
    ctx->train_packs.push_back(e);
    *buf = e.buf;
    return PN_OK;
}

__global__ __launch_bounds__(256, 2) void tconv3_tile_x3_kernel(TConv c, TTile g, const __bf16 *__restrict__ wpx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t_smem8[];
    unsigned char *As_hi = t_smem8, *As_lo = t_smem8 + TX_A_BYTES;          // [3 taps][64 couts][32 ch]
    unsigned char *Hs_hi = t_smem8 + 2 * TX_A_BYTES;                        // [halo pixel][32 ch]
    unsigned char *Hs_lo = Hs_hi + g.HR * g.HC * TX_PITCH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_TILE_CB(b, cblk);
    const int tx = b % g.tiles_x, ty = (b / g.tiles_x) % g.tiles_y, img = b / (g.tiles_x * g.tiles_y);
    const int y0 = ty * g.R, x0 = tx * g.TW, co0 = cblk * 64;
    const int HW = c.H * c.W, HoWo = c.Ho * c.Wo, nhalo = g.HR * g.HC;
    int hb[2], opix[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int sl = 32 * wave + 16 * n + r;
        const int ry = sl / g.TW, rx = sl - ry * g.TW;
        const bool ok = ry < g.R && y0 + ry < c.Ho && x0 + rx < c.Wo;
        hb[n] = (ok ? ry * g.HC + rx : 0) * TX_PITCH + 16 * q;
        opix[n] = ok ? (y0 + ry) * c.Wo + x0 + rx : -1;
    }
    // halo staging role: this wave stages channels 8 wave .. 8 wave + 7 of halo pixels lane + 64 i
    constexpr int NH = 5;               // ceil(320 / 64): halo tiles have at most (2 + 2) x (64 + 2) = 264 pixels
    int hoff[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int e = lane + 64 * i;
        const int hy = e / g.HC, hx = e - hy * g.HC;
        const int iy = y0 - c.pad + hy, ix = x0 - c.pad + hx;
        hoff[i] = (e < nhalo && iy >= 0 && iy < c.H && ix >= 0 && ix < c.W) ? iy * c.W + ix : -1;
    }
    const float *xb = c.x + (size_t)img * c.Cin * HW;
    // weight staging role: per kernel row 3 taps x 64 couts x 4 segments of 8 channels x 2 planes = 1536 16-byte pieces, 6 per thread
    const size_t wplane = (size_t)((c.Cin + 31) / 32) * 9 * c.Cout * 32;

    t_f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = t_f32x4{0.f, 0.f, 0.f, 0.f};

    t_u32x4 wv[6];                                     // the NEXT (chunk, kernel row) step's weight pieces of this thread
    auto load_w = [&](int wc0, int wky) {
        const __bf16 *wrow = wpx + ((size_t)(wc0 >> 5) * 9 + wky * 3) * c.Cout * 32;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int piece = t + 256 * j;                    // < 1536: plane, kx, cout, segment
            const int pl = piece / 768, rem = piece - pl * 768, kx = rem >> 8, co = (rem >> 2) & 63, seg = rem & 3;
            const int ok = (int)(co0 + co < c.Cout);
            const size_t off = ((size_t)pl * wplane + ((size_t)kx * c.Cout + (size_t)((co0 + co) & -ok)) * 32 + seg * 8);
            wv[j] = *reinterpret_cast<const t_u32x4 *>(wrow + off);       // (rows beyond Cout read row 0: zeroed when the piece is stored -- a select here would wait for the load)
        }
    };
    load_w(0, 0);

    for (int c0 = 0; c0 < c.Cin; c0 += 32) {
        const int cbase = c0 + 8 * wave;
        unsigned cmask = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cmask |= (unsigned)(cbase + j < c.Cin) << j;
        // ---- halo tile of 32 channels (all loads first, then split + 16-byte stores) ----
        float hv[NH][8];
#ifdef TX_FAKE_NOHALO                       // timing-only ablation (wrong results): no halo loads
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[i][j] = 1.f;
        if (0)
#endif
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int okp = (int)(hoff[i] >= 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[i][j] = xb[((cbase + j) * HW + hoff[i]) & -(okp & (int)((cmask >> j) & 1u))];
        }
        __syncthreads();                              // the previous chunk's last kernel row has been consumed
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int e = lane + 64 * i;
            if (e < nhalo) {
                t_bf16x8 hi, lo;
                t_split8(hv[i], hoff[i] >= 0 ? cmask : 0u, hi, lo);
                *reinterpret_cast<t_bf16x8 *>(Hs_hi + e * TX_PITCH + 16 * wave) = hi;
                *reinterpret_cast<t_bf16x8 *>(Hs_lo + e * TX_PITCH + 16 * wave) = lo;
            }
        }
        for (int ky = 0; ky < 3; ++ky) {
            // ---- weight slice of kernel row ky: [3 taps][64 couts][32 ch], straight 16-byte copies of the packed planes; fetched one
            // (chunk, kernel row) step AHEAD (round 5): the loads of the next step are in flight under this step's MFMAs instead of in
            // front of them -- three of a chunk's five exposed memory round trips gone ----
            if (ky) __syncthreads();                  // the previous kernel row's fragments have been read
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int piece = t + 256 * j;
                const int pl = piece / 768, rem = piece - pl * 768, kx = rem >> 8, co = (rem >> 2) & 63, seg = rem & 3;
                *reinterpret_cast<t_u32x4 *>((pl ? As_lo : As_hi) + (kx * 64 + co) * TX_PITCH + 16 * seg) = co0 + co < c.Cout ? wv[j] : t_u32x4{0u, 0u, 0u, 0u};
            }
            __syncthreads();
            {   // ONE call site, always taken (past the last step: the current slice again, unused): two conditional sites merged the loaded
                // registers through copies that waited for the loads right here
                const int nc0 = ky == 2 ? c0 + 32 : c0, nky = ky == 2 ? 0 : ky + 1;
                const bool more = nc0 < c.Cin;
                load_w(more ? nc0 : c0, more ? nky : ky);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int toff = (ky * g.HC + kx) * TX_PITCH;
                t_bf16x8 bh[2], bl[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    bh[n] = *reinterpret_cast<const t_bf16x8 *>(Hs_hi + hb[n] + toff);
                    bl[n] = *reinterpret_cast<const t_bf16x8 *>(Hs_lo + hb[n] + toff);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int ao = (kx * 64 + 16 * m + r) * TX_PITCH + 16 * q;
                    const t_bf16x8 ah = *reinterpret_cast<const t_bf16x8 *>(As_hi + ao);
                    const t_bf16x8 al = *reinterpret_cast<const t_bf16x8 *>(As_lo + ao);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[m][n], 0, 0, 0);
#ifndef TX_FAKE_ONEPASS                     // timing-only ablation (wrong results): one MFMA pass instead of three
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[m][n], 0, 0, 0);
#endif
                    }
                }
            }
        }
    }
    // bias of this lane's sixteen couts, gathered once (round 5: `v += c.bias[co]` inside the store loop compiled to load, s_waitcnt vmcnt(0), add --
    // sixteen dependent round trips per pixel tile, as did the accumulate form's old values, which the compiler had sunk to their uses)
    float bv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int co = co0 + 16 * (k >> 2) + 4 * q + (k & 3);
        bv[k] = c.bias ? c.bias[co < c.Cout ? co : 0] : 0.f;
    }
    T_PIN16(bv);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int pok = (int)(opix[n] >= 0);
        float *yb = c.y + (size_t)img * c.Cout * HoWo + (opix[n] & -pok);
        float old[16];
        if (c.accumulate) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + 16 * m + 4 * q + i;
                    old[4 * m + i] = yb[(co * HoWo) & -(int)(co < c.Cout)];
                }
            T_PIN16(old);                               // all sixteen loads issued, THEN used (see T_PIN16)
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * m + 4 * q + i;
                float v = acc[m][n][i];
                v += bv[4 * m + i];
                if (c.accumulate) v += old[4 * m + i];
                if (pok && co < c.Cout) yb[(size_t)co * HoWo] = v;
            }
    }
}

// Wide variant: 64 couts x 256 pixel slots per block, a wave owns 64 couts x 64 pixels (4 x 4 MFMA tiles): per tap 8 + 8 fragment
// reads feed 48 MFMAs (the 128-slot kernel above: 8 + 4 for 24) and a block's packed weight slice -- the larger part of its
// vector-memory bytes -- serves twice the pixels.  LDS images at a 64-byte pitch with the 16-byte segment XOR-swizzled by
// (row >> 2) & 3: sixteen consecutive rows x one segment cover sixteen different 16-byte bank groups, for the staging stores and
// for both operands' fragment reads, and two blocks still fit a CU (76 KB).
#define TXW2_PITCH 64
#define TXW2_A_BYTES (3 * 64 * TXW2_PITCH)
__device__ __forceinline__ int t_swz(int row, int seg) { return row * TXW2_PITCH + 16 * (seg ^ ((row >> 2) & 3)); }

__global__ __launch_bounds__(256, 2) void tconv3_tile_x3w_kernel(TConv c, TTile g, const __bf16 *__restrict__ wpx) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t_smem8[];
    unsigned char *As_hi = t_smem8, *As_lo = t_smem8 + TXW2_A_BYTES;
    unsigned char *Hs_hi = t_smem8 + 2 * TXW2_A_BYTES;
    unsigned char *Hs_lo = Hs_hi + g.HR * g.HC * TXW2_PITCH;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_TILE_CB(b, cblk);
    const int tx = b % g.tiles_x, ty = (b / g.tiles_x) % g.tiles_y, img = b / (g.tiles_x * g.tiles_y);
    const int y0 = ty * g.R, x0 = tx * g.TW, co0 = cblk * 64;
    const int HW = c.H * c.W, HoWo = c.Ho * c.Wo, nhalo = g.HR * g.HC;
    int hbp[4], opix[4];                        // halo pixel of the slot's top-left tap, output pixel (or -1)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int sl = 64 * wave + 16 * n + r;
        const int ry = sl / g.TW, rx = sl - ry * g.TW;
        const bool ok = ry < g.R && y0 + ry < c.Ho && x0 + rx < c.Wo;
        hbp[n] = ok ? ry * g.HC + rx : 0;
        opix[n] = ok ? (y0 + ry) * c.Wo + x0 + rx : -1;
    }
    constexpr int NH = 7;                       // halo tiles have at most 400 pixels (t_tile_geometry_x3w)
    int hoff[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int e = lane + 64 * i;
        const int hy = e / g.HC, hx = e - hy * g.HC;
        const int iy = y0 - c.pad + hy, ix = x0 - c.pad + hx;
        hoff[i] = (e < nhalo && iy >= 0 && iy < c.H && ix >= 0 && ix < c.W) ? iy * c.W + ix : -1;
    }
    const float *xb = c.x + (size_t)img * c.Cin * HW;
    const size_t wplane = (size_t)((c.Cin + 31) / 32) * 9 * c.Cout * 32;

    t_f32x4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = t_f32x4{0.f, 0.f, 0.f, 0.f};

    t_u32x4 wv[6];                                     // the NEXT (chunk, kernel row) step's weight pieces of this thread
    auto load_w = [&](int wc0, int wky) {
        const __bf16 *wrow = wpx + ((size_t)(wc0 >> 5) * 9 + wky * 3) * c.Cout * 32;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int piece = t + 256 * j;
            const int pl = piece / 768, rem = piece - pl * 768, kx = rem >> 8, co = (rem >> 2) & 63, seg = rem & 3;
            const int ok = (int)(co0 + co < c.Cout);
            const size_t off = ((size_t)pl * wplane + ((size_t)kx * c.Cout + (size_t)((co0 + co) & -ok)) * 32 + seg * 8);
            wv[j] = *reinterpret_cast<const t_u32x4 *>(wrow + off);       // (rows beyond Cout read row 0: zeroed when the piece is stored -- a select here would wait for the load)
        }
    };
    load_w(0, 0);

    for (int c0 = 0; c0 < c.Cin; c0 += 32) {
        const int cbase = c0 + 8 * wave;
        unsigned cmask = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) cmask |= (unsigned)(cbase + j < c.Cin) << j;
        // halo of 32 channels, in two halves of the pixel range (56 prefetch registers would not fit next to 64 accumulators)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            constexpr int I0[2] = {0, 4}, I1[2] = {4, NH};
            float hv[4][8];
#pragma unroll
            for (int i = I0[half]; i < I1[half]; ++i) {
                const int okp = (int)(hoff[i] >= 0);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#ifdef TXW2_FAKE_NOHALO                      // timing-only ablations (wrong results): -DTXW2_FAKE_NOHALO / _NOMFMA, variant builds of scripts/r05
                    hv[i - I0[half]][j] = 1.f + (float)okp;
#else
                    hv[i - I0[half]][j] = xb[((cbase + j) * HW + hoff[i]) & -(okp & (int)((cmask >> j) & 1u))];
#endif
            }
            if (half == 0) __syncthreads();       // the previous chunk's last kernel row has been consumed
#pragma unroll
            for (int i = I0[half]; i < I1[half]; ++i) {
                const int e = lane + 64 * i;
                if (e < nhalo) {
                    t_bf16x8 hi, lo;
                    t_split8(hv[i - I0[half]], hoff[i] >= 0 ? cmask : 0u, hi, lo);
                    *reinterpret_cast<t_bf16x8 *>(Hs_hi + t_swz(e, wave)) = hi;
                    *reinterpret_cast<t_bf16x8 *>(Hs_lo + t_swz(e, wave)) = lo;
                }
            }
        }
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            if (ky) __syncthreads();
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int piece = t + 256 * j;
                const int pl = piece / 768, rem = piece - pl * 768, kx = rem >> 8, co = (rem >> 2) & 63, seg = rem & 3;
                *reinterpret_cast<t_u32x4 *>((pl ? As_lo : As_hi) + t_swz(kx * 64 + co, seg)) = co0 + co < c.Cout ? wv[j] : t_u32x4{0u, 0u, 0u, 0u};
            }
            __syncthreads();
            {   // the next step's weights, in flight under this step's MFMAs (see tconv3_tile_x3_kernel)
                const int nc0 = ky == 2 ? c0 + 32 : c0, nky = ky == 2 ? 0 : ky + 1;
                const bool more = nc0 < c.Cin;
                load_w(more ? nc0 : c0, more ? nky : ky);
            }
#pragma unroll 1
            for (int kx = 0; kx < 3; ++kx) {          // not unrolled: with three taps' fragments hoisted the kernel spills (124 B / lane)
                t_bf16x8 bh[4], bl[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int o = t_swz(hbp[n] + ky * g.HC + kx, q);
                    bh[n] = *reinterpret_cast<const t_bf16x8 *>(Hs_hi + o);
                    bl[n] = *reinterpret_cast<const t_bf16x8 *>(Hs_lo + o);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int ao = t_swz(kx * 64 + 16 * m + r, q);
                    const t_bf16x8 ah = *reinterpret_cast<const t_bf16x8 *>(As_hi + ao);
                    const t_bf16x8 al = *reinterpret_cast<const t_bf16x8 *>(As_lo + ao);
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
#ifndef TXW2_FAKE_NOMFMA
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[n], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[n], acc[m][n], 0, 0, 0);
#else
                        acc[m][n][0] += (float)ah[0] * (float)bh[n][0] + (float)al[1] * (float)bl[n][1];
#endif
                    }
                }
            }
        }
    }
    // bias of this lane's sixteen couts, gathered once (round 5: `v += c.bias[co]` inside the store loop compiled to load, s_waitcnt vmcnt(0), add --
    // sixteen dependent round trips per pixel tile, as did the accumulate form's old values, which the compiler had sunk to their uses)
    float bv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int co = co0 + 16 * (k >> 2) + 4 * q + (k & 3);
        bv[k] = c.bias ? c.bias[co < c.Cout ? co : 0] : 0.f;
    }
    T_PIN16(bv);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int pok = (int)(opix[n] >= 0);
        float *yb = c.y + (size_t)img * c.Cout * HoWo + (opix[n] & -pok);
        float old[16];
        if (c.accumulate) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + 16 * m + 4 * q + i;
                    old[4 * m + i] = yb[(co * HoWo) & -(int)(co < c.Cout)];
                }
            T_PIN16(old);                               // all sixteen loads issued, THEN used (see T_PIN16)
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * m + 4 * q + i;
                float v = acc[m][n][i];
                v += bv[4 * m + i];
                if (c.accumulate) v += old[4 * m + i];
                if (pok && co < c.Cout) yb[(size_t)co * HoWo] = v;
            }
    }
}

static bool t_tile_geometry_x3w(int Ho, int Wo, int N, int Cout, TTile *g) {     // 256-slot tiles; only when they still fill the chip
    g->tiles_x = (Wo + 63) / 64;
    g->TW = (Wo + g->tiles_x - 1) / g->tiles_x;
    g->R = 256 / g->TW;
    if (g->R > Ho) g->R = Ho;
    if (g->R < 1) g->R = 1;
    g->tiles_y = (Ho + g->R - 1) / g->R;
    g->HC = g->TW + 2;
    g->HR = g->R + 2;
    g->NI = 0;
    g->CHP = 0;
    const long blocks = (long)N * g->tiles_x * g->tiles_y * ((Cout + 63) / 64);
    const long slots = (long)g->R * g->TW;
    if (getenv("POPNET_TRAIN_X3_WIDE")) return g->HR * g->HC <= 400;      // tests: the wide kernel on every shape it can hold
    return g->HR * g->HC <= 400 && blocks >= 448 && slots >= 192;
}

static bool t_tile_geometry_x3(int Ho, int Wo, TTile *g) {       // tiles of at most 64 columns: the halo tile stays under 320 pixels
    g->tiles_x = (Wo + 63) / 64;
    g->TW = (Wo + g->tiles_x - 1) / g->tiles_x;
    g->R = 128 / g->TW;
    if (g->R > Ho) g->R = Ho;
    if (g->R < 1) g->R = 1;
    g->tiles_y = (Ho + g->R - 1) / g->R;
    g->HC = g->TW + 2;
    g->HR = g->R + 2;
    g->NI = 0;
    g->CHP = 0;
    return g->HR * g->HC <= 320;
}

// Weight gradient on split-bf16 MFMA.  k = pixels, so the X operand of tap (ky, kx) is the channel-major halo row shifted by kx
// ELEMENTS -- not a 16-byte-aligned fragment.  The halo rows are laid out so that every 8-slot pixel group starts 16-byte
// aligned (slots per tile row rounded up to a multiple of 8, halo column 0 = image column x0 - 1); a lane reads the aligned
// group plus the next dword once per (ky, plane) and builds the kx = 1 fragment with four v_alignbyte and the kx = 2 one by
// renaming registers.  Block = 64 couts x (16 input channels x 9 taps), 4 pixel groups of 32 slots per tile.
struct TTileW {
    int TW, SW, R, tiles_x, tiles_y;    // live columns, slots per tile row (multiple of 8), rows
    int HP, HR;                         // halo row pitch in elements (SW + 8), halo rows (R + 2)
    int CHB;                            // bytes per halo channel
};
#define TXW_YP 272                      // bytes per dY row: 128 slots x bf16 + 16


__device__ __forceinline__ t_bf16x8 t_as_bf16x8(t_u32x4 v) {
    union { t_u32x4 u; t_bf16x8 b; } x;
    x.u = v;
    return x.b;
}

#define TXW_CI 32                       // input channels per block (two 16-column MFMA tiles per tap)
__global__ __launch_bounds__(256, 2) void tconv3_wgrad_x3_kernel(TConv c, TTileW g, float *__restrict__ partial, int tiles_per_slice, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t_smem8[];
    unsigned char *Yh = t_smem8, *Yl = t_smem8 + 64 * TXW_YP;                   // dY tile [64 couts][128 slots]
    unsigned char *Xh = t_smem8 + 2 * 64 * TXW_YP, *Xl = Xh + TXW_CI * g.CHB;   // halo [32 channels][HR][HP]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_XYZ(bx_, by_, slice);
    const int c0 = bx_ * TXW_CI, co0 = by_ * 64;
    const int HW = c.H * c.W, HoWo = c.Ho * c.Wo, nh = g.HR * g.HP;
    int hbq[4];                                   // byte offset (inside a channel's halo) of this lane's 8-slot group, per pixel group
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int s0 = 32 * pg + 8 * q, ry = s0 / g.SW, rx0 = s0 - ry * g.SW;
        hbq[pg] = (ry < g.R ? ry * g.HP + rx0 : 0) * 2;
    }
    // staging roles: dY -- slot pair (2 sp, 2 sp + 1), couts wave + 4 j;  halo -- channel t >> 3, elements (t & 7) + 8 i
    const int sp = t & 63, s_a = 2 * sp, ry_a = s_a / g.SW, rx_a = s_a - ry_a * g.SW;
    const int hk = t >> 3, he0 = t & 7;
    const int hk_ok = (int)(c0 + hk < c.Cin);
    t_f32x4 acc[2][9];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[n][k] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    const int tbeg = slice * tiles_per_slice, tend = min(tbeg + tiles_per_slice, ntiles);
    for (int tile = tbeg; tile < tend; ++tile) {
        const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, img = tile / (g.tiles_x * g.tiles_y);
        const int y0 = ty * g.R, x0 = tx * g.TW;
        const bool rowok = ry_a < g.R && y0 + ry_a < c.Ho;
        const int ok0 = (int)(rowok && rx_a < g.TW && x0 + rx_a < c.Wo), ok1 = (int)(rowok && rx_a + 1 < g.TW && x0 + rx_a + 1 < c.Wo);
        const float *dyb = c.y + (size_t)img * c.Cout * HoWo + (rowok ? (y0 + ry_a) * c.Wo + x0 + rx_a : 0);
        const float *xb = c.x + (size_t)img * c.Cin * HW + (size_t)(hk_ok ? c0 + hk : 0) * HW;
        float d0[16], d1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = co0 + wave + 4 * j;
            const int cok = (int)(co < c.Cout);
            d0[j] = dyb[(co * HoWo) & -(ok0 & cok)];
            d1[j] = dyb[(co * HoWo + 1) & -(ok1 & cok)];
        }
        __syncthreads();                          // the previous tile's fragments have been read
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = co0 + wave + 4 * j;
            const float v0 = (ok0 && co < c.Cout) ? d0[j] : 0.f, v1 = (ok1 && co < c.Cout) ? d1[j] : 0.f;
            const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
            const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
            union { __bf16 b[2]; unsigned u; } ph, pl;
            ph.b[0] = h0; ph.b[1] = h1; pl.b[0] = l0; pl.b[1] = l1;
            *reinterpret_cast<unsigned *>(Yh + (wave + 4 * j) * TXW_YP + 4 * sp) = ph.u;
            *reinterpret_cast<unsigned *>(Yl + (wave + 4 * j) * TXW_YP + 4 * sp) = pl.u;
        }
        {   // halo of channel hk, element PAIRS (2 e, 2 e + 1), e = he0 + 8 i: one 4-byte LDS store per plane and pair (2-byte stores
            // of neighbouring lanes into one bank word serialise: 64 % LDS conflict cycles in the first version); HP is even, so a
            // pair never straddles a halo row; (row, column) advance without a division; 3 pairs = 6 loads in flight (all 36
            // loads at once spilled: 22.9 instead of 17.2 ms per step)
            int hy = 0, hx = 2 * he0;
            const int iy0 = y0 - c.pad, ix0 = x0 - c.pad;
            while (hx >= g.HP) { hx -= g.HP; ++hy; }
            for (int e0 = 2 * he0; e0 < nh; e0 += 48) {
                float hv[6];
                int okv[6];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const int rowok = hk_ok & (int)(e0 + 16 * i < nh) & (int)(iy >= 0) & (int)(iy < c.H);
                    okv[2 * i] = rowok & (int)(hx < g.TW + 2) & (int)(ix >= 0) & (int)(ix < c.W);
                    okv[2 * i + 1] = rowok & (int)(hx + 1 < g.TW + 2) & (int)(ix + 1 >= 0) & (int)(ix + 1 < c.W);
                    hv[2 * i] = xb[(iy * c.W + ix) & -okv[2 * i]];
                    hv[2 * i + 1] = xb[(iy * c.W + ix + 1) & -okv[2 * i + 1]];
                    hx += 16;
                    while (hx >= g.HP) { hx -= g.HP; ++hy; }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int e = e0 + 16 * i;
                    if (e < nh) {
                        const float v0 = okv[2 * i] ? hv[2 * i] : 0.f, v1 = okv[2 * i + 1] ? hv[2 * i + 1] : 0.f;
                        const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                        union { __bf16 b[2]; unsigned u; } ph, pl;
                        ph.b[0] = h0; ph.b[1] = h1;
                        pl.b[0] = (__bf16)(v0 - (float)h0); pl.b[1] = (__bf16)(v1 - (float)h1);
                        *reinterpret_cast<unsigned *>(Xh + hk * g.CHB + 2 * e) = ph.u;
                        *reinterpret_cast<unsigned *>(Xl + hk * g.CHB + 2 * e) = pl.u;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            const int ao = (16 * wave + r) * TXW_YP + 64 * pg + 16 * q;
            const t_bf16x8 ah = *reinterpret_cast<const t_bf16x8 *>(Yh + ao), al = *reinterpret_cast<const t_bf16x8 *>(Yl + ao);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int bo = (16 * n + r) * g.CHB + hbq[pg] + ky * g.HP * 2;
                    const t_u32x4 vh = *reinterpret_cast<const t_u32x4 *>(Xh + bo), vl = *reinterpret_cast<const t_u32x4 *>(Xl + bo);
                    const unsigned nh4 = *reinterpret_cast<const unsigned *>(Xh + bo + 16), nl4 = *reinterpret_cast<const unsigned *>(Xl + bo + 16);
                    t_bf16x8 bh[3], bl[3];
                    bh[0] = t_as_bf16x8(vh);
                    bl[0] = t_as_bf16x8(vl);
                    bh[1] = t_as_bf16x8(t_u32x4{__builtin_amdgcn_alignbyte(vh[1], vh[0], 2), __builtin_amdgcn_alignbyte(vh[2], vh[1], 2),
                                                __builtin_amdgcn_alignbyte(vh[3], vh[2], 2), __builtin_amdgcn_alignbyte(nh4, vh[3], 2)});
                    bl[1] = t_as_bf16x8(t_u32x4{__builtin_amdgcn_alignbyte(vl[1], vl[0], 2), __builtin_amdgcn_alignbyte(vl[2], vl[1], 2),
                                                __builtin_amdgcn_alignbyte(vl[3], vl[2], 2), __builtin_amdgcn_alignbyte(nl4, vl[3], 2)});
                    bh[2] = t_as_bf16x8(t_u32x4{vh[1], vh[2], vh[3], nh4});
                    bl[2] = t_as_bf16x8(t_u32x4{vl[1], vl[2], vl[3], nl4});
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[kx], acc[n][ky * 3 + kx], 0, 0, 0);
                        acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[kx], acc[n][ky * 3 + kx], 0, 0, 0);
                        acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[kx], acc[n][ky * 3 + kx], 0, 0, 0);
                    }
                }
        }
    }
    float *pb = partial + (size_t)slice * c.Cout * c.Kdim;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int ci = c0 + 16 * n + r;
        if (ci >= c.Cin) continue;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * wave + 4 * q + i;
                if (co < c.Cout) pb[((size_t)co * c.Cin + ci) * 9 + tap] = acc[n][tap][i];
            }
    }
}

__global__ __launch_bounds__(512, 1) void tconv3_wgrad_x3pp_kernel(TConv c, TTileW g, float *__restrict__ partial, int tiles_per_slice, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t_smem8[];
    // two wave groups of 256 threads, each with its own LDS image set; in phase p group (p & 1) stages tile p while the other
    // group runs the MFMAs of tile p - 1: ONE block barrier per phase, staging and matrix work always overlap inside the block,
    // and the block writes ONE partial tile (the groups' accumulators are added through LDS): half the partial-sum traffic of
    // two independent 4-wave blocks per CU
    const int set_bytes = 2 * 64 * TXW_YP + 2 * TXW_CI * g.CHB;
    const int grp = threadIdx.x >> 8;
    unsigned char *sbase = t_smem8 + grp * set_bytes;
    unsigned char *Yh = sbase, *Yl = sbase + 64 * TXW_YP;                       // dY tile [64 couts][128 slots]
    unsigned char *Xh = sbase + 2 * 64 * TXW_YP, *Xl = Xh + TXW_CI * g.CHB;     // halo [32 channels][HR][HP]
    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_XYZ(bx_, by_, slice);
    const int c0 = bx_ * TXW_CI, co0 = by_ * 64;
    const int HW = c.H * c.W, HoWo = c.Ho * c.Wo, nh = g.HR * g.HP;
    int hbq[4];                                   // byte offset (inside a channel's halo) of this lane's 8-slot group, per pixel group
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int s0 = 32 * pg + 8 * q, ry = s0 / g.SW, rx0 = s0 - ry * g.SW;
        hbq[pg] = (ry < g.R ? ry * g.HP + rx0 : 0) * 2;
    }
    // staging roles: dY -- slot pair (2 sp, 2 sp + 1), couts wave + 4 j;  halo -- channel t >> 3, elements (t & 7) + 8 i
    const int sp = t & 63, s_a = 2 * sp, ry_a = s_a / g.SW, rx_a = s_a - ry_a * g.SW;
    const int hk = t >> 3, he0 = t & 7;
    const int hk_ok = (int)(c0 + hk < c.Cin);
    t_f32x4 acc[2][9];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[n][k] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    const int tbeg = slice * tiles_per_slice, tend = min(tbeg + tiles_per_slice, ntiles), ntl = tend - tbeg;
    for (int ph = 0; ph <= ntl; ++ph) {
      if ((ph & 1) == grp) {
        if (ph < ntl) {
        const int tile = tbeg + ph;
        const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, img = tile / (g.tiles_x * g.tiles_y);
        const int y0 = ty * g.R, x0 = tx * g.TW;
        const bool rowok = ry_a < g.R && y0 + ry_a < c.Ho;
        const int ok0 = (int)(rowok && rx_a < g.TW && x0 + rx_a < c.Wo), ok1 = (int)(rowok && rx_a + 1 < g.TW && x0 + rx_a + 1 < c.Wo);
        const float *dyb = c.y + (size_t)img * c.Cout * HoWo + (rowok ? (y0 + ry_a) * c.Wo + x0 + rx_a : 0);
        const float *xb = c.x + (size_t)img * c.Cin * HW + (size_t)(hk_ok ? c0 + hk : 0) * HW;
        float d0[16], d1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = co0 + wave + 4 * j;
            const int cok = (int)(co < c.Cout);
            d0[j] = dyb[(co * HoWo) & -(ok0 & cok)];
            d1[j] = dyb[(co * HoWo + 1) & -(ok1 & cok)];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = co0 + wave + 4 * j;
            const float v0 = (ok0 && co < c.Cout) ? d0[j] : 0.f, v1 = (ok1 && co < c.Cout) ? d1[j] : 0.f;
            const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
            const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
            union { __bf16 b[2]; unsigned u; } ph, pl;
            ph.b[0] = h0; ph.b[1] = h1; pl.b[0] = l0; pl.b[1] = l1;
            *reinterpret_cast<unsigned *>(Yh + (wave + 4 * j) * TXW_YP + 4 * sp) = ph.u;
            *reinterpret_cast<unsigned *>(Yl + (wave + 4 * j) * TXW_YP + 4 * sp) = pl.u;
        }
        {   // halo of channel hk, element PAIRS (2 e, 2 e + 1), e = he0 + 8 i: one 4-byte LDS store per plane and pair (2-byte stores
            // of neighbouring lanes into one bank word serialise: 64 % LDS conflict cycles in the first version); HP is even, so a
            // pair never straddles a halo row; (row, column) advance without a division; 3 pairs = 6 loads in flight (all 36
            // loads at once spilled: 22.9 instead of 17.2 ms per step)
            int hy = 0, hx = 2 * he0;
            const int iy0 = y0 - c.pad, ix0 = x0 - c.pad;
            while (hx >= g.HP) { hx -= g.HP; ++hy; }
            for (int e0 = 2 * he0; e0 < nh; e0 += 48) {
                float hv[6];
                int okv[6];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const int rowok = hk_ok & (int)(e0 + 16 * i < nh) & (int)(iy >= 0) & (int)(iy < c.H);
                    okv[2 * i] = rowok & (int)(hx < g.TW + 2) & (int)(ix >= 0) & (int)(ix < c.W);
                    okv[2 * i + 1] = rowok & (int)(hx + 1 < g.TW + 2) & (int)(ix + 1 >= 0) & (int)(ix + 1 < c.W);
                    hv[2 * i] = xb[(iy * c.W + ix) & -okv[2 * i]];
                    hv[2 * i + 1] = xb[(iy * c.W + ix + 1) & -okv[2 * i + 1]];
                    hx += 16;
                    while (hx >= g.HP) { hx -= g.HP; ++hy; }
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int e = e0 + 16 * i;
                    if (e < nh) {
                        const float v0 = okv[2 * i] ? hv[2 * i] : 0.f, v1 = okv[2 * i + 1] ? hv[2 * i + 1] : 0.f;
                        const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                        union { __bf16 b[2]; unsigned u; } ph, pl;
                        ph.b[0] = h0; ph.b[1] = h1;
                        pl.b[0] = (__bf16)(v0 - (float)h0); pl.b[1] = (__bf16)(v1 - (float)h1);
                        *reinterpret_cast<unsigned *>(Xh + hk * g.CHB + 2 * e) = ph.u;
                        *reinterpret_cast<unsigned *>(Xl + hk * g.CHB + 2 * e) = pl.u;
                    }
                }
            }
        }
        }
      } else if (ph >= 1) {
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            const int ao = (16 * wave + r) * TXW_YP + 64 * pg + 16 * q;
            const t_bf16x8 ah = *reinterpret_cast<const t_bf16x8 *>(Yh + ao), al = *reinterpret_cast<const t_bf16x8 *>(Yl + ao);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int bo = (16 * n + r) * g.CHB + hbq[pg] + ky * g.HP * 2;
                    const t_u32x4 vh = *reinterpret_cast<const t_u32x4 *>(Xh + bo), vl = *reinterpret_cast<const t_u32x4 *>(Xl + bo);
                    const unsigned nh4 = *reinterpret_cast<const unsigned *>(Xh + bo + 16), nl4 = *reinterpret_cast<const unsigned *>(Xl + bo + 16);
                    t_bf16x8 bh[3], bl[3];
                    bh[0] = t_as_bf16x8(vh);
                    bl[0] = t_as_bf16x8(vl);
                    bh[1] = t_as_bf16x8(t_u32x4{__builtin_amdgcn_alignbyte(vh[1], vh[0], 2), __builtin_amdgcn_alignbyte(vh[2], vh[1], 2),
                                                __builtin_amdgcn_alignbyte(vh[3], vh[2], 2), __builtin_amdgcn_alignbyte(nh4, vh[3], 2)});
                    bl[1] = t_as_bf16x8(t_u32x4{__builtin_amdgcn_alignbyte(vl[1], vl[0], 2), __builtin_amdgcn_alignbyte(vl[2], vl[1], 2),
                                                __builtin_amdgcn_alignbyte(vl[3], vl[2], 2), __builtin_amdgcn_alignbyte(nl4, vl[3], 2)});
                    bh[2] = t_as_bf16x8(t_u32x4{vh[1], vh[2], vh[3], nh4});
                    bl[2] = t_as_bf16x8(t_u32x4{vl[1], vl[2], vl[3], nl4});
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[kx], acc[n][ky * 3 + kx], 0, 0, 0);
                        acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[kx], acc[n][ky * 3 + kx], 0, 0, 0);
                        acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[kx], acc[n][ky * 3 + kx], 0, 0, 0);
                    }
                }
        }
      }
      __syncthreads();
    }
    // add the two groups' accumulators through LDS (72 floats per thread, the image sets are free now)
    float *scr = reinterpret_cast<float *>(t_smem8);
    if (grp == 1) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9)
#pragma unroll
                for (int i = 0; i < 4; ++i) scr[((n * 9 + k9) * 4 + i) * 256 + t] = acc[n][k9][i];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[n][k9][i] += scr[((n * 9 + k9) * 4 + i) * 256 + t];
    float *pb = partial + (size_t)slice * c.Cout * c.Kdim;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int ci = c0 + 16 * n + r;
        if (ci >= c.Cin) continue;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * wave + 4 * q + i;
                if (co < c.Cout) pb[((size_t)co * c.Cin + ci) * 9 + tap] = acc[n][tap][i];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the ping-pong weight gradient with ONE global round trip of staging per tile.  Stamps and the step's kernel stats
// (profiles/r05_train_step_*) said what bounds tconv3_wgrad_x3pp_kernel: not the matrix pipe (216 MFMAs = 3.5 k cycles per tile and
// wave) but the staging group next to it -- 32 + 30 scalar dword loads per thread, the halo ones in five dependent batches of six
// (more in flight spilled), every value split and written to LDS with 4-byte stores: ~5 memory round trips per phase against 1.6 us
// of matrix work.  Here
//   * dY never goes through LDS: a lane's A fragment IS eight consecutive pixels of its cout row (NCHW: 32 contiguous bytes, 16-byte
//     aligned when the map and tile widths are multiples of 4) -- two dwordx4 loads per pixel group, prefetched into registers by the
//     group that will multiply them in its NEXT phase, split into hi / lo right before the MFMAs;
//   * the X halo is fetched by rows: per (channel, halo row) TW / 4 aligned dwordx4 pieces + the two edge columns, <= 9 loads per
//     thread, ALL in flight at once (36 registers), each piece written with one 8-byte store per plane.  Layout per channel:
//     [16 B lead][HR rows x HP bf16], element hx of a row = image column x0 + hx, the left edge column x0 - 1 in the last slot of
//     the previous row's pitch (hx = -1): every dwordx4 piece lands 8-byte aligned, a slot group's fragment for tap kx is the
//     aligned 16-byte group shifted by kx - 1 elements (kx = 1: as read; kx = 0 / 2: five v_alignbyte with the dword before / after).
// Same tiles, same slices, same products in the same order as the x3pp kernel: bit-identical partial sums.  Shapes it does not take
// (widths that are not multiples of 4, pad != 1) stay on x3pp.
// ---------------------------------------------------------------------------------------------------------------------
#define TXV_NPI 9
__global__ __launch_bounds__(512, 1) void tconv3_wgrad_x3v_kernel(TConv c, TTileW g, float *__restrict__ partial, int tiles_per_slice, int ntiles, int ppi, int npieces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char t_smem8[];
    const int set_bytes = 2 * TXW_CI * g.CHB;
    const int grp = threadIdx.x >> 8;
    unsigned char *Xh = t_smem8 + grp * set_bytes, *Xl = Xh + TXW_CI * g.CHB;       // halo [32 channels][16 + HR x HP x 2 bytes]
    const int t = threadIdx.x & 255, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_XYZ(bx_, by_, slice);
    const int c0 = bx_ * TXW_CI, co0 = by_ * 64;
    const int HW = c.H * c.W, HoWo = c.Ho * c.Wo;
    {   // the images start as zeros: slots no piece ever writes (behind the right edge column) are READ by the fragments of the padding
        // slots, whose dY is zero -- the product must not be 0 x NaN
        t_u32x4 *z = reinterpret_cast<t_u32x4 *>(t_smem8);
        for (int i = threadIdx.x; i < 2 * set_bytes / 16; i += 512) z[i] = t_u32x4{0u, 0u, 0u, 0u};
    }
    // this lane's four 8-slot pixel groups: halo byte offset of the aligned fragment group (tap ky adds rows), dY pixel offset in the tile
    int hbq[4], aoff[4], arow[4], acol[4];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        const int s0 = 32 * pg + 8 * q, ry = s0 / g.SW, rx0 = s0 - ry * g.SW;
        const bool in = ry < g.R;
        hbq[pg] = 16 + ((in ? ry : 0) * g.HP + rx0) * 2;
        arow[pg] = in ? ry : -1;
        acol[pg] = rx0;
        aoff[pg] = (in ? ry : 0) * c.Wo + rx0;
    }
    // tile-invariant description of this thread's X pieces: piece t + 256 i = (item = (channel, halo row), pc): pc 0 = left edge column,
    // ppi - 1 = right edge column, else the dwordx4 piece of columns 4 (pc - 1) .. + 3
    int meta[TXV_NPI];                                   // kind | row << 2 | ch << 8 | (colrel + 1) << 14 (kind 3 = none); offsets are rebuilt from it per tile
#pragma unroll
    for (int i = 0; i < TXV_NPI; ++i) {
        const int pidx = t + 256 * i;
        const int item = pidx / ppi, pc = pidx - item * ppi, ch = item / g.HR, row = item - ch * g.HR;
        const int kind = pidx < npieces ? (pc == 0 ? 0 : (pc == ppi - 1 ? 2 : 1)) : 3;
        const int colrel = kind == 0 ? -1 : (kind == 2 ? g.TW : 4 * (pc - 1));
        meta[i] = kind | (row << 2) | (ch << 8) | ((colrel + 1) << 14);
    }
    t_f32x4 acc[2][9];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[n][k] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    t_f32x4 a4[4][2];                                     // this lane's dY fragments of the tile its group multiplies next
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) a4[pg][0] = a4[pg][1] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    const int tbeg = slice * tiles_per_slice, tend = min(tbeg + tiles_per_slice, ntiles), ntl = tend - tbeg;
    __syncthreads();
    for (int ph = 0; ph <= ntl; ++ph) {
      if ((ph & 1) == grp) {
#ifdef TXV_FAKE_NOSTAGE
        if (ph < 2) {
#else
        if (ph < ntl) {
#endif
            const int tile = tbeg + ph;
            const int tx = tile % g.tiles_x, ty = (tile / g.tiles_x) % g.tiles_y, img = tile / (g.tiles_x * g.tiles_y);
            const int y0 = ty * g.R, x0 = tx * g.TW;
            // ---- every load of the phase first: X pieces, then the dY fragments ----
            const float *xb = c.x + (size_t)img * c.Cin * HW + (size_t)c0 * HW + y0 * c.W + x0;
            t_f32x4 xv[TXV_NPI];
            unsigned okm = 0;
#pragma unroll
            for (int i = 0; i < TXV_NPI; ++i) {
                const int kind = meta[i] & 3, row = (meta[i] >> 2) & 63, ch = (meta[i] >> 8) & 63, colrel = (meta[i] >> 14) - 1;
                // every piece is ONE kind of load, a 16-byte group (the edge columns too: the group that holds column x0 - 1 / x0 + TW, one element
                // of it used): a dword load and a dwordx4 load into the same registers on two divergent paths made the compiler wait for each
                // piece before issuing the next -- nine dependent round trips per phase instead of one
                const int goff_i = ch * HW + (row - 1) * c.W + (kind == 0 ? -4 : colrel);
                const int iy = y0 - 1 + row;
                const int ok = (int)(kind != 3) & (int)(c0 + ch < c.Cin) & (int)(iy >= 0) & (int)(iy < c.H) &
                               (int)(kind == 0 ? x0 > 0 : (kind == 2 ? x0 + g.TW < c.W : true));
                okm |= (unsigned)ok << i;
                xv[i] = *reinterpret_cast<const t_f32x4 *>(xb + (goff_i & -ok));
            }
            const int cout = co0 + 16 * wave + r;
            const float *yb = c.y + ((size_t)img * c.Cout + (cout < c.Cout ? cout : 0)) * HoWo + y0 * c.Wo + x0;
            unsigned aok = 0;
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                const int rowok = (int)(cout < c.Cout) & (int)(arow[pg] >= 0) & (int)(y0 + arow[pg] < c.Ho);
                const int ok0 = rowok & (int)(acol[pg] + 3 < g.TW), ok1 = rowok & (int)(acol[pg] + 7 < g.TW);
                aok |= (unsigned)ok0 << (2 * pg) | (unsigned)ok1 << (2 * pg + 1);
                a4[pg][0] = *reinterpret_cast<const t_f32x4 *>(yb + (aoff[pg] & -ok0));
                a4[pg][1] = *reinterpret_cast<const t_f32x4 *>(yb + ((aoff[pg] + 4) & -ok1));
            }
            // ---- X: split and store (zeros where the piece lies outside the image / beyond Cin) ----
#pragma unroll
            for (int i = 0; i < TXV_NPI; ++i) {
                const int kind = meta[i] & 3;
                const bool ok = (okm >> i) & 1u;
                const int loff_i = ((meta[i] >> 8) & 63) * g.CHB + 16 + ((meta[i] >> 2) & 63) * g.HP * 2 + 2 * ((meta[i] >> 14) - 1);
                if (kind == 1) {
                    union { __bf16 b[4]; unsigned long long u; } ph4, pl4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = ok ? xv[i][j] : 0.f;
                        const __bf16 h = (__bf16)v;
                        ph4.b[j] = h;
                        pl4.b[j] = (__bf16)(v - (float)h);
                    }
                    *reinterpret_cast<unsigned long long *>(Xh + loff_i) = ph4.u;
                    *reinterpret_cast<unsigned long long *>(Xl + loff_i) = pl4.u;
                } else if (kind != 3) {
                    const float v = ok ? (kind == 0 ? xv[i][3] : xv[i][0]) : 0.f;
                    const __bf16 h = (__bf16)v;
                    *reinterpret_cast<__bf16 *>(Xh + loff_i) = h;
                    *reinterpret_cast<__bf16 *>(Xl + loff_i) = (__bf16)(v - (float)h);
                }
            }
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {             // the masked loads fetched element 0 of the row: zero them
                if (!((aok >> (2 * pg)) & 1u)) a4[pg][0] = t_f32x4{0.f, 0.f, 0.f, 0.f};
                if (!((aok >> (2 * pg + 1)) & 1u)) a4[pg][1] = t_f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
      } else if (ph >= 1) {
#ifndef TXV_FAKE_NOMFMA                          // timing-only ablations (wrong results): -DTXV_FAKE_NOMFMA / -DTXV_FAKE_NOSTAGE, variant builds of scripts/r05
        // 24 items (pixel group pg, channel tile n, kernel row ky), 9 MFMAs each.  Software-pipelined by hand: the six LDS reads of item
        // it + 1 are issued before the MFMAs of item it (the compiler's own schedule waited for every item's reads right before its
        // MFMAs and separated the dependent triple of an accumulator with s_nop: 45 % matrix-pipe use inside this section), and the three
        // products of a tap are interleaved across the three taps of the row, so that no MFMA reads the accumulator the previous one
        // writes.  Per accumulator the order is still hi*hi, hi*lo, lo*hi: the sums are bit-identical.
        struct Frag { t_u32x4 vh, vl; unsigned mh, ml, nh, nl; };
        auto load_frag = [&](int it) -> Frag {
            const int pg = it / 6, n = (it % 6) / 3, ky = it % 3;
            const int bo = (16 * n + r) * g.CHB + hbq[pg] + ky * g.HP * 2;
            Frag f;
            f.vh = *reinterpret_cast<const t_u32x4 *>(Xh + bo); f.vl = *reinterpret_cast<const t_u32x4 *>(Xl + bo);
            f.mh = *reinterpret_cast<const unsigned *>(Xh + bo - 4); f.ml = *reinterpret_cast<const unsigned *>(Xl + bo - 4);
            f.nh = *reinterpret_cast<const unsigned *>(Xh + bo + 16); f.nl = *reinterpret_cast<const unsigned *>(Xl + bo + 16);
            return f;
        };
        Frag fr[2];
        fr[0] = load_frag(0);
        t_bf16x8 ah, al;
#pragma unroll
        for (int it = 0; it < 24; ++it) {
            const int pg = it / 6, n = (it % 6) / 3, ky = it % 3;
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < 24) fr[(it + 1) & 1] = load_frag(it + 1);
            if (it % 6 == 0) {
                const float v8[8] = {a4[pg][0][0], a4[pg][0][1], a4[pg][0][2], a4[pg][0][3], a4[pg][1][0], a4[pg][1][1], a4[pg][1][2], a4[pg][1][3]};
                t_split8(v8, 0xffu, ah, al);
            }
            const Frag &f = fr[it & 1];
            const unsigned sh1 = __builtin_amdgcn_alignbyte(f.vh[1], f.vh[0], 2), sh2 = __builtin_amdgcn_alignbyte(f.vh[2], f.vh[1], 2), sh3 = __builtin_amdgcn_alignbyte(f.vh[3], f.vh[2], 2);
            const unsigned sl1 = __builtin_amdgcn_alignbyte(f.vl[1], f.vl[0], 2), sl2 = __builtin_amdgcn_alignbyte(f.vl[2], f.vl[1], 2), sl3 = __builtin_amdgcn_alignbyte(f.vl[3], f.vl[2], 2);
            t_bf16x8 bh[3], bl[3];
            bh[0] = t_as_bf16x8(t_u32x4{__builtin_amdgcn_alignbyte(f.vh[0], f.mh, 2), sh1, sh2, sh3});
            bl[0] = t_as_bf16x8(t_u32x4{__builtin_amdgcn_alignbyte(f.vl[0], f.ml, 2), sl1, sl2, sl3});
            bh[1] = t_as_bf16x8(f.vh);
            bl[1] = t_as_bf16x8(f.vl);
            bh[2] = t_as_bf16x8(t_u32x4{sh1, sh2, sh3, __builtin_amdgcn_alignbyte(f.nh, f.vh[3], 2)});
            bl[2] = t_as_bf16x8(t_u32x4{sl1, sl2, sl3, __builtin_amdgcn_alignbyte(f.nl, f.vl[3], 2)});
            __builtin_amdgcn_sched_barrier(0);          // (left to the compiler the fragment arithmetic lands between the MFMAs and the section is 6 % slower)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[kx], acc[n][ky * 3 + kx], 0, 0, 0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[kx], acc[n][ky * 3 + kx], 0, 0, 0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[n][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[kx], acc[n][ky * 3 + kx], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      __syncthreads();
    }
    // add the two groups' accumulators through LDS (72 floats per thread, the image sets are free now)
    float *scr = reinterpret_cast<float *>(t_smem8);
    if (grp == 1) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int k9 = 0; k9 < 9; ++k9)
#pragma unroll
                for (int i = 0; i < 4; ++i) scr[((n * 9 + k9) * 4 + i) * 256 + t] = acc[n][k9][i];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[n][k9][i] += scr[((n * 9 + k9) * 4 + i) * 256 + t];
    float *pb = partial + (size_t)slice * c.Cout * c.Kdim;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int ci = c0 + 16 * n + r;
        if (ci >= c.Cin) continue;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int co = co0 + 16 * wave + 4 * q + i;
                if (co < c.Cout) pb[((size_t)co * c.Cin + ci) * 9 + tap] = acc[n][tap][i];
            }
    }
}

// geometry of the vectorised variant: the x3pp tiles, the row layout described above; false = the shape stays on x3pp
static bool t_tile_geometry_wx3v(const TConv &c, TTileW *g, int *ppi, int *npieces) {
    if (c.pad != 1 || c.Ho != c.H || c.Wo != c.W || (c.W & 3)) return false;
    g->tiles_x = (c.Wo + 63) / 64;
    if (c.Wo % g->tiles_x) return false;
    g->TW = c.Wo / g->tiles_x;
    if (g->TW & 3) return false;
    g->SW = (g->TW + 7) / 8 * 8;
    g->R = 128 / g->SW;
    if (g->R > c.Ho) g->R = c.Ho;
    if (g->R < 1) return false;
    g->tiles_y = (c.Ho + g->R - 1) / g->R;
    g->HP = g->SW + 8;
    g->HR = g->R + 2;
    g->CHB = 16 + g->HR * g->HP * 2;
    // (channel stride 496 / 528 B: a sweep of + 0 .. 128 B found no faster mapping of the fragment reads onto the banks, a multiple of 256 B twice as slow: profiles/r05_notes.txt)
    *ppi = g->TW / 4 + 2;
    *npieces = TXW_CI * g->HR * *ppi;
    if (g->HR > 63 || *npieces > 256 * TXV_NPI) return false;
    const size_t sets = (size_t)2 * 2 * TXW_CI * g->CHB;
    return sets <= 158 * 1024;                              // (the launch asks for at least the 72 KB the accumulator hand-over at the end needs)
}

static bool t_tile_geometry_wx3(int Ho, int Wo, TTileW *g) {
    g->tiles_x = (Wo + 63) / 64;
    g->TW = (Wo + g->tiles_x - 1) / g->tiles_x;
    g->SW = (g->TW + 7) / 8 * 8;
    g->R = 128 / g->SW;
    if (g->R > Ho) g->R = Ho;
    if (g->R < 1) g->R = 1;
    g->tiles_y = (Ho + g->R - 1) / g->R;
    g->HP = g->SW + 8;
    g->HR = g->R + 2;
    g->CHB = g->HR * g->HP * 2 + 16;          // + 16: the dword after the last aligned group; 16 n + 16 keeps 16-byte alignment
    return g->SW <= 64 && 2 * 64 * TXW_YP + 2 * TXW_CI * g->CHB <= 78 * 1024;      // two blocks per CU
}

static int t_tile_lds_ok(pn_ctx *ctx) {       // the tile kernels use up to 76 KB of dynamic LDS: lift the 64 KB default once per context (= per device)
    bool &done = ctx->train_lds_attr;
    if (!done) {
        PN_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)tconv3_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PN_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)tconv3_wgrad_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PN_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)tconv3_tile_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PN_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)tconv3_wgrad_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PN_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)tconv3_tile_x3w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PN_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)tconv3_wgrad_x3pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        PN_HIP_CHECK(ctx, hipFuncSetAttribute((const void *)tconv3_wgrad_x3v_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        done = true;
    }
    return PN_OK;
}

static bool t_tile_geometry(int Ho, int Wo, int mod, TTile *g) {
    g->TW = Wo < 128 ? Wo : 128;
    g->R = 128 / g->TW;
    if (g->R > Ho) g->R = Ho;
    if (g->R < 1) g->R = 1;
    g->tiles_x = (Wo + g->TW - 1) / g->TW;
    g->tiles_y = (Ho + g->R - 1) / g->R;
    g->HC = g->TW + 2;
    g->HR = g->R + 2;
    const int n = g->HR * g->HC;
    g->NI = (n + 255) / 256;
    g->CHP = (n + 63) / 64 * 64 + mod;        // = mod (mod 64)
    if (g->CHP - 64 >= n) g->CHP -= 64;
    return g->NI <= TT_MAXNI;
}

// Weights for the data gradient: Wt[ci][(co, ky', kx')] = W[co][ci][KS-1-ky'][KS-1-kx']
__global__ void wflip_kernel(const float *__restrict__ w, float *__restrict__ wt, int Cout, int Cin, int KS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int KK = KS * KS, total = Cout * Cin * KK;
    if (i >= total) return;
    const int ci = i / (Cout * KK), rem = i - ci * Cout * KK, co = rem / KK, rr = rem - co * KK;
    wt[i] = w[((size_t)co * Cin + ci) * KK + (KK - 1 - rr)];
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient: dW[cout][k] = sum_pixels dY[cout][pixel] * X[k][pixel].  Block = 64 couts x 64 k columns over one slice of
// the pixels (grid.z slices -> partial sums, reduced in slice order by wgrad_reduce_kernel: deterministic, no atomics);
// reduction chunks of 32 pixels, lanes along the pixels for both operands.
// ---------------------------------------------------------------------------------------------------------------------
#define TW_RC 32
#define TW_P 81

template <int KS>
__global__ __launch_bounds__(256) void tconv_wgrad_kernel(TConv c, float *__restrict__ partial, int pix_per_slice) {
    __shared__ float As[TW_RC][TW_P];      // dY  [pixel][cout]
    __shared__ float Bs[TW_RC][TW_P];      // X   [pixel][k column]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    T_DECODE_XYZ(bx_, by_, slice);
    const int kc0 = bx_ * 64, co0 = by_ * 64;
    const int HoWo = c.Ho * c.Wo;
    const int pl = t & 31, g = t >> 5;
    int kci[8], kky[8], kkx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kc0 + g + 8 * j;
        if (k < c.Kdim) {
            kci[j] = k / (KS * KS);
            const int rr = k - kci[j] * (KS * KS);
            kky[j] = rr / KS;
            kkx[j] = rr - kky[j] * KS;
        } else {
            kci[j] = -1; kky[j] = 0; kkx[j] = 0;
        }
    }
    const int pbeg = slice * pix_per_slice, pend = min(pbeg + pix_per_slice, c.P);
    float ra[8], rb[8];
    unsigned okm = 0;          // bit j: rb[j] valid, bit 8 + j: ra[j] valid (the select happens when the values go to LDS)
    auto load = [&](int pc) {
        okm = 0;
        const int p = pc + pl;
        const bool ok = p < pend;
        int img = 0, rem = 0, oy = 0, ox = 0;
        if (ok) {
            img = p / HoWo;
            rem = p - img * HoWo;
            oy = rem / c.Wo;
            ox = rem - oy * c.Wo;
        }
        const float *dyb = c.y + (size_t)img * c.Cout * HoWo + rem;
        const float *xb = c.x + (size_t)img * c.Cin * c.H * c.W;
        const int iy0 = oy * c.stride - c.pad, ix0 = ox * c.stride - c.pad;
#pragma unroll
        for (int j = 0; j < 8; ++j) {            // unconditional loads + selects (see tconv_fwd_kernel)
            const int co = co0 + g + 8 * j;
            const int oka = (int)(ok & (co < c.Cout));
            ra[j] = dyb[(co * HoWo) & -oka];
            const int iy = iy0 + kky[j], ix = ix0 + kkx[j];
            const int okb = (int)(ok & (kci[j] >= 0) & (iy >= 0) & (iy < c.H) & (ix >= 0) & (ix < c.W));
            rb[j] = xb[((kci[j] * c.H + iy) * c.W + ix) & -okb];
            okm |= ((unsigned)oka << (8 + j)) | ((unsigned)okb << j);
        }
    };
    t_f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    if (pbeg < pend) load(pbeg);
    for (int pc = pbeg; pc < pend; pc += TW_RC) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            As[pl][g + 8 * j] = (okm >> (8 + j)) & 1u ? ra[j] : 0.f;
            Bs[pl][g + 8 * j] = (okm >> j) & 1u ? rb[j] : 0.f;
        }
        __syncthreads();
        if (pc + TW_RC < pend) load(pc + TW_RC);
#pragma unroll
        for (int ks = 0; ks < TW_RC / 4; ++ks) {
            const float a = As[4 * ks + q][16 * wave + r];
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[4 * ks + q][16 * n + r], acc[n], 0, 0, 0);
        }
    }
    float *pb = partial + (size_t)slice * c.Cout * c.Kdim;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int k = kc0 + 16 * n + r;
        if (k >= c.Kdim) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + 16 * wave + 4 * q + i;
            if (co < c.Cout) pb[(size_t)co * c.Kdim + k] = acc[n][i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The planes training engine's stem forward (round 6, late): model0.conv1 = 7x7 / 2, ONE input channel, 64 couts, on the same fp32 MFMA with the same
// k order as tconv_fwd_kernel<7> (k = tap, four taps per v_mfma_f32_16x16x4_f32, k-steps in order: bit-identical), without that kernel's staging: it
// gathers the [k][pixel] operand element by element for every 16-tap chunk and restages the weights per chunk (13.7 vector instructions per MFMA by
// SQ_INSTS_VALU / SQ_INSTS_MFMA, 82 us at the head of the step).  With one input channel the operand of tap (ky, kx) IS the input image shifted by
// (ky, kx): a block stages the 21 x 37 input patch of an 8 x 16 output tile once and every lane reads its B value at  base(pixel) + offset(tap)  --
// thirteen per-lane tap offsets for the whole kernel; the 64 x 49 weight matrix is 52 A-fragment registers per lane, loaded once per block, and a
// block walks tiles (two blocks per CU).  Alone 73 -> 41 us; beside the weight packs at the head of the step 82 -> 56 us.
// ---------------------------------------------------------------------------------------------------------------------
template <int F32>
__global__ __launch_bounds__(256) void tstem_fwd_kernel(TConv c, int tiles_x, int tiles_y, int ntiles) {
    constexpr int TR = 8, TC = 16, IR = (TR - 1) * 2 + 7, IC = (TC - 1) * 2 + 7;      // output tile, input patch (21 x 37)
    __shared__ float img[IR * IC];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    float a[4][13];
    int tapoff[13];
#pragma unroll
    for (int s = 0; s < 13; ++s) {
        const int k = 4 * s + q;
        const bool ok = k < 49;
        tapoff[s] = ok ? (k / 7) * IC + (k % 7) : 0;          // (a padding tap multiplies a zero weight with the patch's own first element)
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m][s] = ok ? c.w[(16 * (r >> 2) + 4 * m + (r & 3)) * 49 + k] : 0.f;      // row r of tile m = cout 16 (r >> 2) + 4 m + (r & 3): a lane's sixteen accumulators are couts 16 q ..+15
    }
    const int base0 = (2 * (2 * wave)) * IC + 2 * r, base1 = base0 + 2 * IC;      // this wave's two output rows of the tile, pixel column r
    // the patch of tile n + 1 is fetched (four values per thread, all four loads in flight) while tile n runs: a block's round trip to the input is never exposed
    // (the first form fetched and stored them one after the other at the head of the tile: four dependent round trips, 60 us for 17 us of MFMA work)
    constexpr int NS = (IR * IC + 255) / 256;
    float sv[NS];
    unsigned sok = 0;
    auto fetch = [&](int tile) {
        const int b = tile / (tiles_x * tiles_y), rem = tile - b * (tiles_x * tiles_y), ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int iy0 = ty * TR * 2 - 3, ix0 = tx * TC * 2 - 3;
        const float *xb = c.x + (size_t)b * c.H * c.W;
        sok = 0;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int i = t + 256 * u, rr = i / IC, cc = i - rr * IC, iy = iy0 + rr, ix = ix0 + cc;
            const bool ok = i < IR * IC && (unsigned)iy < (unsigned)c.H && (unsigned)ix < (unsigned)c.W;
            sv[u] = xb[ok ? iy * c.W + ix : 0];
            sok |= (unsigned)ok << u;
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = tile / (tiles_x * tiles_y), rem = tile - b * (tiles_x * tiles_y), ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int oy0 = ty * TR, ox0 = tx * TC;
        __syncthreads();                                       // the previous tile's reads are done
#pragma unroll
        for (int u = 0; u < NS; ++u)
            if (t + 256 * u < IR * IC) img[t + 256 * u] = (sok >> u) & 1u ? sv[u] : 0.f;
        __syncthreads();
        fetch(min(tile + (int)gridDim.x, ntiles - 1));         // (unconditional: a straight-line loop body keeps the compiler's wait counts exact)
        t_f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = t_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 13; ++s) {
            const float b0 = img[base0 + tapoff[s]], b1 = img[base1 + tapoff[s]];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b1, acc[m][1], 0, 0, 0);
            }
        }
        // lane holds couts 16 q + 4 m + i of pixel (row 2 wave + n, column r): 32 contiguous bytes per plane, the four q lanes of a pixel one 128-byte line
        // (with the natural row order -- couts 16 m + 4 q + i -- a line was written by four 8-byte stores per lane quartet: 76 us, as slow as the gather kernel)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int oy = oy0 + 2 * wave + n, ox = ox0 + r;
            if (oy >= c.Ho || ox >= c.Wo) continue;
            const size_t p = ((size_t)b * c.Ho + oy) * c.Wo + ox;
            float v[16];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * m + i] = acc[m][n][i] + 0.f;          // (tconv_fwd_kernel adds its zero bias: -0 becomes +0 there as well)
            if (F32) {
                float *o = (float *)c.pl + p * c.pl_cs + 16 * q;
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<t_f32x4 *>(o + 4 * j) = t_f32x4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
            } else {
                t_bf8 h[2], l[2];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const __bf16 hi = (__bf16)v[j];
                    h[j >> 3][j & 7] = hi;
                    l[j >> 3][j & 7] = (__bf16)(v[j] - (float)hi);
                }
                __bf16 *o = (__bf16 *)c.pl + p * c.pl_cs + 16 * q;
                *reinterpret_cast<t_bf8 *>(o) = h[0];
                *reinterpret_cast<t_bf8 *>(o + 8) = h[1];
                *reinterpret_cast<t_bf8 *>(o + c.pl_split) = l[0];
                *reinterpret_cast<t_bf8 *>(o + c.pl_split + 8) = l[1];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The planes training engine's stem weight gradient (round 6): tconv_wgrad_kernel above -- the same block roles for the MFMAs, the same
// slices, the same order of every sum: bit-identical partials -- with
//   * dY read from a planes tensor ([pixel][cs], two bf16 planes hi | lo or one f32 plane): thread = (pixel t >> 3, couts 8 (t & 7) ..+7),
//     one 16-byte load per plane instead of eight strided 4-byte gathers of an NCHW hand-over tensor;
//   * BN = 1: dY is not read but computed -- the stem's BatchNorm backward (trainx_kernels.h::bn_bwd_apply_kernel: g = dA * ReLU'(x * scale + shift),
//     d = k1 * (g - k2 - xhat * k3), rounded to hi + lo exactly as that kernel stores it) from the activation gradient and the convolution
//     output, so the 103 MB dC0 tensor is neither written nor read;
//   * EIGHT waves per slice instead of four: the kernel is bound by its vector instructions (pixel decomposition, halo tests, the BatchNorm arithmetic, b32
//     LDS stores: ~1 600 issue cycles per wave and chunk against 1 024 of MFMA; four chunks in flight instead of one changed nothing), and 392 slices on
//     256 CUs leave 136 CUs with two blocks: with half the staging work per wave a slice takes half as long.  Wave w accumulates cout tile w & 3 x k-column
//     tiles 2 (w >> 2), 2 (w >> 2) + 1 -- every accumulator sees the MFMAs of tconv_wgrad_kernel in the same order;
//   * D chunks in flight (template parameter; every load of a slot issued unconditionally and the prologue in slot order, or the compiler closes each
//     iteration with s_waitcnt vmcnt(0)): D = 2 is the default -- 83-91 us in the step's traces against 102-163 (D = 1, unsteady) and 105 (D = 4); the
//     launch is bound by instruction issue (~270 instructions per wave and 32-pixel chunk for 16 MFMAs; SQ_INSTS_VALU / SQ_INSTS_MFMA 12.6), not by latency.
// ---------------------------------------------------------------------------------------------------------------------
struct TStemBn {
    const void *x; int x_cs, x_split;            // the convolution output (planes, as dY)
    const float *mean, *invstd, *k1, *k2, *k3, *scale, *shift;
    int act;                                     // 0 none, 1 ReLU, 2 LeakyReLU(0.1); the sign comes from x * scale + shift
};
template <int KS, int F32, int BN, int D>
__global__ __launch_bounds__(512, 1) void tstem_wgrad_kernel(TConv c, TStemBn bn, float *__restrict__ partial, int pix_per_slice) {
    // pitch 80 floats: the four pixel rows q of an MFMA operand read sit 16 banks apart (80 % 32 = 16: lanes (q, r) of a half-wave on 32 different banks; the
    // 81 of tconv_wgrad_kernel gives every such read a two-way conflict, 47 % of this kernel's LDS cycles by SQ_LDS_BANK_CONFLICT), and a thread's four
    // staged values are one aligned 16-byte store
    constexpr int TSP = 80;
    __shared__ __attribute__((aligned(16))) float As[TW_RC][TSP];      // dY  [pixel][cout]
    __shared__ __attribute__((aligned(16))) float Bs[TW_RC][TSP];      // X   [pixel][k column]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = lane >> 4, r = lane & 15;
    const int mt = wave & 3, nh = wave >> 2;            // this wave's cout tile and pair of k-column tiles
    T_DECODE_XYZ(bx_, by_, slice);
    const int kc0 = bx_ * 64, co0 = by_ * 64;
    const int HoWo = c.Ho * c.Wo;
    const int pl = t & 31, g = t >> 5;                  // X staging: pixel, k columns 4 g + j
    // (the vector instructions bound this kernel: everything that does not change from chunk to chunk is decided here -- a k column's offset inside the
    // image and its (ky, kx); an invalid column gets ky = -2^20, which fails the halo test)
    int koff[4], kky[4], kkx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = kc0 + 4 * g + j;
        if (k < c.Kdim) {
            const int ci = k / (KS * KS), rr = k - ci * (KS * KS);
            kky[j] = rr / KS;
            kkx[j] = rr - kky[j] * KS;
            koff[j] = (ci * c.H + kky[j]) * c.W + kkx[j];
        } else {
            koff[j] = 0; kky[j] = -(1 << 20); kkx[j] = 0;
        }
    }
    const int pbeg = slice * pix_per_slice, pend = min(pbeg + pix_per_slice, c.P);
    const int a_px = t >> 4, a_c4 = 4 * (t & 15), a_co = co0 + a_c4;       // dY staging: pixel, couts a_c4 ..+3
    const bool a_cok = a_co < c.Cout;
    float bmean[4], binv[4], bk1[4], bk2[4], bk3[4], bsc[4], bsh[4];
    if (BN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ch = a_cok ? a_co + j : 0;
            bmean[j] = bn.mean[ch]; binv[j] = bn.invstd[ch]; bk1[j] = bn.k1[ch]; bk2[j] = bn.k2[ch]; bk3[j] = bn.k3[ch]; bsc[j] = bn.scale[ch]; bsh[j] = bn.shift[ch];
        }
    }
    // per prefetch slot: raw dY (and x) vectors, the gathered input values, validity bits (the selects happen when the values go to LDS)
    t_f32x4 dyv[D], xv[D];                 // f32: four floats; bf16 planes: .xy = the hi plane's 4 x bf16, .zw = the lo plane's
    float rb[D][4];
    unsigned okm[D];
    const int adv_i = TW_RC / HoWo, adv_y = (TW_RC - adv_i * HoWo) / c.Wo, adv_x = TW_RC - adv_i * HoWo - adv_y * c.Wo;
    int w_img, w_oy, w_ox;
    {
        const int p = min(pbeg + pl, c.P - 1);
        w_img = p / HoWo;
        const int rem = p - w_img * HoWo;
        w_oy = rem / c.Wo;
        w_ox = rem - w_oy * c.Wo;
    }
    auto ldplanes = [&](const void *base, size_t off, int split) -> t_f32x4 {
        if (F32) return *reinterpret_cast<const t_f32x4 *>((const float *)base + off);
        const t_f32x2 h = *reinterpret_cast<const t_f32x2 *>((const __bf16 *)base + off), l = *reinterpret_cast<const t_f32x2 *>((const __bf16 *)base + off + split);
        return t_f32x4{h[0], h[1], l[0], l[1]};
    };
    auto load = [&](int d, int pc) {
        const int pp = pc + a_px;
        const bool aok = pp < pend && a_cok;
        dyv[d] = ldplanes(c.pl, (size_t)(aok ? pp : 0) * c.pl_cs + (aok ? a_co : 0), c.pl_split);
        if (BN) xv[d] = ldplanes(bn.x, (size_t)(aok ? pp : 0) * bn.x_cs + (aok ? a_co : 0), bn.x_split);
        unsigned m = (unsigned)aok << 8;
        // this thread's pixel of the chunk: (img, oy, ox) walk along with the chunks (loads are issued in chunk order), no division per chunk
        const bool ok = pc + pl < pend;
        const int iy0 = w_oy * c.stride - c.pad, ix0 = w_ox * c.stride - c.pad;
        const float *xb = c.x + (size_t)w_img * c.Cin * c.H * c.W + (iy0 * c.W + ix0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int okb = (int)(ok & ((unsigned)(iy0 + kky[j]) < (unsigned)c.H) & ((unsigned)(ix0 + kkx[j]) < (unsigned)c.W));
            const float *src = okb ? xb + koff[j] : c.x;        // (an unconditional load + a select later, see tconv_fwd_kernel)
            rb[d][j] = *src;
            m |= (unsigned)okb << j;
        }
        okm[d] = m;
        w_ox += adv_x; w_oy += adv_y; w_img += adv_i;             // the next chunk's pixel: 32 further (one carry per digit at most)
        if (w_ox >= c.Wo) { w_ox -= c.Wo; ++w_oy; }
        if (w_oy >= c.Ho) { w_oy -= c.Ho; ++w_img; }
    };
    auto value4 = [&](const t_f32x4 raw, float (&v)[4]) {           // a planes vector as trainx_kernels.h::Lay<T>::ld reads it
        if (F32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = raw[j];
        } else {
            const t_bf4 h = __builtin_bit_cast(t_bf4, t_f32x2{raw[0], raw[1]}), l = __builtin_bit_cast(t_bf4, t_f32x2{raw[2], raw[3]});
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (float)h[j] + (float)l[j];
        }
    };
    t_f32x4 acc[2];
    acc[0] = acc[1] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    // every load of a slot is issued UNCONDITIONALLY (a chunk behind the slice's end reads dummy addresses and is never used): with the loads inside
    // `if (pc < pend)` the loop body is not straight-line and the compiler closes every iteration with s_waitcnt vmcnt(0) -- no chunk stays in flight
    // (and the prologue issues the slots IN ORDER: the scheduler had moved the loads the loop needs first to the end of the prologue, which the counter
    // of the loop's first wait then had to allow for)
#pragma unroll
    for (int d = 0; d < D; ++d) {
        load(d, pbeg + d * TW_RC);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int pc0 = pbeg; pc0 < pend; pc0 += D * TW_RC) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int pc = pc0 + d * TW_RC;
            if (pc >= pend) break;
            float a4[4];
            value4(dyv[d], a4);
            if (BN) {
                float x4[4];
                value4(xv[d], x4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float gg = a4[j];
                    if (bn.act) {
                        const float y = x4[j] * bsc[j] + bsh[j];
                        const float neg = bn.act == 2 ? 0.1f : 0.f;
                        gg = y > 0.f ? gg : gg * neg;
                    }
                    const float xh = (x4[j] - bmean[j]) * binv[j];
                    float dd = bk1[j] * (gg - bk2[j] - xh * bk3[j]);
                    if (!F32) {                                            // what Lay<bf>::st stores and Lay<bf>::ld reads back
                        const __bf16 hi = (__bf16)dd;
                        dd = (float)hi + (float)(__bf16)(dd - (float)hi);
                    }
                    a4[j] = dd;
                }
            }
            __syncthreads();
            const unsigned m = okm[d];
            {
                const bool aok = (m >> 8) & 1u;
                *reinterpret_cast<t_f32x4 *>(&As[a_px][a_c4]) = t_f32x4{aok ? a4[0] : 0.f, aok ? a4[1] : 0.f, aok ? a4[2] : 0.f, aok ? a4[3] : 0.f};
                *reinterpret_cast<t_f32x4 *>(&Bs[pl][4 * g]) = t_f32x4{m & 1u ? rb[d][0] : 0.f, m & 2u ? rb[d][1] : 0.f, m & 4u ? rb[d][2] : 0.f, m & 8u ? rb[d][3] : 0.f};
            }
            __syncthreads();
            load(d, pc + D * TW_RC);
#pragma unroll
            for (int ks = 0; ks < TW_RC / 4; ++ks) {
                const float a = As[4 * ks + q][16 * mt + r];
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bs[4 * ks + q][16 * (2 * nh + n) + r], acc[n], 0, 0, 0);
            }
        }
    }
    float *pb = partial + (size_t)slice * c.Cout * c.Kdim;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int k = kc0 + 16 * (2 * nh + n) + r;
        if (k >= c.Kdim) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + 16 * mt + 4 * q + i;
            if (co < c.Cout) pb[(size_t)co * c.Kdim + k] = acc[n][i];
        }
    }
}

template <int V>
__global__ void wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ dw, int n, int slices) {
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (i >= n) return;
    float s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0.f;
    // slice order: deterministic.  Round 5: eight slices' loads in flight, added IN ORDER (the same sums bit for bit): a 64 -> 64 layer's 128 slices were
    // 128 dependent round trips of 36 workgroups (30 us per launch, 0.45 ms per step over the small layers)
    int k = 0;
    for (; k + 8 <= slices; k += 8) {
        float p[8][V];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (V == 4) *reinterpret_cast<float4 *>(p[j]) = *reinterpret_cast<const float4 *>(partial + (size_t)(k + j) * n + i);
            else p[j][0] = partial[(size_t)(k + j) * n + i];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int v = 0; v < V; ++v) s[v] += p[j][v];
    }
    for (; k < slices; ++k) {
        float p[V];
        if (V == 4) *reinterpret_cast<float4 *>(p) = *reinterpret_cast<const float4 *>(partial + (size_t)k * n + i);
        else p[0] = partial[(size_t)k * n + i];
#pragma unroll
        for (int v = 0; v < V; ++v) s[v] += p[v];
    }
    if (V == 4) *reinterpret_cast<float4 *>(dw + i) = *reinterpret_cast<const float4 *>(s);
    else dw[i] = s[0];
}

static void t_wgrad_reduce(hipStream_t s, const float *partial, float *dw, size_t wn, int slices) {
    // (small tensors one element per thread: four times the workgroups, the same per-element sums)
    if ((wn & 3) == 0 && ((((size_t)partial) | ((size_t)dw)) & 15) == 0 && wn >= 256 * 256 * 4)
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)((wn / 4 + 255) / 256)), dim3(256), 0, s, partial, dw, (int)wn, slices);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, s, partial, dw, (int)wn, slices);
}

// ---------------------------------------------------------------------------------------------------------------------
// Reductions per channel over (N, HW): grid (C, slices) partial sums in double, then a one-block-per-channel finish.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double t_block_sum(double v, double *sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += sh[i];
    return s;      // valid on thread 0
}

// MODE 0: sum x, sum x^2 (BN statistics).  MODE 1: sum g, sum g * (x - mean) with g = dy * act'(out) (BN backward).
// MODE 2: sum dy (bias gradient).
// BatchNorm's affine map, ONE definition for the forward and for the mask the backward recomputes from x (explicit fma: the
// three kernels that evaluate it must round identically)
__device__ __forceinline__ float t_bn_affine(float x, float mean, float invstd, float gamma, float beta) {
    return __fmaf_rn((x - mean) * invstd, gamma, beta);
}

// V = 4: four consecutive pixels per thread and 16-byte loads (HW and the slice length multiples of 4); V = 1 otherwise
template <int MODE, int V>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ out,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int act, int N, int C, int HW, int slices,
                                                           double *__restrict__ partial) {
    __shared__ double sh[4];
    const int ch = blockIdx.x, sl = blockIdx.y;
    const long total = (long)N * HW;
    long per = (total + slices - 1) / slices;
    per = (per + V - 1) / V * V;
    const long beg = sl * per, end = min(beg + per, total);
    const float mu = MODE == 1 ? mean[ch] : 0.f;
    // out == nullptr (no residual in the forward): the activation mask is recomputed from x with the forward's own expression
    const bool remask = MODE == 1 && !out && act != PN_ACT_NONE;
    const float a_s = remask ? invstd[ch] : 0.f, a_g = remask ? gamma[ch] : 0.f, a_b = remask ? beta[ch] : 0.f;
    double s0 = 0.0, s1 = 0.0;
    // (image, pixel) walked without a division per element
    long i = beg + (long)threadIdx.x * V;
    int n = (int)(i / HW), p = (int)(i - (long)n * HW);
    for (; i < end; i += 256 * V) {
        const size_t off = ((size_t)n * C + ch) * HW + p;
        float xv[V], gv[V], ov[V];
        if (V == 4) {
            if (MODE != 2) *reinterpret_cast<float4 *>(xv) = *reinterpret_cast<const float4 *>(x + off);
            if (MODE != 0) *reinterpret_cast<float4 *>(gv) = *reinterpret_cast<const float4 *>(dy + off);
            if (MODE == 1 && act != PN_ACT_NONE && out) *reinterpret_cast<float4 *>(ov) = *reinterpret_cast<const float4 *>(out + off);
        } else {
            if (MODE != 2) xv[0] = x[off];
            if (MODE != 0) gv[0] = dy[off];
            if (MODE == 1 && act != PN_ACT_NONE && out) ov[0] = out[off];
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if (MODE == 0) {
                const double v = xv[k];
                s0 += v; s1 += v * v;
            } else if (MODE == 1) {
                float g = gv[k];
                if (act != PN_ACT_NONE) {
                    const float o = out ? ov[k] : t_bn_affine(xv[k], mu, a_s, a_g, a_b);
                    if (act == PN_ACT_RELU) g = o > 0.f ? g : 0.f;
                    else g = o > 0.f ? g : g * 0.1f;
                }
                s0 += g; s1 += (double)g * (double)(xv[k] - mu);
            } else {
                s0 += gv[k];
            }
        }
        p += 256 * V;
        while (p >= HW) { p -= HW; ++n; }
    }
    const double r0 = t_block_sum(s0, sh);
    const double r1 = t_block_sum(s1, sh);
    if (threadIdx.x == 0) {
        partial[((size_t)ch * slices + sl) * 2] = r0;
        partial[((size_t)ch * slices + sl) * 2 + 1] = r1;
    }
}

// launches the reduction with 16-byte loads when the layout allows it
template <int MODE>
static void t_chan_reduce(hipStream_t s, const float *x, const float *dy, const float *out, const float *mean, const float *invstd, const float *gamma,
                          const float *beta, int act, int N, int C, int HW, int slices, double *partial) {
    const bool v4 = (HW & 3) == 0 && ((((size_t)x) | ((size_t)dy) | ((size_t)out)) & 15) == 0;
    if (v4)
        hipLaunchKernelGGL((chan_reduce_kernel<MODE, 4>), dim3((unsigned)C, (unsigned)slices), dim3(256), 0, s, x, dy, out, mean, invstd, gamma, beta, act, N, C, HW, slices, partial);
    else
        hipLaunchKernelGGL((chan_reduce_kernel<MODE, 1>), dim3((unsigned)C, (unsigned)slices), dim3(256), 0, s, x, dy, out, mean, invstd, gamma, beta, act, N, C, HW, slices, partial);
}

// sum of a channel's `slices` partial pairs in slice order, eight pairs' loads in flight (the plain loop waited for every pair: a dependent chain of
// up to 256 round trips in front of every block of the apply kernels); same additions in the same order
__device__ __forceinline__ void t_sum_partials(const double *__restrict__ p, int slices, double &s, double &ss) {
    int k = 0;
    for (; k + 8 <= slices; k += 8) {
        double a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = p[(k + j) * 2]; b[j] = p[(k + j) * 2 + 1]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += a[j]; ss += b[j]; }
    }
    for (; k < slices; ++k) { s += p[k * 2]; ss += p[k * 2 + 1]; }
}

__global__ void sums_finish_kernel(const double *__restrict__ partial, int C, int slices, float *__restrict__ out0, float *__restrict__ out1,
                                   const float *__restrict__ invstd) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C) return;
    double s = 0.0, ss = 0.0;
    t_sum_partials(partial + (size_t)ch * slices * 2, slices, s, ss);
    if (out0) out0[ch] = (float)s;                                     // d beta / d bias
    if (out1) out1[ch] = (float)(invstd ? ss * (double)invstd[ch] : ss);   // d gamma = sum g (x - mean) * invstd
}

// V = 4: four consecutive pixels per thread as one 16-byte access (HW a multiple of 4: every map of the network); V = 1 otherwise
// Round 4: the statistics are FINISHED here (bn_stats_finish_kernel's arithmetic, same order, in every thread of the block: `slices` <= 256
// partial pairs of the block's channel) instead of in a launch of their own -- 33 launches of one 64-thread block each per training step less;
// the block of image 0 / pixel group 0 publishes mean / invstd for the backward pass and updates the running statistics.
template <int V>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                                const double *__restrict__ partial, int slices, double count, float eps, float momentum,
                                float *__restrict__ save_mean, float *__restrict__ save_invstd, float *__restrict__ running_mean, float *__restrict__ running_var,
                                const float *__restrict__ res, int act, int C, int HW, float *__restrict__ y) {
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int ch = blockIdx.y % C;
    double s = 0.0, ss = 0.0;
    t_sum_partials(partial + (size_t)ch * slices * 2, slices, s, ss);
    const double dmean = s / count;
    double var = ss / count - dmean * dmean;
    if (var < 0.0) var = 0.0;
    const float mu = (float)dmean, is = (float)(1.0 / sqrt(var + (double)eps));
    if (blockIdx.x == 0 && (int)blockIdx.y == ch && threadIdx.x == 0) {
        save_mean[ch] = mu;
        save_invstd[ch] = is;
        if (running_mean) running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * dmean);
        if (running_var) running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * var * (count / (count - 1.0)));
    }
    if (p >= HW) return;
    const size_t i = (size_t)blockIdx.y * HW + p;
    const float ga = gamma[ch], be = beta[ch];
    float xv[V], rv[V], o[V];
    if (V == 4) {
        *reinterpret_cast<float4 *>(xv) = *reinterpret_cast<const float4 *>(x + i);
        if (res) *reinterpret_cast<float4 *>(rv) = *reinterpret_cast<const float4 *>(res + i);
    } else {
        xv[0] = x[i];
        if (res) rv[0] = res[i];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float v = t_bn_affine(xv[k], mu, is, ga, be);
        if (res) v += rv[k];
        if (act == PN_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (act == PN_ACT_LEAKY) v = v > 0.f ? v : v * 0.1f;
        o[k] = v;
    }
    if (V == 4) *reinterpret_cast<float4 *>(y + i) = *reinterpret_cast<const float4 *>(o);
    else y[i] = o[0];
}

// dx = gamma * invstd * (g - sum_g / n - (x - mean) * invstd^2 * sum_gx / n);  dres (+)= g   (identity path of a BasicBlock)
// grid (ceil(HW / (256 V)), N * C): the channel is uniform per block, no division per element
// Round 4: the two channel sums are finished here (bn_bwd_finish_kernel's arithmetic: slice-order double sums rounded to float) by every
// thread of the block; the block of image 0 / pixel group 0 writes d beta and d gamma.
template <int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ out,
                                    const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ invstd,
                                    const double *__restrict__ partial, int slices, float *__restrict__ dbeta, float *__restrict__ dgamma,
                                    int act, int C, int HW, float inv_count,
                                    float *__restrict__ dx, float *__restrict__ dres, int dres_accumulate) {
    const int p = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int ch = blockIdx.y % C;
    double s = 0.0, ss = 0.0;
    t_sum_partials(partial + (size_t)ch * slices * 2, slices, s, ss);
    const float sum_g = (float)s, sum_gx = (float)ss;
    if (blockIdx.x == 0 && (int)blockIdx.y == ch && threadIdx.x == 0) {
        dbeta[ch] = sum_g;
        dgamma[ch] = (float)(ss * (double)invstd[ch]);
    }
    if (p >= HW) return;
    const size_t i = (size_t)blockIdx.y * HW + p;
    const float is = invstd[ch], mu = mean[ch], ga = gamma[ch];
    const float be = (act != PN_ACT_NONE && !out) ? beta[ch] : 0.f;
    const float mg = sum_g * inv_count, k2 = sum_gx * inv_count * is * is;
    float xv[V], gv[V], ov[V], dv[V], rv[V];
    if (V == 4) {
        *reinterpret_cast<float4 *>(xv) = *reinterpret_cast<const float4 *>(x + i);
        *reinterpret_cast<float4 *>(gv) = *reinterpret_cast<const float4 *>(dy + i);
        if (act != PN_ACT_NONE && out) *reinterpret_cast<float4 *>(ov) = *reinterpret_cast<const float4 *>(out + i);
        if (dres && dres_accumulate) *reinterpret_cast<float4 *>(rv) = *reinterpret_cast<const float4 *>(dres + i);
    } else {
        xv[0] = x[i];
        gv[0] = dy[i];
        if (act != PN_ACT_NONE && out) ov[0] = out[i];
        if (dres && dres_accumulate) rv[0] = dres[i];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
        float g = gv[k];
        if (act != PN_ACT_NONE) {
            const float o = out ? ov[k] : t_bn_affine(xv[k], mu, is, ga, be);      // the forward's own expression (bn_apply_kernel)
            if (act == PN_ACT_RELU) g = o > 0.f ? g : 0.f;
            else g = o > 0.f ? g : g * 0.1f;
        }
        dv[k] = (g - mg - (xv[k] - mu) * k2) * is * ga;
        gv[k] = (dres && dres_accumulate) ? rv[k] + g : g;
    }
    if (V == 4) {
        *reinterpret_cast<float4 *>(dx + i) = *reinterpret_cast<const float4 *>(dv);
        if (dres) *reinterpret_cast<float4 *>(dres + i) = *reinterpret_cast<const float4 *>(gv);
    } else {
        dx[i] = dv[0];
        if (dres) dres[i] = gv[0];
    }
}

// AvgPool2d(3, stride 2, padding 1), count_include_pad = True
// (grid.y = plane, 32-bit pixel arithmetic: the flat 64-bit index cost three 64-bit divisions per element -- round 5)
__global__ void avgpool_fwd_kernel(const float *__restrict__ x, int H, int W, int Ho, int Wo, size_t total, float *__restrict__ y) {
    const int pp = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (pp >= Ho * Wo) return;
    const int oy = pp / Wo, ox = pp - oy * Wo;
    const size_t planes = total / ((size_t)Ho * Wo);
    for (size_t plane = blockIdx.y; plane < planes; plane += gridDim.y) {       // grid.y = min(planes, 65535): any plane count
        const size_t i = plane * Ho * Wo + pp;
        const float *xp = x + plane * H * W;
        float s = 0.f;
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) s += xp[(size_t)iy * W + ix];
            }
        y[i] = s / 9.f;
    }
}

__global__ void avgpool_bwd_kernel(const float *__restrict__ dy, int H, int W, int Ho, int Wo, size_t total, float *__restrict__ dx) {
    const int pp = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (pp >= H * W) return;
    const int iy = pp / W, ix = pp - iy * W;
    const size_t planes = total / ((size_t)H * W);
    for (size_t plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const size_t i = plane * H * W + pp;
        const float *dp = dy + plane * Ho * Wo;
        float s = 0.f;
        for (int oy = (iy >> 1); oy <= ((iy + 1) >> 1); ++oy)         // outputs whose window [2oy-1, 2oy+1] holds iy
            for (int ox = (ix >> 1); ox <= ((ix + 1) >> 1); ++ox)
                if (oy < Ho && ox < Wo) s += dp[(size_t)oy * Wo + ox];
        dx[i] = s / 9.f;
    }
}

// Heads: s = sigmoid(v); out = kind ? (s - 0.5) * 4 : s  (rtpose_light3d.py:335-337), written into a channel slice of the
// stage-2 input when out_ld > C (torch.cat, :339); loss partials sum w (out - t)^2 per block.
__global__ __launch_bounds__(256) void head_fwd_kernel(const float *__restrict__ v, const float *__restrict__ target, const float *__restrict__ fg, int kind,
                                                       int C, int HW, size_t total, float *__restrict__ s_out, float *__restrict__ out, int out_ld,
                                                       double *__restrict__ partial) {
    __shared__ double sh[4];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0;
    if (i < total) {
        const float s = 1.f / (1.f + expf(-v[i]));
        const float o = kind ? (s - 0.5f) * 4.f : s;
        s_out[i] = s;
        const size_t n = i / ((size_t)C * HW), rem = i - n * (size_t)C * HW;
        out[n * (size_t)out_ld * HW + rem] = o;
        const float d = o - target[i];
        const float w = fg ? 0.1f + fg[i] * 0.9f : 1.f;
        e = (double)(d * d * w);
    }
    const double r = t_block_sum(e, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void loss_finish_kernel(const double *__restrict__ partial, int nblocks, double numel, float *__restrict__ loss) {
    __shared__ double sh[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    const double r = t_block_sum(s, sh);
    if (threadIdx.x == 0) *loss = (float)(r / numel);
}

// dv = (2 w (out - t) / numel + dextra) * (kind ? 4 : 1) * s (1 - s)
__global__ void head_bwd_kernel(const float *__restrict__ s_in, const float *__restrict__ target, const float *__restrict__ fg, const float *__restrict__ dextra,
                                int dextra_ld, int kind, int C, int HW, size_t total, float inv_numel, float *__restrict__ dv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float s = s_in[i];
    const float o = kind ? (s - 0.5f) * 4.f : s;
    const float w = fg ? 0.1f + fg[i] * 0.9f : 1.f;
    float g = 2.f * (o - target[i]) * w * inv_numel;
    if (dextra) {
        const size_t n = i / ((size_t)C * HW), rem = i - n * (size_t)C * HW;
        g += dextra[n * (size_t)dextra_ld * HW + rem];
    }
    if (kind) g *= 4.f;
    dv[i] = g * (1.f - s) * s;
}

// torch.optim.SGD(momentum, nesterov=True, dampening 0, weight decay wd): buf = first ? g : mu buf + g; p -= lr (g + mu buf)
__global__ void sgd_nesterov_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ buf, size_t n, float lr, float mu, float wd,
                                    int first, float gscale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float d = g[i] * gscale;
    if (wd != 0.f) d += wd * p[i];
    const float b = first ? d : mu * buf[i] + d;
    buf[i] = b;
    p[i] -= lr * (d + mu * b);
}

// copy a [N, C, HW] tensor into / out of a channel slice of a [N, ld, HW] tensor (torch.cat and its gradient)
__global__ void slice_copy_kernel(const float *__restrict__ src, int src_ld, float *__restrict__ dst, int dst_ld, int C, int HW, size_t total, int accumulate) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / ((size_t)C * HW), rem = i - n * (size_t)C * HW;
    const float v = src[n * (size_t)src_ld * HW + rem];
    float *d = dst + n * (size_t)dst_ld * HW + rem;
    *d = accumulate ? *d + v : v;
}

static int t_ws(pn_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->train_ws_bytes) {
        if (ctx->train_ws) {
            if (ctx->train_ws_keep) {
                // pn_train_ws_keep: a captured hipGraph still points into the old block -- retire it (freed with the ctx)
                // instead of freeing it; the graph keeps replaying on the old block, eager calls use the new one
                ctx->train_ws_retired.push_back(ctx->train_ws);
            } else {
                PN_HIP_CHECK(ctx, hipDeviceSynchronize());
                (void)hipFree(ctx->train_ws);
            }
            ctx->train_ws = nullptr;
            ctx->train_ws_bytes = 0;
        }
        const size_t want = bytes + bytes / 4;
        PN_HIP_CHECK(ctx, hipMalloc(&ctx->train_ws, want));
        ctx->train_ws_bytes = want;
    }
    *out = ctx->train_ws;
    return PN_OK;
}

static int t_slices(long total, int C) {       // slices per channel so that the reduction fills the chip
    long s = (1024 + C - 1) / C;
    const long cap = (total + 4095) / 4096;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

#define T_CTX_CHECK(name)                                                                                   \
    if (!ctx) return PN_ERR_INVALID;                                                                        \
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");

extern "C" {

int pn_train_ws_keep(pn_ctx *ctx, int keep) {
    T_CTX_CHECK("pn_train_ws_keep")
    ctx->train_ws_keep = keep != 0;
    return PN_OK;
}

int pn_train_pack_cache(pn_ctx *ctx, int enable) {
    T_CTX_CHECK("pn_train_pack_cache")
    if (!enable && ctx->train_pack_cache) {
        PN_HIP_CHECK(ctx, hipDeviceSynchronize());
        for (auto &e : ctx->train_packs) (void)hipFree(e.buf);
        ctx->train_packs.clear();
        if (ctx->train_pack_table) (void)hipFree(ctx->train_pack_table);
        ctx->train_pack_table = nullptr;
        ctx->train_pack_table_entries = 0;
        ctx->train_pack_table_cap = 0;
    }
    ctx->train_pack_cache = enable != 0;
    return PN_OK;
}

int pn_train_pack_refresh(pn_ctx *ctx, void *hip_stream) {
    T_CTX_CHECK("pn_train_pack_refresh")
    if (!ctx->train_pack_cache || ctx->train_packs.empty()) return PN_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    const size_t n = ctx->train_packs.size();
    if (ctx->train_pack_table_entries != n) {
        // the list has grown since the table was uploaded (the engine's first steps): rebuild it -- never under a capture
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cap);
        if (cap != hipStreamCaptureStatusNone)
            return pn_set_error(ctx, PN_ERR_STATE, "pn_train_pack_refresh: the pack list changed since the last eager step; run one eager step before capturing");
        std::vector<TPackDesc> host(n);
        unsigned blocks = 0;
        for (size_t i = 0; i < n; ++i) {
            const auto &e = ctx->train_packs[i];
            const size_t items = e.x3 ? (size_t)((e.Cin + 31) / 32) * 9 * e.Cout * 32 : (size_t)9 * e.Cin * e.Cout;
            host[i] = TPackDesc{e.w, e.buf, e.Cout, e.Cin, e.flip, e.x3, blocks, 0u};
            blocks += (unsigned)((items + 255) / 256);
        }
        PN_HIP_CHECK(ctx, hipStreamSynchronize(s));
        // The table is APPEND-ONLY in a fixed-capacity block: a captured step graph bakes (table pointer, entry count, grid) into its
        // wpack_all_kernel node, and an eager step of another resolution after capture() adds (w, Cout, Cin, flip, x3) keys.  Entries
        // 0 .. n_old - 1 keep their bytes and their block offsets, so the replayed node stays valid (the table used to be
        // freed and reallocated here -- a replay then read freed memory).  Beyond the capacity the old block is retired, not freed,
        // while a graph may point at it (pn_train_ws_keep), exactly like train_ws.
        if (n > ctx->train_pack_table_cap) {
            size_t cap = std::max<size_t>(256, ctx->train_pack_table_cap * 2);
            while (cap < n) cap *= 2;
            if (ctx->train_pack_table) {
                if (ctx->train_ws_keep) ctx->train_ws_retired.push_back(ctx->train_pack_table);
                else (void)hipFree(ctx->train_pack_table);
                ctx->train_pack_table = nullptr;
            }
            PN_HIP_CHECK(ctx, hipMalloc(&ctx->train_pack_table, cap * sizeof(TPackDesc)));
            ctx->train_pack_table_cap = cap;
        }
        PN_HIP_CHECK(ctx, hipMemcpy(ctx->train_pack_table, host.data(), n * sizeof(TPackDesc), hipMemcpyHostToDevice));
        ctx->train_pack_table_entries = n;
        ctx->train_pack_blocks = blocks;
    }
    hipLaunchKernelGGL(wpack_all_kernel, dim3(ctx->train_pack_blocks), dim3(256), 0, s, (const TPackDesc *)ctx->train_pack_table, (int)n);
    PN_HIP_CHECK(ctx, hipGetLastError());
    for (auto &e : ctx->train_packs) e.fresh = true;
    return PN_OK;
}

int pn_train_set_precision(pn_ctx *ctx, int precision) {
    T_CTX_CHECK("pn_train_set_precision")
    if (precision != PN_PREC_F32 && precision != PN_PREC_BF16X3)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_train_set_precision: PN_PREC_F32 or PN_PREC_BF16X3");
    ctx->train_x3 = precision == PN_PREC_BF16X3;
    return PN_OK;
}

int pn_conv2d_forward(pn_ctx *ctx, const float *x_dev, const float *w_dev, const float *bias_dev, float *y_dev, int N, int Cin, int H, int W,
                      int Cout, int ks, int stride, int pad, int accumulate, void *hip_stream) {
    T_CTX_CHECK("pn_conv2d_forward")
    if (!x_dev || !w_dev || !y_dev || N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || stride < 1 || pad < 0 || (ks != 1 && ks != 3 && ks != 7))
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_conv2d_forward: bad arguments (kernel sizes 1, 3, 7)");
    TConv c;
    c.x = x_dev; c.w = w_dev; c.bias = bias_dev; c.y = y_dev;
    c.N = N; c.Cin = Cin; c.H = H; c.W = W; c.Cout = Cout; c.stride = stride; c.pad = pad; c.accumulate = accumulate;
    c.Ho = (H + 2 * pad - ks) / stride + 1;
    c.Wo = (W + 2 * pad - ks) / stride + 1;
    c.Kdim = Cin * ks * ks;
    const long P = (long)N * c.Ho * c.Wo;
    if (c.Ho < 1 || c.Wo < 1 || P > 0x7fffffffL)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_conv2d_forward: size out of range");
    c.P = (int)P;
    hipStream_t s = (hipStream_t)hip_stream;
    TTile g;
    if (ks == 3 && stride == 1 && pad <= 2 && Cin >= 16 && t_tile_geometry(c.Ho, c.Wo, 16, &g)) {
        // second-generation 3x3 kernel: weights to [tap][ci][cout] in the scratch, then halo tiles
        const size_t wn = (size_t)Cout * Cin * 9;
        const size_t wx = (size_t)((Cin + 31) / 32) * 9 * Cout * 32;        // elements per plane of the split-bf16 pack
        void *ws = nullptr;
        bool fresh = false;
        int rc;
        if ((rc = t_tile_lds_ok(ctx)) != PN_OK) return rc;
        TTile gx;
        if (ctx->train_x3 && Cin >= 32 && t_tile_geometry_x3(c.Ho, c.Wo, &gx)) {
            if ((rc = t_pack_get(ctx, w_dev, Cout, Cin, 0, 1, 4 * wx, s, &ws, &fresh)) != PN_OK) return rc;
            if (!fresh) hipLaunchKernelGGL(wpack3_x3_kernel, dim3((unsigned)((wx + 255) / 256)), dim3(256), 0, s, w_dev, (__bf16 *)ws, Cout, Cin, 0);
            TTile gw2;
            if (!getenv("POPNET_TRAIN_X3_NARROW") && t_tile_geometry_x3w(c.Ho, c.Wo, N, Cout, &gw2)) {
                const size_t ldsw2 = (size_t)2 * TXW2_A_BYTES + (size_t)2 * gw2.HR * gw2.HC * TXW2_PITCH;
                hipLaunchKernelGGL(tconv3_tile_x3w_kernel, dim3((unsigned)(N * gw2.tiles_x * gw2.tiles_y), (unsigned)((Cout + 63) / 64)), dim3(256), ldsw2, s, c, gw2, (const __bf16 *)ws);
                PN_HIP_CHECK(ctx, hipGetLastError());
                return PN_OK;
            }
            const size_t ldsx = (size_t)2 * TX_A_BYTES + (size_t)2 * gx.HR * gx.HC * TX_PITCH;
            hipLaunchKernelGGL(tconv3_tile_x3_kernel, dim3((unsigned)(N * gx.tiles_x * gx.tiles_y), (unsigned)((Cout + 63) / 64)), dim3(256), ldsx, s, c, gx, (const __bf16 *)ws);
            PN_HIP_CHECK(ctx, hipGetLastError());
            return PN_OK;
        }
        if ((rc = t_pack_get(ctx, w_dev, Cout, Cin, 0, 0, wn * sizeof(float), s, &ws, &fresh)) != PN_OK) return rc;
        if (!fresh) hipLaunchKernelGGL(wpack3_kernel, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, s, w_dev, (float *)ws, Cout, Cin, 0);
        const size_t lds = (size_t)(144 * TT_AP + 16 * g.CHP) * sizeof(float);
        hipLaunchKernelGGL(tconv3_tile_kernel, dim3((unsigned)(N * g.tiles_x * g.tiles_y), (unsigned)((Cout + 63) / 64)), dim3(256), lds, s, c, g, (const float *)ws);
        PN_HIP_CHECK(ctx, hipGetLastError());
        return PN_OK;
    }
    dim3 grid((unsigned)((P + 127) / 128), (unsigned)((Cout + 63) / 64)), block(256);
    if (ks == 1) hipLaunchKernelGGL(tconv_fwd_kernel<1>, grid, block, 0, s, c);
    else if (ks == 3) hipLaunchKernelGGL(tconv_fwd_kernel<3>, grid, block, 0, s, c);
    else hipLaunchKernelGGL(tconv_fwd_kernel<7>, grid, block, 0, s, c);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_conv2d_dgrad(pn_ctx *ctx, const float *dy_dev, const float *w_dev, float *dx_dev, int N, int Cin, int H, int W, int Cout, int ks, int pad,
                    int accumulate, void *hip_stream) {
    T_CTX_CHECK("pn_conv2d_dgrad")
    if (!dy_dev || !w_dev || !dx_dev || (ks != 1 && ks != 3 && ks != 7) || pad > ks - 1)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_conv2d_dgrad: bad arguments (stride 1 only)");
    const int Ho = H + 2 * pad - ks + 1, Wo = W + 2 * pad - ks + 1;
    const size_t wn = (size_t)Cout * Cin * ks * ks;
    const size_t wx = (size_t)((Cout + 31) / 32) * 9 * Cin * 32;          // elements per plane of the split-bf16 pack (conv input channels = Cout)
    void *ws = nullptr;
    bool fresh = false;
    int rc;
    hipStream_t s = (hipStream_t)hip_stream;
    TTile g;
    if (ks == 3 && Cout >= 16 && t_tile_geometry(H, W, 16, &g)) {
        // dX = conv(dY, rotated transposed weights, padding 2 - pad) on the halo-tile kernel: the packing IS the rotation
        TConv c;
        c.x = dy_dev; c.w = nullptr; c.bias = nullptr; c.y = dx_dev;
        c.N = N; c.Cin = Cout; c.H = Ho; c.W = Wo; c.Cout = Cin; c.stride = 1; c.pad = 2 - pad; c.accumulate = accumulate;
        c.Ho = H; c.Wo = W; c.Kdim = Cout * 9; c.P = N * H * W;
        if ((rc = t_tile_lds_ok(ctx)) != PN_OK) return rc;
        TTile gx;
        if (ctx->train_x3 && Cout >= 32 && t_tile_geometry_x3(H, W, &gx)) {
            if ((rc = t_pack_get(ctx, w_dev, Cin, Cout, 1, 1, 4 * wx, s, &ws, &fresh)) != PN_OK) return rc;
            if (!fresh) hipLaunchKernelGGL(wpack3_x3_kernel, dim3((unsigned)((wx + 255) / 256)), dim3(256), 0, s, w_dev, (__bf16 *)ws, Cin, Cout, 1);
            TTile gw2;
            if (!getenv("POPNET_TRAIN_X3_NARROW") && t_tile_geometry_x3w(H, W, N, Cin, &gw2)) {
                const size_t ldsw2 = (size_t)2 * TXW2_A_BYTES + (size_t)2 * gw2.HR * gw2.HC * TXW2_PITCH;
                hipLaunchKernelGGL(tconv3_tile_x3w_kernel, dim3((unsigned)(N * gw2.tiles_x * gw2.tiles_y), (unsigned)((Cin + 63) / 64)), dim3(256), ldsw2, s, c, gw2, (const __bf16 *)ws);
                PN_HIP_CHECK(ctx, hipGetLastError());
                return PN_OK;
            }
            const size_t ldsx = (size_t)2 * TX_A_BYTES + (size_t)2 * gx.HR * gx.HC * TX_PITCH;
            hipLaunchKernelGGL(tconv3_tile_x3_kernel, dim3((unsigned)(N * gx.tiles_x * gx.tiles_y), (unsigned)((Cin + 63) / 64)), dim3(256), ldsx, s, c, gx, (const __bf16 *)ws);
            PN_HIP_CHECK(ctx, hipGetLastError());
            return PN_OK;
        }
        if ((rc = t_pack_get(ctx, w_dev, Cin, Cout, 1, 0, wn * sizeof(float), s, &ws, &fresh)) != PN_OK) return rc;
        if (!fresh) hipLaunchKernelGGL(wpack3_kernel, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, s, w_dev, (float *)ws, Cin, Cout, 1);
        const size_t lds = (size_t)(144 * TT_AP + 16 * g.CHP) * sizeof(float);
        hipLaunchKernelGGL(tconv3_tile_kernel, dim3((unsigned)(N * g.tiles_x * g.tiles_y), (unsigned)((Cin + 63) / 64)), dim3(256), lds, s, c, g, (const float *)ws);
        PN_HIP_CHECK(ctx, hipGetLastError());
        return PN_OK;
    }
    if ((rc = t_ws(ctx, wn * sizeof(float), &ws)) != PN_OK) return rc;
    hipLaunchKernelGGL(wflip_kernel, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, s, w_dev, (float *)ws, Cout, Cin, ks);
    // dX = conv(dY [N, Cout, Ho, Wo], Wt [Cin, Cout, ks, ks], padding ks - 1 - pad)
    return pn_conv2d_forward(ctx, dy_dev, (const float *)ws, nullptr, dx_dev, N, Cout, Ho, Wo, Cin, ks, 1, ks - 1 - pad, accumulate, hip_stream);
}

int pn_conv2d_wgrad(pn_ctx *ctx, const float *x_dev, const float *dy_dev, float *dw_dev, float *dbias_dev, int N, int Cin, int H, int W, int Cout,
                    int ks, int stride, int pad, void *hip_stream) {
    T_CTX_CHECK("pn_conv2d_wgrad")
    if (!x_dev || !dy_dev || !dw_dev || N < 1 || (ks != 1 && ks != 3 && ks != 7) || stride < 1)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_conv2d_wgrad: bad arguments");
    TConv c;
    c.x = x_dev; c.w = nullptr; c.bias = nullptr; c.y = (float *)dy_dev;
    c.N = N; c.Cin = Cin; c.H = H; c.W = W; c.Cout = Cout; c.stride = stride; c.pad = pad; c.accumulate = 0;
    c.Ho = (H + 2 * pad - ks) / stride + 1;
    c.Wo = (W + 2 * pad - ks) / stride + 1;
    c.Kdim = Cin * ks * ks;
    const long P = (long)N * c.Ho * c.Wo;
    if (c.Ho < 1 || c.Wo < 1 || P > 0x7fffffffL) return pn_set_error(ctx, PN_ERR_INVALID, "pn_conv2d_wgrad: size out of range");
    c.P = (int)P;
    const int csl0 = t_slices(P, Cout);
    TTile g;
    if (ks == 3 && stride == 1 && pad <= 2 && Cin >= 16 && t_tile_geometry(c.Ho, c.Wo, 4, &g)) {
        const int ntiles = N * g.tiles_x * g.tiles_y, groups = ((Cin + 15) / 16) * ((Cout + 63) / 64);
        int S = (512 + groups - 1) / groups;             // two blocks per CU: 512 fill the chip
        if (S > ntiles) S = ntiles;
        if (S < 1) S = 1;
        const int tps = (ntiles + S - 1) / S;
        S = (ntiles + tps - 1) / tps;
        const size_t wn = (size_t)Cout * c.Kdim;
        // slice counts of the split-bf16 variants (their own tile grid): the partial-sum buffer is sized for the largest
        // grid any of the three kernels may be launched with
        TTileW gw;
        const bool x3 = ctx->train_x3 && t_tile_geometry_wx3(c.Ho, c.Wo, &gw);
        int S2 = 0, S3 = 0, tps2 = 0, tps3 = 0, nt = 0;
        size_t setb = 0;
        bool pp = false;
        if (x3) {
            nt = N * gw.tiles_x * gw.tiles_y;
            const int groups2 = ((Cin + TXW_CI - 1) / TXW_CI) * ((Cout + 63) / 64);
            S2 = (512 + groups2 - 1) / groups2;             // two blocks per CU: 512 fill the chip; fewer slices = less partial-sum traffic
            if (S2 > nt) S2 = nt;
            if (S2 < 1) S2 = 1;
            tps2 = (nt + S2 - 1) / S2;
            S2 = (nt + tps2 - 1) / tps2;
            setb = (size_t)2 * 64 * TXW_YP + (size_t)2 * TXW_CI * gw.CHB;
            S3 = (256 + groups2 - 1) / groups2;             // ping-pong variant: one 8-wave block per CU
            if (S3 > nt / 2) S3 = nt / 2;                   // at least two tiles per block, or the second wave group has nothing to do
            pp = !getenv("POPNET_TRAIN_WGRAD_4WAVE") && S3 >= 1 && 2 * setb <= 158 * 1024 && 2 * setb >= 73728;
            if (pp) {
                tps3 = (nt + S3 - 1) / S3;
                S3 = (nt + tps3 - 1) / tps3;
            }
        }
        const int Smax = std::max(S, x3 ? (pp ? S3 : S2) : 0);
        const size_t part_off = (wn * (size_t)Smax * sizeof(float) + 15) & ~(size_t)15;
        void *ws = nullptr;
        int rc = t_ws(ctx, part_off + 16 + (size_t)Cout * csl0 * 2 * sizeof(double), &ws);
        if (rc != PN_OK) return rc;
        if ((rc = t_tile_lds_ok(ctx)) != PN_OK) return rc;
        hipStream_t s = (hipStream_t)hip_stream;
        bool done = false;
        if (x3) {
            TTileW gv;
            int ppi = 0, npieces = 0;
            if (pp && !getenv("POPNET_TRAIN_WGRAD_NOVEC") && t_tile_geometry_wx3v(c, &gv, &ppi, &npieces) && gv.tiles_x == gw.tiles_x && gv.tiles_y == gw.tiles_y && gv.R == gw.R && gv.TW == gw.TW) {
                // the same tiles and slices as x3pp (bit-identical partial sums), one round trip of staging per tile
                const size_t lds = std::max<size_t>((size_t)2 * 2 * TXW_CI * gv.CHB, (size_t)72 * 256 * 4);
                hipLaunchKernelGGL(tconv3_wgrad_x3v_kernel, dim3((unsigned)((Cin + TXW_CI - 1) / TXW_CI), (unsigned)((Cout + 63) / 64), (unsigned)S3), dim3(512), lds, s, c, gv, (float *)ws, tps3, nt, ppi, npieces);
                t_wgrad_reduce(s, (const float *)ws, dw_dev, wn, S3);
            } else if (pp) {
                hipLaunchKernelGGL(tconv3_wgrad_x3pp_kernel, dim3((unsigned)((Cin + TXW_CI - 1) / TXW_CI), (unsigned)((Cout + 63) / 64), (unsigned)S3), dim3(512), 2 * setb, s, c, gw, (float *)ws, tps3, nt);
                t_wgrad_reduce(s, (const float *)ws, dw_dev, wn, S3);
            } else {
                hipLaunchKernelGGL(tconv3_wgrad_x3_kernel, dim3((unsigned)((Cin + TXW_CI - 1) / TXW_CI), (unsigned)((Cout + 63) / 64), (unsigned)S2), dim3(256), setb, s, c, gw, (float *)ws, tps2, nt);
                t_wgrad_reduce(s, (const float *)ws, dw_dev, wn, (int)S2);
            }
            done = true;
        }
        if (!done) {
            const size_t lds = (size_t)(128 * TT_YP + g.HR * g.HC * TT_HP + 128) * sizeof(float);
            hipLaunchKernelGGL(tconv3_wgrad_tile_kernel, dim3((unsigned)((Cin + 15) / 16), (unsigned)((Cout + 63) / 64), (unsigned)S), dim3(256), lds, s, c, g, (float *)ws, tps, ntiles);
            t_wgrad_reduce(s, (const float *)ws, dw_dev, wn, (int)S);
        }
        if (dbias_dev) {
            double *part = (double *)((char *)ws + part_off);
            t_chan_reduce<2>(s, nullptr, dy_dev, nullptr, nullptr, nullptr, nullptr, nullptr, 0, N, Cout, c.Ho * c.Wo, csl0, part);
            hipLaunchKernelGGL(sums_finish_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(64), 0, s, (const double *)part, Cout, csl0, dbias_dev, nullptr, nullptr);
        }
        PN_HIP_CHECK(ctx, hipGetLastError());
        return PN_OK;
    }
    const int tiles = ((c.Kdim + 63) / 64) * ((Cout + 63) / 64);
    long slices = (1024 + tiles - 1) / tiles;
    const long cap = (P + 1023) / 1024;
    if (slices > cap) slices = cap;
    if (slices < 1) slices = 1;
    long pps = (P + slices - 1) / slices;
    pps = (pps + TW_RC - 1) / TW_RC * TW_RC;
    slices = (P + pps - 1) / pps;
    const size_t wn = (size_t)Cout * c.Kdim;
    const int csl = t_slices(P, Cout);
    void *ws = nullptr;
    int rc = t_ws(ctx, wn * slices * sizeof(float) + 16 + (size_t)Cout * csl * 2 * sizeof(double), &ws);
    if (rc != PN_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    dim3 grid((unsigned)((c.Kdim + 63) / 64), (unsigned)((Cout + 63) / 64), (unsigned)slices), block(256);
    if (ks == 1) hipLaunchKernelGGL(tconv_wgrad_kernel<1>, grid, block, 0, s, c, (float *)ws, (int)pps);
    else if (ks == 3) hipLaunchKernelGGL(tconv_wgrad_kernel<3>, grid, block, 0, s, c, (float *)ws, (int)pps);
    else hipLaunchKernelGGL(tconv_wgrad_kernel<7>, grid, block, 0, s, c, (float *)ws, (int)pps);
    t_wgrad_reduce(s, (const float *)ws, dw_dev, wn, (int)slices);
    if (dbias_dev) {
        double *part = (double *)((char *)ws + ((wn * slices * sizeof(float) + 15) & ~(size_t)15));
        t_chan_reduce<2>(s, nullptr, dy_dev, nullptr, nullptr, nullptr, nullptr, nullptr, 0, N, Cout, c.Ho * c.Wo, csl, part);
        hipLaunchKernelGGL(sums_finish_kernel, dim3((unsigned)((Cout + 63) / 64)), dim3(64), 0, s, (const double *)part, Cout, csl, dbias_dev, nullptr, nullptr);
    }
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

}   // extern "C"
// ---- the planes training engine's stem (trainx.hip; internal, not part of the C ABI) --------------------------------------------------
// model0.conv1 (rtpose_light3d.py:144: 7x7 / 2, Cin = 1) with its output / its output gradient as a planes tensor (TConv::pl): the kernels,
// grids, slices and summation order of pn_conv2d_forward / pn_conv2d_wgrad's generic path, without the NCHW f32 hand-over tensor.
static int t_stem_conv(pn_ctx *ctx, TConv *c, const float *x_dev, void *planes, int cs, int split, int N, int Cin, int H, int W, int Cout, int ks, int stride, int pad) {
    if (!x_dev || !planes || N < 1 || Cin < 1 || ks != 7 || stride < 1 || pad < 0 || Cout < 8 || (Cout & 7) || cs < Cout || (cs & 7) || (split & 7))
        return pn_set_error(ctx, PN_ERR_INVALID, "planes stem: bad arguments (7x7, Cout and strides multiples of 8)");
    c->x = x_dev; c->w = nullptr; c->bias = nullptr; c->y = nullptr;
    c->N = N; c->Cin = Cin; c->H = H; c->W = W; c->Cout = Cout; c->stride = stride; c->pad = pad; c->accumulate = 0;
    c->Ho = (H + 2 * pad - ks) / stride + 1;
    c->Wo = (W + 2 * pad - ks) / stride + 1;
    c->Kdim = Cin * ks * ks;
    const long P = (long)N * c->Ho * c->Wo;
    if (c->Ho < 1 || c->Wo < 1 || P > 0x7fffffffL) return pn_set_error(ctx, PN_ERR_INVALID, "planes stem: size out of range");
    c->P = (int)P;
    c->pl = planes; c->pl_cs = cs; c->pl_split = split;
    return PN_OK;
}

int pn_stem_forward_planes(pn_ctx *ctx, const float *x_dev, const float *w_dev, void *y_planes, int cs, int split, int f32, int N, int Cin, int H, int W, int Cout,
                           int ks, int stride, int pad, int gather, hipStream_t s) {
    T_CTX_CHECK("pn_stem_forward_planes")
    TConv c;
    if (int rc = t_stem_conv(ctx, &c, x_dev, y_planes, cs, split, N, Cin, H, W, Cout, ks, stride, pad)) return rc;
    if (!w_dev) return pn_set_error(ctx, PN_ERR_INVALID, "pn_stem_forward_planes: no weights");
    c.w = w_dev;
    if (!gather && Cin == 1 && Cout == 64 && stride == 2 && pad == 3) {         // the input patch in LDS (tstem_fwd_kernel)
        const int tiles_x = (c.Wo + 15) / 16, tiles_y = (c.Ho + 7) / 8;
        const long ntiles = (long)N * tiles_x * tiles_y;
        // blocks per CU, one-stream traces of the step: 1: 49.1, 2: 41.5, 3: 42.8, 4: 45.0, 6: 48.3, 12: 62.2 us (three are resident at 148 registers; every block
        // loads the 52 weight registers once) -- POPNET_STEM_FWD_BLOCKS overrides
        const char *eb = getenv("POPNET_STEM_FWD_BLOCKS");
        const unsigned blocks = (unsigned)std::min<long>(ntiles, (long)ctx->num_cus * (eb ? std::max(1, atoi(eb)) : 2));
        if (f32) hipLaunchKernelGGL(tstem_fwd_kernel<1>, dim3(blocks), dim3(256), 0, s, c, tiles_x, tiles_y, (int)ntiles);
        else hipLaunchKernelGGL(tstem_fwd_kernel<0>, dim3(blocks), dim3(256), 0, s, c, tiles_x, tiles_y, (int)ntiles);
        PN_HIP_CHECK(ctx, hipGetLastError());
        return PN_OK;
    }
    dim3 grid((unsigned)((c.P + 127) / 128), (unsigned)((Cout + 63) / 64)), block(256);
    if (f32) hipLaunchKernelGGL((tconv_fwd_kernel<7, 2>), grid, block, 0, s, c);
    else hipLaunchKernelGGL((tconv_fwd_kernel<7, 1>), grid, block, 0, s, c);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_stem_wgrad_planes(pn_ctx *ctx, const float *x_dev, const void *dy_planes, int cs, int split, int f32, const PnStemBn *bn, float *dw_dev, int N, int Cin, int H,
                         int W, int Cout, int ks, int stride, int pad, int depth, hipStream_t s) {
    T_CTX_CHECK("pn_stem_wgrad_planes")
    TConv c;
    if (int rc = t_stem_conv(ctx, &c, x_dev, (void *)dy_planes, cs, split, N, Cin, H, W, Cout, ks, stride, pad)) return rc;
    if (!dw_dev) return pn_set_error(ctx, PN_ERR_INVALID, "pn_stem_wgrad_planes: no output");
    const long P = c.P;
    // (the slices of pn_conv2d_wgrad's generic path: the same partial sums in the same order)
    const int tiles = ((c.Kdim + 63) / 64) * ((Cout + 63) / 64);
    long slices = (1024 + tiles - 1) / tiles;
    const long cap = (P + 1023) / 1024;
    if (slices > cap) slices = cap;
    if (slices < 1) slices = 1;
    long pps = (P + slices - 1) / slices;
    pps = (pps + TW_RC - 1) / TW_RC * TW_RC;
    slices = (P + pps - 1) / pps;
    const size_t wn = (size_t)Cout * c.Kdim;
    void *ws = nullptr;
    if (int rc = t_ws(ctx, wn * slices * sizeof(float) + 16, &ws)) return rc;
    dim3 grid((unsigned)((c.Kdim + 63) / 64), (unsigned)((Cout + 63) / 64), (unsigned)slices), block(512);
    TStemBn b = TStemBn();
    if (bn) {
        if (!bn->x || !bn->mean || !bn->invstd || !bn->k1 || !bn->k2 || !bn->k3 || !bn->scale || !bn->shift || (bn->x_cs & 7) || (bn->x_split & 7))
            return pn_set_error(ctx, PN_ERR_INVALID, "pn_stem_wgrad_planes: incomplete BatchNorm description");
        b.x = bn->x; b.x_cs = bn->x_cs; b.x_split = bn->x_split; b.mean = bn->mean; b.invstd = bn->invstd; b.k1 = bn->k1; b.k2 = bn->k2; b.k3 = bn->k3;
        b.scale = bn->scale; b.shift = bn->shift; b.act = bn->act;
    }
    float *wsf = (float *)ws;
    const int pp = (int)pps;
#define TSTEM_LAUNCH(F, B, D) hipLaunchKernelGGL((tstem_wgrad_kernel<7, F, B, D>), grid, block, 0, s, c, b, wsf, pp)
    if (depth <= 1) {        // (one chunk in flight: tconv_wgrad_kernel's schedule, for A/B runs)
        if (f32) { if (bn) TSTEM_LAUNCH(1, 1, 1); else TSTEM_LAUNCH(1, 0, 1); }
        else { if (bn) TSTEM_LAUNCH(0, 1, 1); else TSTEM_LAUNCH(0, 0, 1); }
    } else if (depth < 4) {
        if (f32) { if (bn) TSTEM_LAUNCH(1, 1, 2); else TSTEM_LAUNCH(1, 0, 2); }
        else { if (bn) TSTEM_LAUNCH(0, 1, 2); else TSTEM_LAUNCH(0, 0, 2); }
    } else {
        if (f32) { if (bn) TSTEM_LAUNCH(1, 1, 4); else TSTEM_LAUNCH(1, 0, 4); }
        else { if (bn) TSTEM_LAUNCH(0, 1, 4); else TSTEM_LAUNCH(0, 0, 4); }
    }
#undef TSTEM_LAUNCH
    t_wgrad_reduce(s, (const float *)ws, dw_dev, wn, (int)slices);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

extern "C" {
int pn_bn_train_forward(pn_ctx *ctx, const float *x_dev, const float *gamma_dev, const float *beta_dev, const float *res_dev, float *y_dev,
                        float *save_mean_dev, float *save_invstd_dev, float *running_mean_dev, float *running_var_dev, float momentum, float eps,
                        int act, int N, int C, int HW, void *hip_stream) {
    T_CTX_CHECK("pn_bn_train_forward")
    if (!x_dev || !gamma_dev || !beta_dev || !y_dev || !save_mean_dev || !save_invstd_dev || N < 1 || C < 1 || HW < 1 || (long)N * HW < 2)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_bn_train_forward: bad arguments");
    const long cnt = (long)N * HW;
    const int sl = t_slices(cnt, C);
    void *ws = nullptr;
    int rc = t_ws(ctx, (size_t)C * sl * 2 * sizeof(double), &ws);
    if (rc != PN_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    t_chan_reduce<0>(s, x_dev, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, N, C, HW, sl, (double *)ws);
    if ((long)N * C > 65535) return pn_set_error(ctx, PN_ERR_INVALID, "pn_bn_train_forward: N * C out of range");
    const bool v4 = (HW & 3) == 0 && ((((size_t)x_dev) | ((size_t)y_dev) | ((size_t)res_dev)) & 15) == 0;
    if (v4)
        hipLaunchKernelGGL(bn_apply_kernel<4>, dim3((unsigned)((HW / 4 + 255) / 256), (unsigned)(N * C)), dim3(256), 0, s, x_dev, gamma_dev, beta_dev, (const double *)ws, sl,
                           (double)cnt, eps, momentum, save_mean_dev, save_invstd_dev, running_mean_dev, running_var_dev, res_dev, act, C, HW, y_dev);
    else
        hipLaunchKernelGGL(bn_apply_kernel<1>, dim3((unsigned)((HW + 255) / 256), (unsigned)(N * C)), dim3(256), 0, s, x_dev, gamma_dev, beta_dev, (const double *)ws, sl,
                           (double)cnt, eps, momentum, save_mean_dev, save_invstd_dev, running_mean_dev, running_var_dev, res_dev, act, C, HW, y_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_bn_train_backward(pn_ctx *ctx, const float *x_dev, const float *dy_dev, const float *out_dev, const float *gamma_dev, const float *beta_dev,
                         const float *save_mean_dev, const float *save_invstd_dev, int act, int N, int C, int HW, float *dx_dev, float *dgamma_dev, float *dbeta_dev,
                         float *dres_dev, int dres_accumulate, void *hip_stream) {
    T_CTX_CHECK("pn_bn_train_backward")
    if (!x_dev || !dy_dev || !gamma_dev || !save_mean_dev || !save_invstd_dev || !dx_dev || !dgamma_dev || !dbeta_dev || (act != PN_ACT_NONE && !out_dev && !beta_dev))
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_bn_train_backward: bad arguments (an activation needs out, or beta to recompute its mask)");
    const long cnt = (long)N * HW;
    const int sl = t_slices(cnt, C);
    void *ws = nullptr;
    int rc = t_ws(ctx, (size_t)C * sl * 2 * sizeof(double), &ws);
    if (rc != PN_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    t_chan_reduce<1>(s, x_dev, dy_dev, out_dev, save_mean_dev, save_invstd_dev, gamma_dev, beta_dev, act, N, C, HW, sl, (double *)ws);
    // d beta = sum g, d gamma = sum g (x - mean) * invstd: finished inside the apply kernel
    if ((long)N * C > 65535) return pn_set_error(ctx, PN_ERR_INVALID, "pn_bn_train_backward: N * C out of range");
    const bool v4 = (HW & 3) == 0 && ((((size_t)x_dev) | ((size_t)dy_dev) | ((size_t)out_dev) | ((size_t)dx_dev) | ((size_t)dres_dev)) & 15) == 0;
    if (v4)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3((unsigned)((HW / 4 + 255) / 256), (unsigned)(N * C)), dim3(256), 0, s, x_dev, dy_dev, out_dev, gamma_dev, beta_dev,
                           save_mean_dev, save_invstd_dev, (const double *)ws, sl, dbeta_dev, dgamma_dev, act, C, HW, (float)(1.0 / (double)cnt), dx_dev, dres_dev, dres_accumulate);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3((unsigned)((HW + 255) / 256), (unsigned)(N * C)), dim3(256), 0, s, x_dev, dy_dev, out_dev, gamma_dev, beta_dev,
                           save_mean_dev, save_invstd_dev, (const double *)ws, sl, dbeta_dev, dgamma_dev, act, C, HW, (float)(1.0 / (double)cnt), dx_dev, dres_dev, dres_accumulate);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_avgpool3s2_forward(pn_ctx *ctx, const float *x_dev, float *y_dev, int planes, int H, int W, void *hip_stream) {
    T_CTX_CHECK("pn_avgpool3s2_forward")
    if (!x_dev || !y_dev || planes < 1 || H < 1 || W < 1) return pn_set_error(ctx, PN_ERR_INVALID, "pn_avgpool3s2_forward: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)planes * Ho * Wo;
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((unsigned)((Ho * Wo + 255) / 256), (unsigned)std::min(planes, 65535)), dim3(256), 0, (hipStream_t)hip_stream, x_dev, H, W, Ho, Wo, total, y_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_avgpool3s2_backward(pn_ctx *ctx, const float *dy_dev, float *dx_dev, int planes, int H, int W, void *hip_stream) {
    T_CTX_CHECK("pn_avgpool3s2_backward")
    if (!dy_dev || !dx_dev || planes < 1 || H < 1 || W < 1) return pn_set_error(ctx, PN_ERR_INVALID, "pn_avgpool3s2_backward: bad arguments");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)planes * H * W;
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3((unsigned)((H * W + 255) / 256), (unsigned)std::min(planes, 65535)), dim3(256), 0, (hipStream_t)hip_stream, dy_dev, H, W, Ho, Wo, total, dx_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_head_forward(pn_ctx *ctx, const float *v_dev, const float *target_dev, const float *fg_dev, int kind, int N, int C, int HW, float *s_dev,
                    float *out_dev, int out_ld, float *loss_dev, void *hip_stream) {
    T_CTX_CHECK("pn_head_forward")
    if (!v_dev || !target_dev || !s_dev || !out_dev || !loss_dev || out_ld < C || N < 1) return pn_set_error(ctx, PN_ERR_INVALID, "pn_head_forward: bad arguments");
    const size_t total = (size_t)N * C * HW;
    const unsigned nb = (unsigned)((total + 255) / 256);
    void *ws = nullptr;
    int rc = t_ws(ctx, (size_t)nb * sizeof(double), &ws);
    if (rc != PN_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(nb), dim3(256), 0, s, v_dev, target_dev, fg_dev, kind, C, HW, total, s_dev, out_dev, out_ld, (double *)ws);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, s, (const double *)ws, (int)nb, (double)total, loss_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_head_backward(pn_ctx *ctx, const float *s_dev, const float *target_dev, const float *fg_dev, const float *dextra_dev, int dextra_ld, int kind,
                     int N, int C, int HW, float *dv_dev, void *hip_stream) {
    T_CTX_CHECK("pn_head_backward")
    if (!s_dev || !target_dev || !dv_dev || (dextra_dev && dextra_ld < C)) return pn_set_error(ctx, PN_ERR_INVALID, "pn_head_backward: bad arguments");
    const size_t total = (size_t)N * C * HW;
    hipLaunchKernelGGL(head_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, s_dev, target_dev, fg_dev, dextra_dev,
                       dextra_ld, kind, C, HW, total, (float)(1.0 / (double)total), dv_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_slice_copy(pn_ctx *ctx, const float *src_dev, int src_ld, float *dst_dev, int dst_ld, int N, int C, int HW, int accumulate, void *hip_stream) {
    T_CTX_CHECK("pn_slice_copy")
    if (!src_dev || !dst_dev || src_ld < C || dst_ld < C) return pn_set_error(ctx, PN_ERR_INVALID, "pn_slice_copy: bad arguments");
    const size_t total = (size_t)N * C * HW;
    hipLaunchKernelGGL(slice_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, src_dev, src_ld, dst_dev, dst_ld, C, HW,
                       total, accumulate);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_sgd_nesterov(pn_ctx *ctx, float *param_dev, const float *grad_dev, float *momentum_buf_dev, size_t n, float lr, float momentum, float weight_decay,
                    int first_step, float grad_scale, void *hip_stream) {
    T_CTX_CHECK("pn_sgd_nesterov")
    if (!param_dev || !grad_dev || !momentum_buf_dev) return pn_set_error(ctx, PN_ERR_INVALID, "pn_sgd_nesterov: bad arguments");
    if (n == 0) return PN_OK;
    for (auto &e : ctx->train_packs) e.fresh = false;         // the weights move: cached packs are stale until the next pn_train_pack_refresh (or per-call pack)
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, param_dev, grad_dev, momentum_buf_dev, n,
                       lr, momentum, weight_decay, first_step, grad_scale);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

}  // extern "C"
