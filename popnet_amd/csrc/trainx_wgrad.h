// Weight gradient of a 3x3 / 1x1 stride-1 "same" convolution on NHWC [hi | lo] bf16 planes: a pixel-K GEMM on the matrix cores.
//
//   dW[co][ci][ky][kx] = sum over (n, y, x) of dy[n, y, x, co] * x[n, y + ky - pad, x + kx - pad, ci]      (autograd of nn.Conv2d:
//   tpm/lib/network/rtpose_light3d.py:24-34,222-246 under tpm/train_rtpose_light3d_kdh3d_mpaug.py:160-180 (CR))
// with every product taken as dy_hi x_hi + dy_lo x_hi + dy_hi x_lo (split-bf16, fp32 accumulate).
//
// MFMA view: D[co][ci] += A[co][k] B[k][ci] with k = PIXEL.  Both operands are channel-minor in memory, so both fragments are
// "k-strided": they are read from LDS with ds_read_b64_tr_b16 (gfx950's transposing read: a 4-pixel x 16-channel block arrives
// pixel-minor), the one instruction that makes an NHWC weight gradient cheap -- no second (pixel-minor) copy of any tensor exists.
//   * block = 4 waves = 64 couts x 64 cins x all taps; wave = 32 x 32 x KS*KS accumulator tiles (144 VGPRs for 3x3);
//   * a k-chunk is one ROW SEGMENT of <= 30 pixels (32 slots, the surplus slots hold dy = 0): the dy fragment of a row is read once and
//     multiplied with the nine shifted x fragments (tap = an immediate LDS offset: the halo image is [row][plane][32 px][128 B]);
//   * images are filled by LDS-DMA (global_load_lds_dwordx4, no staging registers), 16-byte slots XOR-swizzled by the pixel so that the
//     eight pixels a half-wave's transposed read touches cover all 64 banks once -- for every tap shift;
//   * k -> pixel map of a fragment: lane group g, read j, row q  <->  pixel 16 j + 4 g + q (any map works as long as A and B share it;
//     this one keeps a half-wave on eight CONSECUTIVE pixels);
//   * deterministic split-K: block s of a tile pair sums its strips into partial[s], wgrad_reduce_kernel adds the slices in order and
//     scatters to the reference's [Cout][Cin][k][k] layout (undoing this engine's channel order of the stage-2 input).
#pragma once
#include <functional>
#include <vector>
#include "pn_internal.h"
#include "trainx_kernels.h"

namespace tx {

typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

struct WgArgs {
    const bf *x; int x_cs, x_split; unsigned x_zero;        // input tensor, byte offset of its zero page
    const bf *dy; int dy_cs, dy_split; unsigned dy_zero;    // output gradient
    int B, H, W;
    int Wt, tiles_x, strips_per_img, nstrips, strips_per_block;
    int ncit;                                               // 64-channel tiles on the cin side (blockIdx.y = cot * ncit + cit)
    int co_pad, ci_pad;
    float *partial;                                         // [split][tap][co_pad][ci_pad]
};

__device__ __forceinline__ void glds16(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int KS, int R>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgArgs a) {
    typedef __attribute__((address_space(3))) bf4 lds4;
    constexpr int KK = KS * KS, PAD = KS / 2, HR = R + KS - 1;
    constexpr int ROWB = 2 * 32 * 128;                    // bytes of one image row: 2 planes x 32 pixel slots x 64 channels
    constexpr int XIMG = HR * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [x halo image][dy image]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;              // cout half, cin half of the 64 x 64 block tile
    const int cot = blockIdx.y / a.ncit, cit = blockIdx.y - cot * a.ncit;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int pxl = 4 * g + q;

    // transposed-read addresses (bytes): lane 4q + p of group g supplies row (= pixel) q, channels 4p .. 4p + 3 of the 16-channel window
    int addrA[2], addrB[KS][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int win = wm * 2 + mt, swz = (pxl >> 1) & 3;
        addrA[mt] = XIMG + pxl * 128 + ((((win ^ swz) << 1) | (p >> 1)) << 4) + ((p & 1) << 3);
    }
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int win = wn * 2 + nt, hp = pxl + kx, swz = (hp >> 1) & 3;
            addrB[kx][nt] = hp * 128 + ((((win ^ swz) << 1) | (p >> 1)) << 4) + ((p & 1) << 3);
        }

    f4 acc[2][2][KK];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int tp = 0; tp < KK; ++tp) acc[mt][nt][tp] = f4{0.f, 0.f, 0.f, 0.f};

    // DMA lane roles: one instruction = 8 pixels x 8 slots of 16 B (one plane); slot sp of pixel px holds logical slot ((sp >> 1) ^ ((px >> 1) & 3)) << 1 | (sp & 1)
    const int dpx = lane >> 3, dsp = lane & 7;
    const int dma_px = wave * 8 + dpx;
    const int dma_slot = (((dsp >> 1) ^ ((dma_px >> 1) & 3)) << 1) | (dsp & 1);
    const unsigned lane_x = (unsigned)(((dma_px - PAD) * a.x_cs + ci0 + dma_slot * 8) * 2);      // (two's complement: added to the strip base modulo 2^32)
    const unsigned lane_y = (unsigned)((dma_px * a.dy_cs + co0 + dma_slot * 8) * 2);
    const unsigned rowx = (unsigned)(a.W * a.x_cs * 2), rowy = (unsigned)(a.W * a.dy_cs * 2);
    const unsigned planex = (unsigned)(a.x_split * 2), planey = (unsigned)(a.dy_split * 2);
    const unsigned dma_lds = (unsigned)__builtin_amdgcn_readfirstlane(wave * 1024);
    const int s0 = blockIdx.x * a.strips_per_block, s1 = min(s0 + a.strips_per_block, a.nstrips);
    for (int sid = s0; sid < s1; ++sid) {
        const int b = sid / a.strips_per_img, rem = sid - b * a.strips_per_img;
        const int ty = rem / a.tiles_x, tx_ = rem - ty * a.tiles_x;
        const int oy0 = ty * R, ox0 = tx_ * a.Wt;
        const int Wc = min(a.Wt, a.W - ox0);
        __syncthreads();                                   // every wave has finished reading the previous strip's images
        // x halo image: HR rows x 2 planes, dy image: R rows x 2 planes; each (row, plane) is four 8-pixel pieces: wave w fetches piece w of every one.
        // Addresses in 32-bit arithmetic (a tensor is < 4 GiB): strip base and row validity on the scalar unit, one lane constant, one select per piece
        // (the first version spent 1.9 vector instructions per MFMA on 64-bit address arithmetic: 31 % matrix-pipe use from one wave per SIMD)
        {
            const unsigned bx0 = (unsigned)(((long)(b * a.H + oy0 - PAD) * a.W + ox0) * (long)a.x_cs * 2);
            const unsigned by0 = (unsigned)(((long)(b * a.H + oy0) * a.W + ox0) * (long)a.dy_cs * 2);
            const bool okx = dma_px < Wc + 2 * PAD && (unsigned)(ox0 - PAD + dma_px) < (unsigned)a.W;
            const bool oky = dma_px < Wc;
#pragma unroll
            for (int i = 0; i < HR * 2; ++i) {
                const int row = i >> 1, pl = i & 1;
                const bool rowok = (unsigned)(oy0 - PAD + row) < (unsigned)a.H;            // wave-uniform
                const unsigned off = (okx && rowok) ? bx0 + (unsigned)row * rowx + (unsigned)pl * planex + lane_x : a.x_zero;
                glds16(a.x, off, (unsigned)(i * 4096) + dma_lds);
            }
#pragma unroll
            for (int i = 0; i < R * 2; ++i) {
                const int row = i >> 1, pl = i & 1;
                const bool rowok = oy0 + row < a.H;
                const unsigned off = (oky && rowok) ? by0 + (unsigned)row * rowy + (unsigned)pl * planey + lane_y : a.dy_zero;
                glds16(a.dy, off, (unsigned)(XIMG + i * 4096) + dma_lds);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // Software-pipelined over the R * KK (row, tap) items of the strip: the fragments of item i + 1 are requested BEFORE the twelve MFMAs of item i are issued
        // (one wave per SIMD: nothing else hides an LDS round trip; the compiler's own schedule read two fragments, waited, issued four MFMAs -- 31 % matrix-pipe use).
        // sched_group_barrier pins the interleave: two reads, three MFMAs, four times per item.
        const int rows = min(R, a.H - oy0);                // wave-uniform: rows below the map hold dy = 0 and are skipped
        auto load_a = [&](int r, bf8 (&A)[2][2]) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const bf4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrA[mt] + r * ROWB + pl * 4096));
                    const bf4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrA[mt] + r * ROWB + pl * 4096 + 2048));
                    A[mt][pl] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
        };
        auto load_b = [&](int r, int tp, bf8 (&Bf)[2][2]) {
            const int ky = tp / KS, kx = tp - ky * KS;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const bf4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrB[kx][nt] + (r + ky) * ROWB + pl * 4096));
                    const bf4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrB[kx][nt] + (r + ky) * ROWB + pl * 4096 + 2048));
                    Bf[nt][pl] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
        };
        bf8 A[2][2][2], Bq[2][2][2];                       // [item parity]: current and next fragments
        load_a(0, A[0]);
        load_b(0, 0, Bq[0]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r >= rows) break;
#pragma unroll
            for (int tp = 0; tp < KK; ++tp) {
                const int cur = (r * KK + tp) & 1, nxt = cur ^ 1, ac = r & 1;
                const bool more = tp + 1 < KK || r + 1 < rows;
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    if (tp + 1 < KK) load_b(r, tp + 1, Bq[nxt]);
                    else { load_b(r + 1, 0, Bq[nxt]); load_a(r + 1, A[ac ^ 1]); }
                }
                // product-major: the three MFMAs into one accumulator tile sit four instructions apart (a dependent MFMA issued back to back waits out the first one's passes)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ac][mt][pr == 1 ? 1 : 0], Bq[cur][nt][pr == 2 ? 1 : 0], acc[mt][nt][tp], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // two LDS reads of the next item ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);      // ... then three MFMAs of this one
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // partial[split][tap][co][ci]: a lane's 16 neighbours write 64 contiguous bytes
    float *part = a.partial + (size_t)blockIdx.x * KK * a.co_pad * a.ci_pad;
#pragma unroll
    for (int tp = 0; tp < KK; ++tp)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + wm * 32 + mt * 16 + 4 * g + i, ci = ci0 + wn * 32 + nt * 16 + (lane & 15);
                    part[((size_t)tp * a.co_pad + co) * a.ci_pad + ci] = acc[mt][nt][tp][i];
                }
}

// ---- row-streaming form -------------------------------------------------------------------------------------------------------------------------
// Same tiles, same fragments, same products as wgrad_kernel; what differs is how the images reach LDS.  wgrad_kernel fetches a 3-row strip (5 halo rows
// of x, 3 rows of dy: 64 KB), waits, computes, and -- alone on its CU since the split-K went to one block per CU -- exposes that round trip once per strip
// (36 % of wave cycles waiting).  Here a block walks a COLUMN STRIP top-down: rings of KS + D x rows and 1 + D dy rows (D = 2: the same 64 KB), per row ONE
// group of four LDS-DMA instructions per wave for the row D ahead, ONE barrier, a counted `s_waitcnt vmcnt` (the D - 1 newer groups stay in flight) --
// and no halo row is ever fetched twice.  A block owns `rows_per_block` consecutive rows of the flattened (image, column strip, row) space; where its range
// crosses into the next column strip the rings are re-primed.
struct WgsArgs {
    const bf *x; int x_cs, x_split; unsigned x_zero;
    const bf *dy; int dy_cs, dy_split; unsigned dy_zero;
    int B, H, W;
    int Wt, tiles_x, rows_total, rows_per_block;
    int ncit, co_pad, ci_pad;
    float *partial;
};

template <int KS, int D>
__global__ __launch_bounds__(256, 2) void wgrad_stream_kernel(WgsArgs a) {
    typedef __attribute__((address_space(3))) bf4 lds4;
    constexpr int KK = KS * KS, PAD = KS / 2, RING = 5;       // both rings hold RING rows: ring positions are compile-time in a body unrolled RING rows deep
    static_assert(KS + D <= RING && 1 + D <= RING, "ring too small for the prefetch distance");
    constexpr int ROWB = 2 * 32 * 128;
    constexpr int XRING = RING * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];      // [x ring][dy ring]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int cot = blockIdx.y / a.ncit, cit = blockIdx.y - cot * a.ncit;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int pxl = 4 * g + q;
    int addrA[2], addrB[KS][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int win = wm * 2 + mt, swz = (pxl >> 1) & 3;
        addrA[mt] = XRING + pxl * 128 + ((((win ^ swz) << 1) | (p >> 1)) << 4) + ((p & 1) << 3);
    }
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int win = wn * 2 + nt, hp = pxl + kx, swz = (hp >> 1) & 3;
            addrB[kx][nt] = hp * 128 + ((((win ^ swz) << 1) | (p >> 1)) << 4) + ((p & 1) << 3);
        }
    f4 acc[2][2][KK];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int tp = 0; tp < KK; ++tp) acc[mt][nt][tp] = f4{0.f, 0.f, 0.f, 0.f};

    // every LDS byte a fragment read can touch holds finite data from the start: a tap-shifted read of the surplus pixel slots runs a few pixels past its row
    // (multiplied by dy = 0 there -- but 0 x NaN is NaN, and LDS starts as whatever the previous kernel left)
    for (int o = threadIdx.x * 16; o < 2 * XRING; o += 256 * 16) *reinterpret_cast<u32x4_t *>(smem + o) = u32x4_t{0u, 0u, 0u, 0u};

    const int dpx = lane >> 3, dsp = lane & 7;
    const int dma_px = wave * 8 + dpx;
    const int dma_slot = (((dsp >> 1) ^ ((dma_px >> 1) & 3)) << 1) | (dsp & 1);
    const unsigned lane_x = (unsigned)(((dma_px - PAD) * a.x_cs + ci0 + dma_slot * 8) * 2);
    const unsigned lane_y = (unsigned)((dma_px * a.dy_cs + co0 + dma_slot * 8) * 2);
    const unsigned rowx = (unsigned)(a.W * a.x_cs * 2), rowy = (unsigned)(a.W * a.dy_cs * 2);
    const unsigned planex = (unsigned)(a.x_split * 2), planey = (unsigned)(a.dy_split * 2);
    const unsigned dma_lds = (unsigned)__builtin_amdgcn_readfirstlane(wave * 1024);

    int row = blockIdx.x * a.rows_per_block;
    const int row_end = min(row + a.rows_per_block, a.rows_total);
    while (row < row_end) {
        // segment: rows [ya, yb) of column strip cs
        const int cs = row / a.H, ya = row - cs * a.H;
        const int yb = min(a.H, ya + (row_end - row));
        const int nrows = yb - ya;
        const int b = cs / a.tiles_x, tx_ = cs - b * a.tiles_x;
        const int ox0 = tx_ * a.Wt, Wc = min(a.Wt, a.W - ox0);
        const unsigned bx0 = (unsigned)(((long)b * a.H * a.W + ox0) * (long)a.x_cs * 2);      // image row 0 of the strip (row offsets added below, modulo 2^32)
        const unsigned by0 = (unsigned)(((long)b * a.H * a.W + ox0) * (long)a.dy_cs * 2);
        const bool okx = dma_px < Wc + 2 * PAD && (unsigned)(ox0 - PAD + dma_px) < (unsigned)a.W;
        const bool oky = dma_px < Wc;
        // x row number n of the segment = image row ya - PAD + n -> ring slot n % RING; dy row number n = image row ya + n -> slot n % RING
        auto fetch_x = [&](int n, int slot) {
            const int Y = ya - PAD + n;
            const bool rowok = (unsigned)Y < (unsigned)a.H;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const unsigned off = (okx && rowok) ? bx0 + (unsigned)Y * rowx + (unsigned)pl * planex + lane_x : a.x_zero;
                glds16(a.x, off, (unsigned)(slot * ROWB + pl * 4096) + dma_lds);
            }
        };
        auto fetch_y = [&](int n, int slot) {
            const int Y = ya + n;
            const bool rowok = Y < yb;            // rows past the segment: zeros (their group is issued only to keep the vmcnt arithmetic uniform)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const unsigned off = (oky && rowok) ? by0 + (unsigned)Y * rowy + (unsigned)pl * planey + lane_y : a.dy_zero;
                glds16(a.dy, off, (unsigned)(XRING + slot * ROWB + pl * 4096) + dma_lds);
            }
        };
        __syncthreads();                          // the previous segment's fragment reads are done: the rings may be overwritten
#pragma unroll
        for (int n = 0; n < KS - 1; ++n) fetch_x(n, n % RING);
#pragma unroll
        for (int j = 0; j < D; ++j) { fetch_x(KS - 1 + j, (KS - 1 + j) % RING); fetch_y(j, j % RING); }
        for (int k0 = 0; k0 < nrows; k0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {      // row k = k0 + u: k % RING == u
                const int k = k0 + u;
                if (k >= nrows) break;
                // group k (x row number KS - 1 + k, dy row number k) has landed once at most D - 1 newer groups (4 instructions each) are outstanding
                if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __syncthreads();                  // ... for every wave's pieces; and every wave has finished row k - 1, whose slots group k + D overwrites
                fetch_x(KS - 1 + k + D, (KS - 1 + u + D) % RING);
                fetch_y(k + D, (u + D) % RING);
                auto load_a = [&](bf8 (&A)[2][2]) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            const bf4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrA[mt] + u * ROWB + pl * 4096));
                            const bf4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrA[mt] + u * ROWB + pl * 4096 + 2048));
                            A[mt][pl] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                };
                auto load_b = [&](int tp, bf8 (&Bf)[2][2]) {
                    const int ky = tp / KS, kx = tp - ky * KS;
                    const int xs = ((u + ky) % RING) * ROWB;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) {
                            const bf4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrB[kx][nt] + xs + pl * 4096));
                            const bf4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4 *)(smem + addrB[kx][nt] + xs + pl * 4096 + 2048));
                            Bf[nt][pl] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                };
                bf8 A[2][2], Bq[2][2][2];
                load_a(A);
                load_b(0, Bq[0]);
#pragma unroll
                for (int tp = 0; tp < KK; ++tp) {
                    const int cur = tp & 1, nxt = cur ^ 1;
                    __builtin_amdgcn_sched_barrier(0);
                    if (tp + 1 < KK) load_b(tp + 1, Bq[nxt]);
#pragma unroll
                    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[mt][nt][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt][pr == 1 ? 1 : 0], Bq[cur][nt][pr == 2 ? 1 : 0], acc[mt][nt][tp], 0, 0, 0);
                    if (tp + 1 < KK) {
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the D surplus groups of the segment's tail
        row += nrows;
    }
    float *part = a.partial + (size_t)blockIdx.x * KK * a.co_pad * a.ci_pad;
#pragma unroll
    for (int tp = 0; tp < KK; ++tp)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + wm * 32 + mt * 16 + 4 * g + i, ci = ci0 + wn * 32 + nt * 16 + (lane & 15);
                    part[((size_t)tp * a.co_pad + co) * a.ci_pad + ci] = acc[mt][nt][tp][i];
                }
}

// ---- fp32 form (precision "fp32": exact fp32 FMA chains on v_mfma_f32_16x16x4_f32) -------------------------------------------------------------------
// K = 4 pixels per MFMA and ONE value per lane and operand: lane (m, kg) multiplies dy[pixel 4 s + kg][co m] with x[pixel 4 s + kg + tap][ci n] -- a channel-minor
// tensor needs no transpose at all here (16 lanes read 16 consecutive channels).  Same row-streaming rings (five rows of x, five of dy, 256 B per pixel: 80 KB),
// 16-byte slots XOR-swizzled by the pixel's parity so that the two pixels a half-wave's ds_read_b32 touches fall on different banks.
struct WgfArgs {
    const float *x; int x_cs; unsigned x_zero;
    const float *dy; int dy_cs; unsigned dy_zero;
    int B, H, W;
    int Wt, tiles_x, rows_total, rows_per_block;
    int ncit, co_pad, ci_pad;
    float *partial;
};

template <int KS, int D>
__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(WgfArgs a) {
    typedef __attribute__((address_space(3))) float ldsf;
    constexpr int KK = KS * KS, PAD = KS / 2, RING = 5;
    static_assert(KS + D <= RING && 1 + D <= RING, "ring too small for the prefetch distance");
    constexpr int ROWB = 32 * 256;                          // one row: 32 pixel slots x 64 channels x 4 B
    constexpr int XRING = RING * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int cot = blockIdx.y / a.ncit, cit = blockIdx.y - cot * a.ncit;
    const int co0 = cot * 64, ci0 = cit * 64;
    const int m = lane & 15, kg = lane >> 4;
    // fragment addresses: pixel 4 s + kg (+ kx), channel window of 16; the channel is XORed with 16 on odd pixels
    int addrA[2], addrB[KS][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) addrA[mt] = XRING + kg * 256 + ((((wm * 2 + mt) * 16 + m) ^ ((kg & 1) << 4)) << 2);
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) addrB[kx][nt] = (kg + kx) * 256 + ((((wn * 2 + nt) * 16 + m) ^ (((kg + kx) & 1) << 4)) << 2);
    f4 acc[2][2][KK];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int tp = 0; tp < KK; ++tp) acc[mt][nt][tp] = f4{0.f, 0.f, 0.f, 0.f};
    for (int o = threadIdx.x * 16; o < 2 * XRING; o += 256 * 16) *reinterpret_cast<u32x4_t *>(smem + o) = u32x4_t{0u, 0u, 0u, 0u};

    // DMA: one instruction = 4 pixels x 16 slots of 16 B; wave w fetches pixel groups w and w + 4 of every row
    const int dq = lane >> 4, dsp = lane & 15;
    unsigned lane_x[2], lane_y[2];
    int dpx[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        dpx[h] = (wave + 4 * h) * 4 + dq;
        const int ls = dsp ^ ((dpx[h] & 1) << 2);
        lane_x[h] = (unsigned)(((dpx[h] - PAD) * a.x_cs + ci0 + ls * 4) * 4);
        lane_y[h] = (unsigned)((dpx[h] * a.dy_cs + co0 + ls * 4) * 4);
    }
    const unsigned rowx = (unsigned)(a.W * a.x_cs * 4), rowy = (unsigned)(a.W * a.dy_cs * 4);
    const unsigned dma_lds0 = (unsigned)__builtin_amdgcn_readfirstlane(wave * 1024);

    int row = blockIdx.x * a.rows_per_block;
    const int row_end = min(row + a.rows_per_block, a.rows_total);
    while (row < row_end) {
        const int cs = row / a.H, ya = row - cs * a.H;
        const int yb = min(a.H, ya + (row_end - row));
        const int nrows = yb - ya;
        const int b = cs / a.tiles_x, tx_ = cs - b * a.tiles_x;
        const int ox0 = tx_ * a.Wt, Wc = min(a.Wt, a.W - ox0);
        const unsigned bx0 = (unsigned)(((long)b * a.H * a.W + ox0) * (long)a.x_cs * 4);
        const unsigned by0 = (unsigned)(((long)b * a.H * a.W + ox0) * (long)a.dy_cs * 4);
        bool okx[2], oky[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            okx[h] = dpx[h] < Wc + 2 * PAD && (unsigned)(ox0 - PAD + dpx[h]) < (unsigned)a.W;
            oky[h] = dpx[h] < Wc;
        }
        auto fetch_x = [&](int n, int slot) {
            const int Y = ya - PAD + n;
            const bool rowok = (unsigned)Y < (unsigned)a.H;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned off = (okx[h] && rowok) ? bx0 + (unsigned)Y * rowx + lane_x[h] : a.x_zero;
                glds16(a.x, off, (unsigned)(slot * ROWB + h * 4096) + dma_lds0);
            }
        };
        auto fetch_y = [&](int n, int slot) {
            const int Y = ya + n;
            const bool rowok = Y < yb;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned off = (oky[h] && rowok) ? by0 + (unsigned)Y * rowy + lane_y[h] : a.dy_zero;
                glds16(a.dy, off, (unsigned)(XRING + slot * ROWB + h * 4096) + dma_lds0);
            }
        };
        __syncthreads();
#pragma unroll
        for (int n = 0; n < KS - 1; ++n) fetch_x(n, n % RING);
#pragma unroll
        for (int j = 0; j < D; ++j) { fetch_x(KS - 1 + j, (KS - 1 + j) % RING); fetch_y(j, j % RING); }
        for (int k0 = 0; k0 < nrows; k0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int k = k0 + u;
                if (k >= nrows) break;
                if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (D == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                __syncthreads();
                fetch_x(KS - 1 + k + D, (KS - 1 + u + D) % RING);
                fetch_y(k + D, (u + D) % RING);
                // items (k-step s, tap): the two x values of item i + 1 are requested before the four MFMAs of item i
                auto ld_a = [&](int s_, float (&A)[2]) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) A[mt] = *(const ldsf *)(smem + addrA[mt] + u * ROWB + s_ * 1024);
                };
                auto ld_b = [&](int s_, int tp, float (&Bv)[2]) {
                    const int ky = tp / KS, kx = tp - ky * KS;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) Bv[nt] = *(const ldsf *)(smem + addrB[kx][nt] + ((u + ky) % RING) * ROWB + s_ * 1024);
                };
                float A[2][2], Bq[2][2];
                ld_a(0, A[0]);
                ld_b(0, 0, Bq[0]);
#pragma unroll
                for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
                    for (int tp = 0; tp < KK; ++tp) {
                        const int it = s_ * KK + tp, cur = it & 1, nxt = cur ^ 1, ac = s_ & 1;
                        __builtin_amdgcn_sched_barrier(0);
                        if (tp + 1 < KK) ld_b(s_, tp + 1, Bq[nxt]);
                        else if (s_ + 1 < 8) { ld_b(s_ + 1, 0, Bq[nxt]); ld_a(s_ + 1, A[ac ^ 1]); }
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                acc[mt][nt][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[ac][mt], Bq[cur][nt], acc[mt][nt][tp], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        row += nrows;
    }
    const int g = lane >> 4;
    float *part = a.partial + (size_t)blockIdx.x * KK * a.co_pad * a.ci_pad;
#pragma unroll
    for (int tp = 0; tp < KK; ++tp)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + wm * 32 + mt * 16 + 4 * g + i, ci = ci0 + wn * 32 + nt * 16 + (lane & 15);
                    part[((size_t)tp * a.co_pad + co) * a.ci_pad + ci] = acc[mt][nt][tp][i];
                }
}

// dw[co][ref ci][tap] = sum over the split slices, in a fixed order; k_map: this engine's input channel -> the reference's (nullptr = identity).
// Block = 16 consecutive elements x 16 split lanes: lane l adds slices l, l + 16, ... (independent loads in flight), thread (element, lane 0) adds the
// 16 lane sums in order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ partial, int S, int KK, int co_pad, int ci_pad, int Cout, int Cin, const int *__restrict__ k_map,
                                                            float *__restrict__ dw) {
    __shared__ float sh[16][17];
    const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const size_t tap_stride = (size_t)co_pad * ci_pad, split_stride = tap_stride * KK;
    const size_t e = (size_t)blockIdx.x * 16 + el;                   // element of one slice: (tap, co, ci)
    float s = 0.f;
    if (e < split_stride) {
        const float *p = partial + e;
        int sp = sl;
        for (; sp + 48 < S; sp += 64) {
            const float v0 = p[(size_t)sp * split_stride], v1 = p[(size_t)(sp + 16) * split_stride], v2 = p[(size_t)(sp + 32) * split_stride], v3 = p[(size_t)(sp + 48) * split_stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; sp < S; sp += 16) s += p[(size_t)sp * split_stride];
    }
    sh[sl][el] = s;
    __syncthreads();
    if (sl != 0 || e >= split_stride) return;
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < 16; ++l) t += sh[l][el];
    const int tp = (int)(e / tap_stride);
    const int rem = (int)(e - (size_t)tp * tap_stride);
    const int co = rem / ci_pad, ci = rem - co * ci_pad;
    if (co >= Cout) return;
    const int ref = k_map ? k_map[ci] : (ci < Cin ? ci : -1);
    if (ref < 0) return;
    dw[((size_t)co * Cin + ref) * KK + tp] = t;
}

// Plans the weight gradient of one layer and appends its two launches to `ops`.  *partial_floats grows to what the layer needs; the
// buffer itself (*partial) is allocated by the caller after every layer has been planned (the kernels read the pointer at launch time).
inline int plan_wgrad_f32(pn_ctx *ctx, int B, int H, int W, const float *x, int x_plane, const float *dy, int dy_plane, int Cin, int Cout, int ks, const int *k_map, float *dw,
                          float *const *partial, size_t *partial_floats, std::vector<std::function<int(hipStream_t)>> &ops) {
    if (ks != 1 && ks != 3) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "fp32 weight gradient: kernel size %d not built", ks);
    const int KK = ks * ks, pad = ks / 2, seg_max = 32 - 2 * pad;
    WgfArgs w;
    memset(&w, 0, sizeof w);
    w.x = x; w.x_cs = x_plane; w.x_zero = (unsigned)((size_t)B * H * W * x_plane * 4);
    w.dy = dy; w.dy_cs = dy_plane; w.dy_zero = (unsigned)((size_t)B * H * W * dy_plane * 4);
    w.B = B; w.H = H; w.W = W;
    w.tiles_x = (W + seg_max - 1) / seg_max;
    w.Wt = (W + w.tiles_x - 1) / w.tiles_x;
    const int ci_my = k_map ? x_plane : Cin;
    const int ncot = (Cout + 63) / 64;
    w.ncit = (ci_my + 63) / 64;
    w.co_pad = ncot * 64; w.ci_pad = w.ncit * 64;
    if (w.co_pad > dy_plane || w.ci_pad > x_plane) return pn_set_error(ctx, PN_ERR_INVALID, "fp32 weight gradient: channel tiles exceed the tensors");
    const int pairs = ncot * w.ncit;
    w.rows_total = B * w.tiles_x * H;
    int Sr = std::max(1, std::min(w.rows_total, (ctx->num_cus + pairs - 1) / pairs));          // one block per CU (see plan_wgrad)
    w.rows_per_block = (w.rows_total + Sr - 1) / Sr;
    Sr = (w.rows_total + w.rows_per_block - 1) / w.rows_per_block;
    *partial_floats = std::max(*partial_floats, (size_t)Sr * KK * w.co_pad * w.ci_pad);
    const size_t ldss = (size_t)10 * 8192;
    ops.push_back([=](hipStream_t s) {
        WgfArgs ww = w;
        ww.partial = *partial;
        if (ks == 3) {
            static PnLdsAttr attr;
            if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(wgrad_f32_kernel<3, 2>), ldss)) return rc;
            hipLaunchKernelGGL((wgrad_f32_kernel<3, 2>), dim3(Sr, pairs), dim3(256), ldss, s, ww);
        } else {
            static PnLdsAttr attr;
            if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(wgrad_f32_kernel<1, 2>), ldss)) return rc;
            hipLaunchKernelGGL((wgrad_f32_kernel<1, 2>), dim3(Sr, pairs), dim3(256), ldss, s, ww);
        }
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((size_t)KK * ww.co_pad * ww.ci_pad + 15) / 16)), dim3(256), 0, s, (const float *)ww.partial, Sr, KK, ww.co_pad, ww.ci_pad, Cout, Cin, k_map, dw);
        PN_HIP_CHECK(ctx, hipGetLastError());
        return (int)PN_OK;
    });
    return PN_OK;
}

inline int plan_wgrad(pn_ctx *ctx, int B, int H, int W, const bf *x, int x_plane, const bf *dy, int dy_plane, int Cin, int Cout, int ks, const int *k_map, float *dw,
                      float *const *partial, size_t *partial_floats, std::vector<std::function<int(hipStream_t)>> &ops) {
    if (ks != 1 && ks != 3) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "weight gradient on planes: kernel size %d not built", ks);
    const int R = ks == 3 ? 3 : 4, KK = ks * ks, pad = ks / 2;
    const int seg_max = 32 - 2 * pad;                      // a halo row holds 32 pixels
    WgArgs a;
    memset(&a, 0, sizeof a);
    a.x = x; a.x_cs = 2 * x_plane; a.x_split = x_plane; a.x_zero = (unsigned)((size_t)B * H * W * 2 * x_plane * 2);
    a.dy = dy; a.dy_cs = 2 * dy_plane; a.dy_split = dy_plane; a.dy_zero = (unsigned)((size_t)B * H * W * 2 * dy_plane * 2);
    a.B = B; a.H = H; a.W = W;
    a.tiles_x = (W + seg_max - 1) / seg_max;
    a.Wt = (W + a.tiles_x - 1) / a.tiles_x;
    a.strips_per_img = ((H + R - 1) / R) * a.tiles_x;
    a.nstrips = B * a.strips_per_img;
    const int ci_my = k_map ? x_plane : Cin;               // channels of x that carry weights (the stage-2 input: the whole plane, re-ordered)
    const int ncot = (Cout + 63) / 64;
    a.ncit = (ci_my + 63) / 64;
    a.co_pad = ncot * 64; a.ci_pad = a.ncit * 64;
    if (a.co_pad > dy_plane || a.ci_pad > x_plane) return pn_set_error(ctx, PN_ERR_INVALID, "weight gradient on planes: channel tiles exceed the planes");
    const int pairs = ncot * a.ncit;
    // split-K slices: ONE block per CU.  Every slice costs a 147 KB partial tile written and read back, and a lone 64 KB block leaves the rest of the CU to
    // the BatchNorm / data-gradient launches of the step's own stream (same box, eager step: 1/2 block per CU 7.10 ms, 1: 6.49-6.68, 1.5: 6.93, 2: 7.03, 3: 7.39)
    const int per_cu_x2 = getenv("POPNET_TRAINX_WG_BLOCKS") ? atoi(getenv("POPNET_TRAINX_WG_BLOCKS")) : 2;      // blocks per CU, in halves (experiments)
    int S = std::max(1, std::min(a.nstrips, (per_cu_x2 * ctx->num_cus / 2 + pairs - 1) / pairs));
    a.strips_per_block = (a.nstrips + S - 1) / S;
    S = (a.nstrips + a.strips_per_block - 1) / a.strips_per_block;
    *partial_floats = std::max(*partial_floats, (size_t)S * KK * a.co_pad * a.ci_pad);
    const size_t lds = (size_t)((R + ks - 1) + R) * 8192;
    const WgArgs a0 = a;
    const bool stream_form = !(getenv("POPNET_TRAINX_WG_STREAM") && atoi(getenv("POPNET_TRAINX_WG_STREAM")) == 0);
    if (stream_form) {
        constexpr int D = 2;
        WgsArgs w;
        memset(&w, 0, sizeof w);
        w.x = a.x; w.x_cs = a.x_cs; w.x_split = a.x_split; w.x_zero = a.x_zero;
        w.dy = a.dy; w.dy_cs = a.dy_cs; w.dy_split = a.dy_split; w.dy_zero = a.dy_zero;
        w.B = B; w.H = H; w.W = W; w.Wt = a.Wt; w.tiles_x = a.tiles_x;
        w.rows_total = B * a.tiles_x * H;
        int Sr = std::max(1, std::min(w.rows_total, (per_cu_x2 * ctx->num_cus / 2 + pairs - 1) / pairs));
        w.rows_per_block = (w.rows_total + Sr - 1) / Sr;
        Sr = (w.rows_total + w.rows_per_block - 1) / w.rows_per_block;
        w.ncit = a.ncit; w.co_pad = a.co_pad; w.ci_pad = a.ci_pad;
        *partial_floats = std::max(*partial_floats, (size_t)Sr * KK * a.co_pad * a.ci_pad);
        const size_t ldss = (size_t)(5 + 5) * 8192;            // two rings of five rows
        ops.push_back([=](hipStream_t s) {
            WgsArgs ww = w;
            ww.partial = *partial;
            if (ks == 3) {
                static PnLdsAttr attr;
                if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(wgrad_stream_kernel<3, 2>), ldss)) return rc;
                hipLaunchKernelGGL((wgrad_stream_kernel<3, 2>), dim3(Sr, pairs), dim3(256), ldss, s, ww);
            } else {
                static PnLdsAttr attr;
                if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(wgrad_stream_kernel<1, 2>), ldss)) return rc;
                hipLaunchKernelGGL((wgrad_stream_kernel<1, 2>), dim3(Sr, pairs), dim3(256), ldss, s, ww);
            }
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((size_t)KK * ww.co_pad * ww.ci_pad + 15) / 16)), dim3(256), 0, s, (const float *)ww.partial, Sr, KK, ww.co_pad, ww.ci_pad, Cout, Cin, k_map, dw);
            PN_HIP_CHECK(ctx, hipGetLastError());
            return (int)PN_OK;
        });
        return PN_OK;
    }
    ops.push_back([=](hipStream_t s) {
        WgArgs a = a0;
        a.partial = *partial;                              // the host reads the pointer when the step launches (the buffer exists by then)
        if (ks == 3) {
            static PnLdsAttr attr;
            if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(wgrad_kernel<3, 3>), lds)) return rc;
            hipLaunchKernelGGL((wgrad_kernel<3, 3>), dim3(S, pairs), dim3(256), lds, s, a);
        } else {
            static PnLdsAttr attr;
            if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(wgrad_kernel<1, 4>), lds)) return rc;
            hipLaunchKernelGGL((wgrad_kernel<1, 4>), dim3(S, pairs), dim3(256), lds, s, a);
        }
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(((size_t)KK * a.co_pad * a.ci_pad + 15) / 16)), dim3(256), 0, s, (const float *)a.partial, S, KK, a.co_pad, a.ci_pad, Cout, Cin, k_map, dw);
        PN_HIP_CHECK(ctx, hipGetLastError());
        return (int)PN_OK;
    });
    return PN_OK;
}

}  // namespace tx
