// Fused ResNet BasicBlock(64) for the split-bf16 (PN_PREC_BF16X3) nets: the tolerance-meeting mode's 112x112 layers as ONE launch
//     y = relu( conv3x3(relu(conv3x3(x, W1) + b1), W2) + b2 + x )            64 -> 64 -> 64 channels, stride 1
// replaces the two conv3_kernel<3, 2, 2, 1, 7, 4> launches per block of tpm/lib/network/rtpose_light3d.py:48-72 (BasicBlock.forward) for
// model0.layer1 (rtpose_light3d.py:145-152) and resnet.BasicBlock (tpm/lib/network/resnet.py:27-56) for YoloPoseNet's layer1.
//
// bf16x3 (net.hip::prepare_conv): a tensor is two bf16 planes [hi | lo] read as [hi | lo | hi], weights are [W_hi | W_hi | W_lo] along Cin, one
// bf16 MFMA convolution over 3 x 64 channels computes x_hi W_hi + x_lo W_hi + x_hi W_lo in fp32 accumulators.  Unfused, a BasicBlock moves a
// 154 MB three-plane tensor through HBM four times (4 x 127.5 us per step at 0.28 of the physical peak).  The third plane
// is a copy of the first, so the LDS images of the fused kernel hold TWO planes and the K loop addresses the hi image twice:
//   * tile = 6 output rows x <= 28 columns (8-row tiles need 176 KB); the 10-row x 32-pixel input halo, 2 planes, quarter-major
//     [plane][quarter of 16 channels][row][32 px][32 B] = 80 KB; conv1 is evaluated on the 8 x 30 intermediate halo straight into the
//     intermediate image (2 planes, [plane][quarter][8 rows x 30 px][32 B] = 60 KB: its row pitch IS the halo width, so pixel slot s sits
//     at byte 32 s and a wave's 16-slot fragment reads are contiguous), conv2 reads that image; 5 x 4 KB weight ring: 160 KB, one
//     persistent workgroup per CU;
//   * same k order as the generic kernel -- (plane pair, 32-channel half, tap): 54 k-steps per conv -- with the same MFMA and the same
//     epilogue arithmetic ((acc + b) [+ x_hi + x_lo], ReLU, hi = bf16(v), lo = bf16(v - hi)): BIT-IDENTICAL to the two-launch bf16x3 plan
//     (tests/test_gpu_parity.py, POPNET_NO_BBLOCK=1 under precision="bf16x3");
//   * the input image is single-buffered: the residual (centre of the image, both planes) moves to registers after conv1, then the image
//     loaders refill the image with the NEXT tile's halo under conv2 (54 k-steps) and the output epilogue;
//   * waves 0-3 compute (64 couts x 4 / 3 pixel tiles: 16 / 12 MFMAs per k-step), waves 4, 5 stream the weights (W1 | W2 as one periodic
//     108-step stream, the W_hi steps fetched twice from the same L2 lines; step g + 4 is issued during step g into the slot whose
//     fragments were consumed during step g - 1, g + 2 has landed by the barrier that ends g: two k-steps of slack), waves 6, 7 fetch the
//     hi / lo plane of the next input image; one bare s_barrier per k-step.
// Weight pack (net.hip::add_bblock): [conv 2][W_hi, W_lo][18 k-steps = (half, tap)][4 cout tiles][64 lanes][8 bf16], rows permuted with
// pn_conv_row_channel(tile, row, 4) so that a lane's 16 accumulators are 16 consecutive channels.
#pragma once
#include "conv3_kernel.h"

#define BX_ROWS 6
#define BX_INQ (10 * 32 * 32)                 // one 16-channel quarter of one plane of the input image
#define BX_IN (8 * BX_INQ)                    // 2 planes x 4 quarters: 80 KB
#define BX_MIDP 30                            // intermediate image row pitch in pixels
#define BX_MIDQ (8 * BX_MIDP * 32)
#define BX_MID (8 * BX_MIDQ)                  // 60 KB
#define BX_OFF_MID BX_IN
#define BX_OFF_A (BX_IN + BX_MID)
#define BX_NSLOT 5
#define BX_ASLOT 4096
#define BX_LDS (BX_OFF_A + BX_NSLOT * BX_ASLOT)   // 163840 B = all of a CU's LDS
#define BX_KS 54                              // k-steps per convolution: 3 plane pairs x 2 halves x 9 taps
#ifdef BX_FAKE_NOBAR                          // timing-only ablation (scripts/bbx3lab.hip): no per-k-step barrier
#define BX_KBAR do {} while (0)
#elif defined(BX_LAGA)                        // the slot of step g is refilled during step g: its fragment reads (issued in the first MFMA groups of step g - 1) are drained first
#define BX_KBAR asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory")
#else
#define BX_KBAR asm volatile("s_barrier" ::: "memory")
#endif

__global__ __launch_bounds__(512, 1) void bb64x3_kernel(const BBProblem P) {
    typedef __bf16 T;
    constexpr int CT = 4, PT1 = 4, PT2 = 3, BQ = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int W = P.W, H = P.H;

    auto tile_geom = [&](int t, int &b, int &oy0, int &ox0, int &R, int &Wc) {
        b = t / P.tiles_per_img;
        const int rem = t - b * P.tiles_per_img;
        const int ty = rem / P.tiles_x, tx = rem - ty * P.tiles_x;
        oy0 = ty * BX_ROWS; ox0 = tx * P.Wt;
        R = min(BX_ROWS, H - oy0); Wc = min(P.Wt, W - ox0);
    };

    if (wave >= 4) {
        // ================= loader waves: LDS-DMA only =================
        const int lw = wave - 4;
        const unsigned lane16 = (unsigned)lane * 16u;
        int t = blockIdx.x;
        if (t >= P.ntiles) return;
        if (lw < 2) {
            // ---- weight loaders: each fetches half (two 1 KB instructions) of a k-step ----
            const char *wsrc = (const char *)P.wpack + lw * 2048;
            const unsigned ldsw = (unsigned)(BX_OFF_A + lw * 2048);
            int kn = 0, sn = 0;                                  // tile-local index and ring slot of the next k-step to issue
            auto dma_next = [&]() {
                const int cv = kn >= BX_KS ? 1 : 0, kk = kn - cv * BX_KS;
                const int ch = kk >= 36 ? 2 : (kk >= 18 ? 1 : 0), i = kk - 18 * ch;
                const char *src = wsrc + (size_t)((cv * 2 + (ch == 2 ? 1 : 0)) * 18 + i) * BX_ASLOT;
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ldsw + (unsigned)sn * BX_ASLOT));
                pn_glds16_s<0>(src, lane16, dst);
                pn_glds16_s<0>(src + 1024, lane16, dst + 1024u);
                kn = kn == 2 * BX_KS - 1 ? 0 : kn + 1;
                sn = sn == BX_NSLOT - 1 ? 0 : sn + 1;
            };
            dma_next(); dma_next(); dma_next(); dma_next();     // k-steps 0..3
#ifdef BX_LAGA
            dma_next();                                          // ... and 4: the loop issues step g + 5 into the slot of step g (three steps of slack)
            asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");        // 0 and 1 have landed
#endif
            asm volatile("s_barrier" ::: "memory");              // (the compute waves have read k-step 0's fragments)
            for (; t < P.ntiles; t += gridDim.x) {
#pragma clang loop unroll(disable)
                for (int g = 0; g < 2 * BX_KS; ++g) {
#ifndef BX_FAKE_NODMA_A
                    dma_next();                                  // step g + 4 into the slot of step g - 1
#endif
#ifdef BX_FAKE_NOWAIT
                    BX_KBAR;
#elif defined(BX_FAKE_NOBAR)
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#elif defined(BX_LAGA)
                    asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");  // g + 2 has landed; g + 3 .. g + 5 in flight
#else
                    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");  // g + 2 has landed
#endif
                    if (g == BX_KS - 1) asm volatile("s_barrier" ::: "memory");    // the compute waves publish the intermediate image
                }
                asm volatile("s_barrier" ::: "memory");          // end of tile
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        // ---- image loaders: wave 6 fetches the hi plane, wave 7 the lo plane of the NEXT tile's input image (40 pieces of 32 px x 32 B each) ----
        const int pl = lw - 2;
        const int px = lane >> 1;
        const unsigned lane_c = (unsigned)(pl * P.in_split * 2 + (lane & 1) * 16);
        int gb, goy0, gox0, gR, gWc;
        size_t frame_b = 0;
        auto set_tile = [&](int tt) {
            tile_geom(tt, gb, goy0, gox0, gR, gWc);
            frame_b = ((size_t)gb * H * W * P.in_cs + P.in_coff) * 2;
        };
        auto dma_in = [&](int n) {                              // piece n = (quarter n / 10, halo row n % 10)
            const int qu = n / 10, row = n - qu * 10;
            const int iy = goy0 - 2 + row, ix = gox0 - 2 + px;
            const bool inb = px < gWc + 4 && row < gR + 4 && (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
            const unsigned off = inb ? (unsigned)((iy * W + ix) * P.in_cs * 2) + lane_c + (unsigned)(qu * 32) : P.in_zero_off - (unsigned)frame_b;
            pn_glds16_s<0>((const char *)P.in + frame_b, off, (unsigned)__builtin_amdgcn_readfirstlane((pl * 4 + qu) * BX_INQ + row * 1024));
        };
        set_tile(t);
#pragma clang loop unroll(disable)
        for (int n = 0; n < 40; ++n) dma_in(n);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        for (; t < P.ntiles; t += gridDim.x) {
            const bool more = t + (int)gridDim.x < P.ntiles;
            if (more) set_tile(t + (int)gridDim.x);
#pragma clang loop unroll(disable)
            for (int g = 0; g < 2 * BX_KS; ++g) {
#ifndef BX_FAKE_NODMA_IN
                if (more && g >= BX_KS && g < BX_KS + 40) dma_in(g - BX_KS);     // under conv2: the input image is free
#endif
                BX_KBAR;
                if (g == BX_KS - 1) asm volatile("s_barrier" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");         // end of tile: the next image is complete
        }
        return;
    }

    // ================= compute waves =================
    int t = blockIdx.x;
    if (t >= P.ntiles) return;
    float b1[16], b2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { b1[i] = P.bias1[16 * q + i]; b2[i] = P.bias2[16 * q + i]; }
    const int aaddr = BX_OFF_A + lane * 16;
    bf16x8 aq[2][CT], bq[BQ];
    asm volatile("s_barrier" ::: "memory");             // prologue: first image + weight k-steps 0, 1 landed
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) aq[0][ct] = *reinterpret_cast<const bf16x8 *>(smem + aaddr + ct * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int slot = 1;                                        // ring slot of the NEXT k-step's fragments
    const int out_ps = P.out_split * 2;                 // plane distance of the output tensor in bytes
    PN_STAMP_AT(0);
    for (int it = 0; t < P.ntiles; t += gridDim.x, ++it) {
        int b, oy0, ox0, R, Wc;
        tile_geom(t, b, oy0, ox0, R, Wc);
        if (it == 2) PN_STAMP_AT(1);                     // third tile: steady state
        const int MC = Wc + 2, nmid = (R + 2) * MC, nout = R * Wc;
        const float inv_mc = 1.0f / (float)MC, inv_wc = 1.0f / (float)Wc;
        int ba1[PT1], ba2[PT2];
#pragma unroll
        for (int pt = 0; pt < PT1; ++pt) {
            const int s0 = (wave * PT1 + pt) * 16 + c, s = s0 < nmid ? s0 : 0;
            const int r = (int)(((float)s + 0.5f) * inv_mc), x = s - r * MC;
            ba1[pt] = (q >> 1) * BX_INQ + (q & 1) * 16 + (r * 32 + x) * 32;
        }
#pragma unroll
        for (int pt = 0; pt < PT2; ++pt) {
            const int s0 = (wave * PT2 + pt) * 16 + c, s = s0 < nout ? s0 : 0;
            const int r = (int)(((float)s + 0.5f) * inv_wc), x = s - r * Wc;
            ba2[pt] = BX_OFF_MID + (q >> 1) * BX_MIDQ + (q & 1) * 16 + (r * BX_MIDP + x) * 32;
        }
        f32x4 acc[CT][PT1];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---------------- conv1: 54 k-steps on the input image (plane pair, half, tap) ----------------
#define BX_KOFF(ks, quarter, tapoff) ((((ks) / 18) == 1 ? 4 * (quarter) : 0) + ((((ks) % 18) / 9) * 2 * (quarter)) + (tapoff))
#define BX_TAP1(tap) ((((tap) / 3) * 32 + ((tap) % 3)) * 32)
#define BX_TAP2(tap) ((((tap) / 3) * BX_MIDP + ((tap) % 3)) * 32)
#define BX_OFF1(j) BX_KOFF((j) / PT1, BX_INQ, BX_TAP1(((j) / PT1) % 9))
#define BX_OFF2(j) BX_KOFF((j) / PT2, BX_MIDQ, BX_TAP2(((j) / PT2) % 9))
#pragma unroll
        for (int j = 0; j < BQ - 1; ++j) bq[j] = *reinterpret_cast<const bf16x8 *>(smem + ba1[j % PT1] + BX_OFF1(j));
#pragma clang loop unroll(full)
        for (int ph = 0; ph < BX_KS; ++ph) {
            const int an = aaddr + slot * BX_ASLOT;
            __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT1; ++pt) {
                const int j = ph * PT1 + pt, jr = j + BQ - 1;
                // the next step's four A fragments in the first two MFMA groups: the last one is >= 10 MFMAs old when its first MFMA issues
#ifndef BX_FAKE_NOA
#ifdef BX_A_G0
                if (pt == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) aq[(ph + 1) & 1][ct] = *reinterpret_cast<const bf16x8 *>(smem + an + ct * 1024);
                }
#else
                if (pt < 2) {
                    aq[(ph + 1) & 1][2 * pt] = *reinterpret_cast<const bf16x8 *>(smem + an + (2 * pt) * 1024);
                    aq[(ph + 1) & 1][2 * pt + 1] = *reinterpret_cast<const bf16x8 *>(smem + an + (2 * pt + 1) * 1024);
                }
#endif
#endif
#ifndef BX_FAKE_NOB
                if (jr < BX_KS * PT1) bq[jr % BQ] = *reinterpret_cast<const bf16x8 *>(smem + ba1[jr % PT1] + BX_OFF1(jr));
#endif
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ph & 1][ct], bq[j % BQ], acc[ct][pt], 0, 0, 0);
#ifdef BX_A_G0
                if (pt == 0 && jr < BX_KS * PT1) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                else if (pt == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                else if (jr < BX_KS * PT1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#else
                if (pt < 2 && jr < BX_KS * PT1) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                else if (pt < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (jr < BX_KS * PT1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == BX_NSLOT - 1 ? 0 : slot + 1;
            BX_KBAR;     // bare: the fragment reads of the next step stay in flight across it
        }
        if (it == 2) PN_STAMP_AT(2);
        // ---------------- residual: the centre of the input image, both planes, into registers (the image is refilled under conv2) ----------------
        u32x4 rres[PT2][4];
#pragma clang loop unroll(full)
        for (int pt = 0; pt < PT2; ++pt) {
            const int s0 = (wave * PT2 + pt) * 16 + c, s = s0 < nout ? s0 : 0;
            const int r = (int)(((float)s + 0.5f) * inv_wc), x = s - r * Wc;
            const char *rp = smem + q * BX_INQ + ((r + 2) * 32 + x + 2) * 32;
            rres[pt][0] = *reinterpret_cast<const u32x4 *>(rp);
            rres[pt][1] = *reinterpret_cast<const u32x4 *>(rp + 16);
            rres[pt][2] = *reinterpret_cast<const u32x4 *>(rp + 4 * BX_INQ);
            rres[pt][3] = *reinterpret_cast<const u32x4 *>(rp + 4 * BX_INQ + 16);
        }
        // ---------------- intermediate: bias + ReLU -> hi / lo bf16 planes -> LDS image (zero outside the map) ----------------
#pragma clang loop unroll(full)
        for (int pt = 0; pt < PT1; ++pt) {
            const int s = (wave * PT1 + pt) * 16 + c;
            const int r = (int)(((float)s + 0.5f) * inv_mc), x = s - r * MC;
            const int my = oy0 - 1 + r, mx = ox0 - 1 + x;
            const bool inside = (unsigned)my < (unsigned)H && (unsigned)mx < (unsigned)W;
            alignas(16) T hi[16]; alignas(16) T lo[16];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float v = acc[ct][pt][i] + b1[4 * ct + i];
                    v = v > 0.f ? v : 0.f;
                    if (!inside) v = 0.f;                      // conv2's zero padding
                    hi[4 * ct + i] = (T)v;
                    lo[4 * ct + i] = (T)(v - (float)hi[4 * ct + i]);
                }
            if (s < nmid) {
                char *dst = smem + BX_OFF_MID + q * BX_MIDQ + (r * BX_MIDP + x) * 32;
                *reinterpret_cast<u32x4 *>(dst) = *reinterpret_cast<const u32x4 *>(&hi[0]);
                *reinterpret_cast<u32x4 *>(dst + 16) = *reinterpret_cast<const u32x4 *>(&hi[8]);
                *reinterpret_cast<u32x4 *>(dst + 4 * BX_MIDQ) = *reinterpret_cast<const u32x4 *>(&lo[0]);
                *reinterpret_cast<u32x4 *>(dst + 4 * BX_MIDQ + 16) = *reinterpret_cast<const u32x4 *>(&lo[8]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // intermediate image published; the input image is free
        if (it == 2) PN_STAMP_AT(3);
        // ---------------- conv2: 54 k-steps on the intermediate image ----------------
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < BQ - 1; ++j) bq[j] = *reinterpret_cast<const bf16x8 *>(smem + ba2[j % PT2] + BX_OFF2(j));
#pragma clang loop unroll(full)
        for (int ph = 0; ph < BX_KS; ++ph) {
            const int an = aaddr + slot * BX_ASLOT;
            __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT2; ++pt) {
                const int j = ph * PT2 + pt, jr = j + BQ - 1;
#ifndef BX_FAKE_NOA
#ifdef BX_A_G0
                if (pt == 0) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) aq[(ph + 1) & 1][ct] = *reinterpret_cast<const bf16x8 *>(smem + an + ct * 1024);
                }
#else
                if (pt < 2) {
                    aq[(ph + 1) & 1][2 * pt] = *reinterpret_cast<const bf16x8 *>(smem + an + (2 * pt) * 1024);
                    aq[(ph + 1) & 1][2 * pt + 1] = *reinterpret_cast<const bf16x8 *>(smem + an + (2 * pt + 1) * 1024);
                }
#endif
#endif
#ifndef BX_FAKE_NOB
                if (jr < BX_KS * PT2) bq[jr % BQ] = *reinterpret_cast<const bf16x8 *>(smem + ba2[jr % PT2] + BX_OFF2(jr));
#endif
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ph & 1][ct], bq[j % BQ], acc[ct][pt], 0, 0, 0);
#ifdef BX_A_G0
                if (pt == 0 && jr < BX_KS * PT2) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                else if (pt == 0) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                else if (jr < BX_KS * PT2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#else
                if (pt < 2 && jr < BX_KS * PT2) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                else if (pt < 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (jr < BX_KS * PT2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
                __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            slot = slot == BX_NSLOT - 1 ? 0 : slot + 1;
            BX_KBAR;
        }
        if (it == 2) PN_STAMP_AT(4);
        // ---------------- output: (acc + b2) + x_hi + x_lo, ReLU, hi / lo split, two planes of 2 x 16-B stores per pixel ----------------
        {
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                (char *)P.out + ((size_t)b * H * W * P.out_cs + P.out_coff) * 2, 0, (int)((size_t)H * W * P.out_cs * 2), 0x00020000);
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT2; ++pt) {
                const int s = (wave * PT2 + pt) * 16 + c;
                const int r = (int)(((float)s + 0.5f) * inv_wc), x = s - r * Wc;
                const bool valid = s < nout;
                alignas(16) T hi[16]; alignas(16) T lo[16];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = 4 * ct + i;
                        // a bf16 is the upper half of its float: channel k of the lane's 16 is half (k & 1) of dword k / 2
                        const unsigned wh = rres[pt][k >> 3][(k >> 1) & 3], wl = rres[pt][2 + (k >> 3)][(k >> 1) & 3];
                        float v = acc[ct][pt][i] + b2[k];
                        v += __builtin_bit_cast(float, (k & 1) ? (wh & 0xffff0000u) : (wh << 16));
                        v += __builtin_bit_cast(float, (k & 1) ? (wl & 0xffff0000u) : (wl << 16));
                        v = v > 0.f ? v : 0.f;
                        hi[k] = (T)v;
                        lo[k] = (T)(v - (float)hi[k]);
                    }
                // out-of-range offset for the unused slots: the store is issued unconditionally and dropped by the hardware
                const unsigned voff = valid ? (unsigned)(((oy0 + r) * W + ox0 + x) * P.out_cs * 2 + 32 * q) : 0x80000000u;
                const u32x4 h0 = *reinterpret_cast<const u32x4 *>(&hi[0]), h1 = *reinterpret_cast<const u32x4 *>(&hi[8]);
                __builtin_amdgcn_raw_buffer_store_b128(h0, orsrc, voff, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(h1, orsrc, voff + 16u, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&lo[0]), orsrc, voff + (unsigned)out_ps, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4 *>(&lo[8]), orsrc, voff + (unsigned)out_ps + 16u, 0, 0);
            }
        }
        asm volatile("s_barrier" ::: "memory");         // end of tile: the next input image has landed (the image loaders waited for it)
        if (it == 2) PN_STAMP_AT(5);
    }
    PN_STAMP_AT(12);
#undef BX_KOFF
#undef BX_TAP1
#undef BX_TAP2
#undef BX_OFF1
#undef BX_OFF2
}

static int bb64x3_launch(pn_ctx *ctx, const BBProblem &P, int num_cus, hipStream_t stream) {
    static PnLdsAttr attr;
    if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(bb64x3_kernel), BX_LDS)) return rc;
    const int grid = P.ntiles < num_cus ? P.ntiles : num_cus;
    hipLaunchKernelGGL(bb64x3_kernel, dim3(grid), dim3(512), BX_LDS, stream, P);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}
