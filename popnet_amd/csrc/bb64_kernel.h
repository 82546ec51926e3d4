// Fused ResNet BasicBlock(64) on the CDNA4 matrix cores (bf16 storage, fp32 accumulate):
//     y = relu( conv3x3(relu(conv3x3(x, W1) + b1), W2) + b2 + x )            64 -> 64 -> 64 channels, stride 1
// replaces the two launches per block of tpm/lib/network/rtpose_light3d.py:48-72 (BasicBlock.forward) for model0.layer1
// (tpm/lib/network/rtpose_light3d.py:145-152) and resnet.BasicBlock (tpm/lib/network/resnet.py:27-56) for YoloPoseNet's
// layer1.  BatchNorm is folded into W / b on the host (net.hip).
//
// Why: at 112x112 these layers are not arithmetic: unfused, a BasicBlock moves 297 MB per 32 frames (conv1 reads x with
// its halo and writes the 51 MB intermediate, conv2 reads it back with halo, reads x again as the residual and writes
// y) in memory-bound prologue / epilogue phases that every resident block runs in lockstep (45 us per conv for 12 us of
// MFMA work, profiles/README.md r01 v21).  Here the intermediate never leaves the CU and x is read once:
//   * one PERSISTENT workgroup per CU walks its share of the 8-row x <=28-column output tiles;
//   * per tile the 12 x 32-pixel input halo (48 KB, quarter-major [quarter][row][32 px][32 B]) sits in LDS, conv1 is evaluated on
//     the 10 x 30 intermediate halo (1.34x recompute) straight into a second LDS image (40 KB, zero outside the map =
//     conv2's padding), conv2 reads that image, the residual comes from the CENTRE of the input image (no second read
//     of x), and y leaves with 16-B stores;
//   * the input image is double-buffered: the next tile's image is fetched during this tile's conv1, the stores of
//     this tile drain under the next tile's conv1 -- no lockstep memory phases;
//   * 8 waves (HV = 1) with FIXED ROLES: waves 0-3 compute (each 64 couts x 5 / 4 pixel tiles: 20 / 16 MFMAs per k-step, B
//     fragments from the images, A fragments from a 6-slot LDS ring), waves 4-7 only issue LDS-DMA and count their own
//     vmcnt -- a compute wave's instruction stream is MFMA + ds_read only (conv4_kernel's ablations: DMA issue and its
//     waits cost a lone wave 80 of 717 cycles per k-step).  Round 3: the loaders are split by STREAM -- waves 4, 5 fetch
//     weights (W1 | W2 as one periodic 36-step stream, L2-resident: one 4 KB k-step each per phase, landed one phase later),
//     waves 6, 7 the next tile's input image (HBM / Infinity Cache latency) and wait for it only at the end of the tile.
//     vmcnt retires in issue order, so with both streams on one wave (round 2) every phase of conv1 waited for an image
//     piece issued one phase earlier: 530 cycles per 320-cycle phase against 355 per 256 in conv2 (scripts/bblab.hip);
//   * one barrier per TWO k-steps (a bare s_barrier for the compute waves: their fragment reads stay in flight across it):
//     20 barriers per tile instead of 38; 160 KB of LDS, one workgroup per CU.
//   * round 5: the tiles are handed out by TICKET (one atomic per tile, fetched a tile ahead by an image-loader wave and published through sixteen spare
//     bytes of the intermediate image), not by blockIdx.x + k * gridDim.x: a persistent workgroup can only start on a CU that holds no other wave, so among
//     the other in-flight batches' launches some workgroups start late -- with the static schedule the launch ended when the LATEST starter had walked its
//     seven tiles; now the early ones take its share.  Which workgroup computes which tile changes, nothing else.
// Weight pack (net.hip): [36 k-steps = (conv, half, tap)][4 cout tiles][64 lanes][8 bf16], rows permuted with
// pn_conv_row_channel(tile, row, 4) so that a lane's 16 accumulators are 16 consecutive channels.
#pragma once
#include "conv3_kernel.h"


// LDS images are QUARTER-major: [4 quarters of 16 channels][row][32 px][32 B].  A lane's 16-B B fragment (8 channels) is
// half of a pixel's 32-B quarter entry, so the 16 lanes one ds_read_b128 phase serves (8 with an even, 8 with an odd lane
// quarter q) cover 8 consecutive pixels x 32 B = one whole 256-B bank row: conflict-free, where the half-major image of
// conv3 / conv4 ([row][px][64 B]) costs 2 LDS cycles per read -- this kernel's 4 compute waves would keep the LDS array
// ~100 % busy with it (20 MFMAs per 9 fragment reads).  One DMA instruction = one 32-pixel row of one quarter.
#define BB_INQ (12 * 32 * 32)
#define BB_IN (4 * BB_INQ)                    // 48 KB
#define BB_MIDQ (10 * 32 * 32)
#define BB_MID (4 * BB_MIDQ)                  // 40 KB
#define BB_ASLOT 4096
#define BB_OFF_MID (2 * BB_IN)
#define BB_OFF_A (BB_OFF_MID + BB_MID)
#define BB_NSLOT 6                              // weight ring: a phase = 2 k-steps; phase p + 2 lands in the slots of phase p - 1
#define BB_LDS (BB_OFF_A + BB_NSLOT * BB_ASLOT) // 163840 B = all of a CU's LDS

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;
// ReLU of two packed bf16: the sign bit of a bf16 is the sign bit of the int16 with the same bits
__device__ __forceinline__ unsigned bb_relu_pk(unsigned w) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(i16x2, w), i16x2{0, 0}));
}

// HV = 1 (default): four compute waves of 64 couts.  HV = 2 (round 5, POPNET_BB64_HALVES=2): EIGHT compute waves -- the four pixel groups times two
// cout halves, two per SIMD, so that one wave's fragment reads and barrier waits hide behind the other's MFMAs.  Same LDS images, same weight ring,
// same k order per output: bit-identical.  Measured (scripts/bblab.hip, 32 x 112 x 112): the third tile takes 17 877 shader cycles instead of 19 595
// (the intermediate write 1 944 instead of 3 057), the whole kernel 126 k instead of 136 k -- and 73.2-74.2 us instead of 71.8-72.1: the in-kernel
// clock read 1.56 GHz against 1.71.  The package draws 1 370 W of its 1 400 W limit under this kernel (rocm-smi, profiles/r05_power.txt): twelve
// waves and 55 % more LDS bytes per k-step (both halves read every B fragment) cost the clock what they save in cycles.  Kept as a switch.
template <int HV>
__global__ __launch_bounds__((4 * HV + 4) * 64, 1) void bb64_kernel(const BBProblem P) {
    typedef __bf16 T;
    constexpr int CT = 4 / HV, NCW = 4 * HV, NP = CT / 2, PT1 = 5, PT2 = 4, KK = 9, BQ = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    const int W = P.W, H = P.H;

    // ticket slots: pixel column 31 of row 0 of the intermediate image's first quarter plane is never written (MC <= 30) nor read (x + kx <= 29)
    volatile int *s_tile = reinterpret_cast<volatile int *>(smem + BB_OFF_MID + 31 * 32);
    const bool dyn = P.tickets != nullptr;
    // tile k + 1 of this workgroup, as every wave sees it at the start of tile k (static schedule: t + gridDim.x)
    auto next_tile = [&](int t, int k) -> int { return dyn ? __builtin_amdgcn_readfirstlane(s_tile[(k + 1) & 3]) : t + (int)gridDim.x; };
    // (the compute and weight-loader waves need it at the END of tile k only: they issue the LDS read at the start and look at it there)
    auto next_tile_raw = [&](int t, int k) -> int { return dyn ? s_tile[(k + 1) & 3] : t + (int)gridDim.x; };
    auto tile_geom = [&](int t, int &b, int &oy0, int &ox0, int &R, int &Wc) {
        b = t / P.tiles_per_img;
        const int rem = t - b * P.tiles_per_img;
        const int ty = rem / P.tiles_x, tx = rem - ty * P.tiles_x;
        oy0 = ty * 8; ox0 = tx * P.Wt;
        R = min(8, H - oy0); Wc = min(P.Wt, W - ox0);
    };

    if (wave >= NCW) {
        // ================= loader waves: LDS-DMA only =================
        const int lw = wave - NCW;
        const unsigned lane16 = (unsigned)lane * 16u;
        int t = blockIdx.x;
        if (t >= P.ntiles) return;
        if (lw < 2) {
            // ---- weight loaders: each fetches half (two 1 KB instructions) of both k-steps of a phase ----
            // Ring of 6 slots, k-step k in slot k % 6; the compute waves read the fragments of k-step k during k - 1.  In phase q
            // (k-steps 2q, 2q + 1) the loaders issue k-step 2q + 4 and then 2q + 5 into the slots phase q - 1 has just released;
            // 2q + 4 is read in phase q + 1 and must have landed by the barrier that ends phase q (vmcnt leaves only the two
            // newer instructions = 2q + 5 in flight), 2q + 5 is read in phase q + 2 and gets a phase of slack.
            const char *wsrc = (const char *)P.wpack + lw * 2048;
            auto dma_a = [&](int kstep) {                        // this wave's half of a k-step of the periodic 36-step stream
                pn_glds16_s<0>(wsrc + (size_t)kstep * BB_ASLOT, lane16, (unsigned)(BB_OFF_A + (kstep % BB_NSLOT) * BB_ASLOT) + (unsigned)__builtin_amdgcn_readfirstlane(lw * 2048));
                pn_glds16_s<0>(wsrc + (size_t)kstep * BB_ASLOT + 1024, lane16, (unsigned)(BB_OFF_A + (kstep % BB_NSLOT) * BB_ASLOT + 1024) + (unsigned)__builtin_amdgcn_readfirstlane(lw * 2048));
            };
            dma_a(0); dma_a(1); dma_a(2); dma_a(3);             // phases 0 and 1
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");              // (the compute waves have read k-step 0's fragments)
            for (int k = 0; t < P.ntiles; ++k) {
                const int tnx = next_tile_raw(t, k);
#pragma clang loop unroll(full)
                for (int p = 0; p < 18; ++p) {                   // phase p = k-steps 2p, 2p + 1 of the tile's 36
#ifndef BB_FAKE_NODMA_A
                    dma_a((2 * p + 4) % 36);
                    dma_a((2 * p + 5) % 36);
                    asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
#else
                    asm volatile("s_barrier" ::: "memory");
#endif
                    if (p == 8) asm volatile("s_barrier" ::: "memory");          // the compute waves publish the intermediate image
                }
                asm volatile("s_barrier" ::: "memory");                             // end of tile
                t = __builtin_amdgcn_readfirstlane(tnx);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        // ---- image loaders: wave 6 / 7 fetches pieces 0..23 / 24..47 of the NEXT tile's input image, two per phase ----
        auto dma_in = [&](int tt, int buf, int n) {            // piece n = (quarter n / 12, halo row n % 12): 32 pixels x 32 B
            int b, oy0, ox0, R, Wc;
            tile_geom(tt, b, oy0, ox0, R, Wc);
            const int qu = n / 12, row = n % 12;
            const int px = lane >> 1;
            const int iy = oy0 - 2 + row, ix = ox0 - 2 + px;
            const size_t frame_b = ((size_t)b * H * W * P.in_cs + P.in_coff) * 2;
            const bool inb = px < Wc + 4 && (unsigned)ix < (unsigned)W && (unsigned)iy < (unsigned)H;
            const unsigned off = inb ? (unsigned)((iy * W + ix) * P.in_cs * 2 + qu * 32 + (lane & 1) * 16) : P.in_zero_off - (unsigned)frame_b;
            pn_glds16_s<0>((const char *)P.in + frame_b, off, (unsigned)__builtin_amdgcn_readfirstlane(buf * BB_IN + qu * BB_INQ + row * 1024));
        };
        const int n0 = (lw - 2) * 24;
        const bool ticketer = dyn && lw == 2 && lane == 0;      // this lane fetches the tickets: tile k + 2 during tile k
        int ticket = 0;
        // (hybrids -- the first tiles of a workgroup by position, only its last 1, 2 or 3 by ticket -- measured no better than the static schedule:
        //  profiles/r05_notes.txt)
        // The first TWO tiles of a workgroup are its position (the prologue waits for no atomic).  Launches with fewer than four tiles per workgroup
        // (56 x 56 maps: 448 tiles on 256 workgroups) keep the static schedule altogether (net.hip passes no counters): with the second tile a ticket
        // YoloPoseNet lost 1.5 % (96.1 / 95.9 / 96.3 k against 98.0 / 97.7 / 96.4 k frames/s, same box, profiles/r05_notes.txt).
        auto claim = [&]() -> int { return 2 * (int)gridDim.x + atomicAdd(P.tickets, 1); };
        if (ticketer) ticket = t + (int)gridDim.x;
#pragma unroll
        for (int j = 0; j < 24; ++j) dma_in(t, 0, n0 + j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (ticketer) s_tile[1] = ticket;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        int cur = 0;
        for (int k = 0; t < P.ntiles; ++k) {
            const int tnx = next_tile(t, k);
            const int tn = tnx < P.ntiles ? tnx : t;             // past the last tile: a harmless refetch into the idle image
            // a workgroup on its LAST tile has no use for another ticket: it signs off instead (looked at after the end-of-tile wait, ten microseconds
            // later: the kernel's end does not wait for an atomic's round trip).  The workgroup that signs off last knows every claim has been made and
            // leaves both counters at zero for the next launch (graph replay).
            int signed_off = -1;
            if (ticketer) {
                if (tnx < P.ntiles) ticket = claim();
                else signed_off = atomicAdd(P.tickets + 1, 1);
            }
#pragma clang loop unroll(full)
            for (int p = 0; p < 18; ++p) {
#ifndef BB_FAKE_NODMA_IN
                if (p < 12) { dma_in(tn, cur ^ 1, n0 + 2 * p); dma_in(tn, cur ^ 1, n0 + 2 * p + 1); }
#endif
                asm volatile("s_barrier" ::: "memory");
                if (p == 8) asm volatile("s_barrier" ::: "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // end of tile: the next image is complete (and the ticket is back)
            if (ticketer) {
                s_tile[(k + 2) & 3] = ticket;
                if (signed_off == (int)gridDim.x - 1) {
                    __hip_atomic_store(P.tickets, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(P.tickets + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            cur ^= 1;
            t = tnx;
        }
        return;
    }

    // ================= compute waves =================
    int t = blockIdx.x;
    if (t >= P.ntiles) return;
    const float *bias1 = P.bias1, *bias2 = P.bias2;
    const int pw = HV == 2 ? (wave & 3) : wave, hv = HV == 2 ? (wave >> 2) : 0;            // pixel group, cout half
    float b1[4 * CT], b2[4 * CT];
#pragma unroll
    for (int i = 0; i < 4 * CT; ++i) { b1[i] = bias1[16 * q + 4 * CT * hv + i]; b2[i] = bias2[16 * q + 4 * CT * hv + i]; }
    const int aaddr = BB_OFF_A + lane * 16 + hv * CT * 1024;
    bf16x8 aq[2][CT], bq[BQ];
    asm volatile("s_barrier" ::: "memory");             // prologue: first image + weight k-steps 0..2 landed
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) aq[0][ct] = *reinterpret_cast<const bf16x8 *>(smem + aaddr + ct * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int cur = 0;
    PN_STAMP_AT(0);
    for (int it = 0; t < P.ntiles; ++it) {
        const int tnx = next_tile_raw(t, it);
        int b, oy0, ox0, R, Wc;
        tile_geom(t, b, oy0, ox0, R, Wc);
        if (it == 2) PN_STAMP_AT(1);                     // third tile: steady state
        const int MC = Wc + 2, nmid = (R + 2) * MC, nout = R * Wc;
        const float inv_mc = 1.0f / (float)MC, inv_wc = 1.0f / (float)Wc;
        const int inb = cur * BB_IN;
        // per-lane image addresses (tap and half are immediates)
        int ba1[PT1], ba2[PT2];
#pragma unroll
        for (int pt = 0; pt < PT1; ++pt) {
            const int s0 = (pw * PT1 + pt) * 16 + c, s = s0 < nmid ? s0 : 0;
            const int r = (int)(((float)s + 0.5f) * inv_mc), x = s - r * MC;
            ba1[pt] = inb + (q >> 1) * BB_INQ + (q & 1) * 16 + (r * 32 + x) * 32;
        }
#pragma unroll
        for (int pt = 0; pt < PT2; ++pt) {
            const int s0 = (pw * PT2 + pt) * 16 + c, s = s0 < nout ? s0 : 0;
            const int r = (int)(((float)s + 0.5f) * inv_wc), x = s - r * Wc;
            ba2[pt] = BB_OFF_MID + (q >> 1) * BB_MIDQ + (q & 1) * 16 + (r * 32 + x) * 32;
        }
        f32x4 acc[CT][PT1];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT1; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (aq[0] holds k-step 0's weight fragments: read in the prologue / prefetched by the previous tile's last phase)
        // ---------------- conv1: 18 k-steps on the input image ----------------
#define BB_TAPOFF(tap) ((((tap) / 3) * 32 + ((tap) % 3)) * 32)
#define BB_OFF1(j) ((((j) / PT1) / KK) * 2 * BB_INQ + BB_TAPOFF(((j) / PT1) % KK))
#pragma unroll
        for (int j = 0; j < BQ - 1; ++j) bq[j] = *reinterpret_cast<const bf16x8 *>(smem + ba1[j % PT1] + BB_OFF1(j));
#pragma clang loop unroll(full)
        for (int ph = 0; ph < 18; ++ph) {
            __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT1; ++pt) {
                const int j = ph * PT1 + pt, jr = j + BQ - 1;
                if (pt < CT)
                    aq[(ph + 1) & 1][pt] = *reinterpret_cast<const bf16x8 *>(smem + aaddr + ((ph + 1) % BB_NSLOT) * BB_ASLOT + pt * 1024);
                if (jr < 18 * PT1) bq[jr % BQ] = *reinterpret_cast<const bf16x8 *>(smem + ba1[jr % PT1] + BB_OFF1(jr));
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ph & 1][ct], bq[j % BQ], acc[ct][pt], 0, 0, 0);
                if (pt < CT && jr < 18 * PT1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (pt < CT || jr < 18 * PT1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // one barrier per two k-steps (no lgkmcnt wait: the ring slots refilled after it were last read a phase ago, the
            // images are stable)
            if (ph & 1) asm volatile("s_barrier" ::: "memory");
        }
        if (it == 2) PN_STAMP_AT(2);
        // ---------------- intermediate: bias + ReLU -> bf16 -> LDS image (zero outside the map) ----------------
#pragma clang loop unroll(full)
        for (int pt = 0; pt < PT1; ++pt) {
            const int s = (pw * PT1 + pt) * 16 + c;
            const int r = (int)(((float)s + 0.5f) * inv_mc), x = s - r * MC;
            const int my = oy0 - 1 + r, mx = ox0 - 1 + x;
            const bool inside = (unsigned)my < (unsigned)H && (unsigned)mx < (unsigned)W;
            const bool border = oy0 == 0 || ox0 == 0 || oy0 + R >= H || ox0 + Wc >= W;        // wave-uniform
            // bias + ReLU as add + max, two channels per v_cvt_pk_bf16_f32; the zero padding outside the map is applied to the
            // 8 packed dwords (per pixel), not per channel: a lone wave pays every VALU instruction of this block in full
            // ReLU on the PACKED bf16 pair (a negative bf16 is a negative int16: one v_pk_max_i16 per two channels; rounding is
            // monotonic, so relu(bf16(v)) == bf16(relu(v)))
            u32x4 o[NP];                                 // this lane's 8 * NP channels of the pixel: 16-byte pieces hv * NP .. of its quarter entry
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f32x2 lo = {acc[ct][pt][0] + b1[4 * ct + 0], acc[ct][pt][1] + b1[4 * ct + 1]};
                f32x2 hi = {acc[ct][pt][2] + b1[4 * ct + 2], acc[ct][pt][3] + b1[4 * ct + 3]};
                o[ct >> 1][2 * (ct & 1)] = bb_relu_pk(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)));
                o[ct >> 1][2 * (ct & 1) + 1] = bb_relu_pk(__builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2)));
            }
            if (border && !inside) {                     // conv2's zero padding; interior tiles skip the test
#pragma unroll
                for (int k = 0; k < NP; ++k) o[k] = u32x4{0u, 0u, 0u, 0u};
            }
            if (s < nmid) {
                u32x4 *dst = reinterpret_cast<u32x4 *>(smem + BB_OFF_MID + q * BB_MIDQ + (r * 32 + x) * 32) + hv * NP;
#pragma unroll
                for (int k = 0; k < NP; ++k) dst[k] = o[k];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (it == 2) PN_STAMP_AT(3);
        // ---------------- conv2: 18 k-steps on the intermediate image ----------------
#define BB_OFF2(j) ((((j) / PT2) / KK) * 2 * BB_MIDQ + BB_TAPOFF(((j) / PT2) % KK))
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < PT2; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < BQ - 1; ++j) bq[j] = *reinterpret_cast<const bf16x8 *>(smem + ba2[j % PT2] + BB_OFF2(j));
#pragma clang loop unroll(full)
        for (int p2 = 0; p2 < 18; ++p2) {
            const int ph = 18 + p2;
            __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT2; ++pt) {
                const int j = p2 * PT2 + pt, jr = j + BQ - 1;
                if (pt < CT)
                    aq[(ph + 1) & 1][pt] = *reinterpret_cast<const bf16x8 *>(smem + aaddr + ((ph + 1) % BB_NSLOT) * BB_ASLOT + pt * 1024);
                if (jr < 18 * PT2) bq[jr % BQ] = *reinterpret_cast<const bf16x8 *>(smem + ba2[jr % PT2] + BB_OFF2(jr));
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ph & 1][ct], bq[j % BQ], acc[ct][pt], 0, 0, 0);
                if (pt < CT && jr < 18 * PT2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (pt < CT || jr < 18 * PT2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ph & 1) asm volatile("s_barrier" ::: "memory");
        }
        if (it == 2) PN_STAMP_AT(4);
        // ---------------- output: bias + residual (centre of the input image) + ReLU, 2 x 16-B stores per pixel ----------------
        {
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
                (char *)P.out + ((size_t)b * H * W * P.out_cs + P.out_coff) * 2, 0, (int)((size_t)H * W * P.out_cs * 2), 0x00020000);
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT2; ++pt) {
                const int s = (pw * PT2 + pt) * 16 + c;
                const int r = (int)(((float)s + 0.5f) * inv_wc), x = s - r * Wc;
                const bool valid = s < nout;
                const u32x4 *rp = reinterpret_cast<const u32x4 *>(smem + inb + q * BB_INQ + (((valid ? r : 0) + 2) * 32 + (valid ? x : 0) + 2) * 32);
                u32x4 rr[NP], o[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) rr[k] = rp[hv * NP + k];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const unsigned ra = rr[ct >> 1][2 * (ct & 1)], rb = rr[ct >> 1][2 * (ct & 1) + 1];
                    // a bf16 is the upper half of its float: residual channels by shift / mask, no conversion instruction
                    f32x2 lo = {acc[ct][pt][0] + b2[4 * ct + 0] + __builtin_bit_cast(float, ra << 16),
                                acc[ct][pt][1] + b2[4 * ct + 1] + __builtin_bit_cast(float, ra & 0xffff0000u)};
                    f32x2 hi = {acc[ct][pt][2] + b2[4 * ct + 2] + __builtin_bit_cast(float, rb << 16),
                                acc[ct][pt][3] + b2[4 * ct + 3] + __builtin_bit_cast(float, rb & 0xffff0000u)};
                    o[ct >> 1][2 * (ct & 1)] = bb_relu_pk(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2)));
                    o[ct >> 1][2 * (ct & 1) + 1] = bb_relu_pk(__builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2)));
                }
                // out-of-range offset for the unused slots: the store is issued unconditionally and dropped by the hardware
                const unsigned voff = valid ? (unsigned)(((oy0 + r) * W + ox0 + x) * P.out_cs * 2 + 32 * q) : 0x80000000u;
#pragma unroll
                for (int k = 0; k < NP; ++k) __builtin_amdgcn_raw_buffer_store_b128(o[k], orsrc, voff + 16u * (unsigned)(hv * NP + k), 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the loader may now refill this image's buffer
        if (it == 2) PN_STAMP_AT(5);
        cur ^= 1;
        t = __builtin_amdgcn_readfirstlane(tnx);
    }
    PN_STAMP_AT(12);
#undef BB_OFF1
#undef BB_OFF2
#undef BB_TAPOFF
}

static int bb64_launch(pn_ctx *ctx, const BBProblem &P, int num_cus, hipStream_t stream) {
    static PnLdsAttr attr[2];
    const int halves = P.halves == 2 ? 2 : 1;        // 2 (POPNET_BB64_HALVES=2 when the net was compiled): the eight-compute-wave form -- 7 % fewer cycles, 2-3 % MORE time
    if (int rc = pn_lds_attr(ctx, attr[halves - 1], halves == 2 ? reinterpret_cast<const void *>(bb64_kernel<2>) : reinterpret_cast<const void *>(bb64_kernel<1>), BB_LDS)) return rc;
    // experiment switch: POPNET_BB64_CUS = workgroups of the persistent launch (default: one per CU).  Fewer leave whole CUs to the
    // kernels of other streams while this one runs (a bb64 workgroup owns its CU's LDS).
    static int cap = -1;
    if (cap < 0) { const char *e = getenv("POPNET_BB64_CUS"); cap = e ? atoi(e) : 0; }
    if (cap > 0 && cap < num_cus) num_cus = cap;
    const int grid = P.ntiles < num_cus ? P.ntiles : num_cus;
    if (halves == 2) hipLaunchKernelGGL(bb64_kernel<2>, dim3(grid), dim3(768), BB_LDS, stream, P);
    else hipLaunchKernelGGL(bb64_kernel<1>, dim3(grid), dim3(512), BB_LDS, stream, P);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}
