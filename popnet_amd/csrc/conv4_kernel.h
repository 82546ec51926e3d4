// bf16 3x3 stride-1 "same" convolution on the CDNA4 matrix cores, third generation: BOTH operands through LDS.
//
// Same math and epilogue contract as conv3_kernel.h / conv_mfma_kernel.h (it replaces the same reference code:
// tpm/lib/network/rtpose_light3d.py:24-72,222-246; yolo_posenet.py:101-126; resnet.py:27-56) for the layers that
// dominate the conv time: 3x3, Cin >= 128, >= 64 couts, maps that split into <= 30-column strips.
//
// Why (round-1 counters, profiles/r01_final_pmc_ta_vs_mfma.txt): conv3_kernel's 128-cout x 112-pixel blocks stream their
// whole weight slice global -> VGPR per block (146 B per MFMA through the per-CU vector-memory path: TA busy 1.18x the
// matrix pipe) and every wave re-reads all 7 activation fragments per 14 MFMAs (LDS array ~100 % busy with the 2-way
// conflict of the half-major image).  Here
//   * a wave owns 64 couts x 112 pixels (4 x 7 accumulator tiles, 28 MFMAs per k-step): 11 LDS fragment reads per 28
//     MFMAs instead of 7 per 14;
//   * a block = 2 cout halves x 2 pixel strips (128 couts x 224 pixels, 4 waves); the weight fragments of a k-step
//     (8 KB) are fetched ONCE per block by LDS-DMA into a 3-slot ring and shared by the two strips: 95 B per MFMA
//     through the vector-memory path (weights + halo) instead of 146 + halo;
//   * the two strips are independent 4-row x <= 28-column tiles (own halo images), so a 28-row map wastes nothing;
//   * activations: quarter-major halo images ([2 quarters][halo row][32 px][32 B] per 32-channel half: conflict-free
//     fragment reads, see below), double-buffered
//     per HALF (12 KB per strip and buffer): the next half's image is fetched, one DMA instruction per k-step, while
//     the current half's 9 taps run -- no hand-over stall;
//   * ONE barrier per k-step (28 MFMAs per wave), LDS-DMA kept in flight across it with counted vmcnt (the guide's
//     "pipelining across barriers": raw s_barrier, never vmcnt(0) inside the loop);
//   * <= 256 VGPRs, 2 waves / SIMD, 72 KB of LDS: two blocks per CU.
// Weight pack (net.hip::pack_conv4): [cout block of 128][k-step][8 cout tiles][64 lanes][8 bf16]: one k-step of a block
// is 8 KB contiguous, each DMA instruction copies 1 KB of consecutive bytes.
// k order: (32-channel half hh, tap) -- identical to conv3's (chunk, half, tap).
#pragma once
#include "conv3_kernel.h"

#define PN4_ASLOT 8192                  // one k-step of weight fragments: 8 cout tiles x 1 KB
#ifndef PN4_RING
#define PN4_RING 3                      // weight ring slots (3: slot = immediate; 4: runtime slot, no LDS-read drain at the barrier)
#endif
#ifndef PN4_PITCH
#define PN4_PITCH 32                    // halo pixels per LDS row (36: a 16-pixel tile that wraps a 28-pixel row stays conflict-free)
#endif
#ifndef PN4_LGKM
#define PN4_LGKM 0                      // LDS reads left in flight at the k-step barrier (3 = the three B reads issued last)
#endif
#define PN4_ARING (PN4_RING * PN4_ASLOT)
#define PN4_BQUART (5 * PN4_PITCH * 32 + 1024)   // one 16-channel quarter plane of a strip image: 6 halo rows x PITCH px x 32 B (the last row: the 1 KiB a DMA writes)
#define PN4_BSTRIP (2 * PN4_BQUART)     // one 32-channel half image of one strip
#define PN4_BBUF (2 * PN4_BSTRIP)       // both strips
#define PN4_LDS (PN4_ARING + 2 * PN4_BBUF)

// One block of problem P; block_x = the block's index within the problem's launch (blockIdx.x of conv4_kernel; the persistent
// experiment of scripts/stagelab.hip walks several (problem, block_x) pairs per workgroup).
__device__ __forceinline__ void conv4_body(const ConvProblem &P, const int block_x) {
    typedef __bf16 T;
    constexpr int KS = 3, KK = 9, PT = 7, CT = 4, PITCH = PN4_PITCH, RING = PN4_RING;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    if (block_x >= P.nblocks) return;
    int bx;
    {   // XCD-aware remap, see conv_mfma_kernel.h
        const int nb = P.nblocks, xcd = block_x & 7, idx = block_x >> 3;
        const int qq = nb >> 3, rr = nb & 7;
        bx = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int tid = threadIdx.x;
    PN_STAMP_AT(0);
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wc = wave & 1;            // strip, cout half
    const int c = lane & 15, q = lane >> 4;

    const int cb = bx % P.cout_blocks;
    const int pair = bx / P.cout_blocks;
    const int nstrips = P.B * P.tiles_per_img;
    const int sid_raw = pair * 2 + wp;
    const bool strip_ok = sid_raw < nstrips;             // an odd strip count leaves the last block half empty
    const int sid = strip_ok ? sid_raw : nstrips - 1;
    const int tile = sid % P.tiles_per_img;
    const int b = sid / P.tiles_per_img;
    const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const int oy0 = ty * P.R, ox0 = tx * P.Wt;
    const int R = min(P.R, P.Ho - oy0);
    const int Wo = P.Wo;
    const int Wc = min(P.Wt, Wo - ox0);
    const int npix = R * Wc;
    const int HC = Wc + KS - 1;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const float inv_wc = 1.0f / (float)Wc;
    const int nchunks = P.cin_chunks;

    // ---- weight ring: wave w fetches cout tiles 2w, 2w+1 of every k-step (2 x 1 KB of consecutive bytes) ----
    const char *aptr = (const char *)P.wpack + ((size_t)cb * (size_t)(P.ksteps + 3) * 8 + 2 * wave) * 1024;   // + 3 spare k-steps behind every cout block
    const unsigned alane16 = (unsigned)lane * 16u;
    auto dma_a = [&](int slot) {
        const unsigned dst = (unsigned)(slot * PN4_ASLOT) + (unsigned)__builtin_amdgcn_readfirstlane(wave * 2048);
        // (no instruction offset: the immediate of an LDS-DMA is added to the LDS address as well as to the global one)
        pn_glds16_s<0>(aptr, alane16, dst);
        pn_glds16_s<0>(aptr + 1024, alane16, dst + 1024u);
        aptr += PN4_ASLOT;
    };

    // ---- halo images: this wave fetches DMA pieces n = wc*6 + j (j = 0..5) of ITS strip: halo row n >> 1, pixels 16*(n&1).. ----
    const size_t frame_b = ((size_t)b * P.H * P.W * P.in_cs + P.in_coff) * 2;
    const char *img = (const char *)P.in + frame_b;                          // + hh * 64 per half (scalar)
    const unsigned zero_rel = P.in_zero_off - (unsigned)frame_b;                // zero page (>= 64 * halves + 16 bytes of zeros) relative to img
    // QUARTER-major images ([2 quarters of 16 channels][halo row][32 px][32 B] per 32-channel half): the 16 lanes one
    // ds_read_b128 phase serves (8 with an even, 8 with an odd lane quarter) cover 8 pixels x 32 B = a whole bank row,
    // conflict-free, where the half-major image of conv3 ([row][px][64 B]) costs 2 LDS cycles per read.  One DMA
    // instruction = one 32-pixel halo row of one quarter; wave wc fetches quarter wc (6 rows).
    unsigned boff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int px = lane >> 1;
        const int iy = iy0 + j, ix = ix0 + px;
        const bool inb = px < HC && (unsigned)ix < (unsigned)P.W && (unsigned)iy < (unsigned)P.H;
        boff[j] = inb ? (unsigned)((iy * P.W + ix) * P.in_cs * 2 + wc * 32 + (lane & 1) * 16) : zero_rel;
    }
    const int nhalves = nchunks * 2, wrap_h = P.in_wrap * 2;
    auto dma_b = [&](int j, int buf, int hh) {       // piece j of half hh -> image `buf` of this strip
        const unsigned dst = (unsigned)(PN4_ARING + buf * PN4_BBUF + j * (PITCH * 32)) +
                             (unsigned)__builtin_amdgcn_readfirstlane(wp * PN4_BSTRIP + wc * PN4_BQUART);
        const int hs = hh < nhalves ? (hh >= wrap_h ? hh - wrap_h : hh) : 0;        // bf16x3: the third plane pair reads the hi plane again; past the last half: a harmless refetch into the idle image
        pn_glds16_s<0>(img + (size_t)hs * 64, boff[j], dst);
    };

    // ---- prologue: image of half 0, weight k-steps 0..2 ----
#pragma unroll
    for (int j = 0; j < 6; ++j) dma_b(j, 0, 0);
    dma_a(0); dma_a(1); dma_a(2);

    // per-lane LDS read addresses: tap / buffer / slot are immediates
    int baddr[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int slot = pt * 16 + c;
        const int s = slot < npix ? slot : 0;
        const int ry = (int)(((float)s + 0.5f) * inv_wc);
        const int rx = s - ry * Wc;
        baddr[pt] = PN4_ARING + wp * PN4_BSTRIP + (q >> 1) * PN4_BQUART + (q & 1) * 16 + (ry * PITCH + rx) * 32;
    }
    const int aaddr = wc * 4096 + lane * 16;
    static_assert(RING == 3 || RING == 4, "weight ring: 3 slots (immediate slot offsets) or 4 (runtime slot)");
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // a wave whose 64 couts lie beyond the layer's cout (a 64-cout conv sharing the launch of a 128-cout sibling) only
    // takes part in the DMA and the barriers
    const bool active = (cb * 2 + wc) * 64 < P.cout;

    PN_STAMP_AT(1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    PN_STAMP_AT(2);

    // item j = (phase ph, pixel tile pt): B fragment of tap ph % 9 from image (ph / 9) & 1
#define PN4_BOFF(j) (((((j) / PT) / KK) & 1) * PN4_BBUF + (((((j) / PT) % KK) / KS) * PITCH + ((((j) / PT) % KK) % KS)) * 32)
    // B fragments are read BQ - 1 items (4 MFMAs each) ahead of their use: with 2 ahead a lone wave on its SIMD spent
    // 147 of 717 cycles per k-step waiting for LDS (profiles/README.md, conv4 ablations)
#ifndef PN4_BQ
#define PN4_BQ 6
#endif
    constexpr int BQ = PN4_BQ;
    static_assert((2 * KK * PT) % BQ == 0, "queue slot of an item must not depend on the chunk");
    bf16x8 aq[2][CT], bq[BQ];
    if (active) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) aq[0][ct] = *reinterpret_cast<const bf16x8 *>(smem + aaddr + ct * 1024);
#pragma unroll
        for (int j = 0; j < BQ - 1; ++j) bq[j] = *reinterpret_cast<const bf16x8 *>(smem + baddr[j % PT] + PN4_BOFF(j));
    }
    // phase 0 refills ring slot 0: every wave's reads of k-step 0's fragments must have returned first
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // the K loop, instantiated with and without the arithmetic (an inactive wave keeps the DMA schedule and the barriers)
    auto kloop = [&](auto mathc) {
        constexpr bool MATH = decltype(mathc)::value;
        for (int chunk = 0; chunk < nchunks; ++chunk) {
#pragma clang loop unroll(full)
            for (int ph = 0; ph < 2 * KK; ++ph) {            // k-step s = chunk * 18 + ph
                const int tap = ph % KK, half = ph / KK;
                const int hh = chunk * 2 + half;
                // RING == 4: k-step s lives in slot s & 3 (wave-uniform, computed on the scalar unit + one v_add per step)
                const int sstep = chunk * (2 * KK) + ph;
                const int anext = RING == 3 ? aaddr + ((ph + 1) % 3) * PN4_ASLOT : aaddr + (int)__builtin_amdgcn_readfirstlane(((sstep + 1) & 3) * PN4_ASLOT);
                // (1) staging for later steps: one halo piece of the NEXT half (taps 0..5), the weight fragments of step s + 3
                // (-DPN4_DMA_MID: issued between the 3rd and the 4th pixel tile's MFMAs instead of in front of the step's first MFMA)
                auto stage = [&]() {
#ifndef PN4_FAKE_NODMA_B                        // -DPN4_FAKE_*: timing-only ablations (wrong results), scripts/conv4lab.hip
                    if (tap < 6) dma_b(tap, (half + 1) & 1, hh + 1);
#endif
#ifndef PN4_FAKE_NODMA_A
                    dma_a(RING == 3 ? ph % 3 : (int)__builtin_amdgcn_readfirstlane((sstep + 3) & 3));
#endif
                };
#ifndef PN4_DMA_MID
                stage();
#else
                if (!MATH) stage();
#endif
                // (2) this step's 28 MFMAs; fragment reads for the next step / the next items between them
                if (MATH) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
                    for (int pt = 0; pt < PT; ++pt) {
                        const int j = ph * PT + pt, jr = j + BQ - 1;
#ifdef PN4_DMA_MID
                        if (pt == 3) { __builtin_amdgcn_sched_barrier(0); stage(); __builtin_amdgcn_sched_barrier(0); }
#endif
#ifndef PN4_FAKE_NOLDS
                        if (pt < CT)
                            aq[(ph + 1) & 1][pt] = *reinterpret_cast<const bf16x8 *>(smem + anext + pt * 1024);
                        bq[jr % BQ] = *reinterpret_cast<const bf16x8 *>(smem + baddr[jr % PT] + PN4_BOFF(jr % (2 * KK * PT)));
#endif
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[ph & 1][ct], bq[j % BQ], acc[ct][pt], 0, 0, 0);
                        if (pt < CT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // (3) everything older than the newest weight step (and this phase's halo piece) has landed; every fragment
                // read issued so far has returned (the slot / image it came from may be overwritten after the barrier)
#if defined(PN4_FAKE_NOBAR)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(PN4_FAKE_NODMA_A) || defined(PN4_FAKE_NODMA_B)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
#if PN4_RING == 4
                // 4 slots: the slot refilled in step s + 1 was last read in step s - 1 and the image refilled during a half was
                // last read before the previous half's final MFMAs -- no LDS read has to drain here
                if (tap < 6) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
#elif PN4_LGKM == 3
                // the three LDS reads issued last in a step are B fragments of later items (checked in the ISA): only the A reads
                // (the slot refilled right after this barrier) have to be back
                if (tap < 6) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(3)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(3)\n\ts_barrier" ::: "memory");
#else
                if (tap < 6) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#endif
            }
            PN_STAMP_AT(3 + 2 * (chunk & 3));
        }
    };
    if (active) kloop(std::true_type{});
    else kloop(std::false_type{});
#undef PN4_BOFF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the surplus prefetches must not land in LDS after this block has exited
    if (!active || !strip_ok) return;

    // ---- epilogue: a lane holds 16 CONSECUTIVE couts of each of its pixels (net.hip packs the rows with
    // pn_conv_row_channel(tile, row, 4)): bias + residual + activation + two 16-B NHWC stores per pixel tile; the four
    // lane quarters cover one whole 128-B line of the pixel ----
    PN_STAMP_AT(11);
    constexpr int LC = CT * 4;
    const int wave_c0 = (cb * 2 + wc) * 64;
    const int cw = wave_c0 + LC * q;
    const int cout = P.cout, act = P.act;
    float bias[LC];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 b4 = *reinterpret_cast<const PN_GLOBAL f32x4 *>((const PN_GLOBAL float *)P.bias + cw + 4 * ct);
        bias[4 * ct + 0] = b4[0]; bias[4 * ct + 1] = b4[1]; bias[4 * ct + 2] = b4[2]; bias[4 * ct + 3] = b4[3];
    }
    const int res_cs = P.res_cs, out_cs = P.out_cs, Ho = P.Ho;
    const int split = P.split, res_split = P.res_split;
    const unsigned pix0 = (unsigned)((b * Ho + oy0) * Wo + ox0);
    const bool full = wave_c0 + 64 <= cout && P.out && !P.out_nchw;
    if (full) {
        PN_GLOBAL T *ob = (PN_GLOBAL T *)P.out + P.out_coff + cw;
        const PN_GLOBAL T *rb = P.res ? (const PN_GLOBAL T *)P.res + P.res_coff + cw : nullptr;
        auto finish = [&](auto actc, auto resc) {
            constexpr int ACT = decltype(actc)::value;
            constexpr bool RES = decltype(resc)::value;
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT; ++pt) {
                const int slot = pt * 16 + c;
                const unsigned t = (unsigned)(baddr[pt] - PN4_ARING - wp * PN4_BSTRIP - (q >> 1) * PN4_BQUART) >> 5;     // ry * 32 + rx
                const unsigned opix = PITCH == 32 ? pix0 + (t >> 5) * (unsigned)Wo + (t & 31u) : pix0 + (t / (unsigned)PITCH) * (unsigned)Wo + (t % (unsigned)PITCH);
                float v[LC];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[4 * ct + i] = acc[ct][pt][i] + bias[4 * ct + i];
                if (RES) {
                    for (int pl = 0; pl < (res_split ? 2 : 1); ++pl) {       // bf16x3: residual = hi plane + lo plane
                        T rv[LC];
                        const PN_GLOBAL u32x4 *rp = reinterpret_cast<const PN_GLOBAL u32x4 *>(rb + (opix * (unsigned)res_cs + (unsigned)(pl * res_split)));
                        reinterpret_cast<u32x4 *>(rv)[0] = rp[0];
                        reinterpret_cast<u32x4 *>(rv)[1] = rp[1];
#pragma unroll
                        for (int k = 0; k < LC; ++k) v[k] += (float)rv[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < LC; ++k) {
                    if (ACT == PN_ACT_RELU) v[k] = v[k] > 0.f ? v[k] : 0.f;
                    else if (ACT == PN_ACT_LEAKY) v[k] = v[k] > 0.f ? v[k] : v[k] * 0.1f;
                    else if (ACT != PN_ACT_NONE) v[k] = pn_activate(v[k], act, cw + k, P.yolo_naf);
                }
                // bf16x3: two planes [hi | lo] `split` channels apart, hi = bf16(v), lo = bf16(v - hi)
                for (int pl = 0; pl < (split ? 2 : 1); ++pl) {
                    T ov[LC];
#pragma unroll
                    for (int k = 0; k < LC; ++k) {
                        const T hi = (T)v[k];
                        ov[k] = pl == 1 ? (T)(v[k] - (float)hi) : hi;
                    }
                    if (slot < npix) {
                        PN_GLOBAL u32x4 *op = reinterpret_cast<PN_GLOBAL u32x4 *>(ob + (opix * (unsigned)out_cs + (unsigned)(pl * split)));
#ifdef PN4_WT_STORE
                        // write-through (sc1): the tile leaves the XCD's L2 now, while other blocks still compute, instead of in
                        // the end-of-kernel write-back of ~23 MB of dirty lines that the next launch has to wait for
                        pn_store16_wt(op, reinterpret_cast<u32x4 *>(ov)[0]);
                        pn_store16_wt(op + 1, reinterpret_cast<u32x4 *>(ov)[1]);
#elif defined(PN4_NT_STORE)
                        // experiment: non-temporal stores (conv3's v21, profiles/README.md: faster alone, slower in the network)
                        __builtin_nontemporal_store(reinterpret_cast<u32x4 *>(ov)[0], op);
                        __builtin_nontemporal_store(reinterpret_cast<u32x4 *>(ov)[1], op + 1);
#else
                        op[0] = reinterpret_cast<u32x4 *>(ov)[0];
                        op[1] = reinterpret_cast<u32x4 *>(ov)[1];
#endif
                    }
                }
            }
        };
        if (P.res) {
            if (act == PN_ACT_RELU) finish(std::integral_constant<int, PN_ACT_RELU>{}, std::true_type{});
            else if (act == PN_ACT_LEAKY) finish(std::integral_constant<int, PN_ACT_LEAKY>{}, std::true_type{});
            else if (act == PN_ACT_NONE) finish(std::integral_constant<int, PN_ACT_NONE>{}, std::true_type{});
            else finish(std::integral_constant<int, -1>{}, std::true_type{});
        } else {
            if (act == PN_ACT_RELU) finish(std::integral_constant<int, PN_ACT_RELU>{}, std::false_type{});
            else if (act == PN_ACT_LEAKY) finish(std::integral_constant<int, PN_ACT_LEAKY>{}, std::false_type{});
            else if (act == PN_ACT_NONE) finish(std::integral_constant<int, PN_ACT_NONE>{}, std::false_type{});
            else finish(std::integral_constant<int, -1>{}, std::false_type{});
        }
        PN_STAMP_AT(12);
        return;
    }
    // general path (ragged cout, NCHW f32 export): element-wise, same arithmetic
    {
        const PN_GLOBAL T *res_base = P.res ? (const PN_GLOBAL T *)P.res + P.res_coff + cw : nullptr;
        PN_GLOBAL T *out_base = P.out ? (PN_GLOBAL T *)P.out + P.out_coff + cw : nullptr;
        PN_GLOBAL float *nchw = (PN_GLOBAL float *)P.out_nchw;
#pragma clang loop unroll(full)
        for (int pt = 0; pt < PT; ++pt) {
            const int slot = pt * 16 + c;
            if (slot >= npix || cw >= cout) continue;
            const int ry = (int)(((float)slot + 0.5f) * inv_wc);
            const int rx = slot - ry * Wc;
            const unsigned opix = pix0 + (unsigned)(ry * Wo + rx);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = 4 * ct + i;
                    if (cw + k >= cout) continue;
                    float v = acc[ct][pt][i] + bias[k];
                    if (res_base) v += (float)res_base[opix * (unsigned)res_cs + k];
                    if (res_base && res_split) v += (float)res_base[opix * (unsigned)res_cs + res_split + k];
                    v = pn_activate(v, act, cw + k, P.yolo_naf);
                    if (out_base) {
                        const T hi = (T)v;
                        out_base[opix * (unsigned)out_cs + k] = hi;
                        if (split) {
                            out_base[opix * (unsigned)out_cs + split + k] = (T)(v - (float)hi);
                        }
                    }
                    if (nchw) nchw[((size_t)b * cout + cw + k) * ((size_t)Ho * Wo) + (size_t)(oy0 + ry) * Wo + (ox0 + rx)] = v;
                }
        }
    }
    PN_STAMP_AT(12);
}

__global__ __launch_bounds__(256, 2) void conv4_kernel(const ConvProblem *__restrict__ probs) {
    conv4_body(probs[blockIdx.y], (int)blockIdx.x);
}

static int conv4_launch(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    // POPNET_CONV4_LDS=<bytes> (experiment): ask for more LDS than the kernel uses, e.g. 90000 = one block per CU, which leaves half of the
    // CU's registers and 70 KB of its LDS to the small launches of the other batches in flight
    static const size_t lds = getenv("POPNET_CONV4_LDS") ? std::max<size_t>(PN4_LDS, (size_t)atol(getenv("POPNET_CONV4_LDS"))) : PN4_LDS;
    static PnLdsAttr attr;
    if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(conv4_kernel), lds)) return rc;
    hipLaunchKernelGGL(conv4_kernel, dim3(L.max_blocks, L.nprob), dim3(256), lds, stream, L.probs_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}
