// bf16 3x3 / 1x1 stride-1 "same" convolution on the CDNA4 matrix cores, high-occupancy variant.
//
// Same math, weight packing and epilogue contract as conv_mfma_kernel.h (which stays the generic
// kernel: fp32 parity mode, stride 2, odd shapes); this one is what the bf16 perf path runs for
// every stride-1 layer whose map splits into <= 30-column strips.  It replaces the same reference
// code (tpm/lib/network/rtpose_light3d.py:24-72,222-246,335-350; yolo_posenet.py:101-126;
// resnet.py:27-56).  What differs, and why (profiles/README.md, round-1 timeline stamps):
//   * 4 waves per SIMD instead of 2 (<= 128 VGPRs): a block's fixed latencies (first halo fetch,
//     chunk hand-over, epilogue burst) overlap with the MFMAs of the three other blocks of the CU.
//   * LDS halo image is HALF-MAJOR: [2 halves of 32 channels][halo row][32 px][4 pieces x 16 B].  Tap (ky,kx) and
//     the 32-channel half are IMMEDIATE offsets of one address register per pixel tile (7 VGPRs instead of the 42 of
//     the pixel-major + XOR layout), and no swizzle is needed: the 8 pixels one lane quarter reads per ds_read_b128
//     phase fall on 4 bank groups, a 2-way conflict (8 instead of 4 LDS cycles per read; LDS has the slack -- removing
//     every conflict in a timing experiment bought 2 %).  The first version was piece-major ([8 pieces][row][px][16 B],
//     conflict-free reads) but its fill gathered 64 pixels x 16 B per DMA instruction -- 64 requests to the texture
//     addresser; half-major fills 16 pixels x 64 contiguous bytes per instruction (4 adjacent lanes = one 64-B segment):
//     -8 % on the 112x112 layers, -4 % on the stage levels, +5 % end to end (profiles/README.md v16).
//   * The halo is filled by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write; out-of-image pixels
//     read a zero page that lives at the end of every activation buffer (ConvProblem::in_zero_off).
//   * One LDS image per block (24.5 KB for a 4-row strip tile): the chunk hand-over stalls this
//     block only; occupancy hides it.
//   * Weights: global -> VGPR fragments through a 3-slot queue (2 k-steps ahead is enough at 4
//     waves per SIMD).
#pragma once
#include "conv_mfma_kernel.h"

#ifndef PN_CONV3_OCC
#define PN_CONV3_OCC 4
#endif
// -DPN_CONV3_NT_STORE: output tiles leave with the nt hint (experiment v21: faster alone, slower in the network, where the next layer re-reads them)
#ifndef PN_CONV3_FAST_EPILOGUE
#define PN_CONV3_FAST_EPILOGUE 2  // wave-uniform fast epilogue: 0 never, 1 every instantiation, 2 the 128-cout blocks only (profiles/README.md v23)
#endif
#if !defined(PN_CONV3_PIECEMAJOR) && !defined(PN_CONV3_QUARTERMAJOR)
#define PN_CONV3_HALFMAJOR 1      // LDS halo layout, see the header comment; -DPN_CONV3_PIECEMAJOR selects the first layout (experiments)
#endif
// -DPN_CONV3_QUARTERMAJOR (round 5): conv4_kernel's image, [4 planes of 16 channels][halo row][32 px][32 B] -- the 16 lanes of a
// ds_read_b128 phase cover one whole 256-byte bank row (no conflict; the half-major image: 2-way), one DMA instruction = one 32-pixel halo row of a plane
// (32 segments of 32 B instead of 16 of 64 B).  Measured: profiles/r05_notes.txt.

// LDS-DMA: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1024).
// M0 is written in the same statement that uses it (the compiler does not preserve it around asm).
__device__ __forceinline__ void pn_glds16(gcptr src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}

// 16-byte write-through store (sc1): the line is written to memory at once and dropped from the XCD's L2 (no dirty line is
// left for the end-of-kernel write-back).  Inline asm: hipcc keeps no vmcnt bookkeeping for it -- stores need none.
__device__ __forceinline__ void pn_store16_wt(PN_GLOBAL void *p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// LDS-DMA, scalar base + 32-bit lane offset: no 64-bit address registers, the per-step pointer bump is scalar.
template <int IMM>
__device__ __forceinline__ void pn_glds16_s(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst), "n"(IMM) : "memory");
}

// PT = pixel tiles of 16 per wave: 7 (112 pixels = 4 rows of a 28-column strip, 56 accumulator VGPRs, 4 waves / SIMD) or
// 14 (224 pixels = 8 rows, 112 accumulator VGPRs, 2 waves / SIMD: half the weight bytes per MFMA, for Cin = 64 layers
// on large maps, whose 224-pixel tiles otherwise stream their whole weight slice twice).
// RPG = output rows per wave group kept in the halo image: 4 for 24..30-column strips (4 x 28 = 112 pixels), 8 for
// narrow maps (8 x 14 = 112: the 14x14 layers of YoloPoseNet), 8 as well for PT = 14.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;   // native vector: storable through address-space pointers

// TAIL: the instantiation whose blocks run a second 1x1 convolution on their 128-channel output tile instead of storing it
// (ConvProblem::tail_w; its own instantiation so that the plain kernel's register allocation is untouched: with both epilogues
// in one body the 128-register 1x1 kernel spilled 41 VGPRs)
template <int KS, int WC, int WP, int NBUF, int PT, int RPG = PT * 4 / 7, int TAIL = 0>
__device__ __forceinline__ void conv3_body(const ConvProblem &P) {
    typedef __bf16 T;
    typedef Elem<PN_PREC_BF16>::Frag Frag;
    constexpr int CT = 2, NW = WC * WP;
    constexpr int KK = KS * KS, PAD = KS / 2;
    constexpr int PITCH = 32;                          // halo pixels per LDS row
    constexpr int HR = RPG * WP + KS - 1 + ((KS - 1) & 1 ? 1 : 0);   // halo rows (even: one DMA fills two rows)
    constexpr int PS = HR * PITCH * 16;                // bytes of one piece plane (multiple of 256)
    constexpr int NG = 8 * (HR / 2);                   // DMA instructions per chunk, spread over the waves
    constexpr int FRAGB = 1024;
    constexpr int NSTEP = KK * 2;
#ifndef PN_CONV3_NA2
#define PN_CONV3_NA2 6
#endif
#ifndef PN_CONV3_NA14
#define PN_CONV3_NA14 3
#endif
#ifndef PN_CONV3_DB14
#define PN_CONV3_DB14 6
#endif
    constexpr int NA = KS == 1 ? 2 : (NBUF == 2 ? PN_CONV3_NA2 : (PT == 14 ? PN_CONV3_NA14 : 3));   // NSTEP % NA == 0: the queue slot of a k-step must not depend on the chunk
    static_assert(NSTEP % NA == 0, "weight queue depth must divide the k-steps of a chunk");
    constexpr int IMG = 8 * PS;                        // bytes of one halo image
    constexpr int NITEM = NSTEP * PT;
#ifndef PN_CONV3_DB
#define PN_CONV3_DB 3
#endif
    constexpr int DB = PT == 14 ? PN_CONV3_DB14 : PN_CONV3_DB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    if ((int)blockIdx.x >= P.nblocks) return;
    int bx;
    {   // XCD-aware remap, see conv_mfma_kernel.h
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
    // TAIL == 2 (fused average pool): the block computes the (2 * 3 + 1) x (2 * 7 + 1) patch of the 1x1 convolution's output that its 3 x 7
    // pooled pixels need, origin one row / column before the first window (rows and columns outside the map: zero-page input,
    // masked out of the window sums); neighbouring blocks recompute the shared row and column
    const int oy0 = TAIL == 2 ? ty * 6 - 1 : ty * P.R, ox0 = TAIL == 2 ? tx * 14 - 1 : tx * P.Wt;
    const int R = TAIL == 2 ? 7 : min(P.R, P.Ho - oy0);
    const int Wo = P.Wo;
    const int Wc = TAIL == 2 ? 15 : min(P.Wt, Wo - ox0);
    const int npix = R * Wc;
    const int HC = Wc + KS - 1;
    const int iy0 = oy0 - PAD, ix0 = ox0 - PAD;
    const float inv_wc = 1.0f / (float)Wc;
    const int nchunks = P.cin_chunks;
    const int in_wrap = P.in_wrap;

    // ---- weight stream: scalar base per cout tile + lane offset; first NA-1 k-steps in flight ----
    const int ctile0 = (cb * WC + wc) * CT;
    // buffer loads: one resource for the whole pack, scalar fragment offset, 32-bit lane offset -- no vector ALU
    // work and no 64-bit address registers per k-step (the flat form cost 2 v_lshl_add_u64 + 6 VGPRs)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(P.wpack), 0, 0x7fffffff, 0x00020000);
    unsigned wbase[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wbase[ct] = (unsigned)__builtin_amdgcn_readfirstlane((ctile0 + ct) * P.ksteps * FRAGB);
    const unsigned wlane = (unsigned)lane * 16u;
    auto load_w = [&](unsigned soff) -> Frag {
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, soff, 0);
        return __builtin_bit_cast(Frag, r);
    };
    Frag aq[NA][CT];
#pragma unroll
    for (int d = 0; d < NA - 1; ++d)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) aq[d][ct] = load_w(wbase[ct] + d * FRAGB);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wbase[ct] += (NA - 1) * FRAGB;

    // ---- halo DMA: instruction n = wave * NGW + j fills rows (2i, 2i+1) of piece plane pc ----
    constexpr int NGW = (NG + NW - 1) / NW;
    gcptr img = (gcptr)P.in + ((size_t)b * P.H * P.W * P.in_cs + P.in_coff) * 2;
    const unsigned zero_rel = P.in_zero_off - (unsigned)(((size_t)b * P.H * P.W * P.in_cs + P.in_coff) * 2);   // zero page relative to img
    const int row_b = P.W * P.in_cs * 2, col_b = P.in_cs * 2;
    const int hcol = lane & 31, hrow = lane >> 5;
    const bool col_ok = hcol < HC && (unsigned)(ix0 + hcol) < (unsigned)P.W;
    const unsigned coloff = (unsigned)((ix0 + hcol) * col_b);
    const int Hin = P.H;
    // Branch-free on purpose (a branch inside the unrolled K loop splits it into basic blocks and the
    // compiler then drains the weight queue at every block boundary): an instruction that has nothing
    // to fetch (`live` false, or n beyond the image) copies the zero page into a dump slot behind the images.
    auto stage_one = [&](int chunk, int j, int bufoff, bool live) {   // j-th DMA instruction of this wave for `chunk`
        const int n = wave * NGW + j;                    // wave-uniform
        const bool on = live && n < NG;
#if defined(PN_CONV3_QUARTERMAJOR)
        const int pln = n / HR, rr = n - pln * HR;          // plane of 16 channels, halo row
        const int px = lane >> 1;
        const int iy = iy0 + rr;
        const bool inb = (int)on & (int)(px < HC) & (int)((unsigned)(ix0 + px) < (unsigned)P.W) & (int)((unsigned)iy < (unsigned)Hin);
        unsigned off = (unsigned)(iy * row_b + (ix0 + px) * col_b + pln * 32 + (lane & 1) * 16);
        const int pc = 0, i = 0;
        const int dst_hm = pln * (IMG / 4) + rr * (PITCH * 32);
#elif defined(PN_CONV3_HALFMAJOR)
        // half-major image [2 halves][halo row][32 px][4 pieces x 16 B]: one instruction = 16 pixels x 64 contiguous bytes
        // (4 adjacent lanes = one 64-B segment of a pixel line) instead of 64 pixels x 16 B
        const int hh = n / (2 * HR), rr = (n - hh * 2 * HR) >> 1, gg = n & 1;
        const int px = gg * 16 + (lane >> 2);
        const int iy = iy0 + rr;
        const bool inb = (int)on & (int)(px < HC) & (int)((unsigned)(ix0 + px) < (unsigned)P.W) & (int)((unsigned)iy < (unsigned)Hin);
        unsigned off = (unsigned)(iy * row_b + (ix0 + px) * col_b + hh * 64 + (lane & 3) * 16);
        const int pc = 0, i = 0;
        const int dst_hm = hh * (IMG / 2) + (rr * PITCH + gg * 16) * 64;
#else
        const int pc = n / (HR / 2), i = n - pc * (HR / 2);
        const int iy = iy0 + 2 * i + hrow;
        const bool inb = (int)on & (int)col_ok & (int)((unsigned)iy < (unsigned)Hin);
        unsigned off = (unsigned)(iy * row_b + pc * 16) + coloff;
        const int dst_hm = 0;
#endif
#ifdef PN_CONV3_FAKE_LINDMA   // timing experiment (wrong results): each DMA instruction reads 1 KB of consecutive bytes
        off = (unsigned)(min(max(iy0 + 2 * i, 0), Hin - 2) * row_b + max(ix0, 0) * col_b) + lane * 16;
#endif
        asm volatile("" : "+v"(off));                    // materialise: the select below must stay a v_cndmask, not a branch
        const int csrc = chunk >= in_wrap ? chunk - in_wrap : chunk;      // bf16x3: the third plane pair reads the hi plane again
        const unsigned zrel = zero_rel - (unsigned)(csrc * 128);
#ifndef PN_CONV3_FAKE_NODMA                               // timing experiment (wrong results): no halo fetch at all
        pn_glds16(img + csrc * 128 + (inb ? off : zrel),
#if defined(PN_CONV3_HALFMAJOR) || defined(PN_CONV3_QUARTERMAJOR)
                  (unsigned)__builtin_amdgcn_readfirstlane(on ? bufoff + dst_hm : (NBUF >= 3 ? nchunks : NBUF) * IMG));
#else
                  (unsigned)__builtin_amdgcn_readfirstlane(on ? bufoff + pc * PS + i * (2 * PITCH * 16) + dst_hm : (NBUF >= 3 ? nchunks : NBUF) * IMG));
#endif
#endif
    };
    auto stage = [&](int chunk, int bufoff) {
#pragma unroll
        for (int j = 0; j < NGW; ++j) stage_one(chunk, j, bufoff, true);
    };
    stage(0, 0);
    if (NBUF >= 3)                                       // all-resident mode (1x1 convs, Cin <= 64 * NBUF): every chunk's image up front, no hand-over
        for (int ch = 1; ch < nchunks; ++ch) stage(ch, ch * IMG);

    // ---- per-lane LDS read address of each pixel tile (tap / half are immediates) ----
    int baddr[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        int slot = (wp * PT + pt) * 16 + c;
        int s = slot < npix ? slot : 0;
        int ry = (int)(((float)s + 0.5f) * inv_wc);
        int rx = s - ry * Wc;
#if defined(PN_CONV3_QUARTERMAJOR)
        baddr[pt] = (q >> 1) * (IMG / 4) + (q & 1) * 16 + (ry * PITCH + rx) * 32;
#elif defined(PN_CONV3_HALFMAJOR)
        baddr[pt] = q * 16 + (ry * PITCH + rx) * 64;
#else
        baddr[pt] = q * PS + (ry * PITCH + rx) * 16;
#endif
#ifdef PN_CONV3_FAKE_NOWRAP   // timing experiment only (wrong results): every pixel tile reads 16 consecutive entries
        baddr[pt] = q * PS + (pt * 16 + c) * 16;
#endif
    }
    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    PN_STAMP_AT(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PN_STAMP_AT(2);

    static_assert(NBUF != 2 || NGW <= NSTEP, "the next chunk's DMA is spread one instruction per k-step");
    // a wave whose 32 couts lie beyond the layer's cout (a 64-cout conv sharing the launch of a 128-cout
    // sibling, net.hip::harmonize_level) only takes part in the halo DMA and the barriers
    const bool active = (cb * WC + wc) * (CT * 16) < P.cout;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int cur = NBUF == 2 ? (chunk & 1) * IMG : (NBUF >= 3 ? chunk * IMG : 0);
        const int nxt = NBUF == 2 ? IMG - cur : 0;
        const bool more = chunk + 1 < nchunks;
        const char *sm = smem + cur;
        if (!active) {
            if (NBUF == 2 && more) stage(chunk + 1, nxt);
        } else {
        // item j = (k-step s = half * KK + tap, pixel tile pt)
#if defined(PN_CONV3_QUARTERMAJOR)
#define PN3_OFF(j) ((((j) / PT) / KK) * (IMG / 2) + (((((j) / PT) % KK) / KS) * PITCH + ((((j) / PT) % KK) % KS)) * 32)
#elif defined(PN_CONV3_HALFMAJOR)
#define PN3_OFF(j) ((((j) / PT) / KK) * (IMG / 2) + (((((j) / PT) % KK) / KS) * PITCH + ((((j) / PT) % KK) % KS)) * 64)
#else
#define PN3_OFF(j) ((((j) / PT) / KK) * 4 * PS + (((((j) / PT) % KK) / KS) * PITCH + ((((j) / PT) % KK) % KS)) * 16)
#endif
        Frag bq[DB];
#pragma unroll
        for (int j = 0; j < DB - 1; ++j) bq[j] = read_b_frag<PN_PREC_BF16>(sm + PN3_OFF(j), baddr[j % PT]);
        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
        for (int s = 0; s < NSTEP; ++s) {
#pragma clang loop unroll(full)
          for (int pt = 0; pt < PT; ++pt) {
            const int j = s * PT + pt;
            if (pt == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {        // wpack ends with NA-1 spare fragments
#ifndef PN_CONV3_FAKE_NOA                                // timing experiments only (wrong results)
                    aq[(s + NA - 1) % NA][ct] = load_w(wbase[ct]);
#endif
#ifndef PN_CONV3_FAKE_SAMEA                              // timing experiment: every weight load hits the same (L1-resident) KB
                    wbase[ct] += FRAGB;
#endif
                }
                // double-buffered: the next chunk's image is fetched by ONE DMA instruction per k-step (the
                // compiler does not count asm memory ops, so each one shortens the effective depth of the
                // weight queue by half a k-step; a burst would stall the next weight wait for the whole DMA)
                if (NBUF == 2 && s < NGW) stage_one(chunk + 1, s, nxt, more);
            }
            const int jr = j + DB - 1;
#ifndef PN_CONV3_FAKE_NOB
            if (jr < NITEM) bq[jr % DB] = read_b_frag<PN_PREC_BF16>(sm + PN3_OFF(jr), baddr[jr % PT]);
#endif
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[ct][pt] = mma(aq[s % NA][ct], bq[j % DB], acc[ct][pt]);
            if (pt == 0) __builtin_amdgcn_sched_group_barrier(0x020, CT, 0);
            if (jr < NITEM) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, CT, 0);
        }
        }
#undef PN3_OFF
        }
        PN_STAMP_AT(3 + 2 * (chunk & 3));
        if (more && NBUF < 3) {
            if (NBUF == 2) {
                // every DMA is older than the 2*(NA-1) weight loads in flight: wait for exactly those
                if (active) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                         // image c+1 complete, image c free for chunk c+2
            } else {
                __syncthreads();                         // every wave is done reading this chunk's image
                stage(chunk + 1, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
        PN_STAMP_AT(4 + 2 * (chunk & 3));
    }

    // ---- fused 1x1 tail (ConvProblem::tail_w, net.hip::fuse_1x1_tails): this block's 128-channel tile is the whole input of
    // a second 1x1 convolution (128 -> <= 32 channels, rtpose_light3d.py:266-267).  A lane's 8 finished values of a pixel tile
    // (bias + activation + bf16, exactly what the epilogue below would store) are channels 32 wc + 8 q .. + 7 of pixel c: the B
    // fragment of the second convolution's k-step wc.  The four waves park their fragments in LDS, then each wave runs the
    // chained MFMAs of its pixel tiles over k-steps 0..3 -- the same accumulation chain as the two-launch path, bit for bit.
    if constexpr (TAIL == 2) {
        // ---- fused AvgPool2d(3, 2, 1) (ConvProblem::pool_tail, net.hip::fuse_pool_tails; rtpose_light3d.py:157-158): the block's
        // finished values (bias + activation + bf16: what the epilogue would store, what pool_kernel would read back) are parked in
        // LDS in the layout of the 1x1 tail, then every thread sums the nine taps of its pooled pixels x 8 channels in pool_kernel's
        // order (row-major taps, fp32, a tap outside the map adds 0, one division by 9) -- bit-identical to the two-launch path;
        // the full-resolution map is never written.
        static_assert(KS == 1 && WC == 4 && WP == 1 && PT == 7 && RPG == 8, "the fused pool is built for the 128-cout 1x1 block on a 7 x 15 patch");
        constexpr int LC = CT * 4;
        const int cw = wc * (CT * 16) + LC * q;
        float bias[LC];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 b4 = *reinterpret_cast<const PN_GLOBAL f32x4 *>((const PN_GLOBAL float *)P.bias + cw + 4 * ct);
            bias[4 * ct + 0] = b4[0]; bias[4 * ct + 1] = b4[1]; bias[4 * ct + 2] = b4[2]; bias[4 * ct + 3] = b4[3];
        }
        const int act = P.act;
        // bf16x3 (round 5): the convolution's value is parked as its two planes hi = bf16(v), lo = bf16(v - hi) -- what the unfused launch stores --
        // the lo tile PARK bytes behind the hi tile; the window sums add (float)hi + (float)lo per tap like pool_kernel<.., SPLIT> and the pooled
        // value leaves as two planes `tail_split` channels apart
        constexpr int PARK = 4 * PT * 1024;
        const int tsplit = P.tail_split;
        __syncthreads();                                 // every wave is done reading the halo image: its space becomes the parked tile
        auto park = [&](auto actc) {
            constexpr int ACT = decltype(actc)::value;
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT; ++pt) {
                T ov[LC], ol[LC];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = acc[ct][pt][i] + bias[4 * ct + i];
                        if (ACT == PN_ACT_RELU) v = v > 0.f ? v : 0.f;
                        else if (ACT == PN_ACT_LEAKY) v = v > 0.f ? v : v * 0.1f;
                        ov[4 * ct + i] = (T)v;
                        ol[4 * ct + i] = (T)(v - (float)ov[4 * ct + i]);
                    }
                *reinterpret_cast<u32x4 *>(smem + ((wc * PT + pt) * 64 + lane) * 16) = *reinterpret_cast<u32x4 *>(ov);
                if (tsplit) *reinterpret_cast<u32x4 *>(smem + PARK + ((wc * PT + pt) * 64 + lane) * 16) = *reinterpret_cast<u32x4 *>(ol);
            }
        };
        if (act == PN_ACT_RELU) park(std::integral_constant<int, PN_ACT_RELU>{});
        else if (act == PN_ACT_LEAKY) park(std::integral_constant<int, PN_ACT_LEAKY>{});
        else park(std::integral_constant<int, PN_ACT_NONE>{});
        __syncthreads();
        const int Hp = (P.Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;          // pooled map
        PN_GLOBAL T *pout = (PN_GLOBAL T *)P.tail_out + P.tail_out_coff;
        for (int item = tid; item < 3 * 7 * 16; item += 256) {
            const int g8 = item & 15, pp = item >> 4, pi = pp / 7, pj = pp - pi * 7;
            const int py = ty * 3 + pi, px = tx * 7 + pj;
            if (py >= Hp || px >= Wp) continue;
            const int wq = g8 >> 2, qq = g8 & 3;                             // channels 8 g8 .. + 7 live in wave wq's fragments, lane quarter qq
            float a8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a8[k] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ry = 2 * pi + ky, rx = 2 * pj + kx;            // patch coordinates; map coordinates oy0 + ry, ox0 + rx
                    const bool ok = (unsigned)(oy0 + ry) < (unsigned)P.Ho && (unsigned)(ox0 + rx) < (unsigned)Wo;
                    const int slot = ry * 15 + rx;
                    T tv[8];
                    const int toff = ((wq * PT + (slot >> 4)) * 64 + qq * 16 + (slot & 15)) * 16;
                    *reinterpret_cast<u32x4 *>(tv) = *reinterpret_cast<const u32x4 *>(smem + toff);
                    if (tsplit) {
                        T tl[8];
                        *reinterpret_cast<u32x4 *>(tl) = *reinterpret_cast<const u32x4 *>(smem + PARK + toff);
#pragma unroll
                        for (int k = 0; k < 8; ++k) { float f = (float)tv[k]; f += (float)tl[k]; a8[k] += ok ? f : 0.f; }
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) a8[k] += ok ? (float)tv[k] : 0.f;
                    }
                }
            T o8[8], l8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float rr = a8[k] / 9.0f;
                o8[k] = (T)rr;
                l8[k] = (T)(rr - (float)o8[k]);
            }
            PN_GLOBAL T *op = pout + ((size_t)(b * Hp + py) * Wp + px) * (size_t)P.tail_out_cs + g8 * 8;
            *reinterpret_cast<PN_GLOBAL u32x4 *>(op) = *reinterpret_cast<u32x4 *>(o8);
            if (tsplit) *reinterpret_cast<PN_GLOBAL u32x4 *>(op + tsplit) = *reinterpret_cast<u32x4 *>(l8);
        }
        PN_STAMP_AT(12);
        return;
    }
    if constexpr (TAIL == 1) {
        static_assert(KS == 1 && WC == 4 && WP == 1 && PT == 7, "the fused tail is built for the 128-cout 1x1 block");
        {
            constexpr int LC = CT * 4;
            const int cw = wc * (CT * 16) + LC * q;
            float bias[LC];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const f32x4 b4 = *reinterpret_cast<const PN_GLOBAL f32x4 *>((const PN_GLOBAL float *)P.bias + cw + 4 * ct);
                bias[4 * ct + 0] = b4[0]; bias[4 * ct + 1] = b4[1]; bias[4 * ct + 2] = b4[2]; bias[4 * ct + 3] = b4[3];
            }
            const int act = P.act;
            __syncthreads();                             // every wave is done reading the halo image: its space becomes the fragment image
            // (compile-time activation, as in the epilogues below: the run-time pn_activate switch, unrolled over 56 values, made this
            // instantiation 100 KB of code -- more than the instruction cache -- and the launch 10 us slower)
            auto park = [&](auto actc) {
                constexpr int ACT = decltype(actc)::value;
#pragma clang loop unroll(full)
                for (int pt = 0; pt < PT; ++pt) {
                    T ov[LC];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float v = acc[ct][pt][i] + bias[4 * ct + i];
                            if (ACT == PN_ACT_RELU) v = v > 0.f ? v : 0.f;
                            else if (ACT == PN_ACT_LEAKY) v = v > 0.f ? v : v * 0.1f;
                            ov[4 * ct + i] = (T)v;
                        }
                    *reinterpret_cast<u32x4 *>(smem + ((wc * PT + pt) * 64 + lane) * 16) = *reinterpret_cast<u32x4 *>(ov);
                }
            };
            if (act == PN_ACT_RELU) park(std::integral_constant<int, PN_ACT_RELU>{});
            else if (act == PN_ACT_LEAKY) park(std::integral_constant<int, PN_ACT_LEAKY>{});
            else park(std::integral_constant<int, PN_ACT_NONE>{});           // net.hip::fuse_1x1_tails admits these three only
            __syncthreads();
#ifdef PN_TAIL_FAKE_NOMMA
            return;
#endif
            const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(P.tail_w), 0, 2 * 4 * 1024, 0x00020000);
            unsigned tlane = (unsigned)lane * 16u;
            asm volatile("" : "+v"(tlane));              // keep the eight fragment loads BEHIND the barrier: hoisted above it they are live beside the 56 accumulators
            Frag a2[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    a2[t][ks] = __builtin_bit_cast(Frag, __builtin_amdgcn_raw_buffer_load_b128(trs, tlane, (unsigned)((t * 4 + ks) * 1024), 0));
            const int tcout = P.tail_cout, tact = P.tail_act;
            const PN_GLOBAL float *tb = (const PN_GLOBAL float *)P.tail_bias;
            PN_GLOBAL T *tout = P.tail_out ? (PN_GLOBAL T *)P.tail_out + P.tail_out_coff : nullptr;
            PN_GLOBAL float *tnchw = (PN_GLOBAL float *)P.tail_nchw;
            const int Ho = P.Ho;
            const size_t hw = (size_t)Ho * Wo;
            for (int pt = wave; pt < PT; pt += 4) {      // wave-uniform trip count
                f32x4 c2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const Frag bf = read_b_frag<PN_PREC_BF16>(smem, ((ks * PT + pt) * 64 + lane) * 16);
                    c2[0] = mma(a2[0][ks], bf, c2[0]);
                    c2[1] = mma(a2[1][ks], bf, c2[1]);
                }
                const int slot = pt * 16 + c;
                if (slot < npix) {
                    const int ry = (int)(((float)slot + 0.5f) * inv_wc), rx = slot - ry * Wc;
                    const size_t opix = (size_t)(b * Ho + oy0 + ry) * Wo + ox0 + rx;
                    // a lane holds channels 8q .. 8q + 7 of its pixel: one 16-byte NHWC store (8 bytes when only the first four
                    // exist: 28 PAF channels); element-wise 2-byte stores into 384-byte pixel lines cost more than the whole tail
                    float v8[8];
                    T o8[8];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int ch = 8 * q + 4 * t + i;
                            float v = c2[t][i] + tb[ch < tcout ? ch : 0];
#ifndef PN_TAIL_FAKE_NOACT
                            // the two sigmoid casts of the heads (rtpose_light3d.py:335-337) or none: same expressions as pn_activate
                            if (tact == PN_ACT_SIG_PM2) v = (pn_sigmoid(v) - 0.5f) * 4.f;
                            else if (tact == PN_ACT_SIG) v = pn_sigmoid(v);
#endif
                            v8[4 * t + i] = ch < tcout ? v : 0.f;
                            o8[4 * t + i] = (T)v8[4 * t + i];
                        }
                    const int nvalid = tcout - 8 * q;
#ifndef PN_TAIL_FAKE_NOSTORE
                    if (tout && nvalid > 0) {
                        PN_GLOBAL T *op = tout + opix * (size_t)P.tail_out_cs + 8 * q;
                        if (nvalid >= 8) *reinterpret_cast<PN_GLOBAL u32x4 *>(op) = *reinterpret_cast<u32x4 *>(o8);
                        else if (nvalid == 4) *reinterpret_cast<PN_GLOBAL u32x2 *>(op) = *reinterpret_cast<u32x2 *>(o8);
                        else
                            for (int k = 0; k < nvalid; ++k) op[k] = o8[k];
                    }
                    if (tnchw)
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (8 * q + k < tcout) tnchw[((size_t)b * tcout + 8 * q + k) * hw + (size_t)(oy0 + ry) * Wo + (ox0 + rx)] = v8[k];
#else
                    if (v8[0] == 123.456f && nvalid > 0 && tnchw) tnchw[0] = v8[1] + v8[2] + v8[3] + v8[4] + v8[5] + v8[6] + v8[7];
#endif
                }
            }
            PN_STAMP_AT(12);
            return;
        }
    }

    // ---- epilogue: identical contract to conv_mfma_kernel.h (permuted cout rows, direct stores) ----
    PN_STAMP_AT(11);
    constexpr int LC = CT * 4;
    const int cw = (cb * WC + wc) * (CT * 16) + LC * q;
    const int cout = P.cout, act = P.act;
    float bias[LC];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 b4 = *reinterpret_cast<const PN_GLOBAL f32x4 *>((const PN_GLOBAL float *)P.bias + cw + 4 * ct);
        bias[4 * ct + 0] = b4[0]; bias[4 * ct + 1] = b4[1]; bias[4 * ct + 2] = b4[2]; bias[4 * ct + 3] = b4[3];
    }
    const bool full = cw + LC <= cout;
    const PN_GLOBAL T *res_base = P.res ? (const PN_GLOBAL T *)P.res + P.res_coff + cw : nullptr;
    PN_GLOBAL T *out_base = P.out ? (PN_GLOBAL T *)P.out + P.out_coff + cw : nullptr;
    PN_GLOBAL float *nchw = (PN_GLOBAL float *)P.out_nchw;
    const int res_cs = P.res_cs, out_cs = P.out_cs, Ho = P.Ho, naf = P.yolo_naf;
    const int split = P.split, res_split = P.res_split;
    const int pix0 = (b * Ho + oy0) * Wo + ox0;
    auto finish = [&](auto actc) {
        constexpr int ACT = decltype(actc)::value;
#pragma clang loop unroll(full)                          // a rolled loop would index acc[] dynamically = scratch memory
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
                    *reinterpret_cast<u32x4 *>(rv) = *reinterpret_cast<const PN_GLOBAL u32x4 *>(rp);
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
#ifdef PN_CONV3_NT_STORE
                        __builtin_nontemporal_store(*reinterpret_cast<u32x4 *>(ov), reinterpret_cast<PN_GLOBAL u32x4 *>(op));
#else
                        *reinterpret_cast<PN_GLOBAL u32x4 *>(op) = *reinterpret_cast<u32x4 *>(ov);
#endif
                    } else {
                        for (int k = 0; k < LC; ++k)
                            if (cw + k < cout) op[k] = ov[k];
                    }
                }
            }
            if (nchw) {
                const size_t hw = (size_t)Ho * Wo;
                PN_GLOBAL float *np = nchw + ((size_t)b * cout + cw) * hw + (size_t)(oy0 + ry) * Wo + (ox0 + rx);
                for (int k = 0; k < LC; ++k)
                    if (cw + k < cout) np[(size_t)k * hw] = v[k];
            }
        }
    };
#if defined(PN_CONV3_HALFMAJOR) || defined(PN_CONV3_QUARTERMAJOR)
#if defined(PN_CONV3_QUARTERMAJOR)
#define PN3_PIXOF(ba) (((unsigned)(ba) - (unsigned)((q >> 1) * (IMG / 4))) >> 5)
#else
#define PN3_PIXOF(ba) ((unsigned)(ba) >> 6)
#endif
    // Fast path, chosen per WAVE (every condition is wave-uniform, so no exec-mask branching): all 32 couts of the wave
    // exist, NHWC output only, a compile-time activation.  The pixel of a slot is recovered from its LDS read address
    // (baddr = q*16 + (ry*32 + rx)*64) instead of being divided out again; same arithmetic on the values as `finish`,
    // so the results are bit-identical (tests: conv3 == generic kernel).
    const int wave_c0 = (cb * WC + wc) * (CT * 16);
    const bool fast = (PN_CONV3_FAST_EPILOGUE == 1 || (PN_CONV3_FAST_EPILOGUE == 2 && WC == 4)) &&
                      wave_c0 + CT * 16 <= cout && P.out && !nchw && !split && !res_split && (act == PN_ACT_RELU || act == PN_ACT_LEAKY || act == PN_ACT_NONE);
    auto finish_fast = [&](auto actc, auto resc) {
        constexpr int ACT = decltype(actc)::value;
        constexpr bool RES = decltype(resc)::value;
        PN_GLOBAL T *ob = (PN_GLOBAL T *)P.out + P.out_coff + wave_c0;
        const PN_GLOBAL T *rb = (const PN_GLOBAL T *)P.res + P.res_coff + wave_c0;
        const unsigned lane_c = (unsigned)(LC * q);
        int ba[PT];                                       // opaque copies: keep the pixel arithmetic HERE (hoisted above the K loop it only adds spills)
#pragma clang loop unroll(full)
        for (int pt = 0; pt < PT; ++pt) { ba[pt] = baddr[pt]; asm volatile("" : "+v"(ba[pt])); }
        u32x4 rq[PT];
        if (RES) {
#pragma clang loop unroll(full)
            for (int pt = 0; pt < PT; ++pt) {
                const unsigned t = PN3_PIXOF(ba[pt]);
                const unsigned opix = (unsigned)pix0 + (t >> 5) * (unsigned)Wo + (t & 31u);
                rq[pt] = *reinterpret_cast<const PN_GLOBAL u32x4 *>(rb + (opix * (unsigned)res_cs + lane_c));
            }
        }
#pragma clang loop unroll(full)
        for (int pt = 0; pt < PT; ++pt) {
            const int slot = (wp * PT + pt) * 16 + c;
            const unsigned t = PN3_PIXOF(ba[pt]);
            const unsigned opix = (unsigned)pix0 + (t >> 5) * (unsigned)Wo + (t & 31u);
            float v[LC];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * ct + i] = acc[ct][pt][i] + bias[4 * ct + i];
            if (RES) {
                T rv[LC];
                *reinterpret_cast<u32x4 *>(rv) = rq[pt];
#pragma unroll
                for (int k = 0; k < LC; ++k) v[k] += (float)rv[k];
            }
#pragma unroll
            for (int k = 0; k < LC; ++k) {
                if (ACT == PN_ACT_RELU) v[k] = v[k] > 0.f ? v[k] : 0.f;
                else if (ACT == PN_ACT_LEAKY) v[k] = v[k] > 0.f ? v[k] : v[k] * 0.1f;
            }
            T ov[LC];
#pragma unroll
            for (int k = 0; k < LC; ++k) ov[k] = (T)v[k];
            if (slot < npix) *reinterpret_cast<PN_GLOBAL u32x4 *>(ob + (opix * (unsigned)out_cs + lane_c)) = *reinterpret_cast<u32x4 *>(ov);
        }
    };
    if (fast) {
        if (P.res) {
            if (act == PN_ACT_RELU) finish_fast(std::integral_constant<int, PN_ACT_RELU>{}, std::true_type{});
            else if (act == PN_ACT_LEAKY) finish_fast(std::integral_constant<int, PN_ACT_LEAKY>{}, std::true_type{});
            else finish_fast(std::integral_constant<int, PN_ACT_NONE>{}, std::true_type{});
        } else {
            if (act == PN_ACT_RELU) finish_fast(std::integral_constant<int, PN_ACT_RELU>{}, std::false_type{});
            else if (act == PN_ACT_LEAKY) finish_fast(std::integral_constant<int, PN_ACT_LEAKY>{}, std::false_type{});
            else finish_fast(std::integral_constant<int, PN_ACT_NONE>{}, std::false_type{});
        }
        PN_STAMP_AT(12);
        return;
    }
#endif
    if (act == PN_ACT_RELU) finish(std::integral_constant<int, PN_ACT_RELU>{});
    else if (act == PN_ACT_LEAKY) finish(std::integral_constant<int, PN_ACT_LEAKY>{});
    else if (act == PN_ACT_NONE) finish(std::integral_constant<int, PN_ACT_NONE>{});
    else finish(std::integral_constant<int, -1>{});
    PN_STAMP_AT(12);
}

template <int KS, int WC, int WP, int NBUF, int PT, int RPG = PT * 4 / 7, int TAIL = 0>
__global__ __launch_bounds__(WC * WP * 64, PT == 14 ? 2 : (NBUF == 2 ? 3 : PN_CONV3_OCC)) void conv3_kernel(const ConvProblem *__restrict__ probs) {
    conv3_body<KS, WC, WP, NBUF, PT, RPG, TAIL>(probs[blockIdx.y]);
}

// One launch for the kinds of 128-cout blocks a level holds side by side: a stage's fourth level (rtpose_light3d.py:263-309) = the 3x3
// convolutions of the heat / depth branches + the fused 1x1 + 1x1 tail of the PAF branch; layer2's first level = the BasicBlock's
// first 3x3 + its 1x1 shortcut.  Alone, neither fills the chip (448 and 224 blocks of
// 4 waves for 1 024 SIMDs x 4 slots) and each pays its own ramp, drain and kernel boundary; a block picks its body by a scalar
// test of its problem (same block shape, same 128-register budget; LDS = the larger of the two).  Same code per block as the
// separate launches: results are bit-identical (POPNET_NO_MIX=1 keeps the two launches).
template <int UNUSED = 0>      // a template only so that the header can hold it: instantiated in conv3_inst_0.hip alone
__global__ __launch_bounds__(256, PN_CONV3_OCC) void conv3_mix_kernel(const ConvProblem *__restrict__ probs) {
    const ConvProblem &P = probs[blockIdx.y];
    if (P.tail_w) conv3_body<1, 4, 1, 1, 7, 4, 1>(P);
    else if (P.ks == 1) conv3_body<1, 4, 1, 1, 7, 4, 0>(P);      // a plain 1x1 sibling (the 1x1 shortcut next to a BasicBlock's first 3x3, resnet.py:59-77)
    else conv3_body<3, 4, 1, 1, 7, 4, 0>(P);
}

template <int KS, int WC, int WP, int NBUF, int PT = 7, int RPG = PT * 4 / 7, int TAIL = 0>
static int conv3_launch_one(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    auto kern = conv3_kernel<KS, WC, WP, NBUF, PT, RPG, TAIL>;
    if (L.lds_bytes > 48 * 1024) {
        static PnLdsAttr attr;          // per instantiation, per device
        if (int rc = pn_lds_attr(ctx, attr, reinterpret_cast<const void *>(kern), L.lds_bytes)) return rc;
    }
    hipLaunchKernelGGL(kern, dim3(L.max_blocks, L.nprob), dim3(WC * WP * 64), L.lds_bytes, stream, L.probs_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}
#define PN3_CASE(KS, WC, WP, NB) \
    if (!L.mix && L.ks == KS && L.wc == WC && L.wp == WP && L.nbuf == NB && L.pt == 7 && L.rpg == 4 && !L.tail) return conv3_launch_one<KS, WC, WP, NB>(ctx, L, stream);
#define PN3_CASE_TAIL(KS, WC, WP, NB) \
    if (L.ks == KS && L.wc == WC && L.wp == WP && L.nbuf == NB && L.pt == 7 && L.rpg == 4 && L.tail == 1) return conv3_launch_one<KS, WC, WP, NB, 7, 4, 1>(ctx, L, stream);
#define PN3_CASE_POOLTAIL(KS, WC, WP, NB) \
    if (L.ks == KS && L.wc == WC && L.wp == WP && L.nbuf == NB && L.pt == 7 && L.rpg == 8 && L.tail == 2) return conv3_launch_one<KS, WC, WP, NB, 7, 8, 2>(ctx, L, stream);
#define PN3_CASE_PT(KS, WC, WP, NB, PT_) \
    if (L.ks == KS && L.wc == WC && L.wp == WP && L.nbuf == NB && L.pt == PT_ && L.rpg == PT_ * 4 / 7) return conv3_launch_one<KS, WC, WP, NB, PT_>(ctx, L, stream);
#define PN3_CASE_RPG(KS, WC, WP, NB, RPG_) \
    if (L.ks == KS && L.wc == WC && L.wp == WP && L.nbuf == NB && L.pt == 7 && L.rpg == RPG_) return conv3_launch_one<KS, WC, WP, NB, 7, RPG_>(ctx, L, stream);
int pn_launch_conv3_part0(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv3_part1(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv3_part2(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
