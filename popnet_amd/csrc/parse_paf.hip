// Open-Pose+ pose parsing on the GPU: heat-map NMS with bicubic sub-cell refinement, part-affinity
// limb scoring, greedy one-to-one matching, person assembly, depth read-out and back-projection.
//
// Replaces the reference's per-frame NumPy/SciPy/cv2 post-processing (Python loops on the host):
//   find_peaks / NMS              tpm/lib/utils/paf_to_pose.py:33-153
//   find_connected_joints         tpm/lib/utils/paf_to_pose.py:156-264
//   group_limbs_of_same_person    tpm/lib/utils/paf_to_pose.py:267-351
//   paf_to_human_list             tpm/lib/utils/common.py:5-32
//   retrieve_depth_heat_weighted  tpm/lib/utils/common.py:272-293
//   read-out / rescale / pinhole  tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:179-262
//
// Three launches per batch, all HBM/L2-latency bound (185 KB of maps in, <30 KB of records out per
// frame); none materialises the reference's 224x224x28 up-sampled PAF tensor (5.6 MB/frame): the
// x8 bicubic value is evaluated at the ten sample points of each candidate limb only.
//   k1  one workgroup per (frame, joint map): peak flags -> ordered ballot compaction -> one wave
//       per peak up-samples its <=5x5 patch x8 (<=40x40) and wave-reduces the first arg-max.
//   k2  one workgroup per (frame, limb): lanes = (candidate pair, sample point); stable rank sort;
//       greedy matching.
//   k3  one wave per frame: sequential person assembly with wave-parallel row search, pruning,
//       heat-weighted depth read-out, rescale and back-projection into fixed-size records.
//
// Arithmetic contract (bit-exact against oracle/parse_paf.py): float32 for everything the
// reference computes from float32 maps (bicubic taps in OpenCV's order: horizontal pass then
// vertical pass, products summed left to right), float64 for what NumPy promotes to float64
// (sample coordinates, dot products, means, person scores, rescale, back-projection), NumPy's
// pairwise summation order for np.mean / np.sum, round-half-even for np.round, and no fused
// multiply-add anywhere in this file.
#pragma clang fp contract(off)
#include <algorithm>
#include <cmath>
#include "pn_internal.h"

#define J_ PN_NUM_JOINTS
#define L_ PN_NUM_LIMBS
#define MAXP PN_MAX_PEAKS_PER_JOINT
#define MAXC PN_MAX_CONN_PER_LIMB
#define MAX_MAP 4096      // largest h*w the parse kernels stage in LDS (64 KB static-LDS budget)

// limb topology: util/util_functions.py:17-34 == tpm/lib/datasets/datasets_itop_rtpose.py:45-62
__constant__ int c_limb_src[L_] = {8, 9, 11, 8, 10, 12, 8, 1, 2, 4, 1, 3, 5, 1};
__constant__ int c_limb_dst[L_] = {9, 11, 13, 10, 12, 14, 1, 2, 4, 6, 3, 5, 7, 0};

struct CubicTab { float c[8][4]; };   // phase p: fractional offset (2p+1)/16

// LDS hand-over between lanes of ONE wave (a wave's LDS instructions execute in order; the waits and the compiler fence
// make the earlier writes visible to the later reads of other lanes).  Used where the waves of a block run loops of
// different trip counts, so a block barrier is not available.
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

struct ParseWs {                       // per-frame scratch between the three kernels
    int peak_count[J_];                // uncapped count (overflow detection)
    float peak_x[J_][MAXP], peak_y[J_][MAXP], peak_s[J_][MAXP];
    int conn_count[L_];
    int conn_i[L_][MAXC], conn_j[L_][MAXC];
    double conn_s[L_][MAXC];
};

// interpolateCubic(x, coeffs), A = -0.75, float32 -- same expression order as oracle/cv2_resize.py
static void host_cubic_coeffs(float x, float *c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    const float xm = 1.f - x;
    c[2] = ((A + 2.f) * xm - (A + 3.f)) * xm * xm + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

extern "C" void pn_debug_cubic_coeffs(float x, float *out4) { host_cubic_coeffs(x, out4); }

// destination index d of an x8 up-sampling -> first tap (sx - 1) and coefficient phase
__device__ __forceinline__ void up8_src(int d, int &s0, int &phase) {
    int t = 2 * d - 7;             // (d + 0.5) / 8 - 0.5 = t / 16, t odd
    s0 = (t >> 4) - 1;             // floor(t/16) - 1
    phase = (t & 15) >> 1;
}

// value of cv2.resize(src, fx=8, fy=8, INTER_CUBIC)[uy, ux] for a [sh, sw] float32 image with row
// stride `ld` (replicate border): horizontal pass on the four source rows, then vertical pass.
__device__ __forceinline__ float bicubic8(const float *src, int ld, int sh, int sw, int uy, int ux, const CubicTab &tab) {
    int sx0, px, sy0, py;
    up8_src(ux, sx0, px);
    up8_src(uy, sy0, py);
    int cx[4], cy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cx[k] = min(max(sx0 + k, 0), sw - 1);
        cy[k] = min(max(sy0 + k, 0), sh - 1);
    }
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float *row = src + cy[r] * ld;
        float h = row[cx[0]] * tab.c[px][0];
        h = h + row[cx[1]] * tab.c[px][1];
        h = h + row[cx[2]] * tab.c[px][2];
        h = h + row[cx[3]] * tab.c[px][3];
        float t = h * tab.c[py][r];
        v = (r == 0) ? t : v + t;
    }
    return v;
}

// ---------------------------------------------------------------------------------------------
// k1: peaks + refinement.  grid = (J, B), block = 256.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void peaks_refine_kernel(const float *__restrict__ heat, int h, int w, int heat_c,
                                                            float thresh, CubicTab tab, ParseWs *__restrict__ ws) {
    __shared__ float map[MAX_MAP];
    __shared__ int s_wave_cnt[4];
    __shared__ unsigned short s_wlist[4][MAXP];      // per-wave ordered peak lists (cell index)
    __shared__ int s_px[MAXP], s_py[MAXP];
    __shared__ float s_h[4][5 * 40];                 // per wave: horizontal pass of the patch being refined
    __shared__ __attribute__((aligned(16))) float s_tab[8][4];   // the cubic coefficients: indexed by a per-lane phase in the refinement loops -- from the kernel
                                                     // argument that index is a vector-memory load per loop iteration (25 dependent ones per peak, round 5)
    const int joint = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = h * w;
    const float *src = heat + ((size_t)b * heat_c + joint) * hw;
    for (int i = tid; i < hw; i += 256) map[i] = src[i];
    if (tid < 32) s_tab[tid >> 2][tid & 3] = tab.c[tid >> 2][tid & 3];
    __syncthreads();

    // peak <=> v == max over the 4-connected cross (scipy 'reflect': an out-of-image neighbour is
    // the pixel itself) and v > thresh.  Row-major order (= the reference's ids) is kept by an ordered
    // compaction: each wave scans one contiguous quarter of the map and compacts with ballots only
    // (no block barrier inside the scan); the four lists are concatenated by prefix counts.
    {
        const int Q = (hw + 3) / 4;
        const int lo = wave * Q, hi = min(hw, lo + Q);
        int wcnt = 0;
        for (int base = lo; base < hi; base += 64) {
            const int i = base + lane;
            bool pk = false;
            if (i < hi) {
                const int y = i / w, x = i - y * w;
                const float v = map[i];
                float m = v;
                if (y > 0) m = fmaxf(m, map[i - w]);
                if (y < h - 1) m = fmaxf(m, map[i + w]);
                if (x > 0) m = fmaxf(m, map[i - 1]);
                if (x < w - 1) m = fmaxf(m, map[i + 1]);
                pk = (m == v) && (v > thresh);
            }
            const unsigned long long bal = __ballot(pk);
            const int pos = wcnt + __popcll(bal & ((1ull << lane) - 1ull));
            if (pk && pos < MAXP) s_wlist[wave][pos] = (unsigned short)i;
            wcnt += __popcll(bal);
        }
        if (lane == 0) s_wave_cnt[wave] = wcnt;
    }
    __syncthreads();
    const int c0 = s_wave_cnt[0], c1 = s_wave_cnt[1], c2 = s_wave_cnt[2], c3 = s_wave_cnt[3];
    const int s_total = c0 + c1 + c2 + c3;
    if (tid < MAXP && tid < s_total) {              // global order = wave 0's list, then wave 1's, ...
        int p = tid, cell;
        if (p < c0) cell = s_wlist[0][p];
        else if (p < c0 + c1) cell = s_wlist[1][p - c0];
        else if (p < c0 + c1 + c2) cell = s_wlist[2][p - c0 - c1];
        else cell = s_wlist[3][p - c0 - c1 - c2];
        s_px[tid] = cell % w;
        s_py[tid] = cell / w;
    }
    __syncthreads();
    const int total = s_total;
    ParseWs &W = ws[b];
    if (tid == 0) W.peak_count[joint] = total;
    const int n = min(total, MAXP);

    // one wave per peak: x8 bicubic of the clipped 5x5 patch, first arg-max in row-major order.  The up-sampling is
    // evaluated separably, exactly as cv2 does it (and as bicubic8() does per point): the horizontal pass of each of the
    // <= 5 source rows once (<= 5 x 40 values, kept in LDS), then the vertical pass per output pixel -- the same float32
    // operations in the same order as the point-wise form, 4 instead of 16 taps per output pixel.
    for (int p = wave; p < n; p += 4) {
        const int px = s_px[p], py = s_py[p];
        const int x_min = max(0, px - 2), y_min = max(0, py - 2);
        const int x_max = min(w - 1, px + 2), y_max = min(h - 1, py + 2);
        const int pw = x_max - x_min + 1, ph = y_max - y_min + 1;
        const int uw = pw * 8, un = uw * ph * 8;
        const float inv_uw = 1.0f / (float)uw;
        const float *patch = map + y_min * w + x_min;
        float *hb = s_h[wave];
        // (fixed trip counts -- <= 5 x 40 and <= 40 x 40 values -- so that the LDS reads of several iterations are in flight; same values in the
        // same ascending order per lane)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int i = lane + 64 * it;
            if (i < ph * uw) {
                const int r = (int)(((float)i + 0.5f) * inv_uw), ux = i - r * uw;
                int sx0, phx;
                up8_src(ux, sx0, phx);
                const float *row = patch + r * w;
                const float4 cf = *reinterpret_cast<const float4 *>(s_tab[phx]);
                float hv = row[min(max(sx0, 0), pw - 1)] * cf.x;
                hv = hv + row[min(max(sx0 + 1, 0), pw - 1)] * cf.y;
                hv = hv + row[min(max(sx0 + 2, 0), pw - 1)] * cf.z;
                hv = hv + row[min(max(sx0 + 3, 0), pw - 1)] * cf.w;
                hb[r * 40 + ux] = hv;
            }
        }
        WAVE_LDS_SYNC();
        float best = -INFINITY;
        int best_i = 0x7fffffff;
#pragma unroll 5
        for (int it = 0; it < 25; ++it) {
            const int i = lane + 64 * it;
            if (i < un) {
                const int uy = (int)(((float)i + 0.5f) * inv_uw), ux = i - uy * uw;
                int sy0, phy;
                up8_src(uy, sy0, phy);
                const float4 cf = *reinterpret_cast<const float4 *>(s_tab[phy]);
                float v = hb[min(max(sy0, 0), ph - 1) * 40 + ux] * cf.x;
                v = v + hb[min(max(sy0 + 1, 0), ph - 1) * 40 + ux] * cf.y;
                v = v + hb[min(max(sy0 + 2, 0), ph - 1) * 40 + ux] * cf.z;
                v = v + hb[min(max(sy0 + 3, 0), ph - 1) * 40 + ux] * cf.w;
                if (v > best || best_i == 0x7fffffff) { best = v; best_i = i; }   // i ascending: first max kept
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            float ov = __shfl_xor(best, off);
            int oi = __shfl_xor(best_i, off);
            if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
        }
        if (lane == 0) {
            int my = (int)(((float)best_i + 0.5f) * inv_uw), mx = best_i - my * uw;
            W.peak_x[joint][p] = (float)(8 * x_min + mx);
            W.peak_y[joint][p] = (float)(8 * y_min + my);
            W.peak_s[joint][p] = best;
        }
        WAVE_LDS_SYNC();                                 // the next peak of this wave reuses hb
    }
}

// round-half-even of i*step + start (np.round(np.linspace(...))) as int
__device__ __forceinline__ int linspace_round(double start, double stop, double step, int i, int num) {
    double v = (i == num - 1) ? stop : ((double)i * step + start);
    return (int)rint(v);
}

// ---------------------------------------------------------------------------------------------
// k2: limb scoring + greedy matching.  grid = (L, B), block = 256.
// ---------------------------------------------------------------------------------------------
#define PAIRS_PER_PASS 25      // 25 pairs x 10 sample points = 250 lanes busy per pass
__global__ __launch_bounds__(256) void limb_match_kernel(const float *__restrict__ paf, int h, int w, int paf_c,
                                                          float thresh_paf, int up_h, CubicTab tab,
                                                          ParseWs *__restrict__ ws) {
    __shared__ float pmap[2][MAX_MAP];
    __shared__ double s_pts[PAIRS_PER_PASS][10];
    __shared__ double s_cand_s[MAXP * MAXP];
    __shared__ unsigned char s_cand_i[MAXP * MAXP], s_cand_j[MAXP * MAXP];
    __shared__ unsigned short s_order[MAXP * MAXP];
    __shared__ int s_ncand;
    __shared__ CubicTab s_tab;                   // (per-lane phases index it: LDS reads instead of vector-memory loads from the kernel argument)
    const int limb = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    ParseWs &W = ws[b];
    const int jsrc = c_limb_src[limb], jdst = c_limb_dst[limb];
    const int ns = min(W.peak_count[jsrc], MAXP), nd = min(W.peak_count[jdst], MAXP);
    if (ns == 0 || nd == 0) {
        if (tid == 0) W.conn_count[limb] = 0;
        return;
    }
    const int hw = h * w;
    const float *px_map = paf + ((size_t)b * paf_c + 2 * limb) * hw;
    for (int i = tid; i < hw; i += 256) {
        pmap[0][i] = px_map[i];
        pmap[1][i] = px_map[hw + i];
    }
    if (tid == 0) s_ncand = 0;
    if (tid < 32) s_tab.c[tid >> 2][tid & 3] = tab.c[tid >> 2][tid & 3];
    __syncthreads();

    const int npairs = ns * nd;
    for (int pbase = 0; pbase < npairs; pbase += PAIRS_PER_PASS) {
        const int lp = tid / 10, pt = tid - lp * 10;
        const int pair = pbase + lp;
        const bool active = lp < PAIRS_PER_PASS && pair < npairs;
        double sx = 0, sy = 0, dxn = 0, dyn = 0, dist = 1;
        if (active) {
            const int i = pair / nd, j = pair - i * nd;
            sx = (double)W.peak_x[jsrc][i]; sy = (double)W.peak_y[jsrc][i];
            const double ex = (double)W.peak_x[jdst][j], ey = (double)W.peak_y[jdst][j];
            const double ddx = ex - sx, ddy = ey - sy;
            dist = sqrt(ddx * ddx + ddy * ddy) + 1e-8;
            dxn = ddx / dist; dyn = ddy / dist;
            // np.linspace(start, stop, 10): step = (stop - start) / 9; y_i = i * step + start; y_9 = stop
            const double stepx = (ex - sx) / 9.0, stepy = (ey - sy) / 9.0;
            const int qx = linspace_round(sx, ex, stepx, pt, 10);
            const int qy = linspace_round(sy, ey, stepy, pt, 10);
            const float vx = bicubic8(pmap[0], w, h, w, qy, qx, s_tab);
            const float vy = bicubic8(pmap[1], w, h, w, qy, qx, s_tab);
            s_pts[lp][pt] = (double)vx * dxn + (double)vy * dyn;
        }
        __syncthreads();
        // lanes 0..24 of wave 0 finish one pair each; ordered compaction keeps src-major order
        if (tid < 64) {
            bool ok = false;
            double score = 0;
            const int pr = pbase + tid;
            if (tid < PAIRS_PER_PASS && pr < npairs) {
                const double *s = s_pts[tid];
                // np.mean of 10 float64: pairwise sum order of numpy for 8 <= n < 128
                double res = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                res = res + s[8];
                res = res + s[9];
                const double mean = res / 10.0;
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < 10; ++k) cnt += (s[k] > (double)thresh_paf) ? 1 : 0;
                // recompute this pair's length (lane-local): penalty min(0.5*H/dist - 1, 0)
                const int i = pr / nd, j = pr - i * nd;
                const double ax = (double)W.peak_x[jsrc][i], ay = (double)W.peak_y[jsrc][i];
                const double bx = (double)W.peak_x[jdst][j], by = (double)W.peak_y[jdst][j];
                const double ddx = bx - ax, ddy = by - ay;
                const double dd = sqrt(ddx * ddx + ddy * ddy) + 1e-8;
                const double pen = fmin(0.5 * (double)up_h / dd - 1.0, 0.0);
                score = mean + pen;
                ok = (cnt > 8) && (score > 0.0);      // > 0.8 * num_intermed_pts
            }
            unsigned long long bal = __ballot(ok);
            const int basec = s_ncand;
            if (ok) {
                const int pos = basec + __popcll(bal & ((1ull << tid) - 1ull));
                s_cand_s[pos] = score;
                s_cand_i[pos] = (unsigned char)(pr / nd);
                s_cand_j[pos] = (unsigned char)(pr % nd);
            }
            if (tid == 0) s_ncand = basec + __popcll(bal);
        }
        __syncthreads();
    }

    // stable descending sort by score (Python sorted(..., reverse=True) keeps insertion order on ties)
    const int nc = s_ncand;
    for (int k = tid; k < nc; k += 256) {
        const double sk = s_cand_s[k];
        int rank = 0;
        for (int m = 0; m < nc; ++m) {
            const double sm = s_cand_s[m];
            rank += (sm > sk || (sm == sk && m < k)) ? 1 : 0;
        }
        s_order[rank] = (unsigned short)k;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long used_i = 0, used_j = 0;
        const int maxc = min(ns, nd);
        int n = 0;
        for (int r = 0; r < nc && n < maxc; ++r) {
            const int k = s_order[r];
            const int i = s_cand_i[k], j = s_cand_j[k];
            if (!((used_i >> i) & 1ull) && !((used_j >> j) & 1ull)) {
                used_i |= 1ull << i;
                used_j |= 1ull << j;
                W.conn_i[limb][n] = i;
                W.conn_j[limb][n] = j;
                W.conn_s[limb][n] = s_cand_s[k];
                ++n;
            }
        }
        W.conn_count[limb] = n;
    }
}

// ---------------------------------------------------------------------------------------------
// k3: person assembly + read-out.  grid = B, block = 64 (one wave).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_readout_kernel(const float *__restrict__ heat, const float *__restrict__ z,
                                                            int h, int w, int heat_c, int z_c, pn_parse_cfg cfg,
                                                            const ParseWs *__restrict__ ws, pn_pose_frame *__restrict__ frames,
                                                            pn_pose_wire *__restrict__ wire) {
    __shared__ double rows[PN_MAX_PERSONS][J_ + 2];
    __shared__ int s_base[J_ + 1];
    __shared__ __attribute__((aligned(16))) unsigned s_ws[(sizeof(ParseWs) + 7) / 8 * 2];
    __shared__ unsigned long long s_kbal;
    __shared__ int s_nkeep;
    // 256 threads: the scratch staging, the joint list and the read-out use all four waves; the serial person assembly
    // in between runs on wave 0 alone (the other waves wait at the block barrier behind it)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The assembly below is a serial walk over connections: served from global memory every step was a
    // dependent ~1 us load (measured 25 us per launch).  Stage this frame's 13 KB of scratch in LDS once.
    {
        static_assert(sizeof(ParseWs) % 8 == 0, "staged in 8-byte pieces");
        // (round 5: every piece of the thread in flight at once -- the plain loop compiled to one load, one s_waitcnt vmcnt(0) per iteration: seven
        // dependent round trips in front of everything else the block does)
        const uint2 *src = reinterpret_cast<const uint2 *>(&ws[b]);
        constexpr int NW8 = (int)(sizeof(ParseWs) / 8), NIT = (NW8 + 255) / 256;
        uint2 stg[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) stg[it] = src[min(tid + 256 * it, NW8 - 1)];
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (tid + 256 * it < NW8) reinterpret_cast<uint2 *>(s_ws)[tid + 256 * it] = stg[it];
        // rows beyond n_peaks / n_persons are never written below: the record starts as zeros, so that it is a pure function of its frame
        // (bit-identical wherever in a batch, and in whatever buffer, the frame was processed).  Round 4: zeroed HERE, by the block that owns
        // the record (the barrier below drains these stores -- vmcnt(0) -- before any thread writes a result), instead of by a 1 MB memset
        // launch in front of the three parse kernels.
        static_assert(sizeof(pn_pose_frame) % 16 == 0, "zeroed in 16-byte pieces");
        uint4 *fz = reinterpret_cast<uint4 *>(&frames[b]);
        for (int i = tid; i < (int)(sizeof(pn_pose_frame) / 16); i += 256) fz[i] = uint4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    const ParseWs &W = *reinterpret_cast<const ParseWs *>(s_ws);
    pn_pose_frame &F = frames[b];
    unsigned status = 0;

    if (tid == 0) {
        int acc = 0;
        for (int j = 0; j < J_; ++j) {
            s_base[j] = acc;
            acc += min(W.peak_count[j], MAXP);
        }
        s_base[J_] = acc;
    }
    for (int j = 0; j < J_; ++j)
        if (W.peak_count[j] > MAXP) status |= PN_FRAME_OVERFLOW_PEAKS;      // every thread: uniform
    __syncthreads();
    const int npeaks = s_base[J_];
    // joint_list rows (id == row index)
    for (int j = 0; j < J_; ++j) {
        const int n = min(W.peak_count[j], MAXP);
        for (int k = tid; k < n; k += 256) {
            const int id = s_base[j] + k;
            F.peak_x[id] = W.peak_x[j][k];
            F.peak_y[id] = W.peak_y[j][k];
            F.peak_score[id] = W.peak_s[j][k];
            F.peak_type[id] = j;
        }
    }

    // ---- group_limbs_of_same_person (paf_to_pose.py:280-335), wave-uniform control flow, wave 0 only ----
    int np = 0;
    if (wave == 0) {
    for (int limb = 0; limb < L_; ++limb) {
        const int st = c_limb_src[limb], dt = c_limb_dst[limb];
        const int nconn = W.conn_count[limb];
        for (int cidx = 0; cidx < nconn; ++cidx) {
            const int ci = W.conn_i[limb][cidx], cj = W.conn_j[limb][cidx];
            const double src_id = (double)(s_base[st] + ci), dst_id = (double)(s_base[dt] + cj);
            const double lscore = W.conn_s[limb][cidx];
            const double s_src = (double)W.peak_s[st][ci], s_dst = (double)W.peak_s[dt][cj];
            bool hit = false;
            if (lane < np) hit = (rows[lane][st] == src_id) || (rows[lane][dt] == dst_id);
            const unsigned long long bal = __ballot(hit);
            const int nh = __popcll(bal);
            if (nh == 1) {
                const int p = __ffsll((long long)bal) - 1;
                if (lane == 0 && rows[p][dt] != dst_id) {
                    rows[p][dt] = dst_id;
                    rows[p][J_ + 1] += 1.0;
                    rows[p][J_] += s_dst + lscore;
                }
            } else if (nh == 2) {
                const int p1 = __ffsll((long long)bal) - 1;
                const int p2 = __ffsll((long long)(bal & (bal - 1))) - 1;
                bool both = false;
                if (lane < J_) both = (rows[p1][lane] >= 0.0) && (rows[p2][lane] >= 0.0);
                const bool overlap = __ballot(both) != 0ull;
                if (!overlap) {
                    if (lane < J_) rows[p1][lane] += rows[p2][lane] + 1.0;
                    if (lane == 0) {
                        rows[p1][J_] += rows[p2][J_];
                        rows[p1][J_ + 1] += rows[p2][J_ + 1];
                        rows[p1][J_] += lscore;
                    }
                    WAVE_LDS_SYNC();
                    // person_to_joint_assoc.pop(p2): shift the later rows down by one
                    double tmp[J_ + 2];
                    const bool mv = lane >= p2 && lane < np - 1;
                    if (mv)
                        for (int k = 0; k < J_ + 2; ++k) tmp[k] = rows[lane + 1][k];
                    WAVE_LDS_SYNC();
                    if (mv)
                        for (int k = 0; k < J_ + 2; ++k) rows[lane][k] = tmp[k];
                    --np;
                } else if (lane == 0) {
                    rows[p1][dt] = dst_id;
                    rows[p1][J_ + 1] += 1.0;
                    rows[p1][J_] += s_dst + lscore;
                }
            } else {
                if (np < PN_MAX_PERSONS) {
                    if (lane < J_) rows[np][lane] = (lane == st) ? src_id : ((lane == dt) ? dst_id : -1.0);
                    if (lane == 0) {
                        rows[np][J_ + 1] = 2.0;
                        rows[np][J_] = (s_src + s_dst) + lscore;      // sum([a, b]) + c
                    }
                    ++np;
                } else {
                    status |= PN_FRAME_OVERFLOW_PERSONS;
                }
            }
            WAVE_LDS_SYNC();
        }
    }

    // ---- prune (paf_to_pose.py:338-346) + read-out ----
    bool keep = false;
    if (lane < np) {
        const double cnt = rows[lane][J_ + 1], sc = rows[lane][J_];
        keep = !(cnt < 3.0 || sc / cnt < 0.2);
    }
    const unsigned long long kbal0 = __ballot(keep);
    if (keep) {
        const unsigned long long kbal = kbal0;
        const int o = __popcll(kbal & ((1ull << lane) - 1ull));
        for (int j = 0; j < J_; ++j) F.person_joint[o][j] = (int)rows[lane][j];
        F.person_score[o] = rows[lane][J_];
        F.person_count[o] = (int)rows[lane][J_ + 1];
    }
    if (lane == 0) {
        F.n_persons = __popcll(kbal0);
        F.n_peaks = npeaks;
        F.status = status;
        F.reserved = 0;
        s_kbal = kbal0;
        s_nkeep = __popcll(kbal0);
    }
    }   // wave 0
    __syncthreads();
    const unsigned long long kbal = s_kbal;
    const int nkeep = s_nkeep;

    const int hw = h * w;
    const float *heat_b = heat + (size_t)b * heat_c * hw;
    const float *z_b = z + (size_t)b * z_c * hw;
    const double dsz = (double)cfg.downsample;
    for (int t = tid; t < nkeep * J_; t += 256) {
        const int o = t / J_, j = t - o * J_;
        // o-th kept row -> source row index
        unsigned long long m = kbal;
        for (int k = 0; k < o; ++k) m &= m - 1;
        const int r = __ffsll((long long)m) - 1;
        const int id = (int)rows[r][j];
        double x2 = -1.0, y2 = -1.0, depth = -1.0, conf = 0.0;
        if (id >= 0) {
            const int k = id - s_base[j];
            const double x = (double)W.peak_x[j][k], y = (double)W.peak_y[j][k];
            conf = (double)W.peak_s[j][k];
            // retrieve_depth_heat_weighted([int(x/8), int(y/8)], z*std+mean, heat, radius=1)
            const int cx0 = (int)(x / dsz), cy0 = (int)(y / dsz);
            const int min_x = min(max(cx0 - 1, 0), w - 1), max_x = max(min(cx0 + 1, w - 1), 0);
            const int min_y = min(max(cy0 - 1, 0), h - 1), max_y = max(min(cy0 + 1, h - 1), 0);
            const float *hm = heat_b + (size_t)j * hw;
            const float *zm = z_b + (size_t)j * hw;
            float pw_[9], ww_[9];
            int n = 0;
            // (round 5: the <= 3 x 3 window's eighteen loads are issued before the first is used -- the nested loops with run-time bounds waited for
            // every cell in turn; same cells, same row-major order, same arithmetic)
            float hraw[9], zraw[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int o = min(min_y + dy, max_y) * w + min(min_x + dx, max_x);
                    hraw[dy * 3 + dx] = hm[o];
                    zraw[dy * 3 + dx] = zm[o];
                }
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (min_y + dy > max_y || min_x + dx > max_x) continue;
                    float hv = hraw[dy * 3 + dx];
                    if (hv < 0.f) hv = 0.f;
                    const float wv = hv + 0.000000001f;
                    float dv = zraw[dy * 3 + dx] * cfg.depth_std;
                    dv = dv + cfg.depth_mean;
                    pw_[n] = dv * wv;
                    ww_[n] = wv;
                    ++n;
                }
            float sp, sw;
            if (n < 8) {            // numpy pairwise_sum, n < 8: sequential from -0.0
                sp = -0.0f; sw = -0.0f;
                for (int k2 = 0; k2 < n; ++k2) { sp = sp + pw_[k2]; sw = sw + ww_[k2]; }
            } else {
                sp = ((pw_[0] + pw_[1]) + (pw_[2] + pw_[3])) + ((pw_[4] + pw_[5]) + (pw_[6] + pw_[7]));
                sw = ((ww_[0] + ww_[1]) + (ww_[2] + ww_[3])) + ((ww_[4] + ww_[5]) + (ww_[6] + ww_[7]));
                for (int k2 = 8; k2 < n; ++k2) { sp = sp + pw_[k2]; sw = sw + ww_[k2]; }
            }
            depth = (double)(sp / sw);
            x2 = x / (double)cfg.input_size * (double)cfg.w_org;
            y2 = y / (double)cfg.input_size * (double)cfg.h_org;
        }
        const double X3 = (x2 - cfg.cx) * depth / cfg.fx, Y3 = (y2 - cfg.cy) * depth / cfg.fy;
        F.joints_2d[o][j][0] = x2;
        F.joints_2d[o][j][1] = y2;
        F.joints_3d[o][j][0] = X3;
        F.joints_3d[o][j][1] = Y3;
        F.joints_3d[o][j][2] = depth;
        F.part_conf[o][j] = conf;
        if (wire && o < PN_WIRE_MAX_PERSONS) {            // the compact record, in the same pass (what pn_pack_pose_frames derives from F)
            pn_pose_wire &Wr = wire[b];
            Wr.person_joint[o][j] = (int16_t)id;
            Wr.vals[o][j][0] = (float)x2; Wr.vals[o][j][1] = (float)y2;
            Wr.vals[o][j][2] = (float)X3; Wr.vals[o][j][3] = (float)Y3; Wr.vals[o][j][4] = (float)depth;
            Wr.vals[o][j][5] = (float)conf;
        }
    }
    if (wire) {
        pn_pose_wire &Wr = wire[b];
        if (tid == 0) {
            Wr.n_persons = nkeep;
            Wr.status = status | (nkeep > PN_WIRE_MAX_PERSONS ? PN_FRAME_OVERFLOW_PERSONS : 0u);
        }
        for (int t = tid; t < PN_WIRE_MAX_PERSONS * J_; t += 256) {      // rows beyond the last person: the same filler pack_pose_kernel writes
            const int o = t / J_, j = t - o * J_;
            if (o < nkeep) continue;
            Wr.person_joint[o][j] = (int16_t)-1;
#pragma unroll
            for (int k = 0; k < 6; ++k) Wr.vals[o][j][k] = 0.f;
        }
    }
}

extern "C" int pn_parse_paf(pn_ctx *ctx, const float *heat_dev, const float *paf_dev, const float *z_dev, int B, int h,
                            int w, const pn_parse_cfg *cfg, pn_pose_frame *frames_dev, void *hip_stream) {
    return pn_parse_paf_wire(ctx, heat_dev, paf_dev, z_dev, B, h, w, cfg, frames_dev, nullptr, hip_stream);
}

extern "C" int pn_parse_paf_wire(pn_ctx *ctx, const float *heat_dev, const float *paf_dev, const float *z_dev, int B, int h,
                                 int w, const pn_parse_cfg *cfg, pn_pose_frame *frames_dev, pn_pose_wire *wire_dev, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!heat_dev || !paf_dev || !z_dev || !cfg || !frames_dev || B < 1)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_parse_paf: bad arguments");
    if (h * w > MAX_MAP || h < 1 || w < 1)
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_parse_paf: map %dx%d exceeds %d cells", h, w, MAX_MAP);
    if (cfg->downsample != 8 || cfg->num_intermed_pts != 10)
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_parse_paf: built for downsample=8, 10 intermediate points");
    const size_t need = (size_t)B * sizeof(ParseWs);
    if (ctx->parse_ws_bytes < need) {
        // never move the scratch under a graph: not while this stream is capturing, not once its size was fixed
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hip_stream && hipStreamIsCapturing((hipStream_t)hip_stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            return pn_set_error(ctx, PN_ERR_STATE, "pn_parse_paf: batch %d needs a larger parse scratch while the stream is being captured (call pn_parse_reserve first)", B);
        if (ctx->parse_ws_fixed)
            return pn_set_error(ctx, PN_ERR_STATE, "pn_parse_paf: batch %d exceeds the batch size given to pn_parse_reserve", B);
        if (ctx->parse_ws) (void)hipFree(ctx->parse_ws);
        ctx->parse_ws = nullptr;
        ctx->parse_ws_bytes = 0;
        PN_HIP_CHECK(ctx, hipMalloc(&ctx->parse_ws, need));
        ctx->parse_ws_bytes = need;
    }
    CubicTab tab;
    for (int p = 0; p < 8; ++p) host_cubic_coeffs((float)(2 * p + 1) / 16.0f, tab.c[p]);
    hipStream_t s = (hipStream_t)hip_stream;
    ParseWs *ws = (ParseWs *)ctx->parse_ws;
    // (rows beyond n_peaks / n_persons read as zero: group_readout_kernel zeroes its own record first)
    hipLaunchKernelGGL(peaks_refine_kernel, dim3(J_, B), dim3(256), 0, s, heat_dev, h, w, J_ + 1, cfg->thresh_heatmap, tab, ws);
    hipLaunchKernelGGL(limb_match_kernel, dim3(L_, B), dim3(256), 0, s, paf_dev, h, w, 2 * L_, cfg->thresh_paf,
                       h * cfg->downsample, tab, ws);
    hipLaunchKernelGGL(group_readout_kernel, dim3(B), dim3(256), 0, s, heat_dev, z_dev, h, w, J_ + 1, L_ + 1, *cfg,
                       (const ParseWs *)ws, frames_dev, wire_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

extern "C" int pn_parse_reserve(pn_ctx *ctx, int max_batch) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (max_batch < 1) return pn_set_error(ctx, PN_ERR_INVALID, "pn_parse_reserve: max_batch must be >= 1");
    const size_t need = (size_t)max_batch * sizeof(ParseWs);
    if (ctx->parse_ws_bytes < need) {
        // an explicit reserve may grow the scratch (engines of different batch sizes share the per-device context); it is
        // the caller's contract that no hipGraph captured on THIS context is still alive then -- graph-captured engines
        // own private contexts (pipeline.StreamingEngine)
        if (ctx->parse_ws) (void)hipFree(ctx->parse_ws);
        ctx->parse_ws = nullptr;
        ctx->parse_ws_bytes = 0;
        PN_HIP_CHECK(ctx, hipMalloc(&ctx->parse_ws, need));
        ctx->parse_ws_bytes = need;
    }
    ctx->parse_ws_fixed = true;
    return PN_OK;
}

// ---------------------------------------------------------------------------------------------
// Unbounded path: the same algorithm WITHOUT the record capacities (the reference has none: paf_to_pose.py:33-153,267-351).
// One frame per call, every list in a global-memory workspace sized from the frame's own counts (up to h*w peaks per joint
// map, min(ns, nd) connections per limb, one person per connection), results fetched as variable-length arrays.  It is the
// second pass for a frame the fixed-size records flag as overflowing (PN_FRAME_OVERFLOW_*): slower (lists in L2 instead of
// LDS, the greedy matching as repeated arg-max instead of a rank sort) but the same arithmetic in the same order, so a frame
// that fits the records gives the same values through either path (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------
struct BigWs {
    int *peak_count;                 // [J]
    float *px, *py, *ps;             // [J][hw]
    int *conn_count;                 // [L]
    int *conn_i, *conn_j;            // [L][hw]
    double *conn_s;                  // [L][hw]
    double *cand_s;                  // candidates of all limbs, limb l at cand_off[l]
    unsigned short *cand_i, *cand_j;
    long long cand_off[L_ + 1];
    double *rows, *rows2;            // [maxp][J + 2]
    int maxp;
    int *keep_idx;                   // [maxp]
    int *counts;                     // [0] peaks, [1] persons
    // results
    float *o_peak;                   // [npeaks][3]: x, y, score
    int *o_peak_type;                // [npeaks]
    int *o_person_joint;             // [P][J]
    double *o_person_score;          // [P]
    int *o_person_count;             // [P]
    double *o_j2d, *o_j3d, *o_conf;  // [P][J][2], [P][J][3], [P][J]
};

__global__ __launch_bounds__(256) void big_peaks_kernel(const float *__restrict__ heat, int h, int w, float thresh, CubicTab tab, BigWs W) {
    __shared__ float map[MAX_MAP];
    __shared__ int s_wave_cnt[4];
    __shared__ unsigned short s_wlist[4][MAX_MAP / 4 + 1];
    __shared__ float s_h[4][5 * 40];
    const int joint = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = h * w;
    const float *src = heat + (size_t)joint * hw;
    for (int i = tid; i < hw; i += 256) map[i] = src[i];
    __syncthreads();
    {
        const int Q = (hw + 3) / 4;
        const int lo = wave * Q, hi = min(hw, lo + Q);
        int wcnt = 0;
        for (int base = lo; base < hi; base += 64) {
            const int i = base + lane;
            bool pk = false;
            if (i < hi) {
                const int y = i / w, x = i - y * w;
                const float v = map[i];
                float m = v;
                if (y > 0) m = fmaxf(m, map[i - w]);
                if (y < h - 1) m = fmaxf(m, map[i + w]);
                if (x > 0) m = fmaxf(m, map[i - 1]);
                if (x < w - 1) m = fmaxf(m, map[i + 1]);
                pk = (m == v) && (v > thresh);
            }
            const unsigned long long bal = __ballot(pk);
            const int pos = wcnt + __popcll(bal & ((1ull << lane) - 1ull));
            if (pk) s_wlist[wave][pos] = (unsigned short)i;
            wcnt += __popcll(bal);
        }
        if (lane == 0) s_wave_cnt[wave] = wcnt;
    }
    __syncthreads();
    const int c0 = s_wave_cnt[0], c1 = s_wave_cnt[1], c2 = s_wave_cnt[2];
    const int total = c0 + c1 + c2 + s_wave_cnt[3];
    if (tid == 0) W.peak_count[joint] = total;
    float *opx = W.px + (size_t)joint * hw, *opy = W.py + (size_t)joint * hw, *ops = W.ps + (size_t)joint * hw;
    for (int p = wave; p < total; p += 4) {
        int cell;
        if (p < c0) cell = s_wlist[0][p];
        else if (p < c0 + c1) cell = s_wlist[1][p - c0];
        else if (p < c0 + c1 + c2) cell = s_wlist[2][p - c0 - c1];
        else cell = s_wlist[3][p - c0 - c1 - c2];
        const int px = cell % w, py = cell / w;
        const int x_min = max(0, px - 2), y_min = max(0, py - 2);
        const int x_max = min(w - 1, px + 2), y_max = min(h - 1, py + 2);
        const int pw = x_max - x_min + 1, ph = y_max - y_min + 1;
        const int uw = pw * 8, un = uw * ph * 8;
        const float inv_uw = 1.0f / (float)uw;
        const float *patch = map + y_min * w + x_min;
        float *hb = s_h[wave];
        for (int i = lane; i < ph * uw; i += 64) {
            const int r = (int)(((float)i + 0.5f) * inv_uw), ux = i - r * uw;
            int sx0, phx;
            up8_src(ux, sx0, phx);
            const float *row = patch + r * w;
            float hv = row[min(max(sx0, 0), pw - 1)] * tab.c[phx][0];
            hv = hv + row[min(max(sx0 + 1, 0), pw - 1)] * tab.c[phx][1];
            hv = hv + row[min(max(sx0 + 2, 0), pw - 1)] * tab.c[phx][2];
            hv = hv + row[min(max(sx0 + 3, 0), pw - 1)] * tab.c[phx][3];
            hb[r * 40 + ux] = hv;
        }
        WAVE_LDS_SYNC();
        float best = -INFINITY;
        int best_i = 0x7fffffff;
        for (int i = lane; i < un; i += 64) {
            const int uy = (int)(((float)i + 0.5f) * inv_uw), ux = i - uy * uw;
            int sy0, phy;
            up8_src(uy, sy0, phy);
            float v = hb[min(max(sy0, 0), ph - 1) * 40 + ux] * tab.c[phy][0];
            v = v + hb[min(max(sy0 + 1, 0), ph - 1) * 40 + ux] * tab.c[phy][1];
            v = v + hb[min(max(sy0 + 2, 0), ph - 1) * 40 + ux] * tab.c[phy][2];
            v = v + hb[min(max(sy0 + 3, 0), ph - 1) * 40 + ux] * tab.c[phy][3];
            if (v > best || best_i == 0x7fffffff) { best = v; best_i = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            float ov = __shfl_xor(best, off);
            int oi = __shfl_xor(best_i, off);
            if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
        }
        if (lane == 0) {
            int my = (int)(((float)best_i + 0.5f) * inv_uw), mx = best_i - my * uw;
            opx[p] = (float)(8 * x_min + mx);
            opy[p] = (float)(8 * y_min + my);
            ops[p] = best;
        }
        WAVE_LDS_SYNC();
    }
}

__global__ __launch_bounds__(256) void big_limb_kernel(const float *__restrict__ paf, int h, int w, float thresh_paf, int up_h, CubicTab tab, BigWs W) {
    __shared__ float pmap[2][MAX_MAP];
    __shared__ double s_pts[PAIRS_PER_PASS][10];
    __shared__ unsigned s_used_i[MAX_MAP / 32], s_used_j[MAX_MAP / 32];
    __shared__ double s_best_s[256];
    __shared__ int s_best_k[256];
    __shared__ int s_ncand;
    const int limb = blockIdx.x, tid = threadIdx.x;
    const int jsrc = c_limb_src[limb], jdst = c_limb_dst[limb];
    const int ns = W.peak_count[jsrc], nd = W.peak_count[jdst];
    if (ns == 0 || nd == 0) {
        if (tid == 0) W.conn_count[limb] = 0;
        return;
    }
    const int hw = h * w;
    const float *sx_ = W.px + (size_t)jsrc * hw, *sy_ = W.py + (size_t)jsrc * hw;
    const float *dx_ = W.px + (size_t)jdst * hw, *dy_ = W.py + (size_t)jdst * hw;
    double *cs = W.cand_s + W.cand_off[limb];
    unsigned short *ci = W.cand_i + W.cand_off[limb], *cj = W.cand_j + W.cand_off[limb];
    const float *px_map = paf + (size_t)(2 * limb) * hw;
    for (int i = tid; i < hw; i += 256) {
        pmap[0][i] = px_map[i];
        pmap[1][i] = px_map[hw + i];
    }
    for (int i = tid; i < MAX_MAP / 32; i += 256) { s_used_i[i] = 0u; s_used_j[i] = 0u; }
    if (tid == 0) s_ncand = 0;
    __syncthreads();
    const long long npairs = (long long)ns * nd;
    for (long long pbase = 0; pbase < npairs; pbase += PAIRS_PER_PASS) {
        const int lp = tid / 10, pt = tid - lp * 10;
        const long long pair = pbase + lp;
        const bool active = lp < PAIRS_PER_PASS && pair < npairs;
        if (active) {
            const int i = (int)(pair / nd), j = (int)(pair - (long long)i * nd);
            const double sx = (double)sx_[i], sy = (double)sy_[i];
            const double ex = (double)dx_[j], ey = (double)dy_[j];
            const double ddx = ex - sx, ddy = ey - sy;
            const double dist = sqrt(ddx * ddx + ddy * ddy) + 1e-8;
            const double dxn = ddx / dist, dyn = ddy / dist;
            const double stepx = (ex - sx) / 9.0, stepy = (ey - sy) / 9.0;
            const int qx = linspace_round(sx, ex, stepx, pt, 10);
            const int qy = linspace_round(sy, ey, stepy, pt, 10);
            const float vx = bicubic8(pmap[0], w, h, w, qy, qx, tab);
            const float vy = bicubic8(pmap[1], w, h, w, qy, qx, tab);
            s_pts[lp][pt] = (double)vx * dxn + (double)vy * dyn;
        }
        __syncthreads();
        if (tid < 64) {
            bool ok = false;
            double score = 0;
            const long long pr = pbase + tid;
            if (tid < PAIRS_PER_PASS && pr < npairs) {
                const double *s = s_pts[tid];
                double res = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
                res = res + s[8];
                res = res + s[9];
                const double mean = res / 10.0;
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < 10; ++k) cnt += (s[k] > (double)thresh_paf) ? 1 : 0;
                const int i = (int)(pr / nd), j = (int)(pr - (long long)i * nd);
                const double ax = (double)sx_[i], ay = (double)sy_[i];
                const double bx = (double)dx_[j], by = (double)dy_[j];
                const double ddx = bx - ax, ddy = by - ay;
                const double dd = sqrt(ddx * ddx + ddy * ddy) + 1e-8;
                const double pen = fmin(0.5 * (double)up_h / dd - 1.0, 0.0);
                score = mean + pen;
                ok = (cnt > 8) && (score > 0.0);
            }
            const unsigned long long bal = __ballot(ok);
            const int basec = s_ncand;
            if (ok) {
                const int pos = basec + __popcll(bal & ((1ull << tid) - 1ull));
                cs[pos] = score;
                ci[pos] = (unsigned short)(pr / nd);
                cj[pos] = (unsigned short)(pr % nd);
            }
            if (tid == 0) s_ncand = basec + __popcll(bal);
        }
        __syncthreads();
    }
    __threadfence_block();
    // greedy matching = walk the candidates in stable descending score order and take every pair whose two peaks are both
    // free (paf_to_pose.py:246-262): equivalently, repeatedly take the best (score desc, insertion order asc) candidate
    // among those whose peaks are both still free
    const int nc = s_ncand, maxc = min(ns, nd);
    int n = 0;
    while (n < maxc) {
        double bs = -INFINITY;
        int bk = 0x7fffffff;
        for (int k = tid; k < nc; k += 256) {
            const int i = ci[k], j = cj[k];
            if (((s_used_i[i >> 5] >> (i & 31)) & 1u) || ((s_used_j[j >> 5] >> (j & 31)) & 1u)) continue;
            const double sk = cs[k];
            if (sk > bs || (sk == bs && k < bk) || bk == 0x7fffffff) { bs = sk; bk = k; }
        }
        s_best_s[tid] = bs;
        s_best_k[tid] = bk;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if (tid < off) {
                const double os = s_best_s[tid + off];
                const int ok_ = s_best_k[tid + off];
                const double ms = s_best_s[tid];
                const int mk = s_best_k[tid];
                if (ok_ != 0x7fffffff && (mk == 0x7fffffff || os > ms || (os == ms && ok_ < mk))) { s_best_s[tid] = os; s_best_k[tid] = ok_; }
            }
            __syncthreads();
        }
        const int k = s_best_k[0];
        if (k == 0x7fffffff) break;                       // no free candidate left (block-uniform)
        if (tid == 0) {
            const int i = ci[k], j = cj[k];
            s_used_i[i >> 5] |= 1u << (i & 31);
            s_used_j[j >> 5] |= 1u << (j & 31);
            W.conn_i[(size_t)limb * hw + n] = i;
            W.conn_j[(size_t)limb * hw + n] = j;
            W.conn_s[(size_t)limb * hw + n] = s_best_s[0];
        }
        ++n;
        __syncthreads();
    }
    if (tid == 0) W.conn_count[limb] = n;
}

__global__ __launch_bounds__(256) void big_group_kernel(const float *__restrict__ heat, const float *__restrict__ z, int h, int w, pn_parse_cfg cfg, BigWs W) {
    __shared__ int s_base[J_ + 1];
    __shared__ int s_nh, s_h1, s_h2, s_overlap, s_np, s_nkeep;
    const int tid = threadIdx.x;
    const int hw = h * w;
    constexpr int RW = J_ + 2;
    if (tid == 0) {
        int acc = 0;
        for (int j = 0; j < J_; ++j) { s_base[j] = acc; acc += W.peak_count[j]; }
        s_base[J_] = acc;
        s_np = 0;
    }
    __syncthreads();
    const int npeaks = s_base[J_];
    for (int j = 0; j < J_; ++j) {
        const int n = W.peak_count[j];
        for (int k = tid; k < n; k += 256) {
            const int id = s_base[j] + k;
            W.o_peak[3 * (size_t)id + 0] = W.px[(size_t)j * hw + k];
            W.o_peak[3 * (size_t)id + 1] = W.py[(size_t)j * hw + k];
            W.o_peak[3 * (size_t)id + 2] = W.ps[(size_t)j * hw + k];
            W.o_peak_type[id] = j;
        }
    }
    double *rows = W.rows;
    // ---- group_limbs_of_same_person (paf_to_pose.py:280-335): serial over connections, row search across the block ----
    for (int limb = 0; limb < L_; ++limb) {
        const int st = c_limb_src[limb], dt = c_limb_dst[limb];
        const int nconn = W.conn_count[limb];
        for (int cidx = 0; cidx < nconn; ++cidx) {
            const int ci = W.conn_i[(size_t)limb * hw + cidx], cj = W.conn_j[(size_t)limb * hw + cidx];
            const double src_id = (double)(s_base[st] + ci), dst_id = (double)(s_base[dt] + cj);
            const double lscore = W.conn_s[(size_t)limb * hw + cidx];
            const double s_src = (double)W.ps[(size_t)st * hw + ci], s_dst = (double)W.ps[(size_t)dt * hw + cj];
            const int np = s_np;
            if (tid == 0) { s_nh = 0; s_h1 = 0x7fffffff; s_h2 = 0x7fffffff; s_overlap = 0; }
            __syncthreads();
            for (int r = tid; r < np; r += 256)
                if (rows[(size_t)r * RW + st] == src_id || rows[(size_t)r * RW + dt] == dst_id) { atomicAdd(&s_nh, 1); atomicMin(&s_h1, r); }
            __syncthreads();
            const int nh = s_nh, p1 = s_h1;
            if (nh == 2) {
                for (int r = tid; r < np; r += 256)
                    if (r != p1 && (rows[(size_t)r * RW + st] == src_id || rows[(size_t)r * RW + dt] == dst_id)) atomicMin(&s_h2, r);
                __syncthreads();
                const int p2 = s_h2;
                if (tid < J_ && rows[(size_t)p1 * RW + tid] >= 0.0 && rows[(size_t)p2 * RW + tid] >= 0.0) atomicOr(&s_overlap, 1);
                __syncthreads();
                if (!s_overlap) {
                    if (tid < J_) rows[(size_t)p1 * RW + tid] += rows[(size_t)p2 * RW + tid] + 1.0;
                    if (tid == 0) {
                        rows[(size_t)p1 * RW + J_] += rows[(size_t)p2 * RW + J_];
                        rows[(size_t)p1 * RW + J_ + 1] += rows[(size_t)p2 * RW + J_ + 1];
                        rows[(size_t)p1 * RW + J_] += lscore;
                    }
                    __syncthreads();
                    // person_to_joint_assoc.pop(p2): the later rows move down by one (through the second buffer)
                    for (long long e = (long long)p2 * RW + tid; e < (long long)(np - 1) * RW; e += 256) W.rows2[e] = rows[e + RW];
                    __threadfence_block();
                    __syncthreads();
                    for (long long e = (long long)p2 * RW + tid; e < (long long)(np - 1) * RW; e += 256) rows[e] = W.rows2[e];
                    if (tid == 0) s_np = np - 1;
                } else if (tid == 0) {
                    rows[(size_t)p1 * RW + dt] = dst_id;
                    rows[(size_t)p1 * RW + J_ + 1] += 1.0;
                    rows[(size_t)p1 * RW + J_] += s_dst + lscore;
                }
            } else if (nh == 1) {
                if (tid == 0 && rows[(size_t)p1 * RW + dt] != dst_id) {
                    rows[(size_t)p1 * RW + dt] = dst_id;
                    rows[(size_t)p1 * RW + J_ + 1] += 1.0;
                    rows[(size_t)p1 * RW + J_] += s_dst + lscore;
                }
            } else {
                if (tid < J_) rows[(size_t)np * RW + tid] = (tid == st) ? src_id : ((tid == dt) ? dst_id : -1.0);
                if (tid == 0) {
                    rows[(size_t)np * RW + J_ + 1] = 2.0;
                    rows[(size_t)np * RW + J_] = (s_src + s_dst) + lscore;
                    s_np = np + 1;
                }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
    // ---- prune (paf_to_pose.py:338-346), order kept ----
    if (tid == 0) {
        int nk = 0;
        for (int r = 0; r < s_np; ++r) {
            const double cnt = rows[(size_t)r * RW + J_ + 1], sc = rows[(size_t)r * RW + J_];
            if (!(cnt < 3.0 || sc / cnt < 0.2)) W.keep_idx[nk++] = r;
        }
        s_nkeep = nk;
        W.counts[0] = npeaks;
        W.counts[1] = nk;
    }
    __threadfence_block();
    __syncthreads();
    const int nkeep = s_nkeep;
    const double dsz = (double)cfg.downsample;
    for (int t = tid; t < nkeep * J_; t += 256) {
        const int o = t / J_, j = t - o * J_;
        const int r = W.keep_idx[o];
        const int id = (int)rows[(size_t)r * RW + j];
        W.o_person_joint[(size_t)o * J_ + j] = id;
        if (j == 0) {
            W.o_person_score[o] = rows[(size_t)r * RW + J_];
            W.o_person_count[o] = (int)rows[(size_t)r * RW + J_ + 1];
        }
        double x2 = -1.0, y2 = -1.0, depth = -1.0, conf = 0.0;
        if (id >= 0) {
            const int k = id - s_base[j];
            const double x = (double)W.px[(size_t)j * hw + k], y = (double)W.py[(size_t)j * hw + k];
            conf = (double)W.ps[(size_t)j * hw + k];
            const int cx0 = (int)(x / dsz), cy0 = (int)(y / dsz);
            const int min_x = min(max(cx0 - 1, 0), w - 1), max_x = max(min(cx0 + 1, w - 1), 0);
            const int min_y = min(max(cy0 - 1, 0), h - 1), max_y = max(min(cy0 + 1, h - 1), 0);
            const float *hm = heat + (size_t)j * hw;
            const float *zm = z + (size_t)j * hw;
            float pw_[9], ww_[9];
            int n = 0;
            for (int yy = min_y; yy <= max_y; ++yy)
                for (int xx = min_x; xx <= max_x; ++xx) {
                    float hv = hm[yy * w + xx];
                    if (hv < 0.f) hv = 0.f;
                    const float wv = hv + 0.000000001f;
                    float dv = zm[yy * w + xx] * cfg.depth_std;
                    dv = dv + cfg.depth_mean;
                    pw_[n] = dv * wv;
                    ww_[n] = wv;
                    ++n;
                }
            float sp, sw;
            if (n < 8) {
                sp = -0.0f; sw = -0.0f;
                for (int k2 = 0; k2 < n; ++k2) { sp = sp + pw_[k2]; sw = sw + ww_[k2]; }
            } else {
                sp = ((pw_[0] + pw_[1]) + (pw_[2] + pw_[3])) + ((pw_[4] + pw_[5]) + (pw_[6] + pw_[7]));
                sw = ((ww_[0] + ww_[1]) + (ww_[2] + ww_[3])) + ((ww_[4] + ww_[5]) + (ww_[6] + ww_[7]));
                for (int k2 = 8; k2 < n; ++k2) { sp = sp + pw_[k2]; sw = sw + ww_[k2]; }
            }
            depth = (double)(sp / sw);
            x2 = x / (double)cfg.input_size * (double)cfg.w_org;
            y2 = y / (double)cfg.input_size * (double)cfg.h_org;
        }
        const double X3 = (x2 - cfg.cx) * depth / cfg.fx, Y3 = (y2 - cfg.cy) * depth / cfg.fy;
        W.o_j2d[2 * (size_t)t + 0] = x2;
        W.o_j2d[2 * (size_t)t + 1] = y2;
        W.o_j3d[3 * (size_t)t + 0] = X3;
        W.o_j3d[3 * (size_t)t + 1] = Y3;
        W.o_j3d[3 * (size_t)t + 2] = depth;
        W.o_conf[t] = conf;
    }
}

namespace {
struct BigHost { BigWs w; size_t bytes_a = 0, bytes_b = 0; void *blk_a = nullptr, *blk_b = nullptr; int hw = 0, npeaks = 0, npersons = 0; };
char *bump(char *&p, size_t bytes) { char *r = p; p += (bytes + 255) & ~(size_t)255; return r; }
}  // namespace

extern "C" int pn_parse_paf_unbounded(pn_ctx *ctx, const float *heat_dev, const float *paf_dev, const float *z_dev, int h, int w,
                                      const pn_parse_cfg *cfg, int *n_peaks, int *n_persons, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!heat_dev || !paf_dev || !z_dev || !cfg || !n_peaks || !n_persons) return pn_set_error(ctx, PN_ERR_INVALID, "pn_parse_paf_unbounded: bad arguments");
    if (h * w > MAX_MAP || h < 1 || w < 1) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_parse_paf_unbounded: map %dx%d exceeds %d cells", h, w, MAX_MAP);
    if (cfg->downsample != 8 || cfg->num_intermed_pts != 10) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_parse_paf_unbounded: built for downsample=8, 10 intermediate points");
    hipStream_t s = (hipStream_t)hip_stream;
    BigHost *B = (BigHost *)ctx->parse_big;
    if (!B) { B = new BigHost(); ctx->parse_big = B; }
    const int hw = h * w;
    // block A: everything whose size depends on the map only
    {
        const size_t need = 256 * 8 + (size_t)J_ * hw * 12 + (size_t)L_ * hw * 16 + 4096;
        if (B->bytes_a < need) {
            if (B->blk_a) { PN_HIP_CHECK(ctx, hipDeviceSynchronize()); (void)hipFree(B->blk_a); B->blk_a = nullptr; B->bytes_a = 0; }
            PN_HIP_CHECK(ctx, hipMalloc(&B->blk_a, need));
            B->bytes_a = need;
        }
        char *p = (char *)B->blk_a;
        B->w.peak_count = (int *)bump(p, J_ * 4);
        B->w.conn_count = (int *)bump(p, L_ * 4);
        B->w.counts = (int *)bump(p, 2 * 4);
        B->w.px = (float *)bump(p, (size_t)J_ * hw * 4);
        B->w.py = (float *)bump(p, (size_t)J_ * hw * 4);
        B->w.ps = (float *)bump(p, (size_t)J_ * hw * 4);
        B->w.conn_i = (int *)bump(p, (size_t)L_ * hw * 4);
        B->w.conn_j = (int *)bump(p, (size_t)L_ * hw * 4);
        B->w.conn_s = (double *)bump(p, (size_t)L_ * hw * 8);
    }
    B->hw = hw;
    CubicTab tab;
    for (int p = 0; p < 8; ++p) host_cubic_coeffs((float)(2 * p + 1) / 16.0f, tab.c[p]);
    hipLaunchKernelGGL(big_peaks_kernel, dim3(J_), dim3(256), 0, s, heat_dev, h, w, cfg->thresh_heatmap, tab, B->w);
    int pc[J_];
    PN_HIP_CHECK(ctx, hipMemcpyAsync(pc, B->w.peak_count, sizeof pc, hipMemcpyDeviceToHost, s));
    PN_HIP_CHECK(ctx, hipStreamSynchronize(s));
    // block B: sized from this frame's peak counts
    static const int lsrc[L_] = {8, 9, 11, 8, 10, 12, 8, 1, 2, 4, 1, 3, 5, 1}, ldst[L_] = {9, 11, 13, 10, 12, 14, 1, 2, 4, 6, 3, 5, 7, 0};
    long long ncand = 0, maxp = 1, npk = 0;
    for (int j = 0; j < J_; ++j) npk += pc[j];
    for (int l = 0; l < L_; ++l) {
        B->w.cand_off[l] = ncand;
        ncand += (long long)pc[lsrc[l]] * pc[ldst[l]];
        maxp += std::min(pc[lsrc[l]], pc[ldst[l]]);
    }
    B->w.cand_off[L_] = ncand;
    B->w.maxp = (int)maxp;
    {
        const size_t need = (size_t)ncand * 12 + (size_t)maxp * ((J_ + 2) * 16 + 4 + J_ * 4 + 8 + 4 + J_ * (16 + 24 + 8)) + (size_t)npk * 16 + 16 * 256;
        if (B->bytes_b < need) {
            if (B->blk_b) { PN_HIP_CHECK(ctx, hipDeviceSynchronize()); (void)hipFree(B->blk_b); B->blk_b = nullptr; B->bytes_b = 0; }
            PN_HIP_CHECK(ctx, hipMalloc(&B->blk_b, need + need / 4));
            B->bytes_b = need + need / 4;
        }
        char *p = (char *)B->blk_b;
        B->w.cand_s = (double *)bump(p, (size_t)ncand * 8);
        B->w.cand_i = (unsigned short *)bump(p, (size_t)ncand * 2);
        B->w.cand_j = (unsigned short *)bump(p, (size_t)ncand * 2);
        B->w.rows = (double *)bump(p, (size_t)maxp * (J_ + 2) * 8);
        B->w.rows2 = (double *)bump(p, (size_t)maxp * (J_ + 2) * 8);
        B->w.keep_idx = (int *)bump(p, (size_t)maxp * 4);
        B->w.o_peak = (float *)bump(p, (size_t)npk * 12);
        B->w.o_peak_type = (int *)bump(p, (size_t)npk * 4);
        B->w.o_person_joint = (int *)bump(p, (size_t)maxp * J_ * 4);
        B->w.o_person_score = (double *)bump(p, (size_t)maxp * 8);
        B->w.o_person_count = (int *)bump(p, (size_t)maxp * 4);
        B->w.o_j2d = (double *)bump(p, (size_t)maxp * J_ * 16);
        B->w.o_j3d = (double *)bump(p, (size_t)maxp * J_ * 24);
        B->w.o_conf = (double *)bump(p, (size_t)maxp * J_ * 8);
    }
    hipLaunchKernelGGL(big_limb_kernel, dim3(L_), dim3(256), 0, s, paf_dev, h, w, cfg->thresh_paf, h * cfg->downsample, tab, B->w);
    hipLaunchKernelGGL(big_group_kernel, dim3(1), dim3(256), 0, s, heat_dev, z_dev, h, w, *cfg, B->w);
    int cnt[2] = {0, 0};
    PN_HIP_CHECK(ctx, hipMemcpyAsync(cnt, B->w.counts, sizeof cnt, hipMemcpyDeviceToHost, s));
    PN_HIP_CHECK(ctx, hipStreamSynchronize(s));
    PN_HIP_CHECK(ctx, hipGetLastError());
    B->npeaks = cnt[0];
    B->npersons = cnt[1];
    *n_peaks = cnt[0];
    *n_persons = cnt[1];
    return PN_OK;
}

extern "C" int pn_parse_paf_unbounded_fetch(pn_ctx *ctx, float *peaks_xys, int *peak_type, int *person_joint, double *person_score,
                                            int *person_count, double *joints_2d, double *joints_3d, double *part_conf) {
    if (!ctx) return PN_ERR_INVALID;
    BigHost *B = (BigHost *)ctx->parse_big;
    if (!B || !B->blk_b) return pn_set_error(ctx, PN_ERR_STATE, "pn_parse_paf_unbounded_fetch: no result (call pn_parse_paf_unbounded first)");
    const size_t np = (size_t)B->npeaks, P = (size_t)B->npersons;
    if (np && peaks_xys) PN_HIP_CHECK(ctx, hipMemcpy(peaks_xys, B->w.o_peak, np * 12, hipMemcpyDeviceToHost));
    if (np && peak_type) PN_HIP_CHECK(ctx, hipMemcpy(peak_type, B->w.o_peak_type, np * 4, hipMemcpyDeviceToHost));
    if (P && person_joint) PN_HIP_CHECK(ctx, hipMemcpy(person_joint, B->w.o_person_joint, P * J_ * 4, hipMemcpyDeviceToHost));
    if (P && person_score) PN_HIP_CHECK(ctx, hipMemcpy(person_score, B->w.o_person_score, P * 8, hipMemcpyDeviceToHost));
    if (P && person_count) PN_HIP_CHECK(ctx, hipMemcpy(person_count, B->w.o_person_count, P * 4, hipMemcpyDeviceToHost));
    if (P && joints_2d) PN_HIP_CHECK(ctx, hipMemcpy(joints_2d, B->w.o_j2d, P * J_ * 16, hipMemcpyDeviceToHost));
    if (P && joints_3d) PN_HIP_CHECK(ctx, hipMemcpy(joints_3d, B->w.o_j3d, P * J_ * 24, hipMemcpyDeviceToHost));
    if (P && part_conf) PN_HIP_CHECK(ctx, hipMemcpy(part_conf, B->w.o_conf, P * J_ * 8, hipMemcpyDeviceToHost));
    return PN_OK;
}

// Stand-alone NMS (tpm/lib/utils/paf_to_pose.py:75-153 with bool_refine_center=True, no Gaussian filter) on ANY number of maps of one
// frame: what `paf_to_pose_cpp` (paf_to_pose.py:381-385) runs before it hands the peaks to `process_paf` -- there with the 18 COCO parts, a
// topology the three fixed-size parse kernels (15 joints / 14 limbs) do not serve.  Same kernel as the unbounded second pass: no capacity, peaks
// of map m at [m][0 .. count[m]) in row-major cell order (= the reference's ids), refined x / y in up-sampled (x8) pixels and the bicubic score.
extern "C" int pn_nms_peaks(pn_ctx *ctx, const float *heat_dev, int n_maps, int h, int w, float thresh, int upsample, int *count_dev,
                            float *peak_x_dev, float *peak_y_dev, float *peak_score_dev, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!heat_dev || !count_dev || !peak_x_dev || !peak_y_dev || !peak_score_dev || n_maps < 1) return pn_set_error(ctx, PN_ERR_INVALID, "pn_nms_peaks: bad arguments");
    if (h * w > MAX_MAP || h < 1 || w < 1) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_nms_peaks: map %dx%d exceeds %d cells", h, w, MAX_MAP);
    if (upsample != 8) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_nms_peaks: built for upsampFactor = 8 (MODEL.DOWNSAMPLE)");
    CubicTab tab;
    for (int p = 0; p < 8; ++p) host_cubic_coeffs((float)(2 * p + 1) / 16.0f, tab.c[p]);
    BigWs W = {};
    W.peak_count = count_dev; W.px = peak_x_dev; W.py = peak_y_dev; W.ps = peak_score_dev;
    hipLaunchKernelGGL(big_peaks_kernel, dim3(n_maps), dim3(256), 0, (hipStream_t)hip_stream, heat_dev, h, w, thresh, tab, W);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

void pn_parse_big_free(pn_ctx *ctx) {
    BigHost *B = (BigHost *)ctx->parse_big;
    if (!B) return;
    if (B->blk_a) (void)hipFree(B->blk_a);
    if (B->blk_b) (void)hipFree(B->blk_b);
    delete B;
    ctx->parse_big = nullptr;
}

// ---------------------------------------------------------------------------------------------
// Stand-alone retrieve_depth_heat_weighted (tpm/lib/utils/common.py:272-293) for the per-call
// Python API: n centres on one (depthmap, heatmap) pair, one lane per centre.  Like the reference
// it clamps negative heat values IN PLACE in the caller's heat map before reading.
// ---------------------------------------------------------------------------------------------
__global__ void clamp_negative_kernel(float *__restrict__ hm, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && hm[i] < 0.f) hm[i] = 0.f;
}

__global__ void retrieve_depth_kernel(const float *__restrict__ dm, const float *__restrict__ hm, int h, int w,
                                      const int *__restrict__ centers, int n, int radius, float *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int cx0 = centers[2 * t], cy0 = centers[2 * t + 1];
    const int min_x = min(max(cx0 - radius, 0), w - 1), max_x = max(min(cx0 + radius, w - 1), 0);
    const int min_y = min(max(cy0 - radius, 0), h - 1), max_y = max(min(cy0 + radius, h - 1), 0);
    // np.sum over the [ny, nx] window: numpy pairwise summation (blocks of 8, then the tail)
    const int nx = max_x - min_x + 1, cnt = nx * (max_y - min_y + 1);
    float sp, sw;
    auto P = [&](int k) { int yy = min_y + k / nx, xx = min_x + k % nx; return dm[yy * w + xx] * (hm[yy * w + xx] + 0.000000001f); };
    auto Wt = [&](int k) { int yy = min_y + k / nx, xx = min_x + k % nx; return hm[yy * w + xx] + 0.000000001f; };
    if (cnt < 8) {
        sp = -0.0f; sw = -0.0f;
        for (int k = 0; k < cnt; ++k) { sp = sp + P(k); sw = sw + Wt(k); }
    } else if (cnt < 128) {
        float rp[8], rw[8];
        for (int k = 0; k < 8; ++k) { rp[k] = P(k); rw[k] = Wt(k); }
        int i = 8;
        for (; i < cnt - (cnt % 8); i += 8)
            for (int k = 0; k < 8; ++k) { rp[k] = rp[k] + P(i + k); rw[k] = rw[k] + Wt(i + k); }
        sp = ((rp[0] + rp[1]) + (rp[2] + rp[3])) + ((rp[4] + rp[5]) + (rp[6] + rp[7]));
        sw = ((rw[0] + rw[1]) + (rw[2] + rw[3])) + ((rw[4] + rw[5]) + (rw[6] + rw[7]));
        for (; i < cnt; ++i) { sp = sp + P(i); sw = sw + Wt(i); }
    } else {
        out[t] = __builtin_nanf("");      // windows of >= 128 cells are refused by the host wrapper
        return;
    }
    out[t] = sp / sw;
}

// one thread per (frame, person slot, joint): float64 record -> float32 wire values
__global__ __launch_bounds__(256) void pack_pose_kernel(const pn_pose_frame *__restrict__ frames, int B, pn_pose_wire *__restrict__ wire) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = PN_WIRE_MAX_PERSONS * PN_NUM_JOINTS;
    if (i >= B * per) return;
    const int f = i / per, r = i - f * per, p = r / PN_NUM_JOINTS, j = r - p * PN_NUM_JOINTS;
    const pn_pose_frame &fr = frames[f];
    pn_pose_wire &w = wire[f];
    const int n = fr.n_persons;
    if (r == 0) {
        w.n_persons = n;
        w.status = fr.status | (n > PN_WIRE_MAX_PERSONS ? PN_FRAME_OVERFLOW_PERSONS : 0u);
    }
    const bool live = p < n;
    w.person_joint[p][j] = live ? (int16_t)fr.person_joint[p][j] : (int16_t)-1;
    w.vals[p][j][0] = live ? (float)fr.joints_2d[p][j][0] : 0.f;
    w.vals[p][j][1] = live ? (float)fr.joints_2d[p][j][1] : 0.f;
    w.vals[p][j][2] = live ? (float)fr.joints_3d[p][j][0] : 0.f;
    w.vals[p][j][3] = live ? (float)fr.joints_3d[p][j][1] : 0.f;
    w.vals[p][j][4] = live ? (float)fr.joints_3d[p][j][2] : 0.f;
    w.vals[p][j][5] = live ? (float)fr.part_conf[p][j] : 0.f;
}

extern "C" int pn_pack_pose_frames(pn_ctx *ctx, const pn_pose_frame *frames_dev, int B, pn_pose_wire *wire_dev, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (!frames_dev || !wire_dev || B < 1) return pn_set_error(ctx, PN_ERR_INVALID, "pn_pack_pose_frames: bad arguments");
    const int total = B * PN_WIRE_MAX_PERSONS * PN_NUM_JOINTS;
    hipLaunchKernelGGL(pack_pose_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, frames_dev, B, wire_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

extern "C" size_t pn_sizeof_pose_wire(void) { return sizeof(pn_pose_wire); }

extern "C" int pn_retrieve_depth(pn_ctx *ctx, const float *depthmap_dev, float *heatmap_dev, int h, int w,
                                 const int *centers_xy_dev, int n, int radius, float *out_dev, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!depthmap_dev || !heatmap_dev || !centers_xy_dev || !out_dev || n < 1 || radius < 0)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_retrieve_depth: bad arguments");
    if ((2 * radius + 1) * (2 * radius + 1) >= 128)
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_retrieve_depth: radius %d too large", radius);
    hipStream_t s = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(clamp_negative_kernel, dim3((h * w + 255) / 256), dim3(256), 0, s, heatmap_dev, h * w);
    hipLaunchKernelGGL(retrieve_depth_kernel, dim3((n + 63) / 64), dim3(64), 0, s, depthmap_dev, (const float *)heatmap_dev,
                       h, w, centers_xy_dev, n, radius, out_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}
