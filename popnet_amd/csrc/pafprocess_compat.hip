// Drop-in for the reference's native plug-in `pafprocess` (SWIG module, COCO-18 topology):
//   process_paf + 6 getters           tpm/lib/pafprocess/pafprocess.cpp:22-246
//   constants / topology / structs    tpm/lib/pafprocess/pafprocess.h:6-59
// Same seven C symbols, same borrowed host float32 arrays, same "results live in globals until
// the next call" contract -- but limb scoring, sorting, greedy matching and person assembly run in
// one HIP workgroup on the GPU instead of the host loops.
//
// Arithmetic follows the C++ reference type for type: int peak coordinates, float32 unit vector and
// scores, "int(v + 0.5)" sampling (the +0.5 happens in double), the double-typed length penalty
// "min(0.0, 0.5*h1/norm - 1.0)" added to a float mean, float32 person rows, and its quirks:
// membership test "> 0" (peak id 0 counts as empty), new persons only for pair_id < 18, nothing
// happens when >= 3 persons match.  std::sort is not stable; equal scores here keep candidate order.
#pragma clang fp contract(off)
#include <cmath>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "pn_internal.h"

#define CP_NUM_PART 18
#define CP_NUM_PAIR 19
#define CP_STEP 10
#define CP_MAXPK 64          // peaks per part handled on the device
#define CP_MAXH 128          // person rows

__constant__ int c_pairs[CP_NUM_PAIR][2] = {{1, 2}, {1, 5}, {2, 3}, {3, 4}, {5, 6}, {6, 7}, {1, 8}, {8, 9}, {9, 10}, {1, 11},
                                            {11, 12}, {12, 13}, {1, 0}, {0, 14}, {14, 16}, {0, 15}, {15, 17}, {2, 16}, {5, 17}};
__constant__ int c_pairs_net[CP_NUM_PAIR][2] = {{12, 13}, {20, 21}, {14, 15}, {16, 17}, {22, 23}, {24, 25}, {0, 1}, {2, 3},
                                                {4, 5}, {6, 7}, {8, 9}, {10, 11}, {28, 29}, {30, 31}, {34, 35}, {32, 33},
                                                {36, 37}, {18, 19}, {26, 27}};

struct CpPeak { int x, y; float score; int id; };

struct CpResult {
    int n_humans;
    int overflow;
    float rows[CP_MAXH][20];
};

__global__ __launch_bounds__(256) void process_paf_kernel(const CpPeak *__restrict__ peaks, const int *__restrict__ part_ofs,
                                                           const float *__restrict__ line_score,
                                                           const float *__restrict__ paf, int h1, int f2, int f3,
                                                           CpResult *__restrict__ res) {
    __shared__ float s_score[CP_MAXPK * CP_MAXPK];
    __shared__ unsigned char s_a[CP_MAXPK * CP_MAXPK], s_b[CP_MAXPK * CP_MAXPK];
    __shared__ unsigned short s_order[CP_MAXPK * CP_MAXPK];
    __shared__ int s_ncand, s_wave_cnt[4];
    __shared__ int s_nconn[CP_NUM_PAIR];
    __shared__ int s_cid1[CP_NUM_PAIR][CP_MAXPK], s_cid2[CP_NUM_PAIR][CP_MAXPK];
    __shared__ float s_cscore[CP_NUM_PAIR][CP_MAXPK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int pair = 0; pair < CP_NUM_PAIR; ++pair) {
        const int pa = c_pairs[pair][0], pb = c_pairs[pair][1];
        const int na = part_ofs[pa + 1] - part_ofs[pa], nb = part_ofs[pb + 1] - part_ofs[pb];
        const CpPeak *A = peaks + part_ofs[pa], *Bp = peaks + part_ofs[pb];
        if (tid == 0) { s_ncand = 0; s_nconn[pair] = 0; }
        __syncthreads();
        if (na == 0 || nb == 0) continue;
        const int ch1 = c_pairs_net[pair][0], ch2 = c_pairs_net[pair][1];
        const int npairs = na * nb;
        for (int base = 0; base < npairs; base += 256) {
            const int k = base + tid;
            bool ok = false;
            float crit2 = 0.f;
            if (k < npairs) {
                const int ia = k / nb, ib = k - ia * nb;
                const CpPeak a = A[ia], bb = Bp[ib];
                float vx = (float)(bb.x - a.x), vy = (float)(bb.y - a.y);
                const float norm = (float)sqrt((double)(vx * vx + vy * vy));
                if (!(norm < 1e-12)) {
                    vx = vx / norm; vy = vy / norm;
                    const float stepx = (bb.x - a.x) / (float)CP_STEP, stepy = (bb.y - a.y) / (float)CP_STEP;
                    float scores = 0.0f;
                    int c1 = 0;
                    for (int i = 0; i < CP_STEP; ++i) {
                        const int lx = (int)((double)((float)a.x + (float)i * stepx) + 0.5);
                        const int ly = (int)((double)((float)a.y + (float)i * stepy) + 0.5);
                        const float px = paf[((size_t)ly * f2 + lx) * f3 + ch1];
                        const float py = paf[((size_t)ly * f2 + lx) * f3 + ch2];
                        const float sc = vx * px + vy * py;
                        scores += sc;
                        if (sc > 0.05f) c1 += 1;
                    }
                    const double pen = fmin(0.0, 0.5 * (double)h1 / (double)norm - 1.0);
                    crit2 = (float)((double)(scores / (float)CP_STEP) + pen);
                    ok = (c1 > 6) && (crit2 > 0.f);
                }
            }
            const unsigned long long bal = __ballot(ok);
            if (lane == 0) s_wave_cnt[wave] = __popcll(bal);
            __syncthreads();
            int before = s_ncand;
            for (int q = 0; q < wave; ++q) before += s_wave_cnt[q];
            const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
            if (ok) {
                s_score[pos] = crit2;
                s_a[pos] = (unsigned char)(k / nb);
                s_b[pos] = (unsigned char)(k % nb);
            }
            __syncthreads();
            if (tid == 0) s_ncand += s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
            __syncthreads();
        }
        const int nc = s_ncand;
        for (int k = tid; k < nc; k += 256) {
            const float sk = s_score[k];
            int rank = 0;
            for (int m = 0; m < nc; ++m) {
                const float sm = s_score[m];
                rank += (sm > sk || (sm == sk && m < k)) ? 1 : 0;
            }
            s_order[rank] = (unsigned short)k;
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long ua = 0, ub = 0;
            int n = 0;
            for (int r = 0; r < nc; ++r) {
                const int k = s_order[r];
                const int ia = s_a[k], ib = s_b[k];
                if (((ua >> ia) & 1ull) || ((ub >> ib) & 1ull)) continue;
                ua |= 1ull << ia; ub |= 1ull << ib;
                s_cid1[pair][n] = A[ia].id;
                s_cid2[pair][n] = Bp[ib].id;
                s_cscore[pair][n] = s_score[k];
                ++n;
            }
            s_nconn[pair] = n;
        }
        __syncthreads();
    }

    // person assembly: literal transcription of pafprocess.cpp:128-186 (float32 rows), one lane
    if (tid == 0) {
        int nh = 0, overflow = 0;
        for (int pair = 0; pair < CP_NUM_PAIR; ++pair) {
            const int p1 = c_pairs[pair][0], p2 = c_pairs[pair][1];
            for (int c = 0; c < s_nconn[pair]; ++c) {
                const int cid1 = s_cid1[pair][c], cid2 = s_cid2[pair][c];
                const float cs = s_cscore[pair][c];
                int found = 0, i1 = 0, i2 = 0;
                for (int s = 0; s < nh; ++s)
                    if (res->rows[s][p1] == (float)cid1 || res->rows[s][p2] == (float)cid2) {
                        if (found == 0) i1 = s;
                        if (found == 1) i2 = s;
                        found += 1;
                    }
                if (found == 1) {
                    if (res->rows[i1][p2] != (float)cid2) {
                        res->rows[i1][p2] = (float)cid2;
                        res->rows[i1][19] += 1;
                        res->rows[i1][18] += line_score[cid2] + cs;
                    }
                } else if (found == 2) {
                    int membership = 0;
                    for (int s = 0; s < 18; ++s)
                        if (res->rows[i1][s] > 0 && res->rows[i2][s] > 0) membership = 2;
                    if (membership == 0) {
                        for (int s = 0; s < 18; ++s) res->rows[i1][s] += (res->rows[i2][s] + 1);
                        res->rows[i1][19] += res->rows[i2][19];
                        res->rows[i1][18] += res->rows[i2][18];
                        res->rows[i1][18] += cs;
                        for (int s = i2; s < nh - 1; ++s)
                            for (int q = 0; q < 20; ++q) res->rows[s][q] = res->rows[s + 1][q];
                        --nh;
                    } else {
                        res->rows[i1][p2] = (float)cid2;
                        res->rows[i1][19] += 1;
                        res->rows[i1][18] += line_score[cid2] + cs;
                    }
                } else if (found == 0 && pair < 18) {
                    if (nh < CP_MAXH) {
                        for (int q = 0; q < 20; ++q) res->rows[nh][q] = -1;
                        res->rows[nh][p1] = (float)cid1;
                        res->rows[nh][p2] = (float)cid2;
                        res->rows[nh][19] = 2;
                        res->rows[nh][18] = line_score[cid1] + line_score[cid2] + cs;
                        ++nh;
                    } else {
                        overflow = 1;
                    }
                }
            }
        }
        for (int i = nh - 1; i >= 0; --i)
            if (res->rows[i][19] < 4 || res->rows[i][18] / res->rows[i][19] < 0.3f) {
                for (int s = i; s < nh - 1; ++s)
                    for (int q = 0; q < 20; ++q) res->rows[s][q] = res->rows[s + 1][q];
                --nh;
            }
        res->n_humans = nh;
        res->overflow = overflow;
    }
}

namespace {
std::mutex g_mu;
pn_ctx *g_ctx = nullptr;
std::vector<std::vector<float>> g_subset;   // rows of 20 floats, as pafprocess.cpp:12
std::vector<CpPeak> g_line;                  // peak_infos_line, pafprocess.cpp:13

template <typename T> struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc((void **)&p, n * sizeof(T)); }
};
}  // namespace

extern "C" {

int process_paf(int p1, int p2, int p3, float *peaks, int h1, int h2, int h3, float *heatmap, int f1, int f2, int f3,
                float *pafmap) {
    (void)h2; (void)h3; (void)heatmap;     // the reference never reads the heat map either
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_ctx) g_ctx = pn_create(0);
    pn_ctx *ctx = g_ctx;
    if (ctx->device < 0) return PN_ERR_STATE;
    if (!peaks || !pafmap || p3 < 5 || f3 < 38 || p1 < 0 || p2 < 0) return pn_set_error(ctx, PN_ERR_INVALID, "process_paf: bad arguments");
    g_subset.clear();
    g_line.clear();
    // Peak records in input order (pafprocess.cpp:25-37), bucketed by part
    std::vector<CpPeak> per_part[CP_NUM_PART];
    int cnt = 0;
    for (int i = 0; i < p1; ++i)
        for (int k = 0; k < p2; ++k) {
            const float *r = peaks + ((size_t)i * p2 + k) * p3;
            CpPeak pk;
            pk.id = cnt++;
            pk.x = (int)r[0]; pk.y = (int)r[1]; pk.score = r[2];
            const int part = (int)r[4];
            if (part < 0 || part >= CP_NUM_PART) return pn_set_error(ctx, PN_ERR_INVALID, "process_paf: part id %d out of range", part);
            if (pk.x < 0 || pk.y < 0 || pk.x >= f2 || pk.y >= f1)
                return pn_set_error(ctx, PN_ERR_INVALID, "process_paf: peak (%d,%d) outside the %dx%d PAF map", pk.x, pk.y, f2, f1);
            per_part[part].push_back(pk);
        }
    std::vector<int> ofs(CP_NUM_PART + 1, 0);
    for (int p = 0; p < CP_NUM_PART; ++p) {
        if (per_part[p].size() > CP_MAXPK) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "process_paf: more than %d peaks of part %d", CP_MAXPK, p);
        ofs[p + 1] = ofs[p] + (int)per_part[p].size();
        for (auto &pk : per_part[p]) g_line.push_back(pk);
    }
    if (g_line.empty()) return 0;
    std::vector<float> line_score(g_line.size());
    for (size_t i = 0; i < g_line.size(); ++i) line_score[i] = g_line[i].score;

    if (hipSetDevice(ctx->device) != hipSuccess) return pn_set_error(ctx, PN_ERR_HIP, "hipSetDevice failed");
    DevBuf<CpPeak> d_peaks; DevBuf<int> d_ofs; DevBuf<float> d_ls; DevBuf<float> d_paf; DevBuf<CpResult> d_res;
    const size_t paf_elems = (size_t)f1 * f2 * f3;
    PN_HIP_CHECK(ctx, d_peaks.alloc(g_line.size()));
    PN_HIP_CHECK(ctx, d_ofs.alloc(ofs.size()));
    PN_HIP_CHECK(ctx, d_ls.alloc(line_score.size()));
    PN_HIP_CHECK(ctx, d_paf.alloc(paf_elems));
    PN_HIP_CHECK(ctx, d_res.alloc(1));
    PN_HIP_CHECK(ctx, hipMemcpy(d_peaks.p, g_line.data(), g_line.size() * sizeof(CpPeak), hipMemcpyHostToDevice));
    PN_HIP_CHECK(ctx, hipMemcpy(d_ofs.p, ofs.data(), ofs.size() * sizeof(int), hipMemcpyHostToDevice));
    PN_HIP_CHECK(ctx, hipMemcpy(d_ls.p, line_score.data(), line_score.size() * 4, hipMemcpyHostToDevice));
    PN_HIP_CHECK(ctx, hipMemcpy(d_paf.p, pafmap, paf_elems * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(process_paf_kernel, dim3(1), dim3(256), 0, 0, d_peaks.p, d_ofs.p, d_ls.p, d_paf.p, h1, f2, f3, d_res.p);
    PN_HIP_CHECK(ctx, hipGetLastError());
    std::vector<unsigned char> hres(sizeof(CpResult));
    PN_HIP_CHECK(ctx, hipMemcpy(hres.data(), d_res.p, sizeof(CpResult), hipMemcpyDeviceToHost));
    const CpResult *R = reinterpret_cast<const CpResult *>(hres.data());
    if (R->overflow) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "process_paf: more than %d person rows", CP_MAXH);
    for (int i = 0; i < R->n_humans; ++i) g_subset.emplace_back(R->rows[i], R->rows[i] + 20);
    return 0;
}

int get_num_humans(void) { return (int)g_subset.size(); }
int get_part_cid(int human_id, int part_id) { return (int)g_subset[human_id][part_id]; }
float get_score(int human_id) { return g_subset[human_id][18] / g_subset[human_id][19]; }
int get_part_x(int cid) { return g_line[cid].x; }
int get_part_y(int cid) { return g_line[cid].y; }
float get_part_score(int cid) { return g_line[cid].score; }

}  // extern "C"
