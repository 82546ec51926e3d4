// Development tool (not part of the product): times ONE convolution launch of the library's MFMA
// kernels on synthetic data, checks it against a naive GPU convolution, and (built with -DPN_STAMP)
// dumps an in-kernel s_memtime timeline per block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipopnet_amd/csrc scripts/convlab.hip -o popnet_amd/build/convlab
//   convlab B H W Cin Cout ks cfg [iters] [kernel]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#ifdef PN_STAMP
__device__ unsigned long long *g_stamps;
#define PN_STAMP_AT(i) do { if (threadIdx.x == 0) { size_t b_ = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16; g_stamps[b_ + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0) g_stamps[b_ + 14] = __builtin_amdgcn_s_memrealtime(); if ((i) == 12) g_stamps[b_ + 15] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define PN_STAMP_AT(i) do {} while (0)
#endif
#include "conv_mfma_kernel.h"
#ifdef HAVE_CONV3
#include "conv3_kernel.h"
#endif

static int pn_cfg_couts_l(int cfg) { return cfg == PN_CFG_C128 ? 128 : ((cfg == PN_CFG_C64 || cfg == PN_CFG_C64W) ? 64 : (cfg == PN_CFG_C32 ? 32 : 16)); }
static int pn_conv_stage_maxpx_l(int ks, int pitch, int cfg) {
    if (cfg == PN_CFG_C64W) return 0;
    const int maxpx = ks == 1 ? 128 : (pitch <= 32 ? 192 : (pitch <= 64 ? 288 : 360));
    return (maxpx * 8 + 255) / 256 <= 12 ? maxpx : 0;
}
static int lab_launch_old(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    PN_CASE(PN_PREC_BF16, 3, 1, 32, PN_CFG_C128) PN_CASE(PN_PREC_BF16, 3, 1, 32, PN_CFG_C64)
    PN_CASE(PN_PREC_BF16, 3, 1, 64, PN_CFG_C128) PN_CASE(PN_PREC_BF16, 3, 1, 120, PN_CFG_C64W)
    PN_CASE(PN_PREC_BF16, 1, 1, 32, PN_CFG_C128)
    fprintf(stderr, "no lab instantiation for ks=%d pitch=%d cfg=%d\n", L.ks, L.pitch, L.cfg);
    return 1;
}

int pn_launch_conv3_part0(pn_ctx *, const ConvLaunch &, hipStream_t) { return 1; }
int pn_set_error(pn_ctx *, int code, const char *fmt, ...) { fprintf(stderr, "error %d: %s\n", code, fmt); return code; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void naive_conv(const __bf16 *in, const float *w, const float *bias, const __bf16 *res, float *out, int B, int H, int W,
                           int cin, int in_cs, int cout, int ks, int act) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * H * W * cout;
    if (i >= total) return;
    int co = i % cout; size_t p = i / cout;
    int x = p % W, y = (p / W) % H, b = p / ((size_t)W * H);
    int pad = ks / 2;
    float acc = 0.f;
    for (int ky = 0; ky < ks; ++ky)
        for (int kx = 0; kx < ks; ++kx) {
            int iy = y + ky - pad, ix = x + kx - pad;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const __bf16 *ip = in + ((size_t)(b * H + iy) * W + ix) * in_cs;
            const float *wp = w + ((size_t)co * ks * ks + ky * ks + kx) * cin;
            for (int c = 0; c < cin; ++c) acc += (float)ip[c] * wp[c];
        }
    acc += bias[co];
    if (res) acc += (float)res[p * cout + co];
    if (act == PN_ACT_RELU) acc = acc > 0 ? acc : 0;
    if (act == PN_ACT_LEAKY) acc = acc > 0 ? acc : 0.1f * acc;
    out[i] = acc;
}

static int pick_pitch(int cols) { const int cl[4] = {16, 32, 64, 120}; for (int c : cl) if (cols <= c) return c; return -1; }

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: convlab B H W Cin Cout ks cfg [iters] [kernel: old|v3] [res]\n"); return 2; }
    int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]), cin = atoi(argv[4]), cout = atoi(argv[5]), ks = atoi(argv[6]), cfg = atoi(argv[7]);
    int iters = argc > 8 ? atoi(argv[8]) : 50;
    const char *kname = argc > 9 ? argv[9] : "old";
    int use_res = argc > 10 ? atoi(argv[10]) : 0;
    const int KK = ks * ks;
    const int cin_pad = (cin + 63) / 64 * 64, chunks = cin_pad / 64;
    srand(1);
    // input NHWC bf16 (channel stride = cin_pad, pad channels hold garbage-free zeros)
    size_t npx = (size_t)B * H * W;
    std::vector<uint16_t> hin(npx * cin_pad, 0);
    for (size_t p = 0; p < npx; ++p) for (int c = 0; c < cin; ++c) hin[p * cin_pad + c] = f2bf((rand() % 2001 - 1000) / 1000.0f);
    std::vector<float> hw((size_t)cout * KK * cin);        // [co][tap][ci], bf16-representable
    for (auto &v : hw) v = bf2f(f2bf((rand() % 2001 - 1000) / 1000.0f * 0.05f));
    std::vector<float> hb(((cout + 127) / 128) * 128, 0.f);
    for (int i = 0; i < cout; ++i) hb[i] = (rand() % 2001 - 1000) / 1000.0f;
    std::vector<uint16_t> hres(npx * cout);
    for (auto &v : hres) v = f2bf((rand() % 2001 - 1000) / 1000.0f);

    __bf16 *din, *dout, *dres; float *dw, *dbias, *dref;
    CK(hipMalloc(&din, hin.size() * 2)); CK(hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dw, hw.size() * 4)); CK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dbias, hb.size() * 4)); CK(hipMemcpy(dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dres, hres.size() * 2)); CK(hipMemcpy(dres, hres.data(), hres.size() * 2, hipMemcpyHostToDevice));
    const int out_cs = (cout + 63) / 64 * 64;
    CK(hipMalloc(&dout, npx * out_cs * 2)); CK(hipMemset(dout, 0, npx * out_cs * 2));
    CK(hipMalloc(&dref, npx * cout * 4));
    const int act = PN_ACT_RELU;
    {
        size_t total = npx * cout;
        hipLaunchKernelGGL(naive_conv, dim3((total + 255) / 256), dim3(256), 0, 0, din, dw, dbias, use_res ? dres : nullptr, dref, B, H, W, cin, cin_pad, cout, ks, act);
        CK(hipDeviceSynchronize());
    }
    unsigned long long *dst = nullptr;
    const size_t max_blocks_stamp = 1 << 16;
    CK(hipMalloc(&dst, max_blocks_stamp * 16 * 8)); CK(hipMemset(dst, 0, max_blocks_stamp * 16 * 8));
#ifdef PN_STAMP
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double flops = 2.0 * npx * cout * (double)cin * KK;
    int nblocks = 0, lab_gridx = 0, lab_gridy = 1;
    auto run = [&]() {};
    (void)run;
    if (!strcmp(kname, "old")) {
        const int BC = pn_cfg_couts_l(cfg);
        const int cout_pad = (cout + BC - 1) / BC * BC, ctiles = cout_pad / 16, ksteps = chunks * KK * 2;
        std::vector<uint16_t> pk((size_t)ctiles * ksteps * 512 + 5 * 512, 0);
        for (int ct = 0; ct < ctiles; ++ct) for (int ch = 0; ch < chunks; ++ch) for (int sub = 0; sub < 2; ++sub) for (int tap = 0; tap < KK; ++tap) {
            size_t kstep = (size_t)(ch * 2 + sub) * KK + tap, frag = (size_t)ct * ksteps + kstep;
            for (int lane = 0; lane < 64; ++lane) { int co = pn_conv_row_channel(ct, lane & 15, pn_cfg_ct(cfg)), q = lane >> 4;
                for (int j = 0; j < 8; ++j) { int ci = ch * 64 + sub * 32 + 8 * q + j;
                    float v = (co < cout && ci < cin) ? hw[((size_t)co * KK + tap) * cin + ci] : 0.f;
                    pk[(frag * 64 + lane) * 8 + j] = f2bf(v); } } }
        void *dpk; CK(hipMalloc(&dpk, pk.size() * 2)); CK(hipMemcpy(dpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        ConvProblem P; memset(&P, 0, sizeof P);
        const int BP = cfg == PN_CFG_C128 ? 112 : (cfg == PN_CFG_C64W ? 224 : 128);
        int wt_cap = std::min(BP, 120 - ks + 1), segs = (W + wt_cap - 1) / wt_cap, Wt = (W + segs - 1) / segs;
        int R = std::max(1, std::min(H, BP / Wt));
        int pitch = pick_pitch(Wt - 1 + ks);
        int maxpx = pn_conv_stage_maxpx_l(ks, pitch, cfg);
        while (maxpx > 0 && R > 1 && (R - 1 + ks) * (Wt - 1 + ks) > maxpx) --R;
        P.in = din; P.wpack = dpk; P.bias = dbias; P.res = use_res ? dres : nullptr; P.out = dout; P.B = B; P.H = H; P.W = W; P.Ho = H; P.Wo = W;
        P.cin_chunks = chunks; P.in_cs = cin_pad; P.cout = cout; P.out_cs = out_cs; P.res_cs = cout; P.act = act; P.R = R; P.Wt = Wt;
        P.tiles_x = (W + Wt - 1) / Wt; P.tiles_per_img = ((H + R - 1) / R) * P.tiles_x; P.cout_blocks = (cout + BC - 1) / BC;
        P.nblocks = B * P.tiles_per_img * P.cout_blocks; P.ksteps = ksteps;
        P.lds_buf_bytes = (R - 1 + ks) * pitch * 128; P.lds_two = (chunks > 1 && 2 * P.lds_buf_bytes <= 160 * 1024) ? 1 : 0;
        ConvProblem *dP; CK(hipMalloc(&dP, sizeof P)); CK(hipMemcpy(dP, &P, sizeof P, hipMemcpyHostToDevice));
        ConvLaunch L; L.prec = PN_PREC_BF16; L.ks = ks; L.stride = 1; L.pitch = pitch; L.cfg = cfg; L.nprob = 1; L.max_blocks = P.nblocks;
        L.lds_bytes = (size_t)P.lds_buf_bytes * (P.lds_two ? 2 : 1); L.probs_dev = dP;
        nblocks = P.nblocks;
        printf("old kernel: cfg %d R %d Wt %d pitch %d blocks %d lds %zu two %d\n", cfg, R, Wt, pitch, nblocks, L.lds_bytes, P.lds_two);
        pn_ctx ctx;
        for (int i = 0; i < 3; ++i) if (lab_launch_old(&ctx, L, 0)) return 1;
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) lab_launch_old(&ctx, L, 0);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    }
#ifdef HAVE_CONV3
    else {
        // cfg: 0 = 4x1 waves (128 couts x 112 px), 1 = 2x2 (64 x 224), 2 = 2x1 (64 x 112, 128 threads), 3 = 1x1 (32 x 112, 64 threads), 4 = 4x2 (128 x 224, 512 threads)
        const int WCs[5] = {4, 2, 2, 1, 4}, WPs[5] = {1, 2, 1, 1, 2};
        const int WC = WCs[cfg], WP = WPs[cfg], BC = WC * 32;
        const int cout_pad = (cout + BC - 1) / BC * BC, ctiles = cout_pad / 16, ksteps = chunks * KK * 2;
        std::vector<uint16_t> pk((size_t)ctiles * ksteps * 512 + 5 * 512, 0);
        for (int ct = 0; ct < ctiles; ++ct) for (int ch = 0; ch < chunks; ++ch) for (int sub = 0; sub < 2; ++sub) for (int tap = 0; tap < KK; ++tap) {
            size_t kstep = (size_t)(ch * 2 + sub) * KK + tap, frag = (size_t)ct * ksteps + kstep;
            for (int lane = 0; lane < 64; ++lane) { int co = pn_conv_row_channel(ct, lane & 15, 2), q = lane >> 4;
                for (int j = 0; j < 8; ++j) { int ci = ch * 64 + sub * 32 + 8 * q + j;
                    float v = (co < cout && ci < cin) ? hw[((size_t)co * KK + tap) * cin + ci] : 0.f;
                    pk[(frag * 64 + lane) * 8 + j] = f2bf(v); } } }
        void *dpk; CK(hipMalloc(&dpk, pk.size() * 2)); CK(hipMemcpy(dpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        // input copy with a zero page behind it
        __bf16 *din2; CK(hipMalloc(&din2, hin.size() * 2 + 256)); CK(hipMemset(din2, 0, hin.size() * 2 + 256));
        CK(hipMemcpy(din2, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
        ConvProblem P; memset(&P, 0, sizeof P);
        int segs = (W + 29) / 30, Wt = (W + segs - 1) / segs;
        if (W % 28 == 0) Wt = 28;
        const int lab_pt = getenv("PT") ? atoi(getenv("PT")) : 7;              // PT=14: 224-pixel wave tiles (8 rows x 28)
        int R = std::min(H, (112 * WP * (lab_pt / 7)) / Wt);
        P.in = din2; P.in_zero_off = (unsigned)(hin.size() * 2);
        P.wpack = dpk; P.bias = dbias; P.res = use_res ? dres : nullptr; P.out = dout; P.B = B; P.H = H; P.W = W; P.Ho = H; P.Wo = W;
        P.cin_chunks = chunks; P.in_cs = cin_pad; P.cout = cout; P.out_cs = out_cs; P.res_cs = cout; P.act = act; P.R = R; P.Wt = Wt;
        P.tiles_x = (W + Wt - 1) / Wt; P.tiles_per_img = ((H + R - 1) / R) * P.tiles_x; P.cout_blocks = (cout + BC - 1) / BC;
        P.nblocks = B * P.tiles_per_img * P.cout_blocks; P.ksteps = ksteps;
        // GROUP="cin:cout,cin:cout": further problems (same map, kernel size and block shape) in the same launch, unchecked
        std::vector<ConvProblem> probs(1, P);
        if (const char *g = getenv("GROUP")) {
            std::string gs(g);
            size_t pos = 0;
            while (pos < gs.size()) {
                size_t e = gs.find(',', pos); if (e == std::string::npos) e = gs.size();
                int ci = 0, co = 0; sscanf(gs.substr(pos, e - pos).c_str(), "%d:%d", &ci, &co);
                pos = e + 1;
                const int cp = (ci + 63) / 64 * 64, ch2 = cp / 64, ks2 = ch2 * KK * 2, cop = (co + BC - 1) / BC * BC;
                std::vector<uint16_t> in2(npx * cp); for (auto &v : in2) v = f2bf((rand() % 2001 - 1000) / 1000.0f);
                std::vector<uint16_t> w2((size_t)(cop / 16) * ks2 * 512 + 5 * 512); for (auto &v : w2) v = f2bf((rand() % 2001 - 1000) / 20000.0f);
                __bf16 *di; void *dw2; __bf16 *do2; float *db2;
                CK(hipMalloc(&di, in2.size() * 2 + 256)); CK(hipMemset(di, 0, in2.size() * 2 + 256)); CK(hipMemcpy(di, in2.data(), in2.size() * 2, hipMemcpyHostToDevice));
                CK(hipMalloc(&dw2, w2.size() * 2)); CK(hipMemcpy(dw2, w2.data(), w2.size() * 2, hipMemcpyHostToDevice));
                CK(hipMalloc(&do2, npx * cop * 2)); CK(hipMalloc(&db2, cop * 4)); CK(hipMemset(db2, 0, cop * 4));
                ConvProblem Q = P;
                Q.in = di; Q.in_zero_off = (unsigned)(in2.size() * 2); Q.wpack = dw2; Q.bias = db2; Q.res = nullptr; Q.out = do2;
                Q.cin_chunks = ch2; Q.in_cs = cp; Q.cout = co; Q.out_cs = cop; Q.cout_blocks = cop / BC; Q.nblocks = B * Q.tiles_per_img * Q.cout_blocks; Q.ksteps = ks2;
                probs.push_back(Q);
                flops += 2.0 * npx * co * (double)ci * KK;
            }
        }
        int maxb = 0; for (auto &q : probs) maxb = std::max(maxb, q.nblocks);
        ConvProblem *dP; CK(hipMalloc(&dP, sizeof(P) * probs.size())); CK(hipMemcpy(dP, probs.data(), sizeof(P) * probs.size(), hipMemcpyHostToDevice));
        nblocks = maxb * (int)probs.size();
        const int gridx = maxb, gridy = (int)probs.size();
        lab_gridx = gridx; lab_gridy = gridy;
        const int nbuf = getenv("NBUF") ? atoi(getenv("NBUF")) : 1;
        size_t lds = (size_t)8 * (4 * (lab_pt / 7) * WP + ks - 1) * 32 * 16 * (nbuf >= 3 ? std::min(nbuf, chunks) : nbuf) + 1024;   // nbuf >= 3: all Cin chunks resident
        printf("v3 kernel: cfg %d (WC %d WP %d) NBUF %d R %d Wt %d blocks %d lds %zu\n", cfg, WC, WP, nbuf, R, Wt, nblocks, lds);
        auto launch = [&]() {
            dim3 grid(gridx, gridy);
#define LAB3(KS_, WC_, WP_, NB_) if (ks == KS_ && WC == WC_ && WP == WP_ && nbuf == NB_) { auto kern = conv3_kernel<KS_, WC_, WP_, NB_, 7, 4>; \
                if (lds > 48 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                hipLaunchKernelGGL(kern, grid, dim3(WC_ * WP_ * 64), lds, 0, dP); return; }
            if (lab_pt == 14 && ks == 3 && WP == 1 && nbuf == 1 && (WC == 4 || WC == 2)) {
                if (WC == 4) { auto kern = conv3_kernel<3, 4, 1, 1, 14, 8>; CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, dP); }
                else { auto kern = conv3_kernel<3, 2, 1, 1, 14, 8>; hipLaunchKernelGGL(kern, grid, dim3(128), lds, 0, dP); }
                return; }
            LAB3(1, 4, 1, 1) LAB3(1, 4, 1, 4)
            LAB3(3, 4, 1, 1) LAB3(3, 2, 2, 1) LAB3(3, 2, 1, 1) LAB3(3, 4, 1, 2) LAB3(3, 2, 2, 2) LAB3(3, 2, 1, 2) LAB3(3, 4, 2, 1)
            fprintf(stderr, "no v3 instantiation\n"); exit(1);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    }
#endif
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("%s B%d %dx%d %d->%d k%d cfg%d: %.2f us/launch  %.1f TFLOP/s  (%d blocks)\n", kname, B, H, W, cin, cout, ks, cfg, us, flops / us / 1e6, nblocks);
    // check
    std::vector<uint16_t> ho(npx * out_cs); std::vector<float> href(npx * cout);
    CK(hipMemcpy(ho.data(), dout, ho.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0; size_t bad = 0;
    for (size_t p = 0; p < npx; ++p) for (int c = 0; c < cout; ++c) {
        double r = href[p * cout + c], g = bf2f(ho[p * out_cs + c]);
        double e = fabs(r - g); maxerr = std::max(maxerr, e); maxref = std::max(maxref, fabs(r));
        if (e > 0.02 * fabs(r) + 0.02) { if (bad < 5) printf("  mismatch p=%zu c=%d ref=%g got=%g\n", p, c, r, g); ++bad; }
    }
    printf("check: max|err| %.4g (max|ref| %.4g), %zu bad of %zu -> %s\n", maxerr, maxref, bad, npx * cout, bad ? "FAIL" : "ok");
#ifdef PN_STAMP
    {
        std::vector<unsigned long long> st((size_t)nblocks * 16);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int b = 0; b < nblocks; ++b) if (st[b * 16]) t0 = std::min(t0, st[b * 16]);
        if (!lab_gridx) lab_gridx = nblocks;
        for (int y = 0; y < lab_gridy; ++y) {
            double avg[16] = {0}; int cnt = 0;
            for (int b = y * lab_gridx; b < (y + 1) * lab_gridx && b < nblocks; ++b) { if (!st[b * 16]) continue; ++cnt; for (int i = 0; i < 14; ++i) avg[i] += st[b * 16 + i] ? (double)(st[b * 16 + i] - st[b * 16]) : 0; }
            printf("stamps problem %d (shader cycles from block start, avg over %d blocks):", y, cnt);
            for (int i = 0; i < 13; ++i) printf(" %.0f", cnt ? avg[i] / cnt : 0.0);
            printf("\n");
        }
        printf("\n first-block-start spread: ");
        unsigned long long tmax = 0; for (int b = 0; b < nblocks; ++b) tmax = std::max(tmax, st[b * 16] - t0);
        printf("%llu\n", tmax);
        {   // global timeline from s_memrealtime (100 MHz): when do blocks start and end
            unsigned long long r0 = ~0ull; for (int b = 0; b < nblocks; ++b) if (st[b * 16 + 14]) r0 = std::min(r0, st[b * 16 + 14]);
            std::vector<double> starts, ends;
            for (int b = 0; b < nblocks; ++b) if (st[b * 16 + 14] && st[b * 16 + 15]) { starts.push_back((st[b * 16 + 14] - r0) * 0.01); ends.push_back((st[b * 16 + 15] - r0) * 0.01); }
            std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
            auto pct = [](std::vector<double> &v, double p) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(p * v.size()))]; };
            printf(" block starts us: p50 %.1f p90 %.1f max %.1f | ends: p10 %.1f p50 %.1f p90 %.1f max %.1f  (%zu blocks)\n", pct(starts, .5), pct(starts, .9), starts.empty() ? 0 : starts.back(),
                   pct(ends, .1), pct(ends, .5), pct(ends, .9), ends.empty() ? 0 : ends.back(), starts.size());
            // in-kernel shader clock (MI355X_MICROARCH.md, DVFS give-back item 6): d(s_memtime) / d(s_memrealtime) x 100 MHz, median over blocks
            std::vector<double> clk;
            for (int b = 0; b < nblocks; ++b) if (st[b * 16 + 15] > st[b * 16 + 14] && st[b * 16 + 12] > st[b * 16])
                clk.push_back((double)(st[b * 16 + 12] - st[b * 16]) / (double)(st[b * 16 + 15] - st[b * 16 + 14]) * 0.1);
            std::sort(clk.begin(), clk.end());
            printf(" in-kernel clock GHz: p10 %.3f p50 %.3f p90 %.3f (after %d back-to-back launches)\n", pct(clk, .1), pct(clk, .5), pct(clk, .9), iters);
        }
        for (int b = 0; b < std::min(nblocks, 4); ++b) { printf(" block %d:", b); for (int i = 0; i < 16; ++i) printf(" %lld", st[b*16+i] ? (long long)(st[b * 16 + i] - t0) : -1); printf("\n"); }
    }
#endif
    return bad ? 1 : 0;
}
