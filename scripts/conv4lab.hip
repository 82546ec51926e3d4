// Development tool (not part of the product): times ONE launch of conv4_kernel (optionally a level of several
// convolutions, GROUP="cin:cout,...") on synthetic data and checks the first problem against a naive GPU convolution.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipopnet_amd/csrc scripts/conv4lab.hip -o popnet_amd/build/conv4lab
//   conv4lab B H W Cin Cout [iters] [res]          (-DPN_STAMP: in-kernel s_memtime timeline, -DLAB_V3: conv3_kernel<3,4,1,1> instead)
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
#endif
#include "conv4_kernel.h"

int pn_set_error(pn_ctx *, int code, const char *fmt, ...) { fprintf(stderr, "error %d: %s\n", code, fmt); return code; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void naive_conv(const __bf16 *in, const float *w, const float *bias, const __bf16 *res, float *out, int B, int H, int W,
                           int cin, int in_cs, int cout, int act) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * H * W * cout;
    if (i >= total) return;
    int co = i % cout; size_t p = i / cout;
    int x = p % W, y = (p / W) % H, b = p / ((size_t)W * H);
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            int iy = y + ky - 1, ix = x + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const __bf16 *ip = in + ((size_t)(b * H + iy) * W + ix) * in_cs;
            const float *wp = w + ((size_t)co * 9 + ky * 3 + kx) * cin;
            for (int c = 0; c < cin; ++c) acc += (float)ip[c] * wp[c];
        }
    acc += bias[co];
    if (res) acc += (float)res[p * cout + co];
    if (act == PN_ACT_RELU) acc = acc > 0 ? acc : 0;
    out[i] = acc;
}

struct Prob { ConvProblem P; std::vector<float> w; std::vector<uint16_t> in; int cin, cout, cin_pad, out_cs; __bf16 *din, *dout; float *dbias; };

static std::vector<uint16_t> pack4(const std::vector<float> &hw, int cin, int cout, int chunks) {
    const int cout_pad = (cout + 127) / 128 * 128, ksteps = chunks * 18;
    std::vector<uint16_t> pk((size_t)(cout_pad / 128) * (ksteps + 3) * 4096, 0);
    for (int cbk = 0; cbk < cout_pad / 128; ++cbk) for (int hh = 0; hh < chunks * 2; ++hh) for (int tap = 0; tap < 9; ++tap) for (int t = 0; t < 8; ++t) for (int lane = 0; lane < 64; ++lane) {
        const int co = cbk * 128 + pn_conv_row_channel(t, lane & 15, 4), q = lane >> 4;
        const size_t base = ((((size_t)cbk * (ksteps + 3) + (size_t)hh * 9 + tap) * 8 + t) * 64 + lane) * 8;
        for (int j = 0; j < 8; ++j) { const int ci = hh * 32 + 8 * q + j; pk[base + j] = f2bf((co < cout && ci < cin) ? hw[((size_t)co * 9 + tap) * cin + ci] : 0.f); }
    }
    return pk;
}
static std::vector<uint16_t> pack3(const std::vector<float> &hw, int cin, int cout, int chunks) {
    const int BC = 128, cout_pad = (cout + BC - 1) / BC * BC, ctiles = cout_pad / 16, ksteps = chunks * 18;
    std::vector<uint16_t> pk((size_t)ctiles * ksteps * 512 + 5 * 512, 0);
    for (int ct = 0; ct < ctiles; ++ct) for (int ch = 0; ch < chunks; ++ch) for (int sub = 0; sub < 2; ++sub) for (int tap = 0; tap < 9; ++tap) {
        size_t kstep = (size_t)(ch * 2 + sub) * 9 + tap, frag = (size_t)ct * ksteps + kstep;
        for (int lane = 0; lane < 64; ++lane) { int co = pn_conv_row_channel(ct, lane & 15, 2), q = lane >> 4;
            for (int j = 0; j < 8; ++j) { int ci = ch * 64 + sub * 32 + 8 * q + j; pk[(frag * 64 + lane) * 8 + j] = f2bf((co < cout && ci < cin) ? hw[((size_t)co * 9 + tap) * cin + ci] : 0.f); } } }
    return pk;
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: conv4lab B H W Cin Cout [iters] [res]\n"); return 2; }
    const int B = atoi(argv[1]), H = atoi(argv[2]), W = atoi(argv[3]);
    const int iters = argc > 6 ? atoi(argv[6]) : 50, use_res = argc > 7 ? atoi(argv[7]) : 0;
    std::vector<std::pair<int, int>> shapes{{atoi(argv[4]), atoi(argv[5])}};
    if (const char *g = getenv("GROUP")) { std::string gs(g); size_t pos = 0; while (pos < gs.size()) { size_t e = gs.find(',', pos); if (e == std::string::npos) e = gs.size(); int ci = 0, co = 0; sscanf(gs.substr(pos, e - pos).c_str(), "%d:%d", &ci, &co); shapes.push_back({ci, co}); pos = e + 1; } }
    srand(1);
    const size_t npx = (size_t)B * H * W;
    const int segs = (W + 29) / 30; int Wt = (W + segs - 1) / segs; if (W % 28 == 0) Wt = 28;
    const int R = std::min(H, 112 / Wt), act = PN_ACT_RELU;
    std::vector<Prob> probs(shapes.size());
    std::vector<ConvProblem> hp;
    double flops = 0; int maxb = 0;
    std::vector<uint16_t> hres; __bf16 *dres = nullptr;
    for (size_t i = 0; i < shapes.size(); ++i) {
        Prob &Q = probs[i]; Q.cin = shapes[i].first; Q.cout = shapes[i].second; Q.cin_pad = (Q.cin + 63) / 64 * 64; Q.out_cs = (Q.cout + 63) / 64 * 64;
        const int chunks = Q.cin_pad / 64;
        Q.in.assign(npx * Q.cin_pad, 0);
        for (size_t p = 0; p < npx; ++p) for (int c = 0; c < Q.cin; ++c) Q.in[p * Q.cin_pad + c] = f2bf((rand() % 2001 - 1000) / 1000.0f);
        Q.w.resize((size_t)Q.cout * 9 * Q.cin); for (auto &v : Q.w) v = bf2f(f2bf((rand() % 2001 - 1000) / 1000.0f * 0.05f));
        std::vector<float> hb((Q.cout + 127) / 128 * 128, 0.f); for (int k = 0; k < Q.cout; ++k) hb[k] = (rand() % 2001 - 1000) / 1000.0f;
        CK(hipMalloc(&Q.din, Q.in.size() * 2 + 2048)); CK(hipMemset(Q.din, 0, Q.in.size() * 2 + 2048)); CK(hipMemcpy(Q.din, Q.in.data(), Q.in.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&Q.dbias, hb.size() * 4)); CK(hipMemcpy(Q.dbias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&Q.dout, npx * Q.out_cs * 2)); CK(hipMemset(Q.dout, 0, npx * Q.out_cs * 2));
#ifdef LAB_V3
        std::vector<uint16_t> pk = pack3(Q.w, Q.cin, Q.cout, chunks);
#else
        std::vector<uint16_t> pk = pack4(Q.w, Q.cin, Q.cout, chunks);
#endif
        void *dpk; CK(hipMalloc(&dpk, pk.size() * 2)); CK(hipMemcpy(dpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        if (i == 0 && use_res) { hres.resize(npx * Q.cout); for (auto &v : hres) v = f2bf((rand() % 2001 - 1000) / 1000.0f); CK(hipMalloc(&dres, hres.size() * 2)); CK(hipMemcpy(dres, hres.data(), hres.size() * 2, hipMemcpyHostToDevice)); }
        ConvProblem &P = Q.P; memset(&P, 0, sizeof P);
        P.in = Q.din; P.in_zero_off = (unsigned)(Q.in.size() * 2); P.wpack = dpk; P.bias = Q.dbias; P.res = (i == 0 && use_res) ? dres : nullptr; P.out = Q.dout;
        P.B = B; P.H = H; P.W = W; P.Ho = H; P.Wo = W; P.cin_chunks = chunks; P.in_cs = Q.cin_pad; P.cout = Q.cout; P.out_cs = Q.out_cs; P.res_cs = Q.cout; P.act = act;
        P.R = R; P.Wt = Wt; P.tiles_x = (W + Wt - 1) / Wt; P.tiles_per_img = ((H + R - 1) / R) * P.tiles_x; P.cout_blocks = (Q.cout + 127) / 128; P.ksteps = chunks * 18;
#ifdef LAB_V3
        P.nblocks = B * P.tiles_per_img * P.cout_blocks;
#else
        P.nblocks = ((B * P.tiles_per_img + 1) / 2) * P.cout_blocks;
#endif
        maxb = std::max(maxb, P.nblocks);
        hp.push_back(P);
        flops += 2.0 * npx * Q.cout * (double)Q.cin * 9;
    }
    ConvProblem *dP; CK(hipMalloc(&dP, sizeof(ConvProblem) * hp.size())); CK(hipMemcpy(dP, hp.data(), sizeof(ConvProblem) * hp.size(), hipMemcpyHostToDevice));
    float *dref; CK(hipMalloc(&dref, npx * probs[0].cout * 4));
    float *dw; CK(hipMalloc(&dw, probs[0].w.size() * 4)); CK(hipMemcpy(dw, probs[0].w.data(), probs[0].w.size() * 4, hipMemcpyHostToDevice));
    { size_t total = npx * probs[0].cout; hipLaunchKernelGGL(naive_conv, dim3((total + 255) / 256), dim3(256), 0, 0, probs[0].din, dw, probs[0].dbias, use_res ? dres : nullptr, dref, B, H, W, probs[0].cin, probs[0].cin_pad, probs[0].cout, act); CK(hipDeviceSynchronize()); }
    const int nblocks = maxb * (int)hp.size();
#ifdef PN_STAMP
    unsigned long long *dst; CK(hipMalloc(&dst, (size_t)nblocks * 16 * 8)); CK(hipMemset(dst, 0, (size_t)nblocks * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
#endif
    auto launch = [&]() {
#ifdef LAB_V3
        auto kern = conv3_kernel<3, 4, 1, 1, 7, 4>;
        hipLaunchKernelGGL(kern, dim3(maxb, (unsigned)hp.size()), dim3(256), (size_t)8 * 6 * 32 * 16 + 1024, 0, dP);
#else
        hipLaunchKernelGGL(conv4_kernel, dim3(maxb, (unsigned)hp.size()), dim3(256), PN4_LDS, 0, dP);
#endif
    };
#ifndef LAB_V3
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PN4_LDS));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
#ifdef LAB_V3
    printf("conv3 ");
#else
    printf("conv4 ");
#endif
    printf("B%d %dx%d", B, H, W); for (auto &s : shapes) printf(" %d->%d", s.first, s.second);
    printf(": %.2f us/launch  %.1f TFLOP/s  (%d blocks)\n", us, flops / us / 1e6, nblocks);
    const Prob &Q = probs[0];
    std::vector<uint16_t> ho(npx * Q.out_cs); std::vector<float> href(npx * Q.cout);
    CK(hipMemcpy(ho.data(), Q.dout, ho.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0; size_t bad = 0;
    for (size_t p = 0; p < npx; ++p) for (int c = 0; c < Q.cout; ++c) {
        double r = href[p * Q.cout + c], g = bf2f(ho[p * Q.out_cs + c]);
        double e = fabs(r - g); maxerr = std::max(maxerr, e); maxref = std::max(maxref, fabs(r));
        if (e > 0.02 * fabs(r) + 0.02) { if (bad < 5) printf("  mismatch p=%zu c=%d ref=%g got=%g\n", p, c, r, g); ++bad; }
    }
    printf("check: max|err| %.4g (max|ref| %.4g), %zu bad of %zu -> %s\n", maxerr, maxref, bad, npx * Q.cout, bad ? "FAIL" : "ok");
#ifdef PN_STAMP
    {
        std::vector<unsigned long long> st((size_t)nblocks * 16);
        CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
        for (int y = 0; y < (int)hp.size(); ++y) {
            double avg[16] = {0}; int cnt = 0;
            for (int b = y * maxb; b < (y + 1) * maxb; ++b) { if (!st[(size_t)b * 16]) continue; ++cnt; for (int i = 0; i < 14; ++i) avg[i] += st[(size_t)b * 16 + i] ? (double)(st[(size_t)b * 16 + i] - st[(size_t)b * 16]) : 0; }
            printf("stamps problem %d (shader cycles from block start, avg over %d blocks):", y, cnt);
            for (int i = 0; i < 13; ++i) printf(" %.0f", cnt ? avg[i] / cnt : 0.0);
            printf("\n");
        }
        unsigned long long r0 = ~0ull; for (int b = 0; b < nblocks; ++b) if (st[(size_t)b * 16 + 14]) r0 = std::min(r0, st[(size_t)b * 16 + 14]);
        std::vector<double> starts, ends, clk;
        for (int b = 0; b < nblocks; ++b) if (st[(size_t)b * 16 + 14] && st[(size_t)b * 16 + 15]) {
            starts.push_back((st[(size_t)b * 16 + 14] - r0) * 0.01); ends.push_back((st[(size_t)b * 16 + 15] - r0) * 0.01);
            if (st[(size_t)b * 16 + 15] > st[(size_t)b * 16 + 14] && st[(size_t)b * 16 + 12] > st[(size_t)b * 16]) clk.push_back((double)(st[(size_t)b * 16 + 12] - st[(size_t)b * 16]) / (double)(st[(size_t)b * 16 + 15] - st[(size_t)b * 16 + 14]) * 0.1);
        }
        std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end()); std::sort(clk.begin(), clk.end());
        auto pct = [](std::vector<double> &v, double p) { return v.empty() ? 0.0 : v[std::min(v.size() - 1, (size_t)(p * v.size()))]; };
        printf(" block starts us: p50 %.1f p90 %.1f max %.1f | ends: p10 %.1f p50 %.1f p90 %.1f max %.1f (%zu blocks) | clock GHz p50 %.3f\n", pct(starts, .5), pct(starts, .9), starts.empty() ? 0 : starts.back(),
               pct(ends, .1), pct(ends, .5), pct(ends, .9), ends.empty() ? 0 : ends.back(), starts.size(), pct(clk, .5));
    }
#endif
    return bad ? 1 : 0;
}
