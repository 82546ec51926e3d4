// Development tool (not part of the product): the three conv4 levels of a stage (28x28, B frames) as
//   (a) three launches back to back on one stream (what the network does),
//   (b) the same three launches on three streams with NO dependencies (wrong as a network, timing only): what tail filling
//       and shared-mode packing across levels could buy if the kernel boundaries were replaced by finer dependencies.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipopnet_amd/csrc scripts/stagelab.hip -o popnet_amd/build/stagelab
//   stagelab [B] [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "conv4_kernel.h"

int pn_set_error(pn_ctx *, int code, const char *fmt, ...) { fprintf(stderr, "error %d: %s\n", code, fmt); return code; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Level { std::vector<ConvProblem> probs; ConvProblem *dev = nullptr; int maxb = 0; double flops = 0; };

static Level make_level(int B, int H, int W, std::vector<std::pair<int, int>> shapes) {
    Level L;
    const size_t npx = (size_t)B * H * W;
    for (auto &sh : shapes) {
        const int cin = sh.first, cout = sh.second, cin_pad = (cin + 63) / 64 * 64, out_cs = (cout + 63) / 64 * 64, chunks = cin_pad / 64;
        const int cout_pad = (cout + 127) / 128 * 128, ksteps = chunks * 18;
        void *din, *dout, *dpk; float *dbias;
        CK(hipMalloc(&din, npx * cin_pad * 2 + 2048)); CK(hipMemset(din, 0x11, npx * cin_pad * 2)); CK(hipMemset((char *)din + npx * cin_pad * 2, 0, 2048));
        CK(hipMalloc(&dout, npx * out_cs * 2));
        const size_t pkb = (size_t)(cout_pad / 128) * (ksteps + 3) * 8192;
        CK(hipMalloc(&dpk, pkb)); CK(hipMemset(dpk, 0x3c, pkb));
        CK(hipMalloc(&dbias, cout_pad * 4)); CK(hipMemset(dbias, 0, cout_pad * 4));
        ConvProblem P; memset(&P, 0, sizeof P);
        P.in = din; P.in_zero_off = (unsigned)(npx * cin_pad * 2); P.wpack = dpk; P.bias = dbias; P.out = dout;
        P.B = B; P.H = H; P.W = W; P.Ho = H; P.Wo = W; P.cin_chunks = chunks; P.in_cs = cin_pad; P.cout = cout; P.out_cs = out_cs; P.act = PN_ACT_LEAKY;
        P.R = 4; P.Wt = 28; P.tiles_x = 1; P.tiles_per_img = 7; P.cout_blocks = cout_pad / 128; P.ksteps = ksteps;
        P.nblocks = ((B * P.tiles_per_img + 1) / 2) * P.cout_blocks;
        L.maxb = std::max(L.maxb, P.nblocks);
        L.probs.push_back(P);
        L.flops += 2.0 * npx * cout * (double)cin * 9;
    }
    CK(hipMalloc(&L.dev, sizeof(ConvProblem) * L.probs.size()));
    CK(hipMemcpy(L.dev, L.probs.data(), sizeof(ConvProblem) * L.probs.size(), hipMemcpyHostToDevice));
    return L;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, iters = argc > 2 ? atoi(argv[2]) : 300;
    std::vector<Level> lv;
    lv.push_back(make_level(B, 28, 28, {{128, 256}, {128, 128}, {128, 128}}));
    lv.push_back(make_level(B, 28, 28, {{256, 256}, {128, 128}, {128, 64}}));
    lv.push_back(make_level(B, 28, 28, {{256, 256}, {128, 128}, {64, 64}}));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(conv4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PN4_LDS));
    hipStream_t st[3];
    for (auto &s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto launch = [&](int l, hipStream_t s) { hipLaunchKernelGGL(conv4_kernel, dim3(lv[l].maxb, (unsigned)lv[l].probs.size()), dim3(256), PN4_LDS, s, lv[l].dev); };
    double flops = 0;
    for (auto &l : lv) flops += l.flops;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < 3; ++i) for (int l = 0; l < 3; ++l) launch(l, st[mode ? l : 0]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        if (mode) { CK(hipStreamWaitEvent(st[1], e0, 0)); CK(hipStreamWaitEvent(st[2], e0, 0)); }
        for (int i = 0; i < iters; ++i) for (int l = 0; l < 3; ++l) launch(l, st[mode ? l : 0]);
        hipEvent_t j1, j2; CK(hipEventCreate(&j1)); CK(hipEventCreate(&j2));
        if (mode) { CK(hipEventRecord(j1, st[1])); CK(hipEventRecord(j2, st[2])); CK(hipStreamWaitEvent(st[0], j1, 0)); CK(hipStreamWaitEvent(st[0], j2, 0)); }
        CK(hipEventRecord(e1, st[0]));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("stage (3 conv4 levels, B%d): %s: %.2f us per stage  %.1f TFLOP/s\n", B, mode ? "three streams, no dependencies (timing only)" : "one stream, back to back", us, flops / us / 1e6);
    }
    return 0;
}
