// Development tool: times bb64_kernel (fused BasicBlock(64)) on synthetic data; -DPN_STAMP dumps an in-kernel timeline
// of the third tile of every workgroup (stamps: 1 tile start, 2 conv1 done, 3 intermediate written, 4 conv2 done, 5 tile end).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipopnet_amd/csrc scripts/bblab.hip -o popnet_amd/build/bblab
//   bblab B H W [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#ifdef PN_STAMP
__device__ unsigned long long *g_stamps;
#define PN_STAMP_AT(i) do { if (threadIdx.x == 0) { size_t b_ = (size_t)blockIdx.x * 16; g_stamps[b_ + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0) g_stamps[b_ + 14] = __builtin_amdgcn_s_memrealtime(); if ((i) == 12) g_stamps[b_ + 15] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#endif
#include "bb64_kernel.h"
#ifndef BBLAB_HV
#define BBLAB_HV 2          // compute-wave halves (bb64_kernel<HV>): -DBBLAB_HV=1 is the four-compute-wave form
#endif
int pn_set_error(pn_ctx *, int code, const char *fmt, ...) { fprintf(stderr, "error %d: %s\n", code, fmt); return code; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 112, W = argc > 3 ? atoi(argv[3]) : 112, iters = argc > 4 ? atoi(argv[4]) : 200;
    srand(1);
    const size_t npx = (size_t)B * H * W;
    std::vector<uint16_t> hin(npx * 64), hw(36 * 4 * 64 * 8);
    for (auto &v : hin) v = f2bf((rand() % 2001 - 1000) / 1000.0f);
    for (auto &v : hw) v = f2bf((rand() % 2001 - 1000) / 20000.0f);
    std::vector<float> hb(128, 0.1f);
    __bf16 *din, *dout; void *dw; float *db;
    CK(hipMalloc(&din, hin.size() * 2 + 2048)); CK(hipMemset(din, 0, hin.size() * 2 + 2048)); CK(hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, npx * 64 * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&db, 512)); CK(hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice));
    BBProblem P; memset(&P, 0, sizeof P);
    P.in = din; P.out = dout; P.wpack = dw; P.bias1 = db; P.bias2 = db + 64; P.B = B; P.H = H; P.W = W; P.in_cs = 64; P.out_cs = 64;
    const int segs = (W + 27) / 28; P.Wt = (W + segs - 1) / segs; P.tiles_x = (W + P.Wt - 1) / P.Wt; P.tiles_per_img = ((H + 7) / 8) * P.tiles_x; P.ntiles = B * P.tiles_per_img;
    P.in_zero_off = (unsigned)(hin.size() * 2); P.halves = BBLAB_HV;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int grid = std::min(P.ntiles, prop.multiProcessorCount);
#ifdef PN_STAMP
    unsigned long long *dst; CK(hipMalloc(&dst, (size_t)grid * 16 * 8)); CK(hipMemset(dst, 0, (size_t)grid * 16 * 8)); CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst)));
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(bb64_kernel<BBLAB_HV>), hipFuncAttributeMaxDynamicSharedMemorySize, BB_LDS));
    auto launch = [&]() { hipLaunchKernelGGL(bb64_kernel<BBLAB_HV>, dim3(grid), dim3((4 * BBLAB_HV + 4) * 64), BB_LDS, 0, P); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, flops = 2.0 * 2.0 * npx * 64 * 64 * 9;
    printf("bb64 B%d %dx%d: %.2f us/launch  %.1f TFLOP/s algorithmic  (%d tiles on %d workgroups)\n", B, H, W, us, flops / us / 1e6, P.ntiles, grid);
#ifdef PN_STAMP
    std::vector<unsigned long long> st((size_t)grid * 16); CK(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
    double avg[16] = {0}; int cnt = 0; std::vector<double> clk;
    for (int b = 0; b < grid; ++b) { if (!st[(size_t)b * 16 + 1]) continue; ++cnt; for (int i = 1; i < 13; ++i) avg[i] += st[(size_t)b * 16 + i] ? (double)(st[(size_t)b * 16 + i] - st[(size_t)b * 16 + 1]) : 0;
        avg[0] += (double)(st[(size_t)b * 16 + 12] - st[(size_t)b * 16]);
        if (st[(size_t)b * 16 + 15] > st[(size_t)b * 16 + 14]) clk.push_back((double)(st[(size_t)b * 16 + 12] - st[(size_t)b * 16]) / (double)(st[(size_t)b * 16 + 15] - st[(size_t)b * 16 + 14]) * 0.1); }
    std::sort(clk.begin(), clk.end());
    printf("third tile, cycles from its start (avg over %d workgroups): conv1 done %.0f, intermediate written %.0f, conv2 done %.0f, tile end %.0f | whole kernel %.0f cycles, clock %.3f GHz\n",
           cnt, avg[2] / cnt, avg[3] / cnt, avg[4] / cnt, avg[5] / cnt, avg[0] / cnt, clk.empty() ? 0.0 : clk[clk.size() / 2]);
#endif
    return 0;
}
