// Development tool (not part of the product): the three conv4 levels of a stage (28x28, B frames) as
//   (a) three launches back to back on one stream (what the network does),
//   (b) the same three launches on three streams with NO dependencies (wrong as a network, timing only): what tail filling
//       and shared-mode packing across levels could buy if the kernel boundaries were replaced by finer dependencies.
//   (c) ONE persistent launch (512 workgroups = two per CU) that walks the three levels with a grid barrier between them: every
//       thread fences its stores (device scope), thread 0 takes a ticket and spins (bounded) until all 512 arrived, then an acquire
//       fence -- what a level boundary costs when it is NOT a kernel boundary (round 3 question: is a persistent stage kernel worth
//       building?), and (d) the same without barriers or fences (wrong as a network: the upper bound of that idea).
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

struct StageDesc {
    const ConvProblem *probs[3];
    const int2 *tiles[3];          // (problem, block_x) of every real block of the level, long blocks first
    int ntiles[3];
    unsigned *counter;
};

// mode 1: every thread fences (the textbook form: 2 048 waves each write back / invalidate their XCD's L2);
// mode 2: every wave waits for its own stores (vmcnt(0)), then ONE wave per workgroup does the release, the ticket, the spin and the acquire
__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target, int mode) {
    if (mode == 1) __threadfence();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        // mode 3 (build with -DPN4_WT_STORE: the epilogue's stores are write-through, nothing dirty is left in the L2): no release
        if (mode == 3) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;                                // bounded: a lab run must never hang the GPU
        // relaxed polls (an acquire load invalidates the L2 on every iteration), one acquire fence once the count is reached
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 400000u) __builtin_amdgcn_s_sleep(16);
        if (mode >= 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (mode == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(256, 2) void stage_persist_kernel(StageDesc D, unsigned base, int sync) {
    for (int l = 0; l < 3; ++l) {
        for (int t = blockIdx.x; t < D.ntiles[l]; t += gridDim.x) {
            const int2 pt = D.tiles[l][t];
            conv4_body(D.probs[l][pt.x], pt.y);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();                               // the next tile reuses the LDS images
        }
        if (sync && l < 2) grid_barrier(D.counter, base + (unsigned)(l + 1) * gridDim.x, sync);
    }
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
    // ---- (c), (d): one persistent launch per stage ----
    StageDesc D;
    for (int l = 0; l < 3; ++l) {
        std::vector<int2> tl;
        for (size_t y = 0; y < lv[l].probs.size(); ++y)          // problems are listed long (most k-steps) first
            for (int x = 0; x < lv[l].probs[y].nblocks; ++x) tl.push_back(int2{(int)y, x});
        int2 *dt;
        CK(hipMalloc(&dt, tl.size() * sizeof(int2)));
        CK(hipMemcpy(dt, tl.data(), tl.size() * sizeof(int2), hipMemcpyHostToDevice));
        D.probs[l] = lv[l].dev; D.tiles[l] = dt; D.ntiles[l] = (int)tl.size();
        printf("level %d: %d real blocks\n", l, D.ntiles[l]);
    }
    CK(hipMalloc(&D.counter, 4)); CK(hipMemset(D.counter, 0, 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(stage_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PN4_LDS));
    const int grid = 512;
    unsigned launches = 0;
    for (int sync = 3; sync >= 0; --sync) {
        for (int i = 0; i < 3; ++i) { hipLaunchKernelGGL(stage_persist_kernel, dim3(grid), dim3(256), PN4_LDS, st[0], D, launches * 2u * grid, sync); if (sync) ++launches; }
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < iters; ++i) { hipLaunchKernelGGL(stage_persist_kernel, dim3(grid), dim3(256), PN4_LDS, st[0], D, launches * 2u * grid, sync); if (sync) ++launches; }
        CK(hipEventRecord(e1, st[0]));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        unsigned cnt = 0; CK(hipMemcpy(&cnt, D.counter, 4, hipMemcpyDeviceToHost));
        printf("stage (3 conv4 levels, B%d): ONE persistent launch of %d workgroups, %s: %.2f us per stage  %.1f TFLOP/s   (barrier arrivals %u, expected %u)\n", B, grid,
               sync == 1 ? "grid barrier, every thread fences" : sync == 2 ? "grid barrier, one release / acquire per workgroup" : sync == 3 ? "grid barrier, no release (meaningful with -DPN4_WT_STORE), one acquire per workgroup" : "no barrier, no fences (timing only)", us, flops / us / 1e6, cnt, launches * 2u * grid);
    }
    return 0;
}
