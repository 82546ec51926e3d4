// What the matrix cores of THIS box sustain on v_mfma_f32_16x16x32_bf16 with nothing else in the loop: the ceiling
// that profiles/README.md sets the conv kernels' TFLOP/s against, next to the 2.5 PFLOP/s spec figure the bench prices with.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_peak.hip -o popnet_amd/build/mfma_peak
//   ./mfma_peak [waves per SIMD = 4] [seconds = 2] [random = 1] [shape = 16 | 32]
// shape 32 (round 5): v_mfma_f32_32x32x16_bf16, the same FLOPs per cycle with half the operand reads per FLOP -- does the power-limited clock notice?
// Operands are random bf16 in [-1, 1) (or zeros with random = 0: the chip holds a higher clock on zeros); 14 independent
// accumulator tiles per wave (the conv kernel's 2 x 7), no memory traffic inside the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mfma_loop(const bf16x8 *src, float *sink, unsigned long long *stamps, int iters) {
    bf16x8 a[2], b[7];
    for (int i = 0; i < 2; ++i) a[i] = src[(threadIdx.x + 256 * i) & 1023];
    for (int i = 0; i < 7; ++i) b[i] = src[(threadIdx.x * 3 + 64 * i + 17) & 1023];
    f32x4 acc[14];
    for (int i = 0; i < 14; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 14; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 14; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void mfma_loop32(const bf16x8 *src, float *sink, unsigned long long *stamps, int iters) {
    bf16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) a[i] = src[(threadIdx.x + 256 * i) & 1023];
    for (int i = 0; i < 2; ++i) b[i] = src[(threadIdx.x * 3 + 64 * i + 17) & 1023];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 7; ++u)               // 28 MFMAs of 32768 FLOPs = the 56 x 16384 of the 16 x 16 loop
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char **argv) {
    int wps = argc > 1 ? atoi(argv[1]) : 4;
    double seconds = argc > 2 ? atof(argv[2]) : 2.0;
    int random = argc > 3 ? atoi(argv[3]) : 1;
    const int shape = argc > 4 ? atoi(argv[4]) : 16;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount, blocks = cus * wps;           // 4-wave blocks: wps blocks per CU = wps waves per SIMD
    std::vector<unsigned short> h(1024 * 8);
    srand(1234);
    for (auto &v : h) { float f = random ? (rand() / (float)RAND_MAX) * 2.f - 1.f : 0.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    bf16x8 *src; float *sink; unsigned long long *st;
    CK(hipMalloc(&src, h.size() * 2)); CK(hipMalloc(&sink, (size_t)blocks * 256 * 4)); CK(hipMalloc(&st, (size_t)blocks * 16));
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const int iters = 2000;                                           // 2000 x 56 MFMAs x 16 cycles = 1.8 M cycles per wave (~1 ms at 4 waves / SIMD)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() { if (shape == 32) mfma_loop32<<<blocks, 256>>>(src, sink, st, iters); else mfma_loop<<<blocks, 256>>>(src, sink, st, iters); };
    launch(); CK(hipDeviceSynchronize());
    int launches = 0; float ms = 0, total = 0;
    while (total < seconds * 1e3) {                                   // back to back until the clock has settled; the last batch is the one reported
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1)); total += ms; launches += 20;
    }
    double flops = (double)blocks * 4 * iters * 56 * 2.0 * 16 * 16 * 32 * 20;
    std::vector<unsigned long long> hs((size_t)blocks * 2); CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk; for (int b = 0; b < blocks; ++b) if (hs[b * 2 + 1]) clk.push_back((double)hs[b * 2] / (double)hs[b * 2 + 1] * 0.1);
    std::sort(clk.begin(), clk.end());
    double ghz = clk.empty() ? 0 : clk[clk.size() / 2];
    printf("%s: %d CUs, shape %d, %d waves/SIMD, %s operands: %.1f TFLOP/s after %d launches; in-kernel clock %.3f GHz (p10 %.3f, p90 %.3f); %.1f flop/cycle/CU of 4096\n",
           prop.gcnArchName, cus, shape, wps, random ? "random" : "zero", flops / (ms * 1e-3) / 1e12, launches, ghz, clk.empty() ? 0 : clk[clk.size() / 10], clk.empty() ? 0 : clk[clk.size() * 9 / 10],
           flops / (ms * 1e-3) / (ghz * 1e9) / cus);
    return 0;
}
