// Development tool (not part of the product): what does ONE ds_read_b128 (or one LDS-DMA instruction) cost a wave that is
// otherwise issuing nothing but MFMAs?  The conv4_kernel ablations (profiles/r02_conv4_ablations.txt) price the 11 fragment
// reads of a 28-MFMA k-step at 147 cycles; this isolates the effect per MFMA shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 docs/lab-archive/mfma_lds_issue.hip -o popnet_amd/build/mfma_lds_issue
// Per variant: cycles per k-step (= 28 x 16x16x32 or 14 x 32x32x16 MFMAs, 448 cycles of matrix-pipe time either way) with
// NR fragment reads interleaved, at 1 and 2 waves per SIMD, every CU busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// SHAPE 0: 28 x v_mfma_f32_16x16x32_bf16 per step (4 A x 7 B accumulator tiles); SHAPE 1: 14 x v_mfma_f32_32x32x16_bf16 (2 x 7)
// NR: ds_read_b128 per step, spread evenly; the data feeds later MFMAs (queue), so every read has a real consumer.
template <int SHAPE, int NR, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(const __bf16 *__restrict__ in, float *__restrict__ out, unsigned long long *__restrict__ cyc, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 16; i += blockDim.x) reinterpret_cast<bf16x8 *>(smem)[i] = reinterpret_cast<const bf16x8 *>(in)[i];
    __syncthreads();
    constexpr int NM = SHAPE == 0 ? 28 : 14;
    constexpr int Q = NR > 0 ? NR : 2;                     // one operand register per read of a step
    constexpr int LEAD = 5;                                 // a fragment is consumed no earlier than LEAD MFMAs after its read was issued
    bf16x8 q[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) q[i] = reinterpret_cast<const bf16x8 *>(smem)[lane + 64 * i];
    f32x4 acc4[SHAPE == 0 ? 28 : 1];
    f32x16 acc16[SHAPE == 1 ? 7 : 1];
#pragma unroll
    for (auto &a : acc4) a = f32x4{0, 0, 0, 0};
#pragma unroll
    for (auto &a : acc16) for (int i = 0; i < 16; ++i) a[i] = 0;
    const int raddr = lane * 16;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
        for (int m = 0; m < NM; ++m) {
            // read r (r = 0 .. NR-1) is issued in front of MFMA floor(r * NM / NR) and lands in register r; MFMA m multiplies the two
            // newest fragments whose reads are at least LEAD MFMAs old (indices wrap into the previous step)
            const int r0 = (m * NR + NM - 1) / NM, r1 = ((m + 1) * NR + NM - 1) / NM;
#pragma clang loop unroll(full)
            for (int r = r0; r < r1; ++r)
                q[r % Q] = *reinterpret_cast<const bf16x8 *>(smem + raddr + ((r * 1024 + (s & 7) * 2048) & 32767));
            const int rr = NR > 0 ? ((m - LEAD + 4 * NM) * NR + NM - 1) / NM : 2;      // reads issued before MFMA m - LEAD (+ 4 steps)
            const int ia = (rr + 4 * Q - 1) % Q, ib = (rr + 4 * Q - 2) % Q;
            if (SHAPE == 0) acc4[m % 28] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q[ia], q[ib], acc4[m % 28], 0, 0, 0);
            else acc16[m % 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q[ia], q[ib], acc16[m % 7], 0, 0, 0);
            if (r1 - r0 == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (r1 - r0 == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
#pragma unroll
    for (auto &a : acc4) sum += a[0] + a[1] + a[2] + a[3];
#pragma unroll
    for (auto &a : acc16) for (int i = 0; i < 16; ++i) sum += a[i];
    out[(size_t)blockIdx.x * blockDim.x + tid] = sum;
    if (lane == 0) cyc[(size_t)blockIdx.x * WAVES + (tid >> 6)] = t1 - t0;
}

template <int SHAPE, int NR, int WAVES>
static void run(const __bf16 *din, float *dout, unsigned long long *dcyc, int steps) {
    auto kern = k<SHAPE, NR, WAVES>;
    const int blocks = 256;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), 32768, 0, din, dout, dcyc, steps);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(WAVES * 64), 32768, 0, din, dout, dcyc, steps);
    CK(hipEventRecord(b, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)blocks * WAVES);
    CK(hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double c = (double)h[h.size() / 2] / steps;
    const double flops = 2.0 * 16 * 16 * 32 * 28 * steps * (double)blocks * WAVES;
    printf("%s  reads/step %2d  waves/SIMD %d : %7.1f cycles per step per wave (448 = matrix pipe alone; %5.1f per read over the 0-read run)  %.0f TFLOP/s\n",
           SHAPE == 0 ? "16x16x32 x28" : "32x32x16 x14", NR, WAVES / 4, c, 0.0, flops / (ms * 1e-3) / 1e12);
}

int main() {
    const int steps = 4000;
    std::vector<unsigned short> h(32768 / 2);
    srand(3);
    for (auto &v : h) v = (unsigned short)(0x3c00 + (rand() & 0x1ff));      // random bf16 around 0.01
    __bf16 *din; float *dout; unsigned long long *dcyc;
    CK(hipMalloc(&din, 32768)); CK(hipMemcpy(din, h.data(), 32768, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, 256 * 512 * 4)); CK(hipMalloc(&dcyc, 256 * 8 * 8));
#define ROW(S, W) run<S, 0, W>(din, dout, dcyc, steps); run<S, 4, W>(din, dout, dcyc, steps); run<S, 7, W>(din, dout, dcyc, steps); \
                  run<S, 11, W>(din, dout, dcyc, steps); run<S, 14, W>(din, dout, dcyc, steps); run<S, 18, W>(din, dout, dcyc, steps);
    ROW(0, 4) ROW(1, 4) ROW(0, 8) ROW(1, 8)
    return 0;
}
