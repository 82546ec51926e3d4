// Bandwidth-bound companions of the MFMA convolution: the 7x7 stride-2 single-channel stem,
// pooling, and NHWC -> NCHW export.  All are HBM/L2-streaming kernels (no matrix cores: Cin = 1
// gives the stem an arithmetic intensity of ~46 flop/B).
//
//   stem   nn.Conv2d(1, 64, 7, stride 2, pad 3, bias=False) + BN + ReLU
//          tpm/lib/network/rtpose_light3d.py:145-147,203-205 ; tpm/lib/network/yolo_posenet.py:33-36,46-48
//   pools  nn.AvgPool2d(3, 2, 1) (count_include_pad -> divisor 9)  rtpose_light3d.py:152,158
//          nn.MaxPool2d(3, 2, 1)                                   yolo_posenet.py:37
//          nn.MaxPool2d(2, 2)                                      yolo_posenet.py:118
#include "pn_internal.h"
#include "preproc_pixel.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---------------------------------------------------------------------------------------------
// Stem.  Block = 16x16 output pixels of one image; the 37x37 input patch and the 49x64 weight
// table sit in LDS.  Each thread produces all 64 output channels of one pixel (weights are read
// as wave-wide LDS broadcasts), then writes one contiguous 128-B / 256-B NHWC pixel.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem7x7_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                       const float *__restrict__ bias, T *__restrict__ out,
                                                       int H, int W, int Ho, int Wo, int out_cs, int split) {
    __shared__ float tile[37][40];
    __shared__ __attribute__((aligned(16))) float wl[49 * 64];
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    for (int i = tid; i < 49 * 64; i += 256) wl[i] = w[i];
    const float *xb = x + (size_t)b * H * W;
    for (int i = tid; i < 37 * 37; i += 256) {
        int r = i / 37, cc = i - r * 37;
        int iy = iy0 + r, ix = ix0 + cc;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xb[(size_t)iy * W + ix];
        tile[r][cc] = v;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int oy = oy0 + ty, ox = ox0 + tx;
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float xv = tile[ty * 2 + ky][tx * 2 + kx];
            const f32x4 *wp = reinterpret_cast<const f32x4 *>(&wl[(ky * 7 + kx) * 64]);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                f32x4 w4 = wp[i];
                acc[4 * i + 0] = fmaf(xv, w4[0], acc[4 * i + 0]);
                acc[4 * i + 1] = fmaf(xv, w4[1], acc[4 * i + 1]);
                acc[4 * i + 2] = fmaf(xv, w4[2], acc[4 * i + 2]);
                acc[4 * i + 3] = fmaf(xv, w4[3], acc[4 * i + 3]);
            }
        }
    }
    if (oy < Ho && ox < Wo) {
        T *op = out + ((size_t)(b * Ho + oy) * Wo + ox) * out_cs;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            float v = acc[i] + bias[i];
            v = v > 0.f ? v : 0.f;
            const T hi = (T)v;
            op[i] = hi;
            if (split) {                                  // bf16x3: planes [hi | lo]
                op[split + i] = (T)(v - (float)hi);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stem on the matrix cores (bf16 mode).  D[cout][pixel] = sum_k W[cout][k] X[k][pixel] with
// k = ky*8 + kx' (kx' = kx + 1: tap column 0 and tap row 7 carry zero weights), i.e. K = 64 = two
// v_mfma_f32_16x16x32_bf16 steps.  Shifting the window by one column makes every lane's 8-tap row
// segment start at an EVEN input column (2*ox - 4): four 4-byte-aligned ds_read_b32 build the B
// fragment straight from a bf16 copy of the input tile -- no im2col buffer.  The 8 weight fragments
// (64 couts x 64 k) live in registers for the whole block.  Tile rows of cout tile t are permuted
// (row 4q+r <-> cout 16q+4t+r) so that a lane's 16 accumulators are 16 CONSECUTIVE channels of its
// pixel: two 16-B NHWC stores.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 stem_bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int stem_u32x4;
#define STEM_PITCH 96      // bf16 elements per LDS row: PITCH/2 = 48 = 16 (mod 32) -> the two row groups of a ds_read_b32 phase hit disjoint banks

// X3 (bf16x3 mode): the input patch is kept as hi + lo bf16 parts, the weights as hi + lo fragments (16 instead of 8), and
// every product is x_hi w_hi + x_lo w_hi + x_hi w_lo -- the stem must not round the depth frame to 8 significant bits.
// SRC: 0 = x is the pre-processed [B, 1, H, W] f32 input; 1 / 2 = the kernel reads the RAW depth frames (f16 / f32) and computes
// every pixel of its input tile with pn_preproc_pixel -- pn_preprocess's own arithmetic, so the tile holds the same floats:
// the pre-processing launch and the 6.4 MB it writes and the stem reads back (B = 32) disappear (pn_*_forward_frames).
template <bool X3, int SRC = 0>
__global__ __launch_bounds__(256) void stem7x7_mfma_kernel(const float *__restrict__ x, const __bf16 *__restrict__ wfrag,
                                                            const float *__restrict__ bias, __bf16 *__restrict__ out,
                                                            int H, int W, int Ho, int Wo, int out_cs, int split, PnFrameSrc src) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[38 * STEM_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 tile_lo[X3 ? 38 * STEM_PITCH : 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b = blockIdx.z, oy0 = blockIdx.y * 16, ox0 = blockIdx.x * 16;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 4;
    const float *xb = x + (size_t)b * H * W;
    if (SRC != 0) {
        // coordinate terms once per tile column / row (threads 0..37 / 64..101), then 4 taps + the interpolation per pixel
        __shared__ PnAxisX axs[38];
        __shared__ PnAxisY ays[38];
        if (tid < 38) axs[tid] = pn_preproc_axis_x(ix0 + tid, src.scale_x, src.W);
        else if (tid >= 64 && tid < 64 + 38) ays[tid - 64] = pn_preproc_axis_y(iy0 + tid - 64, src.scale_y, src.H);
        __syncthreads();
        for (int i = tid; i < 38 * 38; i += 256) {
            const int r = i / 38, cc = i - r * 38;
            const int iy = iy0 + r, ix = ix0 + cc;
            float v = 0.f;
            if (r < 37 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                if (SRC == 1) v = pn_preproc_combine((const _Float16 *)src.frames + (size_t)b * src.H * src.W, src.W, axs[cc], ays[r], src.dmax, src.mean, src.stdv);
                else v = pn_preproc_combine((const float *)src.frames + (size_t)b * src.H * src.W, src.W, axs[cc], ays[r], src.dmax, src.mean, src.stdv);
            }
            const __bf16 vh = (__bf16)v;
            tile[r * STEM_PITCH + cc] = vh;
            if (X3) tile_lo[r * STEM_PITCH + cc] = (__bf16)(v - (float)vh);
        }
    } else if ((W & 3) == 0) {
        // 16-byte loads: ix0 = 32 k - 4 and W are multiples of 4, so a group of four columns is entirely inside or entirely
        // outside the frame; 380 vector loads per block instead of 1 444 scalar ones (round 3: the stem was 25 us of a 0.49 ms
        // step with the texture addresser 7x as busy as the matrix pipe, and not hidden behind the other streams)
        for (int i = tid; i < 38 * 10; i += 256) {
            const int r = i / 10, g = i - r * 10;
            const int iy = iy0 + r, ix = ix0 + 4 * g;
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (r < 37 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const float4 *>(xb + (size_t)iy * W + ix);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            __bf16 h4[4], l4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                h4[k] = (__bf16)vv[k];                   // row 37 (tap row 7) is zero: its weights are zero, the data must be finite
                l4[k] = (__bf16)(vv[k] - (float)h4[k]);
            }
            *reinterpret_cast<uint2 *>(&tile[r * STEM_PITCH + 4 * g]) = *reinterpret_cast<uint2 *>(h4);
            if (X3) *reinterpret_cast<uint2 *>(&tile_lo[r * STEM_PITCH + 4 * g]) = *reinterpret_cast<uint2 *>(l4);
        }
    } else
    for (int i = tid; i < 38 * 38; i += 256) {
        const int r = i / 38, cc = i - r * 38;
        const int iy = iy0 + r, ix = ix0 + cc;
        float v = 0.f;
        if (r < 37 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xb[(size_t)iy * W + ix];
        const __bf16 vh = (__bf16)v;
        tile[r * STEM_PITCH + cc] = vh;                 // row 37 (tap row 7) is zero: its weights are zero, the data must be finite
        if (X3) tile_lo[r * STEM_PITCH + cc] = (__bf16)(v - (float)vh);
    }
    stem_bf16x8 aw[4][2], awl[X3 ? 4 : 1][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            aw[t][s] = *reinterpret_cast<const stem_bf16x8 *>(wfrag + ((t * 2 + s) * 64 + lane) * 8);
            if (X3) awl[t][s] = *reinterpret_cast<const stem_bf16x8 *>(wfrag + ((8 + t * 2 + s) * 64 + lane) * 8);
        }
    float bs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bs[i] = bias[16 * q + i];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int ry = wave * 4 + rr;                    // output row of this pixel tile (16 pixels: ox0 .. ox0+15)
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned *src = reinterpret_cast<const unsigned *>(&tile[(2 * ry + 4 * s + q) * STEM_PITCH + 2 * c]);
            stem_u32x4 raw = {src[0], src[1], src[2], src[3]};
            stem_bf16x8 bf = *reinterpret_cast<stem_bf16x8 *>(&raw);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[t][s], bf, acc[t], 0, 0, 0);
            if (X3) {
                const unsigned *srl = reinterpret_cast<const unsigned *>(&tile_lo[(2 * ry + 4 * s + q) * STEM_PITCH + 2 * c]);
                stem_u32x4 rawl = {srl[0], srl[1], srl[2], srl[3]};
                stem_bf16x8 bl = *reinterpret_cast<stem_bf16x8 *>(&rawl);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[t][s], bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(awl[t][s], bf, acc[t], 0, 0, 0);
                }
            }
        }
        const int oy = oy0 + ry, ox = ox0 + c;
        if (oy < Ho && ox < Wo) {
            // bf16x3: two planes [hi | lo] `split` channels apart (pn_internal.h, ConvProblem::split)
            for (int pl = 0; pl < (split ? 2 : 1); ++pl) {
                __bf16 o[16];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[t][r] + bs[4 * t + r];
                        v = v > 0.f ? v : 0.f;
                        const __bf16 hi = (__bf16)v;
                        o[4 * t + r] = pl == 1 ? (__bf16)(v - (float)hi) : hi;
                    }
                __bf16 *op = out + ((size_t)(b * Ho + oy) * Wo + ox) * out_cs + 16 * q + pl * split;
                reinterpret_cast<stem_u32x4 *>(op)[0] = reinterpret_cast<stem_u32x4 *>(o)[0];
                reinterpret_cast<stem_u32x4 *>(op)[1] = reinterpret_cast<stem_u32x4 *>(o)[1];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stem + MaxPool2d(3, 2, 1) in one launch (YoloPoseNet: conv1 - bn1 - relu - maxpool, resnet.py:134-148 / yolo_posenet.py:101-108;
// net.hip::build_yolo, POPNET_NO_STEMPOOL=1 keeps the two launches).  A block owns 8 x 7 pooled pixels: it computes the 17 x 15
// stem outputs their windows cover (one row and one column shared with the neighbouring blocks are recomputed), parks them as
// bf16 in LDS and takes the window maxima there -- max is exact, so the result equals pool_kernel's on the stored map bit for
// bit, and the 112 x 112 x 64 map (51 MB written and read back at B = 32) is never stored.  Same MFMA formulation as
// stem7x7_mfma_kernel (20 output rows per block instead of 16: wave w owns rows 5w .. 5w + 4, rows 17..19 are skipped).
// ---------------------------------------------------------------------------------------------
template <int SRC>
__global__ __launch_bounds__(256) void stem7x7_pool_kernel(const float *__restrict__ x, const __bf16 *__restrict__ wfrag,
                                                            const float *__restrict__ bias, __bf16 *__restrict__ out,
                                                            int H, int W, int Ho, int Wo, int Hp, int Wp, int out_cs, PnFrameSrc src) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[40 * STEM_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 otile[17 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int b = blockIdx.z, py0 = blockIdx.y * 8, px0 = blockIdx.x * 7;
    const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;                  // first stem output row / column of the block (may be -1)
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 4;
    if (SRC != 0) {
        __shared__ PnAxisX axs[38];
        __shared__ PnAxisY ays[40];
        if (tid < 38) axs[tid] = pn_preproc_axis_x(ix0 + tid, src.scale_x, src.W);
        else if (tid >= 64 && tid < 64 + 40) ays[tid - 64] = pn_preproc_axis_y(iy0 + tid - 64, src.scale_y, src.H);
        __syncthreads();
        for (int i = tid; i < 40 * 38; i += 256) {
            const int r = i / 38, cc = i - r * 38;
            const int iy = iy0 + r, ix = ix0 + cc;
            float v = 0.f;
            if (r < 39 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                if (SRC == 1) v = pn_preproc_combine((const _Float16 *)src.frames + (size_t)b * src.H * src.W, src.W, axs[cc], ays[r], src.dmax, src.mean, src.stdv);
                else v = pn_preproc_combine((const float *)src.frames + (size_t)b * src.H * src.W, src.W, axs[cc], ays[r], src.dmax, src.mean, src.stdv);
            }
            tile[r * STEM_PITCH + cc] = (__bf16)v;
        }
    } else {
        const float *xb = x + (size_t)b * H * W;
        for (int i = tid; i < 40 * 38; i += 256) {
            const int r = i / 38, cc = i - r * 38;
            const int iy = iy0 + r, ix = ix0 + cc;
            float v = 0.f;
            if (r < 39 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xb[(size_t)iy * W + ix];
            tile[r * STEM_PITCH + cc] = (__bf16)v;             // row 39 (tap row 7 of output row 16) is zero: its weights are zero
        }
    }
    stem_bf16x8 aw[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) aw[t][s2] = *reinterpret_cast<const stem_bf16x8 *>(wfrag + ((t * 2 + s2) * 64 + lane) * 8);
    float bs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) bs[i] = bias[16 * q + i];
    __syncthreads();
    for (int rr = 0; rr < 5; ++rr) {
        const int ry = wave * 5 + rr;                           // wave-uniform
        if (ry >= 17) break;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned *sp = reinterpret_cast<const unsigned *>(&tile[(2 * ry + 4 * s2 + q) * STEM_PITCH + 2 * c]);
            stem_u32x4 raw = {sp[0], sp[1], sp[2], sp[3]};
            stem_bf16x8 bf = *reinterpret_cast<stem_bf16x8 *>(&raw);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aw[t][s2], bf, acc[t], 0, 0, 0);
        }
        __bf16 o[16];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[t][r] + bs[4 * t + r];
                o[4 * t + r] = (__bf16)(v > 0.f ? v : 0.f);       // the same bias + ReLU + rounding as stem7x7_mfma_kernel's epilogue
            }
        stem_u32x4 *dst = reinterpret_cast<stem_u32x4 *>(&otile[(ry * 16 + c) * 64 + 16 * q]);
        dst[0] = reinterpret_cast<stem_u32x4 *>(o)[0];
        dst[1] = reinterpret_cast<stem_u32x4 *>(o)[1];
    }
    __syncthreads();
    for (int item = tid; item < 8 * 7 * 8; item += 256) {
        const int g8 = item & 7, pp = item >> 3, pi = pp / 7, pj = pp - pi * 7;
        const int py = py0 + pi, px = px0 + pj;
        if (py >= Hp || px >= Wp) continue;
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ry = 2 * pi + ky, cc = 2 * pj + kx;     // window taps in block coordinates; map coordinates oy0 + ry, ox0 + cc
                const bool ok = (unsigned)(oy0 + ry) < (unsigned)Ho && (unsigned)(ox0 + cc) < (unsigned)Wo;
                __bf16 tv[8];
                *reinterpret_cast<stem_u32x4 *>(tv) = *reinterpret_cast<const stem_u32x4 *>(&otile[(ry * 16 + cc) * 64 + g8 * 8]);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = ok ? fmaxf(m[k], (float)tv[k]) : m[k];
            }
        __bf16 o8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o8[k] = (__bf16)m[k];
        *reinterpret_cast<stem_u32x4 *>(out + ((size_t)(b * Hp + py) * Wp + px) * out_cs + g8 * 8) = *reinterpret_cast<stem_u32x4 *>(o8);
    }
}

int pn_launch_stem_pool(pn_ctx *ctx, const float *x, const void *wfrag, const float *bias, void *out, int B, int H, int W, int Ho, int Wo,
                        int out_cs, hipStream_t stream, const PnFrameSrc *src) {
    const int Hp = (Ho - 1) / 2 + 1, Wp = (Wo - 1) / 2 + 1;
    dim3 grid((Wp + 6) / 7, (Hp + 7) / 8, B), block(256);
    const PnFrameSrc none = {};
    if (!src) hipLaunchKernelGGL(stem7x7_pool_kernel<0>, grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, Hp, Wp, out_cs, none);
    else if (src->dtype == PN_DEPTH_F16) hipLaunchKernelGGL(stem7x7_pool_kernel<1>, grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, Hp, Wp, out_cs, *src);
    else hipLaunchKernelGGL(stem7x7_pool_kernel<2>, grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, Hp, Wp, out_cs, *src);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_launch_stem(pn_ctx *ctx, int prec, const float *x, const float *w, const void *wfrag, const float *bias, void *out,
                   int B, int H, int W, int Ho, int Wo, int out_cs, int split, hipStream_t stream, const PnFrameSrc *src) {
    dim3 grid((Wo + 15) / 16, (Ho + 15) / 16, B), block(256);
    const PnFrameSrc none = {};
    if (src) {                           // raw depth frames in (matrix-core stems only: net.hip refuses the fp32 net)
        const bool f16 = src->dtype == PN_DEPTH_F16;
        if (prec == PN_PREC_BF16 && f16) hipLaunchKernelGGL((stem7x7_mfma_kernel<false, 1>), grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, out_cs, split, *src);
        else if (prec == PN_PREC_BF16) hipLaunchKernelGGL((stem7x7_mfma_kernel<false, 2>), grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, out_cs, split, *src);
        else if (prec == PN_PREC_BF16X3 && f16) hipLaunchKernelGGL((stem7x7_mfma_kernel<true, 1>), grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, out_cs, split, *src);
        else if (prec == PN_PREC_BF16X3) hipLaunchKernelGGL((stem7x7_mfma_kernel<true, 2>), grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, out_cs, split, *src);
        else return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "the frames-in stem is built for the bf16 and bf16x3 nets");
        PN_HIP_CHECK(ctx, hipGetLastError());
        return PN_OK;
    }
    if (prec == PN_PREC_BF16)
        hipLaunchKernelGGL((stem7x7_mfma_kernel<false, 0>), grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, out_cs, split, none);
    else if (prec == PN_PREC_BF16X3)     // split input and split weights on the matrix cores, split bf16 output
        hipLaunchKernelGGL((stem7x7_mfma_kernel<true, 0>), grid, block, 0, stream, x, (const __bf16 *)wfrag, bias, (__bf16 *)out, H, W, Ho, Wo, out_cs, split, none);
    else
        hipLaunchKernelGGL(stem7x7_kernel<float>, grid, block, 0, stream, x, w, bias, (float *)out, H, W, Ho, Wo, out_cs, 0);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

// ---------------------------------------------------------------------------------------------
// Pooling on NHWC.  One thread = one output pixel x 8 channels (16 B of bf16 / 32 B of f32).
// ---------------------------------------------------------------------------------------------
// SPLIT (round 5): the bf16x3 form (value = hi plane + lo plane) as its own instantiation -- with the lo-plane load behind a run-time `if` the
// compiler waited for every tap's pair before issuing the next (nine dependent round trips: 28 us per launch against 15.7 us for twice the bytes)
template <typename T, int MODE, bool SPLIT = false>
__global__ void pool_kernel(const T *__restrict__ in, T *__restrict__ out, int B, int H, int W, int Ho, int Wo,
                            int C8, int in_cs, int out_cs, int out_coff, int in_split, int out_split) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * Ho * Wo * C8;
    if (gid >= total) return;
    int c8 = (int)(gid % C8);
    size_t p = gid / C8;
    int ox = (int)(p % Wo);
    size_t t = p / Wo;
    int oy = (int)(t % Ho);
    int b = (int)(t / Ho);
    constexpr int K = (MODE == 2) ? 2 : 3;
    constexpr int PAD = (MODE == 2) ? 0 : 1;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (MODE == 0) ? 0.f : -INFINITY;
    // Branch-free: every tap is loaded (from a clamped address) before the first is used, so the K*K 16-byte loads of a thread
    // are in flight together; a tap outside the map adds 0 (average: count_include_pad, x + 0 == x) or is skipped by a select
    // (max).  Same taps in the same order as the looped form: same results bit for bit, at HBM rate instead of one dependent
    // load at a time (round 3: the two average pools were 28 us of a 0.49 ms step and not hidden).
    T v[K * K][8], v2[K * K][8];
    bool ok[K * K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int iy = oy * 2 - PAD + ky, ix = ox * 2 - PAD + kx;
            ok[ky * K + kx] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const T *ip = in + ((size_t)(b * H + cy) * W + cx) * in_cs + c8 * 8;
            if (sizeof(T) == 2) {
                *reinterpret_cast<uint4 *>(v[ky * K + kx]) = *reinterpret_cast<const uint4 *>(ip);
                if (SPLIT) *reinterpret_cast<uint4 *>(v2[ky * K + kx]) = *reinterpret_cast<const uint4 *>(ip + in_split);
            } else {
                reinterpret_cast<uint4 *>(v[ky * K + kx])[0] = reinterpret_cast<const uint4 *>(ip)[0];
                reinterpret_cast<uint4 *>(v[ky * K + kx])[1] = reinterpret_cast<const uint4 *>(ip)[1];
            }
        }
#pragma unroll
    for (int k = 0; k < K * K; ++k) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[k][i];
        if (sizeof(T) == 2 && SPLIT) {                    // bf16x3: value = hi plane + lo plane
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] += (float)v2[k][i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] += ok[k] ? f[i] : 0.f;
            else acc[i] = ok[k] ? fmaxf(acc[i], f[i]) : acc[i];
        }
    }
    for (int pl = 0; pl < (out_split ? 2 : 1); ++pl) {      // bf16x3: planes [hi | lo]
        T o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float r = (MODE == 0) ? acc[i] / 9.0f : acc[i];
            const T hi = (T)r;
            o[i] = pl == 1 ? (T)(r - (float)hi) : hi;
        }
        T *op = out + p * out_cs + out_coff + c8 * 8 + pl * out_split;
        if (sizeof(T) == 2) {
            *reinterpret_cast<uint4 *>(op) = *reinterpret_cast<uint4 *>(o);
        } else {
            reinterpret_cast<uint4 *>(op)[0] = reinterpret_cast<uint4 *>(o)[0];
            reinterpret_cast<uint4 *>(op)[1] = reinterpret_cast<uint4 *>(o)[1];
        }
    }
}

template <typename T>
static void launch_pool_t(int mode, const void *in, void *out, int B, int H, int W, int Ho, int Wo, int C,
                          int in_cs, int out_cs, int out_coff, int in_split, int out_split, hipStream_t stream) {
    size_t total = (size_t)B * Ho * Wo * (C / 8);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (sizeof(T) == 2 && in_split) {
        if (mode == 0)
            hipLaunchKernelGGL((pool_kernel<T, 0, true>), grid, block, 0, stream, (const T *)in, (T *)out, B, H, W, Ho, Wo, C / 8, in_cs, out_cs, out_coff, in_split, out_split);
        else if (mode == 1)
            hipLaunchKernelGGL((pool_kernel<T, 1, true>), grid, block, 0, stream, (const T *)in, (T *)out, B, H, W, Ho, Wo, C / 8, in_cs, out_cs, out_coff, in_split, out_split);
        else
            hipLaunchKernelGGL((pool_kernel<T, 2, true>), grid, block, 0, stream, (const T *)in, (T *)out, B, H, W, Ho, Wo, C / 8, in_cs, out_cs, out_coff, in_split, out_split);
        return;
    }
    if (mode == 0)
        hipLaunchKernelGGL((pool_kernel<T, 0>), grid, block, 0, stream, (const T *)in, (T *)out, B, H, W, Ho, Wo, C / 8, in_cs, out_cs, out_coff, in_split, out_split);
    else if (mode == 1)
        hipLaunchKernelGGL((pool_kernel<T, 1>), grid, block, 0, stream, (const T *)in, (T *)out, B, H, W, Ho, Wo, C / 8, in_cs, out_cs, out_coff, in_split, out_split);
    else
        hipLaunchKernelGGL((pool_kernel<T, 2>), grid, block, 0, stream, (const T *)in, (T *)out, B, H, W, Ho, Wo, C / 8, in_cs, out_cs, out_coff, in_split, out_split);
}

int pn_launch_pool(pn_ctx *ctx, int prec, int mode, const void *in, void *out, int B, int H, int W, int C,
                   int in_cs, int out_cs, int out_coff, int in_split, int out_split, hipStream_t stream) {
    if (C % 8 || in_cs % 8 || out_cs % 8 || out_coff % 8)
        return pn_set_error(ctx, PN_ERR_INVALID, "pool: channels must be multiples of 8");
    int Ho = (mode == 2) ? H / 2 : (H + 2 - 3) / 2 + 1;
    int Wo = (mode == 2) ? W / 2 : (W + 2 - 3) / 2 + 1;
    if (prec == PN_PREC_BF16) launch_pool_t<__bf16>(mode, in, out, B, H, W, Ho, Wo, C, in_cs, out_cs, out_coff, in_split, out_split, stream);
    else launch_pool_t<float>(mode, in, out, B, H, W, Ho, Wo, C, in_cs, out_cs, out_coff, 0, 0, stream);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

// ---------------------------------------------------------------------------------------------
// NHWC channel slice -> NCHW f32 (stage-1 outputs / diagnostics; not on the timed path).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T *__restrict__ in, float *__restrict__ out, int B, int HW, int C,
                                    int in_cs, int in_coff, int split) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)B * C * HW;
    if (gid >= total) return;
    int p = (int)(gid % HW);
    size_t t = gid / HW;
    int ch = (int)(t % C);
    int b = (int)(t / C);
    float v = (float)in[((size_t)b * HW + p) * in_cs + in_coff + ch];
    if (split) v += (float)in[((size_t)b * HW + p) * in_cs + in_coff + ch + split];      // bf16x3: hi + lo
    out[gid] = v;
}

int pn_launch_nhwc_to_nchw(pn_ctx *ctx, int prec, const void *in, float *out, int B, int H, int W, int C,
                           int in_cs, int in_coff, int split, hipStream_t stream) {
    size_t total = (size_t)B * C * H * W;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (prec == PN_PREC_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, grid, block, 0, stream, (const __bf16 *)in, out, B, H * W, C, in_cs, in_coff, split);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, block, 0, stream, (const float *)in, out, B, H * W, C, in_cs, in_coff, 0);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

// ---------------------------------------------------------------------------------------------
// NCHW f32 -> ReLU -> NHWC T channel slice (bf16x3: two planes [hi | lo] `split` channels apart).  The multi-channel stem
// (input_dim != 1: the reference constructors' default is input_dim = 3, tpm/lib/network/rtpose_light3d.py:250, yolo_posenet.py:88)
// runs the generic fp32 7x7 convolution of the training primitives (pn_conv2d_forward, any Cin) and hands its map to the NHWC
// layers through this kernel.  Not on the depth path's timed region (north_star: single-channel depth).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_relu_to_nhwc_kernel(const float *__restrict__ in, T *__restrict__ out, int B, int HW, int C, int out_cs, int split) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // gid = (b * HW + p) * C + ch: consecutive threads write consecutive channels
    size_t total = (size_t)B * C * HW;
    if (gid >= total) return;
    int ch = (int)(gid % C);
    size_t t = gid / C;
    int p = (int)(t % HW);
    int b = (int)(t / HW);
    float v = in[((size_t)b * C + ch) * HW + p];
    v = v > 0.f ? v : 0.f;
    T *op = out + ((size_t)b * HW + p) * out_cs + ch;
    const T hi = (T)v;
    op[0] = hi;
    if (split) op[split] = (T)(v - (float)hi);
}

int pn_launch_nchw_relu_to_nhwc(pn_ctx *ctx, int prec, const float *in, void *out, int B, int H, int W, int C, int out_cs, int split, hipStream_t stream) {
    size_t total = (size_t)B * C * H * W;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (prec == PN_PREC_BF16)
        hipLaunchKernelGGL(nchw_relu_to_nhwc_kernel<__bf16>, grid, block, 0, stream, in, (__bf16 *)out, B, H * W, C, out_cs, split);
    else
        hipLaunchKernelGGL(nchw_relu_to_nhwc_kernel<float>, grid, block, 0, stream, in, (float *)out, B, H * W, C, out_cs, 0);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}
