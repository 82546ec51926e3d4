// Context management + depth-frame pre-processing kernel.
//
// pn_preprocess replaces the image half of test-mode KDH3D_Keypoints.__getitem__:
//   np.load(..).astype(float) -> float32                tpm/lib/datasets/datasets_kdh3d_rtpose_mpreal.py:225 (CR)
//   cv2.resize(image, (S, S), INTER_LINEAR)             tpm/lib/datasets/data_augmentation_2d3d.py:507-510
//   image[image < 0] = 0; image[image > depth_max] = .. tpm/lib/datasets/datasets_kdh3d_rtpose_mpreal.py:238-239 (CR)
//   ToTensor + Normalize(depth_mean, depth_std)         ...:192-194,242 (CR)
// The bilinear arithmetic follows OpenCV 4.2 resize.cpp's float path (see oracle/cv2_resize.py):
// source coordinate computed in double and rounded to float, horizontal pass then vertical pass,
// float32 products summed left to right, no fused multiply-add.  HBM-bound: reads 4 taps of the
// f16 frame per output pixel (614 KB per 480x640 frame), writes S*S*4 B.
#pragma clang fp contract(off)
#include <cmath>
#include <cstdio>
#include <cstring>
#include "pn_internal.h"
#include "preproc_pixel.h"

template <typename TIN>
__global__ void preprocess_kernel(const TIN *__restrict__ depth, float *__restrict__ out, int B, int H, int W, int S,
                                  double scale_x, double scale_y, float dmax, float mean, float stdv) {
    // blockIdx.y = frame, blockIdx.x * 256 + tid = pixel of the S x S output: 32-bit index arithmetic only (the first
    // version divided a 64-bit linear index three times per pixel -- most of its 17.7 us per 32 frames)
    const unsigned pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (unsigned)(S * S)) return;
    const int b = blockIdx.y;
    const int dy = (int)(pix / (unsigned)S);
    const int dx = (int)(pix - (unsigned)dy * (unsigned)S);
    const size_t gid = (size_t)b * S * S + pix;

    out[gid] = pn_preproc_pixel(depth + (size_t)b * H * W, H, W, dy, dx, scale_x, scale_y, dmax, mean, stdv);
}

extern "C" {

int pn_abi_version(void) { return PN_ABI_VERSION; }
int pn_build_experiments(void) {
#ifdef PN_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

pn_ctx *pn_create(int device_id) {
    pn_ctx *ctx = new pn_ctx();
    ctx->device = device_id;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || device_id < 0 || device_id >= ndev) {
        // keep the context so the caller can read the message; every later call fails cleanly
        char buf[256];
        snprintf(buf, sizeof buf, "pn_create: device %d not available (%d devices, %s)", device_id, ndev,
                 hipGetErrorString(e));
        ctx->err = buf;
        ctx->device = -1;
        return ctx;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
    return ctx;
}

void pn_destroy(pn_ctx *ctx) {
    if (!ctx) return;
    if (ctx->parse_ws) (void)hipFree(ctx->parse_ws);
    pn_parse_big_free(ctx);
    if (ctx->train_ws) (void)hipFree(ctx->train_ws);
    for (void *p : ctx->train_ws_retired) (void)hipFree(p);
    for (auto &e : ctx->train_packs) (void)hipFree(e.buf);
    if (ctx->train_pack_table) (void)hipFree(ctx->train_pack_table);
    delete ctx;
}

int pn_last_error(pn_ctx *ctx, char *buf, size_t buf_len) {
    if (!ctx) return 0;
    if (buf && buf_len) {
        size_t n = ctx->err.size() < buf_len - 1 ? ctx->err.size() : buf_len - 1;
        memcpy(buf, ctx->err.data(), n);
        buf[n] = 0;
    }
    return (int)ctx->err.size();
}

int pn_preprocess(pn_ctx *ctx, const void *depth_dev, int depth_dtype, int B, int H, int W, float *out_dev, int S,
                  float depth_max, float depth_mean, float depth_std, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!depth_dev || !out_dev || B < 1 || H < 2 || W < 2 || S < 1)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_preprocess: bad arguments");
    if (W == 2 * S && H == 2 * S)   // cv::resize (4.2 resize.cpp) switches INTER_LINEAR to INTER_AREA at exactly 2x decimation: that branch is not built
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_preprocess: %dx%d -> %d is an exact 2x decimation, where cv2.resize(INTER_LINEAR) runs INTER_AREA instead: not built", W, H, S);
    const double inv_x = (double)S / (double)W, inv_y = (double)S / (double)H;
    const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;   // as cv::resize computes them
    if (B > 65535 || (size_t)S * S > 0x7fffffffu) return pn_set_error(ctx, PN_ERR_INVALID, "pn_preprocess: batch or output size out of range");
    dim3 grid((unsigned)(((size_t)S * S + 255) / 256), (unsigned)B), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
    if (depth_dtype == PN_DEPTH_F16)
        hipLaunchKernelGGL(preprocess_kernel<_Float16>, grid, block, 0, s, (const _Float16 *)depth_dev, out_dev, B, H, W, S,
                           scale_x, scale_y, depth_max, depth_mean, depth_std);
    else if (depth_dtype == PN_DEPTH_F32)
        hipLaunchKernelGGL(preprocess_kernel<float>, grid, block, 0, s, (const float *)depth_dev, out_dev, B, H, W, S,
                           scale_x, scale_y, depth_max, depth_mean, depth_std);
    else
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_preprocess: unknown depth dtype %d", depth_dtype);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

void pn_parse_cfg_default(pn_parse_cfg *cfg) {
    if (!cfg) return;
    cfg->thresh_heatmap = 0.1f;
    cfg->thresh_paf = 0.05f;
    cfg->num_intermed_pts = 10;
    cfg->downsample = 8;
    cfg->input_size = 224;
    cfg->w_org = 480;
    cfg->h_org = 640;
    cfg->fx = 504.1189880371094;
    cfg->fy = 504.042724609375;
    cfg->cx = 231.7421875;
    cfg->cy = 320.62640380859375;
    cfg->depth_mean = 3.f;
    cfg->depth_std = 2.f;
}

}  // extern "C"
