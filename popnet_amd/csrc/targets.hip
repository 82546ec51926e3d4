// Training targets on the GPU: the multi-person depth compositor and the ground-truth map rasterisers of the reference's
// training dataset, one thread per output cell instead of Python loops over persons x joints x maps.
//   z-buffer compositor   tpm/lib/datasets/datasets_kdh3d_rtpose_mpaug.py:231-266 (CR line endings)
//   get_ground_truth      tpm/lib/datasets/datasets_kdh3d_rtpose_mpaug.py:318-401 (CR)
//     putGaussianMaps     tpm/lib/datasets/heatmap.py:20-36
//     putVecMaps          tpm/lib/datasets/paf.py:18-69
//     putJointZ           tpm/lib/datasets/posemap.py:83-106
// Arithmetic contract (oracle/targets.py, pinned by the reference's own outputs in tests/golden/targets.npz): heat and PAF
// maps in float64 (NumPy's default) rounded to float32 at the end; the z maps in float32 as inside __getitem__ (they inherit
// the dtype of the resized input); Python round() = round-half-even, int() = truncation; persons are visited in annotation
// order (the clamp at 1, the count-weighted PAF average and the first-writer foreground rule depend on it).  No fused
// multiply-add.  HBM-bound and tiny (B x 59 x 28 x 28 floats out): one launch per batch.
#pragma clang fp contract(off)
#include <cmath>
#include "pn_internal.h"

__constant__ int t_limb_a[PN_NUM_LIMBS] = {8, 9, 11, 8, 10, 12, 8, 1, 2, 4, 1, 3, 5, 1};
__constant__ int t_limb_b[PN_NUM_LIMBS] = {9, 11, 13, 10, 12, 14, 1, 2, 4, 6, 3, 5, 7, 0};

template <typename T>
__global__ void compose_depth_kernel(const T *__restrict__ fg_depth, const unsigned char *__restrict__ fg_mask, const int *__restrict__ n_src,
                                     const T *__restrict__ bg, int S, size_t HW, double depth_max, float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= HW) return;
    double image = 2.0 * depth_max, uni = 0.0;
    const int n = min(n_src[b], S);
    for (int s = 0; s < n; ++s) {
        const double m = (double)fg_mask[((size_t)b * S + s) * HW + i];
        if (m > 0.0) image = fmin((double)fg_depth[((size_t)b * S + s) * HW + i] * m, image);
        uni = fmax(uni, m);
    }
    const double v = image * uni + (double)bg[(size_t)b * HW + i] * (1.0 - uni);
    out[(size_t)b * HW + i] = (float)v;
}

// inside the network input?  remove_illegal_joint (:308-316)
__device__ __forceinline__ bool t_inb(const float *kp, int input_x, int input_y) {
    return !(kp[0] >= (float)input_x || kp[0] < 0.f || kp[1] >= (float)input_y || kp[1] < 0.f);
}

__global__ __launch_bounds__(256) void rasterize_kernel(const float *__restrict__ kp2d, const double *__restrict__ kpz, const int *__restrict__ n_persons,
                                                         int Pmax, const float *__restrict__ depth_resize, pn_target_cfg cfg, int gh, int gw,
                                                         float *__restrict__ heat, float *__restrict__ paf, float *__restrict__ zmap, float *__restrict__ fgm) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (cell >= gh * gw) return;
    const int yi = cell / gw, xi = cell - yi * gw;
    const double xs = (double)xi, ys = (double)yi;
    const int P = min(n_persons[b], Pmax);
    const float *kp = kp2d + (size_t)b * Pmax * PN_NUM_JOINTS * 2;
    const double *kz = kpz + (size_t)b * Pmax * PN_NUM_JOINTS;
    const double stride = (double)cfg.stride, start = stride / 2.0 - 0.5;
    const size_t hw = (size_t)gh * gw;

    // ---- confidence maps (putGaussianMaps), background channel = max(1 - max_i, 0) ----
    double hmax = 0.0;
    for (int i = 0; i < PN_NUM_JOINTS; ++i) {
        double acc = 0.0;
        for (int j = 0; j < P; ++j) {
            const float *c = kp + ((size_t)j * PN_NUM_JOINTS + i) * 2;
            if (!t_inb(c, cfg.input_x, cfg.input_y)) continue;
            const double dx = xs * stride + start - (double)c[0], dy = ys * stride + start - (double)c[1];
            const double d2 = dx * dx + dy * dy;
            const double e = d2 / 2.0 / cfg.sigma / cfg.sigma;
            if (e <= 4.6052) acc = acc + exp(-e);
            if (acc > 1.0) acc = 1.0;
        }
        heat[((size_t)b * (PN_NUM_JOINTS + 1) + i) * hw + cell] = (float)acc;
        hmax = i == 0 ? acc : fmax(hmax, acc);
    }
    heat[((size_t)b * (PN_NUM_JOINTS + 1) + PN_NUM_JOINTS) * hw + cell] = (float)fmax(1.0 - hmax, 0.0);

    // ---- part affinity fields (putVecMaps): count-weighted running average over the persons ----
    for (int l = 0; l < PN_NUM_LIMBS; ++l) {
        double vx = 0.0, vy = 0.0, cnt = 0.0;
        for (int j = 0; j < P; ++j) {
            const float *ca = kp + ((size_t)j * PN_NUM_JOINTS + t_limb_a[l]) * 2, *cb = kp + ((size_t)j * PN_NUM_JOINTS + t_limb_b[l]) * 2;
            if (!t_inb(ca, cfg.input_x, cfg.input_y) || !t_inb(cb, cfg.input_x, cfg.input_y)) continue;
            const double ax = (double)ca[0] / stride, ay = (double)ca[1] / stride, bx = (double)cb[0] / stride, by = (double)cb[1] / stride;
            const double lx = bx - ax, ly = by - ay;
            const double norm = sqrt(lx * lx + ly * ly);
            if (norm == 0.0) continue;
            const double ux = lx / norm, uy = ly / norm;
            const int x0 = max((int)rint(fmin(ax, bx) - 1.0), 0), x1 = min((int)rint(fmax(ax, bx) + 1.0), gw - 1);
            const int y0 = max((int)rint(fmin(ay, by) - 1.0), 0), y1 = min((int)rint(fmax(ay, by) + 1.0), gh - 1);
            double wx = 0.0, wy = 0.0;
            if (xi >= x0 && xi <= x1 && yi >= y0 && yi <= y1) {
                const double width = fabs((xs - ax) * uy - (ys - ay) * ux);
                if (width < 1.0) { wx = ux; wy = uy; }
            }
            const bool hit = fabs(wx) > 0.0 || fabs(wy) > 0.0;
            vx = vx * cnt + wx;
            vy = vy * cnt + wy;
            if (hit) cnt += 1.0;
            const double div = cnt == 0.0 ? 1.0 : cnt;
            vx = vx / div;
            vy = vy / div;
        }
        paf[((size_t)b * 2 * PN_NUM_LIMBS + 2 * l) * hw + cell] = (float)vx;
        paf[((size_t)b * 2 * PN_NUM_LIMBS + 2 * l + 1) * hw + cell] = (float)vy;
    }

    // ---- joint depth maps (putJointZ), float32 like the map they update ----
    const float zin = depth_resize[(size_t)b * hw + cell];
    const float dmax = (float)cfg.depth_max;
    for (int k = 0; k < PN_NUM_JOINTS; ++k) {
        float z = 2.f * dmax, fg = 0.f;
        for (int j = 0; j < P; ++j) {
            const float *c = kp + ((size_t)j * PN_NUM_JOINTS + k) * 2;
            if (!t_inb(c, cfg.input_x, cfg.input_y)) continue;
            const double cx = (double)c[0] / stride, cy = (double)c[1] / stride;
            const int x0 = max((int)(cx - cfg.z_radius), 0), x1 = min((int)(cx + cfg.z_radius), gw - 1);
            const int y0 = max((int)(cy - cfg.z_radius), 0), y1 = min((int)(cy + cfg.z_radius), gh - 1);
            const bool win = xi >= x0 && xi <= x1 && yi >= y0 && yi <= y1;
            const float pz = win ? (float)kz[(size_t)j * PN_NUM_JOINTS + k] : dmax;
            float zk = fminf(pz, z);
            const bool fresh = pz < dmax && fg == 0.f;
            if (fresh) { zk = pz; fg = 1.f; }
            z = zk;
        }
        if (fg == 0.f) z = zin;
        if (z < 0.f) z = 0.f;
        if (z > dmax) z = dmax;
        z = z - (float)cfg.depth_mean;
        z = z / (float)cfg.depth_std;
        zmap[((size_t)b * PN_NUM_JOINTS + k) * hw + cell] = z;
        fgm[((size_t)b * PN_NUM_JOINTS + k) * hw + cell] = fg;
    }
}

extern "C" {

void pn_target_cfg_default(pn_target_cfg *cfg) {
    if (!cfg) return;
    cfg->input_x = 224; cfg->input_y = 224; cfg->stride = 8; cfg->z_radius = 2;
    cfg->sigma = 7.0; cfg->depth_max = 6.0; cfg->depth_mean = 3.0; cfg->depth_std = 2.0;
}

int pn_compose_depth(pn_ctx *ctx, const void *fg_depth_dev, const unsigned char *fg_mask_dev, const int *n_src_dev, const void *bg_dev,
                     int depth_dtype, int B, int S, int H, int W, float depth_max, float *out_dev, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!fg_depth_dev || !fg_mask_dev || !n_src_dev || !bg_dev || !out_dev || B < 1 || B > 65535 || S < 1 || H < 1 || W < 1)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_compose_depth: bad arguments");
    const size_t HW = (size_t)H * W;
    dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B), block(256);
    hipStream_t s = (hipStream_t)hip_stream;
    if (depth_dtype == PN_DEPTH_F16)
        hipLaunchKernelGGL(compose_depth_kernel<_Float16>, grid, block, 0, s, (const _Float16 *)fg_depth_dev, fg_mask_dev, n_src_dev, (const _Float16 *)bg_dev, S, HW, (double)depth_max, out_dev);
    else if (depth_dtype == PN_DEPTH_F32)
        hipLaunchKernelGGL(compose_depth_kernel<float>, grid, block, 0, s, (const float *)fg_depth_dev, fg_mask_dev, n_src_dev, (const float *)bg_dev, S, HW, (double)depth_max, out_dev);
    else
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_compose_depth: unknown depth dtype %d", depth_dtype);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

int pn_rasterize_targets(pn_ctx *ctx, const float *kp2d_dev, const double *kp_z_dev, const int *n_persons_dev, int B, int Pmax,
                         const float *depth_resize_dev, const pn_target_cfg *cfg, float *heat_dev, float *paf_dev, float *z_dev,
                         float *fg_dev, void *hip_stream) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!kp2d_dev || !kp_z_dev || !n_persons_dev || !depth_resize_dev || !cfg || !heat_dev || !paf_dev || !z_dev || !fg_dev || B < 1 || B > 65535 || Pmax < 1)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_rasterize_targets: bad arguments");
    if (cfg->stride < 1 || cfg->input_x < cfg->stride || cfg->input_y < cfg->stride)
        return pn_set_error(ctx, PN_ERR_INVALID, "pn_rasterize_targets: bad geometry");
    const int gh = cfg->input_y / cfg->stride, gw = cfg->input_x / cfg->stride;
    dim3 grid((unsigned)((gh * gw + 255) / 256), (unsigned)B), block(256);
    hipLaunchKernelGGL(rasterize_kernel, grid, block, 0, (hipStream_t)hip_stream, kp2d_dev, kp_z_dev, n_persons_dev, Pmax, depth_resize_dev, *cfg, gh, gw,
                       heat_dev, paf_dev, z_dev, fg_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

}  // extern "C"
