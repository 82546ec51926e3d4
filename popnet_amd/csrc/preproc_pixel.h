// One pixel of the pre-processed depth frame: the arithmetic of pn_preprocess (api.hip, where the reference lines are cited),
// shared by preprocess_kernel and by the 7x7 stem when it reads the raw frames itself (pn_rtpose_forward_frames /
// pn_yolo_forward_frames: conv_misc.hip).  OpenCV 4.2 resize.cpp float path: source coordinate in double rounded to float,
// horizontal pass then vertical pass, float32 products summed left to right, NO fused multiply-add -- the pragmas below keep the
// compiler from contracting a * b + c whatever the including file's setting is (results bit-exact against the oracle).
//
// The arithmetic is split along the axes: pn_preproc_axis_x / _y are the coordinate terms (a function of the column / the row
// alone), pn_preproc_combine the interpolation.  preprocess_kernel evaluates all three per pixel; the stem evaluates the axis
// terms once per tile row / column (38 + 38 instead of 1 444 double-precision coordinate computations) -- the same float
// operations in the same order either way, so the same bits.
#pragma once

struct PnFrameSrc {
    const void *frames;        // [B, H, W] f16 or f32 metres (device)
    int dtype;                 // pn_depth_dtype
    int H, W;                  // frame size
    double scale_x, scale_y;   // W / S, H / S as cv::resize computes them
    float dmax, mean, stdv;
};

struct PnAxisX { int sx; float fx; };                 // source column (clamped) and its weight; fx == 0 at the borders
struct PnAxisY { int y0, y1; float fy; };             // the two source rows (clamped) and the weight of the second

__device__ __forceinline__ PnAxisX pn_preproc_axis_x(int dx, double scale_x, int W) {
#pragma clang fp contract(off)
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= W - 1) { sx = W - 1; fx = 0.f; }
    return PnAxisX{sx, fx};
}

__device__ __forceinline__ PnAxisY pn_preproc_axis_y(int dy, double scale_y, int H) {
#pragma clang fp contract(off)
    float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    return PnAxisY{min(max(sy, 0), H - 1), min(max(sy + 1, 0), H - 1), fy};
}

template <typename TIN>
__device__ __forceinline__ float pn_preproc_combine(const TIN *__restrict__ img, int W, PnAxisX ax, PnAxisY ay, float dmax, float mean, float stdv) {
#pragma clang fp contract(off)
    const float a0 = 1.f - ax.fx, a1 = ax.fx;
    float h0, h1;
    if (ax.sx + 1 >= W) {      // HResizeLinear tail: D = S[sx] * 1
        h0 = (float)img[(size_t)ay.y0 * W + ax.sx];
        h1 = (float)img[(size_t)ay.y1 * W + ax.sx];
    } else {
        h0 = (float)img[(size_t)ay.y0 * W + ax.sx] * a0 + (float)img[(size_t)ay.y0 * W + ax.sx + 1] * a1;
        h1 = (float)img[(size_t)ay.y1 * W + ax.sx] * a0 + (float)img[(size_t)ay.y1 * W + ax.sx + 1] * a1;
    }
    float v = h0 * (1.f - ay.fy) + h1 * ay.fy;
    if (v < 0.f) v = 0.f;
    if (v > dmax) v = dmax;
    return (v - mean) / stdv;
}

template <typename TIN>
__device__ __forceinline__ float pn_preproc_pixel(const TIN *__restrict__ img, int H, int W, int dy, int dx, double scale_x, double scale_y,
                                                  float dmax, float mean, float stdv) {
    return pn_preproc_combine(img, W, pn_preproc_axis_x(dx, scale_x, W), pn_preproc_axis_y(dy, scale_y, H), dmax, mean, stdv);
}
