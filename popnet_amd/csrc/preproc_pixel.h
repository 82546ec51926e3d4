// One pixel of the pre-processed depth frame: the arithmetic of pn_preprocess (api.hip, where the reference lines are cited),
// shared by preprocess_kernel and by the 7x7 stem when it reads the raw frames itself (pn_rtpose_forward_frames /
// pn_yolo_forward_frames: conv_misc.hip).  OpenCV 4.2 resize.cpp float path: source coordinate in double rounded to float,
// horizontal pass then vertical pass, float32 products summed left to right, NO fused multiply-add -- the pragma below keeps the
// compiler from contracting a * b + c whatever the including file's setting is (results bit-exact against the oracle).
#pragma once

struct PnFrameSrc {
    const void *frames;        // [B, H, W] f16 or f32 metres (device)
    int dtype;                 // pn_depth_dtype
    int H, W;                  // frame size
    double scale_x, scale_y;   // W / S, H / S as cv::resize computes them
    float dmax, mean, stdv;
};

template <typename TIN>
__device__ __forceinline__ float pn_preproc_pixel(const TIN *__restrict__ img, int H, int W, int dy, int dx, double scale_x, double scale_y,
                                                  float dmax, float mean, float stdv) {
#pragma clang fp contract(off)
    float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= W - 1) { sx = W - 1; fx = 0.f; }
    float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int y0 = min(max(sy, 0), H - 1), y1 = min(max(sy + 1, 0), H - 1);
    const float a0 = 1.f - fx, a1 = fx;
    float h0, h1;
    if (sx + 1 >= W) {      // HResizeLinear tail: D = S[sx] * 1
        h0 = (float)img[(size_t)y0 * W + sx];
        h1 = (float)img[(size_t)y1 * W + sx];
    } else {
        h0 = (float)img[(size_t)y0 * W + sx] * a0 + (float)img[(size_t)y0 * W + sx + 1] * a1;
        h1 = (float)img[(size_t)y1 * W + sx] * a0 + (float)img[(size_t)y1 * W + sx + 1] * a1;
    }
    float v = h0 * (1.f - fy) + h1 * fy;
    if (v < 0.f) v = 0.f;
    if (v > dmax) v = dmax;
    return (v - mean) / stdv;
}
