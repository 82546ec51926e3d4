// Yolo-Pose+ decode, greedy box NMS and skeleton extraction on the GPU -- replaces
// parse_prior_pose (tpm/lib/utils/prior_pose_align.py:10-168, pred_vis False and True), which the reference
// runs as a chain of small in-place torch ops plus a Python suppression loop per image.
//
// One workgroup per frame.  The network map ([A*(5+3J), h, w] f32, 78 KB at 14x14) is read once;
// only cells whose objectness passes the threshold are decoded.  float32 throughout, operations in
// the reference's order (".add_(lin_x).div_(w)" = (v + gx) / w, ...), no fused multiply-add.
// Reference quirks kept:
//   * candidate order is anchor-major, then cell index (:54-77);
//   * keep = column sums of triu(iou > thr, 1); then "for i in 1..n-2: if keep[i] > 0:
//     keep -= conflicting[i]" -- rows 0 and n-1 never subtract (:108-115); survivors: keep == 0;
//   * visibility is the inclusive test 0+m <= x <= w_out-1-m (:160-161).
// Difference: torch.sort(descending) is not stable for equal scores; ties here resolve to candidate
// order (the oracle does the same).  The input map is NOT modified.
#pragma clang fp contract(off)
#include "pn_internal.h"

#define YMAXC 512          // max candidates per frame (A*h*w must not exceed it)
#define YWORDS (YMAXC / 32)

__global__ __launch_bounds__(256) void parse_yolo_kernel(const float *__restrict__ pm, int h, int w, int A, int J,
                                                          float aw0, float ah0, float aw1, float ah1, float aw2, float ah2,
                                                          float w_out, float h_out, float depth_mean, float depth_std,
                                                          float conf_thr, float nms_thr, float vis_margin,
                                                          int glue, float g_in, float g_worg, float g_horg, float g_fx, float g_fy,
                                                          float g_cx, float g_cy, pn_yolo_frame *__restrict__ frames,
                                                          int F, float *__restrict__ vis_pred) {
    __shared__ unsigned short s_cell[YMAXC];            // candidate -> a*hw + cell
    __shared__ float s_score[YMAXC];
    __shared__ float s_x1[YMAXC], s_y1[YMAXC], s_x2[YMAXC], s_y2[YMAXC];   // in sorted order
    __shared__ unsigned short s_sorted[YMAXC];          // sorted position -> candidate
    __shared__ unsigned s_conf[YMAXC][YWORDS];          // conflicting[i][j] bit rows (i < j)
    __shared__ int s_keep[YMAXC];
    __shared__ int s_wave_cnt[4];
    __shared__ int s_total;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = h * w, ncell = A * hw;          // F = 5 + 3 J features per anchor, 5 + 4 J with predicted visibilities (pred_vis)
    const float *map = pm + (size_t)b * A * F * hw;
    pn_yolo_frame &out = frames[b];
    const float fw = (float)w, fh = (float)h;
    if (tid == 0) s_total = 0;
    __syncthreads();

    // ---- 1. ordered compaction of cells with conf > threshold ----
    for (int base = 0; base < ncell; base += 256) {
        const int k = base + tid;
        bool ok = false;
        float conf = 0.f;
        if (k < ncell) {
            const int a = k / hw, cell = k - a * hw;
            conf = map[(size_t)(a * F + 4) * hw + cell];
            ok = conf > conf_thr;
        }
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int before = s_total;
        for (int q = 0; q < wave; ++q) before += s_wave_cnt[q];
        const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
        if (ok && pos < YMAXC) {
            s_cell[pos] = (unsigned short)k;
            s_score[pos] = conf;
        }
        __syncthreads();
        if (tid == 0) s_total += s_wave_cnt[0] + s_wave_cnt[1] + s_wave_cnt[2] + s_wave_cnt[3];
        __syncthreads();
    }
    const int n = min(s_total, YMAXC);
    if (n == 0) {
        if (tid == 0) { out.n_det = 0; out.n_candidates = 0; out.status = 0; out.reserved = 0; }
        return;
    }

    // ---- 2. stable descending rank by score ----
    for (int k = tid; k < n; k += 256) {
        const float sk = s_score[k];
        int rank = 0;
        for (int m = 0; m < n; ++m) {
            const float sm = s_score[m];
            rank += (sm > sk || (sm == sk && m < k)) ? 1 : 0;
        }
        s_sorted[rank] = (unsigned short)k;
    }
    __syncthreads();

    // ---- 3. decode boxes of the sorted candidates ----
    for (int r = tid; r < n; r += 256) {
        const int k = s_cell[s_sorted[r]];
        const int a = k / hw, cell = k - a * hw;
        const float gx = (float)(cell % w), gy = (float)(cell / w);
        const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2), ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
        const float *f = map + (size_t)a * F * hw + cell;
        const float cx = (f[0] + gx) / fw, cy = (f[hw] + gy) / fh;
        const float bw = (f[2 * hw] * aw) / fw, bh = (f[3 * hw] * ah) / fh;
        s_x1[r] = cx - bw / 2.f; s_y1[r] = cy - bh / 2.f;
        s_x2[r] = cx + bw / 2.f; s_y2[r] = cy + bh / 2.f;
    }
    for (int i = tid; i < n * YWORDS; i += 256) s_conf[i / YWORDS][i % YWORDS] = 0u;
    __syncthreads();

    // ---- 4. conflicting = triu(iou > thr, 1); keep = column sums ----
    for (int j = tid; j < n; j += 256) {
        const float x1 = s_x1[j], y1 = s_y1[j], x2 = s_x2[j], y2 = s_y2[j];
        const float area_j = (x2 - x1) * (y2 - y1);
        int cnt = 0;
        for (int i = 0; i < j; ++i) {
            float dx = fminf(s_x2[i], x2) - fmaxf(s_x1[i], x1);
            float dy = fminf(s_y2[i], y2) - fmaxf(s_y1[i], y1);
            dx = dx < 0.f ? 0.f : dx;
            dy = dy < 0.f ? 0.f : dy;
            const float inter = dx * dy;
            const float area_i = (s_x2[i] - s_x1[i]) * (s_y2[i] - s_y1[i]);
            const float uni = (area_i + area_j) - inter;
            const float iou = inter / uni;
            if (iou > nms_thr) {
                atomicOr(&s_conf[i][j >> 5], 1u << (j & 31));
                ++cnt;
            }
        }
        s_keep[j] = cnt;
    }
    __syncthreads();

    // ---- 5. the reference's suppression loop (rows 1 .. n-2), one wave, columns across lanes ----
    if (wave == 0) {
        for (int i = 1; i < n - 1; ++i) {
            const int ki = s_keep[i];           // wave-uniform read after the previous row's writes
            if (ki > 0)
                for (int j = lane; j < n; j += 64)
                    s_keep[j] -= (int)((s_conf[i][j >> 5] >> (j & 31)) & 1u);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // ---- 6. survivors, in sorted order ----
    if (wave == 0) {
        int n_out = 0;
        unsigned status = 0;
        for (int base = 0; base < n; base += 64) {
            const int r = base + lane;
            const bool keep = r < n && s_keep[r] == 0;
            const unsigned long long bal = __ballot(keep);
            const int o = n_out + __popcll(bal & ((1ull << lane) - 1ull));
            if (keep && o < PN_YOLO_MAX_DET) {
                const int k = s_cell[s_sorted[r]];
                const int a = k / hw, cell = k - a * hw;
                const float gx = (float)(cell % w), gy = (float)(cell / w);
                const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2), ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
                const float *f = map + (size_t)a * F * hw + cell;
                float b0 = (f[0] + gx) / fw, b1 = (f[hw] + gy) / fh;
                float b2 = (f[2 * hw] * aw) / fw, b3 = (f[3 * hw] * ah) / fh;
                b0 = b0 * w_out; b2 = b2 * w_out; b1 = b1 * h_out; b3 = b3 * h_out;
                b0 = b0 - b2 / 2.f; b1 = b1 - b3 / 2.f;
                b2 = b2 + b0; b3 = b3 + b1;
                out.bbox[o][0] = b0; out.bbox[o][1] = b1; out.bbox[o][2] = b2; out.bbox[o][3] = b3;
                out.bbox[o][4] = f[4 * hw];
                if (glue) {
                    out.bbox_org[o][0] = b0 / g_in * g_worg; out.bbox_org[o][2] = b2 / g_in * g_worg;
                    out.bbox_org[o][1] = b1 / g_in * g_horg; out.bbox_org[o][3] = b3 / g_in * g_horg;
                }
                const float awh = aw / 2.0f, ahh = ah / 2.0f;
                for (int jn = 0; jn < J; ++jn) {
                    float x = ((f[(size_t)(5 + jn) * hw] * awh + gx) / fw) * w_out;
                    float y = ((f[(size_t)(5 + J + jn) * hw] * ahh + gy) / fh) * h_out;
                    float zz = f[(size_t)(5 + 2 * J + jn) * hw] * depth_std + depth_mean;
                    out.human[o][jn][0] = x; out.human[o][jn][1] = y; out.human[o][jn][2] = zz;
                    if (glue) {      // evaluation_yolo_posenet_kdh3d_mpreal.py:194-195, common.py:107-115, all float32
                        const float x2 = x / g_in * g_worg, y2 = y / g_in * g_horg;
                        out.joints_2d[o][jn][0] = x2; out.joints_2d[o][jn][1] = y2;
                        out.joints_3d[o][jn][0] = (x2 - g_cx) / g_fx * zz;
                        out.joints_3d[o][jn][1] = (y2 - g_cy) / g_fy * zz;
                        out.joints_3d[o][jn][2] = zz;
                    }
                    const bool inside = x >= 0.f + vis_margin && x <= w_out - 1.f - vis_margin &&
                                        y >= 0.f + vis_margin && y <= h_out - 1.f - vis_margin;
                    out.visibility[o][jn] = inside ? 1 : 0;
                    // pred_vis (:153-157): the in-bounds test TIMES the network's visibility channel (numpy bool * float32)
                    if (vis_pred) vis_pred[((size_t)b * PN_YOLO_MAX_DET + o) * J + jn] = (inside ? 1.f : 0.f) * f[(size_t)(5 + 3 * J + jn) * hw];
                }
            }
            n_out += __popcll(bal);
        }
        if (n_out > PN_YOLO_MAX_DET) status |= 1u;
        if (lane == 0) {
            out.n_det = min(n_out, PN_YOLO_MAX_DET);
            out.n_candidates = s_total;
            out.status = status | (s_total > YMAXC ? 2u : 0u);
            out.reserved = 0;
        }
    }
}

static int parse_yolo_impl(pn_ctx *ctx, const float *posemaps_dev, int B, int h, int w, const float *anchors_wh,
                           int num_anchors, int num_joints, int w_out, int h_out, float depth_mean, float depth_std,
                           float conf_threshold, float nms_threshold, int vis_margin, const pn_parse_cfg *glue,
                           pn_yolo_frame *frames_dev, float *vis_pred_dev, void *hip_stream, const char *who) {
    if (!ctx) return PN_ERR_INVALID;
    if (ctx->device < 0) return pn_set_error(ctx, PN_ERR_STATE, "context has no device");
    if (!posemaps_dev || !anchors_wh || !frames_dev || B < 1) return pn_set_error(ctx, PN_ERR_INVALID, "%s: bad arguments", who);
    if (num_anchors < 1 || num_anchors > 3) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "%s: 1..3 anchors supported", who);
    if (num_joints != PN_NUM_JOINTS) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "%s: built for %d joints", who, PN_NUM_JOINTS);
    if (num_anchors * h * w > YMAXC) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "%s: %d cells exceed %d", who, num_anchors * h * w, YMAXC);
    float a[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 2 * num_anchors; ++i) a[i] = anchors_wh[i];
    PN_HIP_CHECK(ctx, hipMemsetAsync(frames_dev, 0, (size_t)B * sizeof(pn_yolo_frame), (hipStream_t)hip_stream));   // unused rows read as zero
    if (vis_pred_dev) PN_HIP_CHECK(ctx, hipMemsetAsync(vis_pred_dev, 0, (size_t)B * PN_YOLO_MAX_DET * PN_NUM_JOINTS * sizeof(float), (hipStream_t)hip_stream));
    hipLaunchKernelGGL(parse_yolo_kernel, dim3(B), dim3(256), 0, (hipStream_t)hip_stream, posemaps_dev, h, w, num_anchors,
                       num_joints, a[0], a[1], a[2], a[3], a[4], a[5], (float)w_out, (float)h_out, depth_mean, depth_std,
                       conf_threshold, nms_threshold, (float)vis_margin, glue ? 1 : 0, glue ? (float)glue->input_size : 1.f,
                       glue ? (float)glue->w_org : 1.f, glue ? (float)glue->h_org : 1.f, glue ? (float)glue->fx : 1.f,
                       glue ? (float)glue->fy : 1.f, glue ? (float)glue->cx : 0.f, glue ? (float)glue->cy : 0.f, frames_dev,
                       vis_pred_dev ? 5 + 4 * num_joints : 5 + 3 * num_joints, vis_pred_dev);
    PN_HIP_CHECK(ctx, hipGetLastError());
    return PN_OK;
}

extern "C" int pn_parse_yolo(pn_ctx *ctx, const float *posemaps_dev, int B, int h, int w, const float *anchors_wh,
                             int num_anchors, int num_joints, int w_out, int h_out, float depth_mean, float depth_std,
                             float conf_threshold, float nms_threshold, int vis_margin, const pn_parse_cfg *glue,
                             pn_yolo_frame *frames_dev, void *hip_stream) {
    return parse_yolo_impl(ctx, posemaps_dev, B, h, w, anchors_wh, num_anchors, num_joints, w_out, h_out, depth_mean, depth_std, conf_threshold,
                           nms_threshold, vis_margin, glue, frames_dev, nullptr, hip_stream, "pn_parse_yolo");
}

extern "C" int pn_parse_yolo_predvis(pn_ctx *ctx, const float *posemaps_dev, int B, int h, int w, const float *anchors_wh,
                                     int num_anchors, int num_joints, int w_out, int h_out, float depth_mean, float depth_std,
                                     float conf_threshold, float nms_threshold, int vis_margin, const pn_parse_cfg *glue,
                                     pn_yolo_frame *frames_dev, float *vis_pred_dev, void *hip_stream) {
    if (!vis_pred_dev) return pn_set_error(ctx, PN_ERR_INVALID, "pn_parse_yolo_predvis: vis_pred_dev is required");
    return parse_yolo_impl(ctx, posemaps_dev, B, h, w, anchors_wh, num_anchors, num_joints, w_out, h_out, depth_mean, depth_std, conf_threshold,
                           nms_threshold, vis_margin, glue, frames_dev, vis_pred_dev, hip_stream, "pn_parse_yolo_predvis");
}

extern "C" size_t pn_sizeof_pose_frame(void) { return sizeof(pn_pose_frame); }
extern "C" size_t pn_sizeof_yolo_frame(void) { return sizeof(pn_yolo_frame); }
