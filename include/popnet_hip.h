/*
 * popnet_hip.h -- C ABI of libpopnet_hip.so, the MI355X (gfx950) implementation of the
 * PoP-Net / MP-3DHP inference hot path:
 *
 *     depth frame -> resize/clamp/normalise -> CNN forward -> pose parsing -> 2D/3D joints
 *
 * Plain C types only (pointers, sizes, a HIP stream passed as void*): no torch, no C++ types.
 * Device pointers are raw HIP allocations owned by the CALLER; the library owns only the
 * workspace inside a pn_ctx / pn_net.  Every call returns 0 on success or a negative
 * pn_status; the message is retrievable with pn_last_error().  Nothing aborts, nothing is
 * global except the seven legacy `pafprocess` symbols at the end of this file.
 *
 * Each entry point cites the reference interface (file:line under /root/reference) it
 * replaces.  "tpm/" = third_party_methods/.
 */
#ifndef POPNET_HIP_H
#define POPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN_ABI_VERSION 1

/* ---- status codes ------------------------------------------------------------------------ */
enum pn_status {
    PN_OK = 0,
    PN_ERR_INVALID = -1,     /* bad argument / unknown tensor name / shape mismatch */
    PN_ERR_HIP = -2,         /* a HIP runtime call failed (message has the hipError string) */
    PN_ERR_STATE = -3,       /* call order violated (e.g. forward before finalize) */
    PN_ERR_UNSUPPORTED = -4  /* shape outside what the kernels are built for */
};

/* ---- compute / storage precision of the conv stack ---------------------------------------- */
enum pn_precision {
    PN_PREC_F32 = 0,   /* fp32 storage, fp32-input MFMA (v_mfma_f32_16x16x4_f32): parity mode */
    PN_PREC_BF16 = 1,  /* bf16 storage, bf16 MFMA (v_mfma_f32_16x16x32_bf16), fp32 accumulate: throughput mode */
    PN_PREC_BF16X3 = 2 /* split-bf16 ("3 x bf16"): every tensor and weight is kept as hi + lo bf16
                          parts (16 significant bits), a product is x_hi w_hi + x_lo w_hi + x_hi w_lo on the bf16 matrix
                          cores with fp32 accumulation; fp32-class results (north_star's 1e-3 m) at ~1/3 of the bf16 rate */
};

enum pn_net_kind {
    PN_NET_RTPOSE_LIGHT3D = 0,  /* "Open-Pose+"  tpm/lib/network/rtpose_light3d.py:249-356 */
    PN_NET_YOLO_POSENET = 1     /* "Yolo-Pose+"  tpm/lib/network/yolo_posenet.py:87-158   */
};

enum pn_depth_dtype { PN_DEPTH_F16 = 0, PN_DEPTH_F32 = 1 };

typedef struct pn_ctx pn_ctx;
typedef struct pn_net pn_net;

/* ---- context -------------------------------------------------------------------------------
 * One context per GPU / stream owner.  Not thread-safe per context; independent contexts are. */
int pn_abi_version(void);
/* 1 when the library was built with -DPN_EXPERIMENTS (lab builds: the timing-only ablation and mixed-precision environment
 * switches POPNET_ABLATE_SKIP / POPNET_X3_BF16_CONVS are compiled in and can change results), 0 for the shipped library, which
 * honours no result-changing environment variable.  bench.py refuses to time a lab build. */
int pn_build_experiments(void);
/* Measurement aid (bench.py `roofline.peak_sustained_tflops`; no reference counterpart -- the reference has no device code): runs a loop of
 * nothing but v_mfma_f32_16x16x32_bf16 on random bf16 operands (`waves_per_simd` waves on every SIMD of every CU, no memory traffic)
 * back to back for `seconds` and reports what the LAST batch of launches sustained, i.e. the matrix-core ceiling of THIS box at the
 * clock its power limit allows, next to the 2.5 PFLOP/s spec figure; *in_kernel_ghz (may be NULL) = median shader clock inside the
 * kernel (s_memtime / s_memrealtime).  Synchronises the stream. */
int pn_mfma_sustained(pn_ctx *ctx, double seconds, int waves_per_simd, double *tflops, double *in_kernel_ghz, void *hip_stream);
pn_ctx *pn_create(int device_id);
void pn_destroy(pn_ctx *ctx);
/* Copies the last error message of this context into buf (NUL-terminated). Returns its length. */
int pn_last_error(pn_ctx *ctx, char *buf, size_t buf_len);

/* ---- pre-processing -------------------------------------------------------------------------
 * Replaces test-mode KDH3D_Keypoints.__getitem__ image path:
 *   tpm/lib/datasets/datasets_kdh3d_rtpose_mpreal.py:225-246 (CR line endings)
 *   tpm/lib/datasets/data_augmentation_2d3d.py:76-89 (Cvt2ndarray), :507-522 (Resize, cv2 INTER_LINEAR)
 * depth_dev: [B, H, W] f16 or f32 metres (device).  out_dev: [B, 1, S, S] f32 (device),
 * = (clip(bilinear(depth), 0, depth_max) - depth_mean) / depth_std.                            */
int pn_preprocess(pn_ctx *ctx, const void *depth_dev, int depth_dtype, int B, int H, int W,
                  float *out_dev, int S, float depth_max, float depth_mean, float depth_std,
                  void *hip_stream);

/* ---- networks -------------------------------------------------------------------------------
 * pn_net_create + pn_net_set_tensor + pn_net_finalize replace
 *   model = rtpose_light3d(num_parts, num_limbs, num_stages, input_dim); model.load_state_dict(sd)
 *   (tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:135-144)
 * and the YoloPoseNet twin (tpm/evaluate/evaluation_yolo_posenet_kdh3d_mpreal.py:119-128).
 * Tensor names are the reference state_dict keys, without "module.".
 * `a` = num_limbs for rtpose_light3d, number of anchors for YoloPoseNet.  input_dim = channels of the input batch [B, input_dim, H, W]
 * (1 = depth, the path north_star names: fused matrix-core stem, frames-in forward; 2..16, e.g. the reference constructors' default 3
 * (rtpose_light3d.py:250, yolo_posenet.py:88): the 7x7 stem on the generic fp32 convolution, pn_*_forward only).                          */
pn_net *pn_net_create(pn_ctx *ctx, int kind, int num_parts, int a, int input_dim);
void pn_net_destroy(pn_net *net);
/* Host fp32 data, copied.  Unknown names are rejected, except keys the reference builds but
 * never executes (YoloPoseNet model0.layer3.*, *.num_batches_tracked), which are accepted and
 * ignored. */
int pn_net_set_tensor(pn_net *net, const char *name, const float *host_data,
                      const int64_t *shape, int ndim);
/* Folds BatchNorm (eval mode, eps 1e-5) into the convolutions, packs weights into MFMA
 * fragment order, uploads, allocates NHWC activation workspace for `max_batch` frames of
 * in_h x in_w.  Must be called once after all tensors are set. */
int pn_net_finalize(pn_net *net, int precision, int max_batch, int in_h, int in_w);

/* rtpose_light3d.forward (tpm/lib/network/rtpose_light3d.py:326-356), stage-2 outputs:
 *   x_dev [B,1,in_h,in_w] f32 NCHW -> paf [B,2L,h,w], heat [B,J+1,h,w], z [B,L+1,h,w]
 *   f32 NCHW device buffers (h = in_h/8).  Sigmoid range casts applied.                        */
int pn_rtpose_forward(pn_net *net, const float *x_dev, int B, float *paf_dev, float *heat_dev,
                      float *z_dev, void *hip_stream);
/* YoloPoseNet.forward (tpm/lib/network/yolo_posenet.py:131-158):
 *   x_dev [B,1,in_h,in_w] -> out [B, A*(5+3J), in_h/16, in_w/16] f32 NCHW, slice casts applied. */
int pn_yolo_forward(pn_net *net, const float *x_dev, int B, float *out_dev, void *hip_stream);
/* Frames in (round 3): pn_preprocess + pn_rtpose_forward / pn_yolo_forward as ONE call on the raw depth frames -- the image half of
 * test-mode KDH3D_Keypoints.__getitem__ (tpm/lib/datasets/datasets_kdh3d_rtpose_mpreal.py:225-246 (CR), data_augmentation_2d3d.py:
 * 76-89,507-522) followed by model(img) (tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:171-178 /
 * evaluation_yolo_posenet_kdh3d_mpreal.py:160-166).  The 7x7 stem computes its input tile with pn_preprocess's arithmetic, so the
 * maps are bit-identical to the two-call form; the pre-processed [B,1,S,S] tensor is never written.  depth_dev [B,H,W] f16 / f32
 * metres (device), resized to the square S = in_h = in_w the net was finalized for.  bf16 and bf16x3 nets (the fp32 parity mode
 * keeps the two calls: PN_ERR_UNSUPPORTED).                                                                                       */
int pn_rtpose_forward_frames(pn_net *net, const void *depth_dev, int depth_dtype, int B, int H, int W, float depth_max,
                             float depth_mean, float depth_std, float *paf_dev, float *heat_dev, float *z_dev,
                             void *hip_stream);
int pn_yolo_forward_frames(pn_net *net, const void *depth_dev, int depth_dtype, int B, int H, int W, float depth_max,
                           float depth_mean, float depth_std, float *out_dev, void *hip_stream);
/* Test/diagnostic: copies a named internal activation of the last forward to the host as
 * f32 NCHW.  Names: "feat" (stem output), "paf1" "heat1" "z1" (stage-1 outputs) for
 * rtpose_light3d; "feat" (layer2 output) for YoloPoseNet.  Synchronises the stream.           */
int pn_net_read_activation(pn_net *net, const char *name, int B, float *host_out,
                           size_t host_elems, void *hip_stream);
/* Same, device to device: writes f32 NCHW into dev_out on the stream, no synchronisation.  This is
 * how the Python module materialises forward()'s `saved_for_loss` stage-1 entries
 * (tpm/lib/network/rtpose_light3d.py:340-342).                                                   */
int pn_net_copy_activation(pn_net *net, const char *name, int B, float *dev_out, void *hip_stream);
/* Algorithmic FLOPs (2*MAC, convolutions only) of one frame through the finalized net. */
double pn_net_flops_per_frame(pn_net *net);
/* Freezes the launch descriptors at the batch size / output pointers of the last forward: afterwards a forward with
 * another batch size or other output buffers returns PN_ERR_STATE instead of rewriting descriptors that a captured
 * hipGraph reads at replay time.  pn_net_lock(net, 0) lifts the freeze.                                               */
int pn_net_lock(pn_net *net, int locked);
/* Measurement hooks: between begin and end every kernel launch of pn_*_forward is bracketed by
 * HIP events recorded on the caller's stream.  end() waits for them and returns the summed
 * durations: conv_* = the MFMA convolution launches (with the algorithmic FLOPs they covered),
 * other_* = stem + pooling launches.  Event pairs are recycled; nothing is allocated per call once
 * the first profiled forward has run. */
int pn_net_profile_begin(pn_net *net);
int pn_net_profile_end(pn_net *net, double *conv_ms, int64_t *conv_launches, double *conv_flops,
                       double *other_ms, int64_t *other_launches);
/* After pn_net_profile_end: the convolution kernel instantiation with the rank-th largest summed
 * duration (rank 0 = the dominant kernel), its launches and the algorithmic FLOPs they covered.
 * PN_ERR_INVALID when rank is past the last instantiation. */
int pn_net_profile_kernel(pn_net *net, int rank, char *name, size_t name_cap, double *ms,
                          int64_t *launches, double *flops);

/* ---- Open-Pose+ parsing ---------------------------------------------------------------------
 * Replaces, per frame, paf_to_pose + paf_to_human_list + the depth read-out / rescale /
 * back-projection glue:
 *   tpm/lib/utils/paf_to_pose.py:33-377, tpm/lib/utils/common.py:5-32,272-293,
 *   tpm/evaluate/evaluation_rtpose_light3d_kdh3d_mpreal_ablation.py:179-262.
 * Limits are compile-time; exceeding one sets PN_FRAME_OVERFLOW in the frame's status and
 * truncates deterministically (first-come in reference order).                                  */
#define PN_NUM_JOINTS 15
#define PN_NUM_LIMBS 14
#define PN_MAX_PEAKS_PER_JOINT 32
#define PN_MAX_PEAKS (PN_NUM_JOINTS * PN_MAX_PEAKS_PER_JOINT)
#define PN_MAX_CONN_PER_LIMB PN_MAX_PEAKS_PER_JOINT
#define PN_MAX_PERSONS 32

#define PN_FRAME_OVERFLOW_PEAKS 1u
#define PN_FRAME_OVERFLOW_PERSONS 2u

typedef struct pn_parse_cfg {
    float thresh_heatmap;      /* cfg.TEST.THRESH_HEATMAP  = 0.1   tpm/lib/config/default.py:126 */
    float thresh_paf;          /* cfg.TEST.THRESH_PAF      = 0.05  :127 */
    int   num_intermed_pts;    /* must be 10               :128 */
    int   downsample;          /* must be 8  cfg.MODEL.DOWNSAMPLE :41 */
    int   input_size;          /* 224: network input side, divisor of the rescale */
    int   w_org, h_org;        /* original frame size for the rescale (480, 640) */
    double fx, fy, cx, cy;     /* pinhole intrinsics  util/util_functions.py:4 */
    float depth_mean, depth_std; /* 3, 2   util/util_functions.py:11-12 */
} pn_parse_cfg;

/* One frame of results.  Doubles where the reference keeps float64 (everything after NMS). */
typedef struct pn_pose_frame {
    int32_t  n_persons;
    int32_t  n_peaks;                                   /* rows of joint_list */
    uint32_t status;                                    /* PN_FRAME_OVERFLOW_* bits */
    int32_t  reserved;
    /* joint_list (paf_to_pose return value 0): x, y, score, id, type */
    float    peak_x[PN_MAX_PEAKS];                      /* integer-valued pixel coords, 224 frame */
    float    peak_y[PN_MAX_PEAKS];
    float    peak_score[PN_MAX_PEAKS];
    int32_t  peak_type[PN_MAX_PEAKS];
    /* person_to_joint_assoc (return value 1): J joint ids (-1 = missing), score, count */
    int32_t  person_joint[PN_MAX_PERSONS][PN_NUM_JOINTS];
    double   person_score[PN_MAX_PERSONS];
    int32_t  person_count[PN_MAX_PERSONS];
    /* glue outputs, already rescaled to the original frame / back-projected */
    double   joints_2d[PN_MAX_PERSONS][PN_NUM_JOINTS][2];   /* human_pred_set_2d entry   */
    double   joints_3d[PN_MAX_PERSONS][PN_NUM_JOINTS][3];   /* human_pred_set_3d entry   */
    double   part_conf[PN_MAX_PERSONS][PN_NUM_JOINTS];      /* human_pred_set_part_conf  */
} pn_pose_frame;

void pn_parse_cfg_default(pn_parse_cfg *cfg);
/* heat [B,J+1,h,w], paf [B,2L,h,w], z [B,L+1,h,w]: f32 NCHW device buffers (z normalised, as
 * the network emits it).  frames_dev: device array of B pn_pose_frame.                          */
int pn_parse_paf(pn_ctx *ctx, const float *heat_dev, const float *paf_dev, const float *z_dev,
                 int B, int h, int w, const pn_parse_cfg *cfg, pn_pose_frame *frames_dev,
                 void *hip_stream);
/* Sizes the context's parse scratch for batches of up to max_batch frames, once, so that pn_parse_paf never
 * allocates on the launch path (a captured hipGraph keeps the scratch pointer: growing it later would free memory
 * the graph still uses).  pn_parse_paf returns PN_ERR_STATE instead of growing the scratch while its stream is being
 * captured, and after pn_parse_reserve has fixed the size.                                                          */
int pn_parse_reserve(pn_ctx *ctx, int max_batch);

/* The same parse WITHOUT the record capacities (the reference has none): one frame (heat [J+1,h,w], paf [2L,h,w], z [L+1,h,w]
 * device maps), every list in a workspace sized from the frame's own counts (up to h*w peaks per joint map).  The second pass
 * for a frame whose fixed-size record carries PN_FRAME_OVERFLOW_*: same arithmetic, same order, variable-length result.
 * Synchronises the stream; *n_peaks / *n_persons size the arrays pn_parse_paf_unbounded_fetch copies out (any of them may be
 * NULL): peaks_xys [n_peaks][3] (x, y, score; id = row, as paf_to_pose's joint_list), peak_type [n_peaks], person_joint
 * [n_persons][J], person_score / person_count [n_persons], joints_2d [n][J][2], joints_3d [n][J][3], part_conf [n][J].
 * The result lives in the context until the next pn_parse_paf_unbounded call (not re-entrant per context). */
int pn_parse_paf_unbounded(pn_ctx *ctx, const float *heat_dev, const float *paf_dev, const float *z_dev, int h, int w,
                           const pn_parse_cfg *cfg, int *n_peaks, int *n_persons, void *hip_stream);
int pn_parse_paf_unbounded_fetch(pn_ctx *ctx, float *peaks_xys, int *peak_type, int *person_joint, double *person_score,
                                 int *person_count, double *joints_2d, double *joints_3d, double *part_conf);

/* NMS(heatmaps, upsampFactor, bool_refine_center=True, config) (tpm/lib/utils/paf_to_pose.py:75-153) alone, on n_maps maps
 * [n_maps, h, w] f32 of one frame (any topology: `paf_to_pose_cpp`, paf_to_pose.py:381-385, runs it on the 18 COCO parts before
 * `process_paf`).  No capacity: count_dev [n_maps] int32 receives every map's peak count, peak_x / peak_y / peak_score_dev
 * [n_maps][h*w] f32 the peaks of map m at [m][0 .. count[m]) in the reference's order (row-major cells): refined x, y in
 * up-sampled pixels and the x8 bicubic score.  upsample must be 8.  Asynchronous on hip_stream.                              */
int pn_nms_peaks(pn_ctx *ctx, const float *heat_dev, int n_maps, int h, int w, float thresh, int upsample, int *count_dev,
                 float *peak_x_dev, float *peak_y_dev, float *peak_score_dev, void *hip_stream);

/* retrieve_depth_heat_weighted(center, depthmap, heatmap, radius) (tpm/lib/utils/common.py:272-293)
 * for n centres (x, y int32 pairs) on one [h, w] f32 map pair; like the reference it first clamps
 * negative heat values in place.  out_dev: n float32 (the reference's np.sum/np.sum is float32).   */
int pn_retrieve_depth(pn_ctx *ctx, const float *depthmap_dev, float *heatmap_dev, int h, int w,
                      const int *centers_xy_dev, int n, int radius, float *out_dev, void *hip_stream);
size_t pn_sizeof_pose_frame(void);

/* Compact wire form of pn_pose_frame for the multi-GPU gather ("[frames, P_max, 15, 6] float32 +
 * count"): what crosses xGMI / PCIe when only the per-person results are needed (6.2 KB instead of 33 KB per
 * frame).  vals = (x, y) in the original frame, (X, Y, Z) metres, part confidence, rounded to float32 from the
 * float64 record; person_joint = the peak ids of person_to_joint_assoc (-1 = missing), so assignments stay exact.
 * Frames with more than PN_WIRE_MAX_PERSONS persons set PN_FRAME_OVERFLOW_PERSONS in status and keep the first
 * PN_WIRE_MAX_PERSONS rows (n_persons still holds the true count). */
#define PN_WIRE_MAX_PERSONS 16
typedef struct pn_pose_wire {
    int32_t  n_persons;
    uint32_t status;
    int16_t  person_joint[PN_WIRE_MAX_PERSONS][PN_NUM_JOINTS];
    float    vals[PN_WIRE_MAX_PERSONS][PN_NUM_JOINTS][6];
} pn_pose_wire;
int pn_pack_pose_frames(pn_ctx *ctx, const pn_pose_frame *frames_dev, int B, pn_pose_wire *wire_dev, void *hip_stream);
/* pn_parse_paf that also writes the compact records (wire_dev: B pn_pose_wire, may be NULL) in the same pass of the
 * read-out kernel -- identical bytes to pn_parse_paf followed by pn_pack_pose_frames, one launch fewer.               */
int pn_parse_paf_wire(pn_ctx *ctx, const float *heat_dev, const float *paf_dev, const float *z_dev,
                      int B, int h, int w, const pn_parse_cfg *cfg, pn_pose_frame *frames_dev,
                      pn_pose_wire *wire_dev, void *hip_stream);
size_t pn_sizeof_pose_wire(void);

/* ---- training targets ----------------------------------------------------------
 * The CPU data-loader work of the reference's training dataset, as device kernels:
 *   pn_compose_depth      the z-buffer multi-person compositor of KDH3D_Keypoints.__getitem__
 *                         (tpm/lib/datasets/datasets_kdh3d_rtpose_mpaug.py:231-266, CR line endings): per output frame up to S
 *                         source frames fg_depth [B,S,H,W] (f16 / f32 metres) with foreground masks fg_mask [B,S,H,W]
 *                         (uint8 0 / 1), the first n_src[b] of them used, pasted over bg [B,H,W]; out [B,H,W] f32.
 *   pn_rasterize_targets  get_ground_truth (:318-401) = putGaussianMaps (heatmap.py:20-36) + putVecMaps (paf.py:18-69) +
 *                         putJointZ (posemap.py:83-106): kp2d [B,Pmax,15,2] f32 joint positions in network-input pixels
 *                         (as Resize leaves them), kp_z [B,Pmax,15] f64 joint depths (metres), n_persons [B],
 *                         depth_resize [B,h,w] f32 (the clamped input at stride resolution) ->
 *                         heat [B,16,h,w], paf [B,28,h,w], z [B,15,h,w] (normalised), fg [B,15,h,w]  f32 NCHW, h = input / stride. */
typedef struct pn_target_cfg {
    int input_x, input_y;      /* network input size (224) */
    int stride;                /* 8 */
    int z_radius;              /* 2   train_rtpose_light3d_kdh3d_mpaug.py default */
    double sigma;              /* 7.0 pixels  datasets_kdh3d_rtpose_mpaug.py:357 */
    double depth_max, depth_mean, depth_std;   /* 6, 3, 2 */
} pn_target_cfg;
void pn_target_cfg_default(pn_target_cfg *cfg);
int pn_compose_depth(pn_ctx *ctx, const void *fg_depth_dev, const unsigned char *fg_mask_dev, const int *n_src_dev,
                     const void *bg_dev, int depth_dtype, int B, int S, int H, int W, float depth_max, float *out_dev,
                     void *hip_stream);
int pn_rasterize_targets(pn_ctx *ctx, const float *kp2d_dev, const double *kp_z_dev, const int *n_persons_dev, int B, int Pmax,
                         const float *depth_resize_dev, const pn_target_cfg *cfg, float *heat_dev, float *paf_dev,
                         float *z_dev, float *fg_dev, void *hip_stream);

/* ---- training step primitives (BASELINE configs[4]) ------------------------------
 * fp32, NCHW, contiguous -- the layout of the reference's own tensors, so every intermediate can be laid next to the
 * reference module's.  One entry per differentiable primitive of rtpose_light3d in train mode; popnet_amd/train.py
 * strings them together (forward, rtpose_light3d_loss_fgweight, backward, Nesterov SGD).  All asynchronous on the stream.
 * They share ONE scratch buffer inside the pn_ctx (rotated weights, split-reduction partials), reused in stream order: drive a
 * pn_ctx's training primitives from one stream at a time (TrainEngine owns a private pn_ctx for that reason).
 *   pn_conv2d_forward   nn.Conv2d (tpm/lib/network/rtpose_light3d.py:31-40,144,232-246): kernel 1 / 3 / 7, any stride and
 *                       padding; accumulate != 0 adds into y.  w [Cout, Cin, k, k], bias [Cout] or NULL.
 *   pn_conv2d_dgrad     its input gradient (stride 1): dx [N,Cin,H,W] (+)= conv_transpose(dy [N,Cout,Ho,Wo], w)
 *   pn_conv2d_wgrad     its weight (and bias, if dbias != NULL) gradient
 *   pn_bn_train_forward nn.BatchNorm2d in train mode: batch statistics (biased variance, eps), running statistics updated
 *                       with `momentum` (unbiased variance) when given, y = act(bn(x) + res); act = 0 none, 1 ReLU,
 *                       2 LeakyReLU(0.1); res (BasicBlock identity, rtpose_light3d.py:66-67) or NULL
 *   pn_bn_train_backward  gradient of the above: dy is the gradient w.r.t. y, out = y (for the activation mask) -- or NULL when the
 *                       forward had no residual input: the mask is then recomputed from x with the forward's own expression
 *                       (same float operations, same result; one tensor less to read, beta required);
 *                       dres (or NULL) receives / accumulates the gradient of the residual input
 *   pn_avgpool3s2_*     nn.AvgPool2d(3, 2, 1) (rtpose_light3d.py:152,158), planes = N * C
 *   pn_head_forward     s = sigmoid(v); out = kind ? (s - 0.5) * 4 : s (rtpose_light3d.py:335-337) written with an image
 *                       stride of out_ld channels (a slice of the stage-2 input, :339); *loss = mean(w (out - target)^2),
 *                       w = 0.1 + 0.9 fg when fg != NULL (rtpose_light3d_loss_fgweight, losses.py:65-90)
 *   pn_head_backward    dv = (d loss / d out + dextra) * d out / d v; dextra [N, dextra_ld, HW] slice or NULL
 *   pn_slice_copy       torch.cat / its gradient: copy (or add) a [N, C, HW] tensor between channel slices
 *   pn_sgd_nesterov     torch.optim.SGD(momentum, nesterov=True) on a flat buffer (train_rtpose_light3d_kdh3d_mpaug.py:313-316);
 *                       grad_scale multiplies the gradient first (1 / world size after the all-reduce)
 *   pn_train_set_precision  PN_PREC_F32 (default: every product on v_mfma_f32_16x16x4_f32, exact fp32 FMA chains) or
 *                       PN_PREC_BF16X3: the 3x3 forward / data-gradient convolutions split every operand into two bf16
 *                       (16 mantissa bits) and run three v_mfma_f32_16x16x32_bf16 per product block -- fp32 tensors in and
 *                       out, results within ~1e-5 relative of the fp32 kernels, 5.3x less matrix-pipe time */
int pn_train_set_precision(pn_ctx *ctx, int precision);
/* keep != 0: a hipGraph captured from the training primitives points into the context's scratch; from now on a call that
 * needs a larger scratch retires the old block (freed by pn_destroy) instead of freeing it, so the graph stays replayable. */
int pn_train_ws_keep(pn_ctx *ctx, int keep);
/* Weight-pack cache of the 3x3 training convolutions (round 4).  Off (default): pn_conv2d_forward / _dgrad re-pack their weights
 * (transpose / rotate / split) into the context's scratch on every call.  On: the packs live in persistent buffers keyed by (weight
 * pointer, shape, rotation, precision); pn_train_pack_refresh re-packs ALL of them in one launch (call it once per step, before the
 * first convolution: popnet_amd.train.TrainEngine does), pn_sgd_nesterov marks them stale, a stale or unknown pack is rebuilt by the
 * call that needs it.  Weights changed by anything but pn_sgd_nesterov need a refresh before the next convolution.  Replaces the
 * 62 pack launches per step of train_rtpose_light3d_kdh3d_mpaug.py:160-180's forward / backward (tpm/lib/network/rtpose_light3d.py). */
int pn_train_pack_cache(pn_ctx *ctx, int enable);
int pn_train_pack_refresh(pn_ctx *ctx, void *hip_stream);
int pn_conv2d_forward(pn_ctx *ctx, const float *x_dev, const float *w_dev, const float *bias_dev, float *y_dev, int N, int Cin,
                      int H, int W, int Cout, int ks, int stride, int pad, int accumulate, void *hip_stream);
int pn_conv2d_dgrad(pn_ctx *ctx, const float *dy_dev, const float *w_dev, float *dx_dev, int N, int Cin, int H, int W, int Cout,
                    int ks, int pad, int accumulate, void *hip_stream);
int pn_conv2d_wgrad(pn_ctx *ctx, const float *x_dev, const float *dy_dev, float *dw_dev, float *dbias_dev, int N, int Cin, int H,
                    int W, int Cout, int ks, int stride, int pad, void *hip_stream);
int pn_bn_train_forward(pn_ctx *ctx, const float *x_dev, const float *gamma_dev, const float *beta_dev, const float *res_dev,
                        float *y_dev, float *save_mean_dev, float *save_invstd_dev, float *running_mean_dev,
                        float *running_var_dev, float momentum, float eps, int act, int N, int C, int HW, void *hip_stream);
int pn_bn_train_backward(pn_ctx *ctx, const float *x_dev, const float *dy_dev, const float *out_dev, const float *gamma_dev,
                         const float *beta_dev, const float *save_mean_dev, const float *save_invstd_dev, int act, int N, int C, int HW, float *dx_dev,
                         float *dgamma_dev, float *dbeta_dev, float *dres_dev, int dres_accumulate, void *hip_stream);
int pn_avgpool3s2_forward(pn_ctx *ctx, const float *x_dev, float *y_dev, int planes, int H, int W, void *hip_stream);
int pn_avgpool3s2_backward(pn_ctx *ctx, const float *dy_dev, float *dx_dev, int planes, int H, int W, void *hip_stream);
int pn_head_forward(pn_ctx *ctx, const float *v_dev, const float *target_dev, const float *fg_dev, int kind, int N, int C, int HW,
                    float *s_dev, float *out_dev, int out_ld, float *loss_dev, void *hip_stream);
int pn_head_backward(pn_ctx *ctx, const float *s_dev, const float *target_dev, const float *fg_dev, const float *dextra_dev,
                     int dextra_ld, int kind, int N, int C, int HW, float *dv_dev, void *hip_stream);
int pn_slice_copy(pn_ctx *ctx, const float *src_dev, int src_ld, float *dst_dev, int dst_ld, int N, int C, int HW, int accumulate,
                  void *hip_stream);
int pn_sgd_nesterov(pn_ctx *ctx, float *param_dev, const float *grad_dev, float *momentum_buf_dev, size_t n, float lr,
                    float momentum, float weight_decay, int first_step, float grad_scale, void *hip_stream);

/* ---- training step on NHWC [hi | lo] bf16 planes (round 6; csrc/trainx.hip) -------------------------------------------------
 * One object runs forward + loss + backward of rtpose_light3d(15, 14, 2, input_dim = 1) in train mode for ONE batch shape:
 * the per-batch body of tpm/train_rtpose_light3d_kdh3d_mpaug.py:160-180 (CR) up to (not including) optimizer.step(), i.e.
 * model(img) (tpm/lib/network/rtpose_light3d.py:326-356, BatchNorm on batch statistics, running statistics updated),
 * rtpose_light3d_loss_fgweight (tpm/lib/network/losses.py:65-106) and loss.backward().  Split-bf16 arithmetic (PN_PREC_BF16X3:
 * 16 significant bits per operand, fp32 accumulate); forward and data-gradient convolutions run the inference kernels
 * (conv3_kernel / conv4_kernel / conv_mfma_kernel), the weight gradient a pixel-K MFMA GEMM (trainx_wgrad.h).
 * Parameters and gradients live in the CALLER's flat fp32 buffers (one float per parameter element, any tensor order):
 *   pn_trainer_set_param   where tensor `name` ("model0.layer1.0.conv1.weight", ...; the reference's state_dict keys) starts in
 *                          both flat buffers (offset in floats; 16-byte aligned offsets recommended) and how many elements it has
 *   pn_trainer_set_stat    device pointer of a BatchNorm running_mean / running_var tensor (updated in place every step)
 *   pn_trainer_finalize    plans the step for batch B of H x W inputs (multiples of 8), allocates every activation / gradient
 *                          tensor and descriptor; nothing is allocated afterwards (the step can be captured in a hipGraph)
 *   pn_trainer_forward_backward   img [B,1,H,W], heat_gt [B,16,H/8,W/8], paf_gt [B,28,..], z_gt [B,15,..], fg_mask [B,15,..] f32 NCHW;
 *                          writes the six loss terms (l1_paf, l1_heat, l1_z, l2_paf, l2_heat, l2_z) and every gradient
 *                          (asynchronous on hip_stream); apply them with pn_sgd_nesterov. */
typedef struct pn_trainer pn_trainer;
pn_trainer *pn_trainer_create(pn_ctx *ctx);
void pn_trainer_destroy(pn_trainer *t);
/* PN_PREC_BF16X3 (default): tensors as [hi | lo] bf16 planes, split-bf16 MFMA.  PN_PREC_F32: one fp32 plane per tensor, every product an exact fp32 FMA
 * chain on v_mfma_f32_16x16x4_f32 (the generic fp32 inference kernel for forward / data gradient, a K = 4 pixel weight gradient): the parity mode.
 * Call before pn_trainer_finalize. */
int pn_trainer_set_precision(pn_trainer *t, int precision);
int pn_trainer_set_param(pn_trainer *t, const char *name, size_t offset, size_t numel);
int pn_trainer_set_stat(pn_trainer *t, const char *name, float *stat_dev);
int pn_trainer_finalize(pn_trainer *t, float *flat_param_dev, float *flat_grad_dev, int B, int H, int W, float bn_momentum, float bn_eps);
int pn_trainer_forward_backward(pn_trainer *t, const float *img_dev, const float *heat_gt_dev, const float *paf_gt_dev, const float *z_gt_dev,
                                const float *fg_mask_dev, float *loss_terms_dev, void *hip_stream);
/* split-bf16 MFMA FLOPs (3 x 2 MAC) of the convolutions one step runs on the matrix cores (forward + data gradient; the weight gradient has the forward's count) */
double pn_trainer_conv_flops(pn_trainer *t);

/* ---- Yolo-Pose+ decode ----------------------------------------------------------------------
 * Replaces parse_prior_pose (tpm/lib/utils/prior_pose_align.py:10-168, pred_vis=False), quirks
 * included (candidate order anchor-major; suppression loop over rows 1..n-2; inclusive
 * visibility test).  Unlike the reference it does NOT modify posemaps in place.               */
#define PN_YOLO_MAX_DET 64
typedef struct pn_yolo_frame {
    int32_t n_det;                 /* surviving boxes (<= PN_YOLO_MAX_DET) */
    int32_t n_candidates;          /* conf > threshold before NMS */
    uint32_t status;               /* bit0: more than PN_YOLO_MAX_DET survivors (truncated) */
    int32_t reserved;
    float   bbox[PN_YOLO_MAX_DET][5];                 /* x1, y1, x2, y2, conf (pixels) */
    float   human[PN_YOLO_MAX_DET][PN_NUM_JOINTS][3]; /* x, y (pixels), Z (metres) */
    int32_t visibility[PN_YOLO_MAX_DET][PN_NUM_JOINTS];
    /* per-frame glue of tpm/evaluate/evaluation_yolo_posenet_kdh3d_mpreal.py:182-217 (float32, as the
     * script computes it): joints rescaled to the original frame and back-projected; part confidence =
     * bbox[4] for every joint.  Filled when pn_parse_yolo is given a glue configuration. */
    float   joints_2d[PN_YOLO_MAX_DET][PN_NUM_JOINTS][2];
    float   joints_3d[PN_YOLO_MAX_DET][PN_NUM_JOINTS][3];
    float   bbox_org[PN_YOLO_MAX_DET][4];             /* box corners rescaled to the original frame (:197-200) */
} pn_yolo_frame;

int pn_parse_yolo(pn_ctx *ctx, const float *posemaps_dev, int B, int h, int w,
                  const float *anchors_wh, int num_anchors, int num_joints, int w_out, int h_out,
                  float depth_mean, float depth_std, float conf_threshold, float nms_threshold,
                  int vis_margin, const pn_parse_cfg *glue /* may be NULL: input_size, w_org, h_org, intrinsics */,
                  pn_yolo_frame *frames_dev, void *hip_stream);

/* pred_vis=True (prior_pose_align.py:62,120,153-157): the maps carry 5 + 4 J channels per anchor, the last J being predicted
 * visibilities; vis_pred_dev [B][PN_YOLO_MAX_DET][J] float32 receives (in-bounds test) * (that channel), the value the
 * reference returns as `visibility` in this mode.  Everything else as pn_parse_yolo. */
int pn_parse_yolo_predvis(pn_ctx *ctx, const float *posemaps_dev, int B, int h, int w,
                          const float *anchors_wh, int num_anchors, int num_joints, int w_out, int h_out,
                          float depth_mean, float depth_std, float conf_threshold, float nms_threshold,
                          int vis_margin, const pn_parse_cfg *glue, pn_yolo_frame *frames_dev, float *vis_pred_dev,
                          void *hip_stream);

size_t pn_sizeof_yolo_frame(void);

/* Host-only diagnostic: the four float32 bicubic taps (OpenCV interpolateCubic, A = -0.75) the
 * parse kernels use for fractional offset x.  Lets CPU tests pin the table against the oracle.   */
void pn_debug_cubic_coeffs(float x, float *out4);

/* ---- legacy plug-in ABI: the SWIG module `pafprocess` -----------------------------------------
 * Same seven symbols, same argument meaning and the same (non re-entrant, global-state)
 * semantics as tpm/lib/pafprocess/pafprocess.h:53-59, COCO-18 topology (pafprocess.h:6-24).
 * Inputs are HOST float32 C-contiguous arrays, borrowed for the duration of the call:
 *   peaks [p1][p2][p3>=5] (x, y, score, -, part), heatmap [h1][h2][h3] (unused, as in the
 *   reference), pafmap [f1][f2][f3=38].
 * Differences: returns a negative pn_status instead of 0 when the GPU path fails; peaks that
 * would index outside the PAF map are rejected instead of read out of bounds
 * (pafprocess.cpp:232-233 has no bounds check).                                                 */
int process_paf(int p1, int p2, int p3, float *peaks, int h1, int h2, int h3, float *heatmap,
                int f1, int f2, int f3, float *pafmap);
int get_num_humans(void);
int get_part_cid(int human_id, int part_id);
float get_score(int human_id);
int get_part_x(int cid);
int get_part_y(int cid);
float get_part_score(int cid);

#ifdef __cplusplus
}
#endif
#endif /* POPNET_HIP_H */
