// Internal declarations shared by the HIP translation units of libpopnet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/popnet_hip.h"

struct pn_ctx {
    int device = 0;
    std::string err;
    int num_cus = 256;
    // scratch for the parse kernels (grown on demand, owned by the ctx)
    void *parse_ws = nullptr;
    size_t parse_ws_bytes = 0;
    bool parse_ws_fixed = false;     // pn_parse_reserve: the scratch never moves again
    void *parse_big = nullptr;       // workspace of the unbounded second pass (parse_paf.hip::BigHost), freed by pn_parse_big_free
    // scratch of the training kernels (flipped weights, split-reduction partials; train.hip), stream-ordered reuse
    void *train_ws = nullptr;
    size_t train_ws_bytes = 0;
    bool train_ws_keep = false;      // pn_train_ws_keep: a captured graph points into train_ws -- a growth retires the old block instead of freeing it
    std::vector<void *> train_ws_retired;
    bool train_lds_attr = false;     // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done for the tile kernels on this device
    bool train_x3 = false;           // pn_train_set_precision: 3x3 forward / data-gradient convolutions on split-bf16 MFMA
    // pn_train_pack_cache: the packed (transposed / rotated / split) weights of the 3x3 training convolutions live in persistent buffers keyed by
    // (weight pointer, shape, flip, precision) and are all refreshed by ONE launch per step (pn_train_pack_refresh) instead of one launch per
    // convolution call (train.hip)
    struct PackEntry { const float *w; int Cout, Cin, flip, x3; void *buf; size_t bytes; bool fresh; };
    bool train_pack_cache = false;
    std::vector<PackEntry> train_packs;
    void *train_pack_table = nullptr;      // device copy of the descriptor table
    size_t train_pack_table_entries = 0;   // entries the device table holds (re-uploaded when the host list has grown)
    size_t train_pack_table_cap = 0;       // entries the block has room for: append-only inside it (a captured graph keeps pointing at it)
    unsigned train_pack_blocks = 0;
};

int pn_set_error(pn_ctx *ctx, int code, const char *fmt, ...);
void pn_parse_big_free(pn_ctx *ctx);     // parse_paf.hip

#define PN_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return pn_set_error((ctx), PN_ERR_HIP, "%s failed: %s (%s:%d)", #expr,           \
                                hipGetErrorString(_e), __FILE__, __LINE__);                  \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: every launch site keeps one of these (a static
// per kernel instantiation) and opts in once per device it launches on (a process-wide flag left a second GPU unconfigured).
struct PnLdsAttr { size_t bytes[64] = {}; };
inline int pn_lds_attr(pn_ctx *ctx, PnLdsAttr &st, const void *kern, size_t bytes) {
    const int d = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if (st.bytes[d] >= bytes) return PN_OK;
    PN_HIP_CHECK(ctx, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    st.bytes[d] = bytes;
    return PN_OK;
}

// ---------------------------------------------------------------------------------------------
// Convolution problem descriptor (device-visible).  One launch handles a GROUP of problems that
// share a kernel instantiation (blockIdx.y = problem), e.g. the three branches of a stage.
// Activations are NHWC with a channel stride (so a tensor can be a channel slice of a wider
// buffer: the stage-2 "concat" is never materialised by a copy).
// ---------------------------------------------------------------------------------------------
enum pn_act {
    PN_ACT_NONE = 0,
    PN_ACT_RELU = 1,
    PN_ACT_LEAKY = 2,        // LeakyReLU(0.1)
    PN_ACT_SIG_PM2 = 3,      // (sigmoid(x) - 0.5) * 4   rtpose_light3d.py:335,337
    PN_ACT_SIG = 4,          // sigmoid(x)               rtpose_light3d.py:336
    PN_ACT_YOLO = 5          // per-slice casts          yolo_posenet.py:146-156
};

struct ConvProblem {
    const void *in;        // NHWC activations (element type T)
    const void *wpack;     // weights in MFMA fragment order (see pack_conv_weights)
    const float *bias;     // folded bias, padded to cout tiles
    const void *res;       // residual NHWC (T) or nullptr; added before the activation
    void *out;             // NHWC (T) output or nullptr
    float *out_nchw;       // NCHW f32 output or nullptr
    int B, H, W;           // input spatial size
    int Ho, Wo;            // output spatial size
    int cin_chunks;        // padded Cin / 64
    int in_cs, in_coff;    // input channel stride / first channel
    int cout;              // valid output channels
    int out_cs, out_coff;
    int res_cs, res_coff;
    int act;
    int yolo_naf;          // channels per anchor for PN_ACT_YOLO (5 + 3J)
    int R;                 // output rows per block
    int Wt, tiles_x;       // output columns per block, ceil(Wo / Wt)
    int tiles_per_img;     // ceil(Ho / R) * tiles_x
    int cout_blocks;       // ceil(cout / (WC*CT*16))
    int nblocks;           // B * tiles_per_img * cout_blocks
    int ksteps;            // cin_chunks * KS*KS * 2   (k32 steps per cout tile in wpack)
    int ks;                // kernel size of this problem (conv3_mix_kernel picks the block's body by it)
    int lds_buf_bytes;     // bytes of one LDS halo image
    int lds_two;           // 1: a second image follows (double-buffered chunks), 0: single image
    unsigned in_zero_off;  // conv3_kernel: byte offset from `in` to >= 16 zero bytes (padding source of the halo DMA)
    // bf16x3 mode (PN_PREC_BF16X3): a tensor is the planes hi = bf16(v), lo = bf16(v - hi), `split` channels apart; 0 = plain
    // bf16 / f32 tensor.  The K loop sees three times the channels, [x_hi | x_lo | x_hi] against packed weights [W_hi | W_hi | W_lo]
    // (net.hip::prepare_conv).
    // Round 4: the third plane is a copy of the first, so only TWO planes [hi | lo] are stored; the K loop still runs over three
    // plane pairs and fetches input chunk c from chunk (c >= in_wrap ? c - in_wrap : c) of the buffer -- the hi plane twice.  A third
    // less HBM traffic on every bf16x3 tensor, same values in the same k order.
    int split;             // plane distance (channels) of the OUTPUT tensor, 0 = not split
    int res_split;         // plane distance of the residual tensor (hi + lo are added), 0 = not split
    int in_wrap;           // 64-channel chunks of the INPUT buffer after which the source chunk index wraps to 0 (2 x plane / 64; > cin_chunks = never)
    // fused 1x1 tail (conv3_kernel<1, 4, 1, ...> only; net.hip::fuse_1x1_tails): when tail_w != nullptr the block's 128-channel
    // output tile is NOT stored; a second 1x1 convolution (128 -> tail_cout <= 32 channels) runs on it from LDS and only ITS
    // result leaves the kernel -- the `1x1 256 -> 128 + BN + LeakyReLU, then 1x1 128 -> 28` tail of the PAF branch
    // (tpm/lib/network/rtpose_light3d.py:266-267) as one launch instead of two.
    const void *tail_w;    // [2 cout tiles][4 k-steps][64 lanes][8 bf16], rows permuted with pn_conv_row_channel(tile, row, 2)
    const float *tail_bias;
    void *tail_out;        // NHWC (T) or nullptr
    float *tail_nchw;      // NCHW f32 or nullptr
    int tail_cout, tail_act, tail_out_cs, tail_out_coff;
    int tail_split;        // fused average pool in a bf16x3 net: plane distance (channels) of the pooled [hi | lo] output, 0 = plain bf16
};

// Tile configuration ids (see conv_mfma.hip).
enum pn_conv_cfg {
    PN_CFG_C128 = 0,  // 4x1 waves, 2 cout tiles x 7 pixel tiles per wave: 128 couts x 112 px
    PN_CFG_C64 = 1,   // 2x2 waves, 2 cout tiles x 4 pixel tiles per wave:  64 couts x 128 px
    PN_CFG_C32 = 2,   // 1x4 waves, 2 cout tiles x 2 pixel tiles per wave:  32 couts x 128 px
    PN_CFG_C16 = 3,   // 1x4 waves, 1 cout tile  x 2 pixel tiles per wave:  16 couts x 128 px
    PN_CFG_C64W = 4   // 2x2 waves, 2 cout tiles x 7 pixel tiles per wave:  64 couts x 224 px (wide maps)
};
int pn_cfg_couts(int cfg);
// cout tiles per wave of a configuration (TileCfg<>::CT in conv_mfma_kernel.h)
inline int pn_cfg_ct(int cfg) { return cfg == PN_CFG_C16 ? 1 : 2; }
// Output channel computed by row `row` (0..15) of packed cout tile `tile`.  The CT tiles a wave owns
// form a group of 16*CT channels; lane quarter q = row >> 2 of every tile in the group holds channels
// [4*CT*q, 4*CT*(q+1)) of it, so a lane's CT accumulators are 4*CT consecutive channels (one 16-B
// bf16 store for CT = 2).  CT = 1 is the natural order.
inline int pn_conv_row_channel(int tile, int row, int CT) {
    return (tile / CT) * CT * 16 + 4 * CT * (row >> 2) + 4 * (tile % CT) + (row & 3);
}
int pn_cfg_pixels(int cfg);

struct ConvLaunch {
    int prec;      // pn_precision
    int ks;        // 1 or 3
    int stride;    // 1 or 2
    int pitch;     // LDS halo row pitch in pixels (multiple of 8, >= halo columns)
    int cfg;       // pn_conv_cfg
    int kern = 0, wc = 0, wp = 0, nbuf = 0, pt = 7, rpg = 4;   // kern 3: conv3_kernel<ks, wc, wp, nbuf, pt, rpg>
    int tail = 0;            // kern 3: 1 = every problem of the launch carries a fused 1x1 tail (ConvProblem::tail_w); 2 = a fused average pool (net.hip::fuse_pool_tails)
    int mix = 0;             // kern 3: 3x3 problems and fused-tail 1x1 problems share the launch (conv3_mix_kernel)
    int nprob;
    int max_blocks;          // max nblocks over the group
    size_t lds_bytes;
    const ConvProblem *probs_dev;
};
int pn_launch_conv(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv3(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);     // conv3_inst_*.hip
int pn_launch_conv4(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);     // conv4_inst.hip

// fused BasicBlock(64) (bb64_kernel.h): passed by value as the kernel argument
struct BBProblem {
    const void *in;        // NHWC bf16, channel stride in_cs, first channel in_coff (64 channels used)
    void *out;             // NHWC bf16
    const void *wpack;     // 36 x 4 KB
    const float *bias1, *bias2;
    int B, H, W;
    int in_cs, in_coff, out_cs, out_coff;
    int Wt, tiles_x, tiles_per_img, ntiles;
    unsigned in_zero_off;  // byte offset from `in` to >= 16 zero bytes
    // bb64x3_kernel (bf16x3 nets): plane distance in channels of the three-plane [hi | lo | hi] input / output tensors; wpack = 72 x 4 KB
    int in_split, out_split;
    // bb64_kernel (round 5): two zero-initialised ints in device memory -- [0] the next tile ticket, [1] the number of workgroups that have finished (the last
    // one resets both, so a replayed graph finds zeros again).  nullptr = the static schedule (tile = blockIdx.x + k * gridDim.x)
    int *tickets;
    int halves;          // bb64_kernel<halves>: 2 = eight compute waves (opt-in), anything else = four
};
int pn_launch_bb64(pn_ctx *ctx, const BBProblem &P, hipStream_t stream);       // bb64_inst.hip
int pn_launch_bb64x3(pn_ctx *ctx, const BBProblem &P, hipStream_t stream);     // bb64x3_inst.hip (6-row tiles: tiles_per_img = ceil(H / 6) * tiles_x)
size_t pn_conv3_lds_bytes(int ks, int WP, int nbuf, int rpg = 4);
size_t pn_conv_lds_bytes(int prec, int ks, int stride, int pitch, int R);
int pn_conv_stage_maxpx(int prec, int ks, int stride, int pitch, int cfg);   // 0 = no limit (direct staging)

// stem: 7x7 stride-2 pad-3, Cin = 1, fused folded-BN bias + ReLU.  x NCHW f32 [B,1,H,W] ->
// NHWC T [B,Ho,Wo,64].  w [49][64] f32 (tap-major) for the fp32 VALU kernel, wfrag = the same weights as
// 8 bf16 MFMA A-fragments (see stem7x7_mfma_kernel) for bf16 mode, bias [64].
struct PnFrameSrc;
// stem + MaxPool2d(3, 2, 1) in one launch (bf16 nets; conv_misc.hip): out = the POOLED map [B, (Ho-1)/2+1, (Wo-1)/2+1, out_cs]
int pn_launch_stem_pool(pn_ctx *ctx, const float *x, const void *wfrag, const float *bias, void *out, int B, int H, int W, int Ho, int Wo,
                        int out_cs, hipStream_t stream, const PnFrameSrc *src);
struct PnFrameSrc;       // preproc_pixel.h: raw depth frames + the pre-processing constants (nullptr = x is the pre-processed input)
int pn_launch_stem(pn_ctx *ctx, int prec, const float *x, const float *w, const void *wfrag, const float *bias,
                   void *out, int B, int H, int W, int Ho, int Wo, int out_cs, int split, hipStream_t stream, const PnFrameSrc *src = nullptr);

// pooling on NHWC T.  mode 0: avg 3x3 s2 p1 (count_include_pad), 1: max 3x3 s2 p1, 2: max 2x2 s2.
int pn_launch_pool(pn_ctx *ctx, int prec, int mode, const void *in, void *out, int B, int H, int W,
                   int C, int in_cs, int out_cs, int out_coff, int in_split, int out_split, hipStream_t stream);

// NCHW f32 -> ReLU -> NHWC T (the multi-channel stem's hand-over, conv_misc.hip)
// train.hip: model0.conv1 of the planes training engine (trainx.hip) -- output / output gradient as a planes tensor [pixel][cs] (f32 != 0: one
// f32 plane; else two bf16 planes [hi | lo] `split` elements apart)
// gather != 0: tconv_fwd_kernel's element-by-element operand gather (A/B; bit-identical) instead of the input patch in LDS
int pn_stem_forward_planes(pn_ctx *ctx, const float *x_dev, const float *w_dev, void *y_planes, int cs, int split, int f32, int N, int Cin, int H, int W, int Cout,
                           int ks, int stride, int pad, int gather, hipStream_t s);
// bn != nullptr: `dy_planes` is the gradient w.r.t. the BatchNorm + activation OUTPUT and the kernel applies the BatchNorm backward itself (k1, k2, k3: what
// trainx_kernels.h::bn_bwd_finish_kernel leaves); depth = chunks of 32 pixels in flight per block (1 or 4)
struct PnStemBn {
    const void *x; int x_cs, x_split;
    const float *mean, *invstd, *k1, *k2, *k3, *scale, *shift;
    int act;
};
int pn_stem_wgrad_planes(pn_ctx *ctx, const float *x_dev, const void *dy_planes, int cs, int split, int f32, const PnStemBn *bn, float *dw_dev, int N, int Cin, int H,
                         int W, int Cout, int ks, int stride, int pad, int depth, hipStream_t s);
int pn_launch_nchw_relu_to_nhwc(pn_ctx *ctx, int prec, const float *in, void *out, int B, int H, int W, int C, int out_cs, int split, hipStream_t stream);
// NHWC T channel slice -> NCHW f32 (diagnostics / stage-1 outputs).
int pn_launch_nhwc_to_nchw(pn_ctx *ctx, int prec, const void *in, float *out, int B, int H, int W,
                           int C, int in_cs, int in_coff, int split, hipStream_t stream);
