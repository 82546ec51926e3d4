// The training step of rtpose_light3d on the INFERENCE convolution kernels (round 6).
//
// Replaces, for TrainEngine(precision = "bf16x3"), the NCHW fp32 primitives of train.hip (whose forward / data-gradient kernels spent
// 83 of 110 us per launch gathering an NCHW tensor into a channel-minor LDS image) with one C++ object that keeps every activation and
// gradient as NHWC [hi | lo] bf16 planes -- the layout of the bf16x3 inference net -- so that
//   * forward convolutions AND data gradients run conv3_kernel / conv4_kernel / conv_mfma_kernel unchanged (a data gradient is the same
//     convolution on the 180-degree-rotated, Cin <-> Cout-transposed weight pack: trainx_kernels.h::pack_kernel builds both packs of every
//     layer from the live fp32 parameters in ONE launch per step);
//   * train-mode BatchNorm, pooling, heads and loss are channel-minor 16-byte-vector passes (trainx_kernels.h);
//   * the weight gradient is a pixel-K MFMA GEMM over the same planes (trainx_wgrad.h), deterministic split-K.
// Reference being replaced (tpm/ = third_party_methods/):
//   tpm/train_rtpose_light3d_kdh3d_mpaug.py:153-212 (CR)   the per-batch body: model(img) -> loss -> backward
//   tpm/lib/network/rtpose_light3d.py:124-219,222-246,326-356   the module in train mode
//   tpm/lib/network/losses.py:65-106                        rtpose_light3d_loss_fgweight
// Parameters, gradients and BatchNorm running statistics stay in the caller's flat fp32 buffers (popnet_amd.train.TrainEngine): the
// optimiser, the data-parallel all-reduce and the checkpoint format are untouched.  The 7x7 single-channel stem keeps train.hip's NCHW
// kernels (Cin = 1: nothing for the matrix cores to contract over) behind two layout hand-overs.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "pn_internal.h"
#include "conv_plan.h"
#include "trainx_kernels.h"
#include "trainx_wgrad.h"

namespace {

using tx::bf;

struct TxTensor {                     // [B][H][W][2 * plane] bf16 planes [hi | lo], or (fp32 engine) [B][H][W][plane] float: 4 bytes per channel either way
    char *p = nullptr;
    int H = 0, W = 0, plane = 0;
    bool f32 = false;
    int cs() const { return f32 ? plane : 2 * plane; }      // channel stride in elements
    int split() const { return f32 ? 0 : plane; }           // plane distance in elements (0: one fp32 plane)
    char *at(int ch) const { return p + (size_t)ch * (f32 ? 4 : 2); }
};

struct TxBn {
    std::string name;
    int C = 0;
    const float *gamma = nullptr, *beta = nullptr;
    float *dgamma = nullptr, *dbeta = nullptr, *rm = nullptr, *rv = nullptr;
    float *mean = nullptr, *invstd = nullptr, *scale = nullptr, *shift = nullptr, *k1 = nullptr, *k2 = nullptr, *k3 = nullptr;
};

struct TxLayer {                      // one nn.Conv2d
    std::string name;
    int ks = 3, cin = 0, cout = 0;
    const float *w = nullptr, *b = nullptr;
    float *dw = nullptr, *db = nullptr;
    int x = -1;                       // input tensor
    bool cat = false;                 // reads the stage-2 input in this engine's channel order [feat | paf | heat | z | pad]
    bool bn_follows = false;          // its output goes straight into a train-mode BatchNorm: the batch mean cancels the bias, d loss / d bias == 0 exactly
    bf *pack_f = nullptr, *pack_d = nullptr;      // forward / data-gradient weight packs (built lazily by the group that needs them)
    float *bias_pad = nullptr;
};

struct ConvUse {                      // one convolution launch problem
    int layer; bool dgrad;
    int in, out, out_coff, res;       // tensors (-1 = none)
    int act;
    float *out_nchw;
};

}  // namespace

struct pn_trainer {
    pn_ctx *ctx = nullptr;
    int B = 0, H = 0, W = 0;
    float *flat_p = nullptr, *flat_g = nullptr;
    std::map<std::string, std::pair<size_t, size_t>> params;      // name -> (offset, numel) in the flat buffers
    std::map<std::string, float *> stats;                        // running_mean / running_var device pointers
    bool finalized = false;
    bool legacy_wgrad = false;
    bool f32 = false;                // precision "fp32": one fp32 plane per tensor, the generic fp32 convolution kernel, K = 4 weight gradient
    std::vector<void *> allocs;
    std::vector<TxTensor> T;
    std::vector<TxBn> bns;
    std::vector<TxLayer> layers;
    std::vector<ConvLaunch> launches;
    std::vector<std::function<int(hipStream_t)>> ops;
    std::vector<tx::PackDesc> packs;
    std::vector<tx::BiasDesc> biases;
    tx::PackDesc *packs_dev = nullptr;
    tx::BiasDesc *biases_dev = nullptr;
    unsigned pack_groups = 0, pack_units = 0;
    bool pack_gather = false;          // POPNET_TRAINX_PACK=gather: round 6's first pack kernel (one thread per 16-byte group; bit-identical)
    float *zero_bias = nullptr;
    int *cat_k_map = nullptr, *cat_ref_map = nullptr;           // [192] my channel -> reference channel ; [187] reference -> my channel
    double *partial = nullptr; size_t partial_doubles = 0;
    float *nchw_a = nullptr, *nchw_b = nullptr; size_t nchw_elems = 0;      // NCHW f32 scratch: the stem hand-over and the legacy weight gradient
    float *wg_partial[4] = {nullptr, nullptr, nullptr, nullptr}; size_t wg_partial_floats = 0;      // one split-K scratch per side stream
    // the weight gradients run on a second stream beside the BatchNorm / data-gradient chain (matrix-core-bound next to bandwidth-bound launches)
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    int nside = 1, next_side = 0;            // POPNET_TRAINX_SIDES: weight gradients of different layers dealt round-robin to this many side streams
    bool two_streams = true;
    std::vector<hipEvent_t> events;
    size_t side_tail = (size_t)-1;            // ops.size() right after the last side-stream op: nothing new on the step's stream since = no new fork needed
    float *head_out[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
    // per-step arguments (read by the ops when they launch)
    const float *img = nullptr, *target[3] = {nullptr, nullptr, nullptr}, *fg = nullptr;
    float *loss = nullptr;
    float momentum = 0.1f, eps = 1e-5f;
    double flops_conv = 0;
};

namespace {

const int HEAD_C[3] = {28, 16, 15};          // paf, heat, z            rtpose_light3d(15, 14, 2)
const int HEAD_KIND[3] = {1, 0, 1};          // (s - 0.5) * 4 | s | (s - 0.5) * 4      rtpose_light3d.py:335-337
const int HEAD_ACT[3] = {PN_ACT_SIG_PM2, PN_ACT_SIG, PN_ACT_SIG_PM2};
const int CAT_OFF[3] = {128, 156, 172};      // channel slices of the stage-2 input (net.hip::build_rtpose)
const int CAT_PLANE = 192;

int tx_alloc(pn_trainer *t, void **p, size_t bytes, bool zero) {
    PN_HIP_CHECK(t->ctx, hipMalloc(p, bytes));
    t->allocs.push_back(*p);
    if (zero) PN_HIP_CHECK(t->ctx, hipMemset(*p, 0, bytes));
    return PN_OK;
}

int new_tensor(pn_trainer *t, int H, int W, int plane, int *id) {
    TxTensor x;
    x.H = H; x.W = W; x.plane = plane; x.f32 = t->f32;
    const size_t bytes = (size_t)t->B * H * W * plane * 4 + 2048;          // (two bf16 planes or one fp32 plane) + zero page: the halo DMA's padding source
    if (bytes >= ((size_t)1 << 32)) return pn_set_error(t->ctx, PN_ERR_UNSUPPORTED, "pn_trainer: a %dx%dx%d tensor at batch %d exceeds the 4 GiB the kernels' 32-bit offsets address", H, W, plane, t->B);
    if (int rc = tx_alloc(t, (void **)&x.p, bytes, true)) return rc;
    t->T.push_back(x);
    *id = (int)t->T.size() - 1;
    return PN_OK;
}

int pad64(int c) { return (c + 63) / 64 * 64; }

int find_param(pn_trainer *t, const std::string &name, size_t numel, const float **p, float **g) {
    auto it = t->params.find(name);
    if (it == t->params.end()) return pn_set_error(t->ctx, PN_ERR_INVALID, "pn_trainer: parameter %s was not set", name.c_str());
    if (it->second.second != numel) return pn_set_error(t->ctx, PN_ERR_INVALID, "pn_trainer: parameter %s has %zu elements, %zu expected", name.c_str(), it->second.second, numel);
    *p = t->flat_p + it->second.first;
    *g = t->flat_g + it->second.first;
    return PN_OK;
}

int new_layer(pn_trainer *t, const std::string &name, int ks, int cin, int cout, bool bias, int x, bool cat, int *id) {
    TxLayer L;
    L.name = name; L.ks = ks; L.cin = cin; L.cout = cout; L.x = x; L.cat = cat;
    if (int rc = find_param(t, name + ".weight", (size_t)cout * cin * ks * ks, &L.w, &L.dw)) return rc;
    if (bias)
        if (int rc = find_param(t, name + ".bias", (size_t)cout, &L.b, &L.db)) return rc;
    t->layers.push_back(L);
    *id = (int)t->layers.size() - 1;
    return PN_OK;
}

int new_bn(pn_trainer *t, const std::string &name, int C, int *id) {
    TxBn b;
    b.name = name; b.C = C;
    float *g = nullptr;
    if (int rc = find_param(t, name + ".weight", (size_t)C, &b.gamma, &b.dgamma)) return rc;
    const float *bp = nullptr;
    if (int rc = find_param(t, name + ".bias", (size_t)C, &bp, &b.dbeta)) return rc;
    b.beta = bp;
    (void)g;
    auto rm = t->stats.find(name + ".running_mean"), rv = t->stats.find(name + ".running_var");
    if (rm == t->stats.end() || rv == t->stats.end()) return pn_set_error(t->ctx, PN_ERR_INVALID, "pn_trainer: running statistics of %s were not set", name.c_str());
    b.rm = rm->second; b.rv = rv->second;
    float *blk = nullptr;
    if (int rc = tx_alloc(t, (void **)&blk, (size_t)7 * C * 4, true)) return rc;
    b.mean = blk; b.invstd = blk + C; b.scale = blk + 2 * C; b.shift = blk + 3 * C; b.k1 = blk + 4 * C; b.k2 = blk + 5 * C; b.k3 = blk + 6 * C;
    t->bns.push_back(b);
    *id = (int)t->bns.size() - 1;
    return PN_OK;
}

unsigned grid_for(long items, int cus) {          // grid-stride elementwise launches: enough blocks to fill the chip, no more
    const long want = (items + 255) / 256;
    return (unsigned)std::max<long>(1, std::min<long>(want, (long)cus * 16));
}

// ---- convolution groups (one launch per kernel instantiation, blockIdx.y = problem) -------------------------------------------------
struct PlannedConv { ConvUse u; ConvGeom g; int rows, kplane, ks, cin_chunks, BC, cout_pad; };

int ensure_pack(pn_trainer *t, const PlannedConv &pc, bf **dst_out) {
    TxLayer &L = t->layers[pc.u.layer];
    bf *&slot = pc.u.dgrad ? L.pack_d : L.pack_f;
    if (slot) { *dst_out = slot; return PN_OK; }
    const int KK = pc.ks * pc.ks, ksteps = pc.cin_chunks * KK * 2;
    const bool k4 = pc.g.kern == 4;
    const size_t real_groups = k4 ? (size_t)(pc.cout_pad / 128) * (ksteps + 3) * 8 * 64 : (size_t)(pc.cout_pad / 16) * ksteps * 64;
    const size_t gbytes = t->f32 ? 32 : 16;                               // one lane's share of a fragment: 8 values
    const size_t bytes = real_groups * gbytes + (k4 ? 0 : 5 * 64 * gbytes);          // + spare fragments: the weight queue prefetches up to 5 k-steps ahead
    if (int rc = tx_alloc(t, (void **)&slot, bytes, true)) return rc;
    tx::PackDesc d;
    memset(&d, 0, sizeof d);
    d.w = L.w; d.dst = slot; d.f32 = t->f32 ? 1 : 0; d.Cout = L.cout; d.Cin = L.cin; d.ks = pc.ks;
    d.transpose = pc.u.dgrad ? 1 : 0; d.conv4 = k4 ? 1 : 0;
    d.CT = pc.g.kern == 3 ? 2 : pn_cfg_ct(pc.g.cfg);
    d.rows_valid = pc.rows; d.kplane = pc.kplane; d.ksteps = ksteps;
    d.row_map = (pc.u.dgrad && L.cat) ? t->cat_k_map : nullptr;
    d.k_map = (!pc.u.dgrad && L.cat) ? t->cat_k_map : nullptr;
    d.first_group = t->pack_groups; d.ngroups = (unsigned)real_groups;
    t->pack_groups += (unsigned)real_groups;
    d.first_unit = t->pack_units; d.nunits = (unsigned)((size_t)pc.cout_pad * pc.kplane / 8);
    t->pack_units += d.nunits;
    t->packs.push_back(d);
    *dst_out = slot;
    return PN_OK;
}

int add_conv_group(pn_trainer *t, const std::vector<ConvUse> &uses) {
    pn_ctx *ctx = t->ctx;
    std::vector<PlannedConv> pcs;
    // what net.hip::harmonize_level decides for a level of independent convolutions
    bool k4 = false, wide[4] = {false, false, false, false};
    long blocks = 0;
    for (const ConvUse &u : uses) {
        const TxLayer &L = t->layers[u.layer];
        const TxTensor &in = t->T[u.in];
        const int rows = u.dgrad ? t->T[L.x].plane : L.cout;
        if (L.ks == 3 && rows >= 64) {
            k4 = true;
            const long strips = (long)t->B * ((in.H + 3) / 4) * ((in.W + 29) / 30);
            blocks += ((strips + 1) / 2) * ((rows + 127) / 128);
        }
        if (rows > 64) wide[L.ks] = true;
    }
    if (blocks < 448) k4 = false;
    for (const ConvUse &u : uses) {
        const TxLayer &L = t->layers[u.layer];
        const TxTensor &in = t->T[u.in];
        PlannedConv pc;
        pc.u = u;
        pc.ks = L.ks;
        pc.rows = u.dgrad ? t->T[L.x].plane : L.cout;
        pc.kplane = in.plane;
        if (pc.kplane % 64) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_trainer: %s: input plane %d is not a multiple of 64", L.name.c_str(), pc.kplane);
        pc.cin_chunks = (t->f32 ? 1 : 3) * pc.kplane / 64;
        pc.g = ConvGeom();
        const int wc_min = (wide[L.ks] && pc.rows > 32) ? 4 : 0;
        const int prec = t->f32 ? PN_PREC_F32 : PN_PREC_BF16;
        pn_plan_conv_kernel(prec, t->B, ctx->num_cus, in.H, in.W, pc.rows, L.ks, 1, pc.cin_chunks, wc_min, 0, k4 ? 1 : 0, pc.g);
        const char *why = "";
        if (int rc = pn_plan_conv_tiles(prec, in.H, in.W, L.ks, 1, pc.g, &why)) return pn_set_error(ctx, rc, "pn_trainer: %s: %s", L.name.c_str(), why);
        if (pc.g.kern == 0 && L.ks == 1 && pc.g.pitch == 16) pc.g.pitch = 32;      // the generic 1x1 kernel is not instantiated for the 16-pixel pitch class (a wider LDS row is always valid)
        pc.BC = pc.g.kern == 4 ? 128 : (pc.g.kern == 3 ? pc.g.wc * 32 : pn_cfg_couts(pc.g.cfg));
        pc.cout_pad = (pc.rows + pc.BC - 1) / pc.BC * pc.BC;
        pcs.push_back(pc);
    }
    std::vector<bool> used(pcs.size(), false);
    for (size_t i = 0; i < pcs.size(); ++i) {
        if (used[i]) continue;
        std::vector<size_t> members;
        const PlannedConv &a = pcs[i];
        for (size_t j = i; j < pcs.size(); ++j) {
            const PlannedConv &b = pcs[j];
            if (used[j]) continue;
            const bool same = b.ks == a.ks && b.g.pitch == a.g.pitch && b.g.R == a.g.R && b.g.Wt == a.g.Wt && b.g.kern == a.g.kern &&
                              (a.g.kern == 4 || (a.g.kern == 3 ? (b.g.wc == a.g.wc && b.g.wp == a.g.wp && b.g.nbuf == a.g.nbuf && b.g.pt == a.g.pt && b.g.rpg == a.g.rpg) : b.g.cfg == a.g.cfg));
            if (same) { members.push_back(j); used[j] = true; }
        }
        std::vector<ConvProblem> probs;
        int max_blocks = 0;
        bool two_bufs = false;
        for (size_t m : members) {
            const PlannedConv &pc = pcs[m];
            TxLayer &L = t->layers[pc.u.layer];
            const TxTensor &in = t->T[pc.u.in];
            ConvProblem P;
            memset(&P, 0, sizeof P);
            bf *pack = nullptr;
            if (int rc = ensure_pack(t, pc, &pack)) return rc;
            P.in = in.p; P.wpack = pack;
            if (!pc.u.dgrad && L.b) {
                if (!L.bias_pad) {
                    if (int rc = tx_alloc(t, (void **)&L.bias_pad, (size_t)pad64(std::max(pc.cout_pad, 128)) * 4, true)) return rc;
                    tx::BiasDesc bd;
                    bd.src = L.b; bd.dst = L.bias_pad; bd.n = L.cout; bd.npad = pad64(std::max(pc.cout_pad, 128));
                    t->biases.push_back(bd);
                }
                P.bias = L.bias_pad;
            } else P.bias = t->zero_bias;
            P.B = t->B; P.H = in.H; P.W = in.W; P.Ho = in.H; P.Wo = in.W;
            P.cin_chunks = pc.cin_chunks; P.in_cs = in.cs(); P.in_coff = 0;
            P.in_wrap = t->f32 ? (1 << 20) : 2 * (in.plane / 64);
            P.cout = pc.rows;
            if (pc.u.out >= 0) { const TxTensor &o = t->T[pc.u.out]; P.out = o.p; P.out_cs = o.cs(); P.out_coff = pc.u.out_coff; P.split = o.split(); }
            if (pc.u.res >= 0) { const TxTensor &r = t->T[pc.u.res]; P.res = r.p; P.res_cs = r.cs(); P.res_coff = 0; P.res_split = r.split(); }
            P.out_nchw = pc.u.out_nchw;
            P.act = pc.u.act;
            P.yolo_naf = 50;
            P.R = pc.g.R; P.Wt = pc.g.Wt;
            P.tiles_x = (P.Wo + pc.g.Wt - 1) / pc.g.Wt;
            P.tiles_per_img = ((P.Ho + pc.g.R - 1) / pc.g.R) * P.tiles_x;
            P.cout_blocks = (pc.rows + pc.BC - 1) / pc.BC;
            P.nblocks = t->B * P.tiles_per_img * P.cout_blocks;
            if (pc.g.kern == 4) P.nblocks = ((t->B * P.tiles_per_img + 1) / 2) * P.cout_blocks;
            P.ksteps = pc.cin_chunks * pc.ks * pc.ks * 2;
            P.ks = pc.ks;
            P.lds_buf_bytes = (int)pn_conv_lds_bytes(t->f32 ? PN_PREC_F32 : PN_PREC_BF16, pc.ks, 1, pc.g.pitch, pc.g.R);
            P.lds_two = (pc.cin_chunks > 1 && 2 * (size_t)P.lds_buf_bytes <= 160 * 1024) ? 1 : 0;
            P.in_zero_off = (unsigned)((size_t)t->B * in.H * in.W * in.plane * 4);
            if (P.lds_two) two_bufs = true;
            max_blocks = std::max(max_blocks, P.nblocks);
            probs.push_back(P);
            t->flops_conv += 2.0 * (t->f32 ? 1.0 : 3.0) * (double)t->B * in.H * in.W * pc.rows * pc.kplane * pc.ks * pc.ks;
        }
        ConvProblem *dev = nullptr;
        if (int rc = tx_alloc(t, (void **)&dev, probs.size() * sizeof(ConvProblem), false)) return rc;
        PN_HIP_CHECK(ctx, hipMemcpy(dev, probs.data(), probs.size() * sizeof(ConvProblem), hipMemcpyHostToDevice));
        ConvLaunch cl;
        cl.prec = t->f32 ? PN_PREC_F32 : PN_PREC_BF16; cl.ks = a.ks; cl.stride = 1; cl.pitch = a.g.pitch; cl.cfg = a.g.cfg;
        cl.kern = a.g.kern; cl.wc = a.g.wc; cl.wp = a.g.wp; cl.nbuf = a.g.nbuf; cl.pt = a.g.pt; cl.rpg = a.g.rpg;
        cl.tail = 0; cl.mix = 0;
        cl.nprob = (int)probs.size(); cl.max_blocks = max_blocks;
        cl.lds_bytes = pn_conv_lds_bytes(cl.prec, a.ks, 1, a.g.pitch, a.g.R) * (two_bufs ? 2 : 1);
        if (a.g.kern == 3) cl.lds_bytes = pn_conv3_lds_bytes(a.ks, a.g.wp, a.g.nbuf, a.g.rpg);
        if (a.g.kern == 4) cl.lds_bytes = 0;
        cl.probs_dev = dev;
        t->launches.push_back(cl);
        const size_t li = t->launches.size() - 1;
        t->ops.push_back([t, li](hipStream_t s) { return pn_launch_conv(t->ctx, t->launches[li], s); });
    }
    return PN_OK;
}

// ---- BatchNorm / reductions ----------------------------------------------------------------------------------------------------------
int red_blocks(const pn_trainer *t, long npix, int C, int *ppb) {
    const int PL = 256 / (C / 8);
    long per = std::max<long>((npix + 4 * t->ctx->num_cus - 1) / (4 * t->ctx->num_cus), 8L * PL);     // >= 8 pixels per thread, ~4 blocks per CU
    per = (per + PL - 1) / PL * PL;
    *ppb = (int)per;
    return (int)((npix + per - 1) / per);
}

int need_partial(pn_trainer *t, size_t doubles) { t->partial_doubles = std::max(t->partial_doubles, doubles); return PN_OK; }

struct BnUse { int bn, x, res, y, act; };                 // forward: y = act(bn(x) [+ res])
struct BnBwdUse { int bn, x, dy, y, dx, dres, act; bool has_res; };

// up to three independent BatchNorm layers (the branches of a stage level) per launch: blockIdx.y = layer
void op_bn_fwd(pn_trainer *t, std::vector<BnUse> uses) {
    const int n = (int)uses.size();
    struct Geo { long npix; int C, ppb, nblk; size_t poff; } g[3];
    size_t poff = 0;
    int max_nblk = 0, max_c = 0;
    long max_items = 0;
    for (int i = 0; i < n; ++i) {
        const TxTensor X = t->T[uses[i].x];
        g[i].npix = (long)t->B * X.H * X.W; g[i].C = t->bns[uses[i].bn].C;
        g[i].nblk = red_blocks(t, g[i].npix, g[i].C, &g[i].ppb);
        g[i].poff = poff; poff += (size_t)g[i].nblk * g[i].C * 2;
        max_nblk = std::max(max_nblk, g[i].nblk); max_c = std::max(max_c, g[i].C); max_items = std::max(max_items, g[i].npix * (g[i].C / 8));
    }
    need_partial(t, poff);
    t->ops.push_back([=](hipStream_t s) {
        tx::Multi<tx::RedArgs> r;
        tx::Multi<tx::BnFinArgs> f;
        tx::Multi<tx::BnApplyArgs> a;
        memset(&r, 0, sizeof r); memset(&f, 0, sizeof f); memset(&a, 0, sizeof a);
        for (int i = 0; i < n; ++i) {
            const TxBn &b = t->bns[uses[i].bn];
            const TxTensor X = t->T[uses[i].x], Y = t->T[uses[i].y];
            tx::RedArgs &ri = r.a[i];
            ri.x = X.p; ri.x_cs = X.cs(); ri.x_split = X.split(); ri.C = g[i].C; ri.npix = g[i].npix; ri.ppb = g[i].ppb; ri.nblk = g[i].nblk; ri.partial = t->partial + g[i].poff;
            tx::BnFinArgs &fi = f.a[i];
            fi.partial = t->partial + g[i].poff; fi.nblk = g[i].nblk; fi.C = g[i].C; fi.n = (double)g[i].npix; fi.gamma = b.gamma; fi.beta = b.beta;
            fi.mean = b.mean; fi.invstd = b.invstd; fi.scale = b.scale; fi.shift = b.shift; fi.running_mean = b.rm; fi.running_var = b.rv;
            fi.momentum = t->momentum; fi.eps = t->eps;
            tx::BnApplyArgs &ai = a.a[i];
            ai.x = X.p; ai.x_cs = X.cs(); ai.x_split = X.split();
            if (uses[i].res >= 0) { const TxTensor R = t->T[uses[i].res]; ai.res = R.p; ai.res_cs = R.cs(); ai.res_split = R.split(); }
            ai.y = Y.p; ai.y_cs = Y.cs(); ai.y_split = Y.split(); ai.scale = b.scale; ai.shift = b.shift; ai.act = uses[i].act; ai.C = g[i].C; ai.npix = g[i].npix;
        }
        if (t->f32) hipLaunchKernelGGL((tx::reduce_kernel<0, float>), dim3(max_nblk, n), dim3(256), 0, s, r);
        else hipLaunchKernelGGL((tx::reduce_kernel<0, bf>), dim3(max_nblk, n), dim3(256), 0, s, r);
        hipLaunchKernelGGL(tx::bn_finish_kernel, dim3((max_c + 3) / 4, n), dim3(256), 0, s, f);
        if (t->f32) hipLaunchKernelGGL(tx::bn_apply_kernel<float>, dim3(grid_for(max_items, t->ctx->num_cus), n), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(tx::bn_apply_kernel<bf>, dim3(grid_for(max_items, t->ctx->num_cus), n), dim3(256), 0, s, a);
        PN_HIP_CHECK(t->ctx, hipGetLastError());
        return (int)PN_OK;
    });
}
void op_bn_fwd(pn_trainer *t, int bn, int x, int res, int y, int act) { op_bn_fwd(t, std::vector<BnUse>{{bn, x, res, y, act}}); }

void op_bn_bwd(pn_trainer *t, std::vector<BnBwdUse> uses) {
    const int n = (int)uses.size();
    struct Geo { long npix; int C, ppb, nblk; size_t poff; } g[3];
    size_t poff = 0;
    int max_nblk = 0, max_c = 0;
    long max_items = 0;
    for (int i = 0; i < n; ++i) {
        const TxTensor X = t->T[uses[i].x];
        g[i].npix = (long)t->B * X.H * X.W; g[i].C = t->bns[uses[i].bn].C;
        g[i].nblk = red_blocks(t, g[i].npix, g[i].C, &g[i].ppb);
        g[i].poff = poff; poff += (size_t)g[i].nblk * g[i].C * 2;
        max_nblk = std::max(max_nblk, g[i].nblk); max_c = std::max(max_c, g[i].C); max_items = std::max(max_items, g[i].npix * (g[i].C / 8));
    }
    need_partial(t, poff);
    t->ops.push_back([=](hipStream_t s) {
        tx::Multi<tx::RedArgs> r;
        tx::Multi<tx::BnBwdFinArgs> f;
        tx::Multi<tx::BnBwdApplyArgs> a;
        memset(&r, 0, sizeof r); memset(&f, 0, sizeof f); memset(&a, 0, sizeof a);
        for (int i = 0; i < n; ++i) {
            const BnBwdUse &u = uses[i];
            const TxBn &b = t->bns[u.bn];
            const TxTensor X = t->T[u.x], DY = t->T[u.dy], Y = t->T[u.y], DX = u.dx >= 0 ? t->T[u.dx] : TxTensor();
            // the activation's sign: from the stored output when a residual went into it, else recomputed from x (one tensor less to read)
            const char *ysrc = (u.act && u.has_res) ? Y.p : nullptr;
            tx::RedArgs &ri = r.a[i];
            ri.x = X.p; ri.x_cs = X.cs(); ri.x_split = X.split(); ri.dy = DY.p; ri.dy_cs = DY.cs(); ri.dy_split = DY.split();
            ri.y = ysrc; ri.y_cs = Y.cs(); ri.mean = b.mean; ri.invstd = b.invstd; ri.scale = b.scale; ri.shift = b.shift; ri.act = u.act;
            ri.C = g[i].C; ri.npix = g[i].npix; ri.ppb = g[i].ppb; ri.nblk = g[i].nblk; ri.partial = t->partial + g[i].poff;
            tx::BnBwdFinArgs &fi = f.a[i];
            fi.partial = t->partial + g[i].poff; fi.nblk = g[i].nblk; fi.C = g[i].C; fi.n = (double)g[i].npix; fi.gamma = b.gamma; fi.invstd = b.invstd;
            fi.dgamma = b.dgamma; fi.dbeta = b.dbeta; fi.k1 = b.k1; fi.k2 = b.k2; fi.k3 = b.k3;
            tx::BnBwdApplyArgs &ai = a.a[i];
            ai.x = X.p; ai.x_cs = X.cs(); ai.x_split = X.split(); ai.dy = DY.p; ai.dy_cs = DY.cs(); ai.dy_split = DY.split();
            ai.y = ysrc; ai.y_cs = Y.cs(); ai.mean = b.mean; ai.invstd = b.invstd; ai.k1 = b.k1; ai.k2 = b.k2; ai.k3 = b.k3; ai.scale = b.scale; ai.shift = b.shift;
            ai.dx = DX.p; ai.dx_cs = DX.cs(); ai.dx_split = DX.split();
            if (u.dres >= 0) { const TxTensor R = t->T[u.dres]; ai.dres = R.p; ai.dres_cs = R.cs(); ai.dres_split = R.split(); }
            ai.act = u.act; ai.C = g[i].C; ai.npix = g[i].npix;
        }
        if (t->f32) hipLaunchKernelGGL((tx::reduce_kernel<1, float>), dim3(max_nblk, n), dim3(256), 0, s, r);
        else hipLaunchKernelGGL((tx::reduce_kernel<1, bf>), dim3(max_nblk, n), dim3(256), 0, s, r);
        hipLaunchKernelGGL(tx::bn_bwd_finish_kernel, dim3((max_c + 3) / 4, n), dim3(256), 0, s, f);
        if (uses[0].dx < 0) {                     // sums and k1 / k2 / k3 only: the consumer applies them itself (the stem's weight gradient)
            PN_HIP_CHECK(t->ctx, hipGetLastError());
            return (int)PN_OK;
        }
        if (t->f32) hipLaunchKernelGGL(tx::bn_bwd_apply_kernel<float>, dim3(grid_for(max_items, t->ctx->num_cus), n), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(tx::bn_bwd_apply_kernel<bf>, dim3(grid_for(max_items, t->ctx->num_cus), n), dim3(256), 0, s, a);
        PN_HIP_CHECK(t->ctx, hipGetLastError());
        return (int)PN_OK;
    });
}
void op_bn_bwd(pn_trainer *t, int bn, int x, int dy, int y, int dx, int dres, int act, bool has_res) {
    op_bn_bwd(t, std::vector<BnBwdUse>{{bn, x, dy, y, dx, dres, act, has_res}});
}

// bias gradient of layer `l` from its output gradient tensor dy (first cout channels)
void op_dbias(pn_trainer *t, int l, int dy) {
    const TxTensor DY = t->T[dy];
    const long npix = (long)t->B * DY.H * DY.W;
    const int C = DY.plane;
    int ppb;
    const int nblk = red_blocks(t, npix, C, &ppb);
    need_partial(t, (size_t)nblk * C * 2);
    t->ops.push_back([=](hipStream_t s) {
        const TxLayer &L = t->layers[l];
        tx::Multi<tx::RedArgs> r;
        memset(&r, 0, sizeof r);
        tx::RedArgs &ri = r.a[0];
        ri.x = DY.p; ri.x_cs = DY.cs(); ri.x_split = DY.split(); ri.C = C; ri.npix = npix; ri.ppb = ppb; ri.nblk = nblk; ri.partial = t->partial;
        if (t->f32) hipLaunchKernelGGL((tx::reduce_kernel<2, float>), dim3(nblk, 1), dim3(256), 0, s, r);
        else hipLaunchKernelGGL((tx::reduce_kernel<2, bf>), dim3(nblk, 1), dim3(256), 0, s, r);
        hipLaunchKernelGGL(tx::sum_finish_kernel, dim3((L.cout + 3) / 4), dim3(256), 0, s, (const double *)t->partial, nblk, C, L.cout, L.db);
        PN_HIP_CHECK(t->ctx, hipGetLastError());
        return (int)PN_OK;
    });
}

void op_add(pn_trainer *t, std::vector<std::array<int, 2>> ins /* (tensor, channel offset) */, int C, int out) {
    const TxTensor O = t->T[out];
    const long npix = (long)t->B * O.H * O.W;
    tx::AddArgs a;
    memset(&a, 0, sizeof a);
    a.n = (int)ins.size();
    for (int i = 0; i < a.n; ++i) { const TxTensor X = t->T[ins[i][0]]; a.in[i] = X.at(ins[i][1]); a.cs[i] = X.cs(); a.split[i] = X.split(); }
    a.out = O.p; a.out_cs = O.cs(); a.out_split = O.split(); a.C = C; a.npix = npix;
    t->ops.push_back([=](hipStream_t s) {
        if (t->f32) hipLaunchKernelGGL(tx::add_kernel<float>, dim3(grid_for(npix * (C / 8), t->ctx->num_cus)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(tx::add_kernel<bf>, dim3(grid_for(npix * (C / 8), t->ctx->num_cus)), dim3(256), 0, s, a);
        PN_HIP_CHECK(t->ctx, hipGetLastError());
        return (int)PN_OK;
    });
}

void op_pool_fwd(pn_trainer *t, int x, int y, int out_coff) {
    const TxTensor X = t->T[x], Y = t->T[y];
    t->ops.push_back([=](hipStream_t s) {
        return pn_launch_pool(t->ctx, t->f32 ? PN_PREC_F32 : PN_PREC_BF16, 0, X.p, Y.p, t->B, X.H, X.W, X.plane, X.cs(), Y.cs(), out_coff, X.split(), Y.split(), s);
    });
}

void op_pool_bwd(pn_trainer *t, int dy, int dx) {
    const TxTensor DY = t->T[dy], DX = t->T[dx];
    t->ops.push_back([=](hipStream_t s) {
        const long items = (long)t->B * DX.H * DX.W * (DX.plane / 8);
        if (t->f32) hipLaunchKernelGGL(tx::avgpool_bwd_kernel<float>, dim3(grid_for(items, t->ctx->num_cus)), dim3(256), 0, s, (const float *)DY.p, DY.cs(), 0, (float *)DX.p, DX.cs(), 0,
                                       t->B, DX.H, DX.W, DY.H, DY.W, DX.plane);
        else hipLaunchKernelGGL(tx::avgpool_bwd_kernel<bf>, dim3(grid_for(items, t->ctx->num_cus)), dim3(256), 0, s, (const bf *)DY.p, DY.cs(), DY.plane, (bf *)DX.p, DX.cs(), DX.plane,
                                t->B, DX.H, DX.W, DY.H, DY.W, DX.plane);
        PN_HIP_CHECK(t->ctx, hipGetLastError());
        return (int)PN_OK;
    });
}

// loss terms and head gradients of a stage's three heads (one launch); dcat = the stage-2 input gradient whose slices reach the stage-1 heads (or -1)
void op_heads(pn_trainer *t, int stage, int dcat, const int dv[3]) {
    int nblk[3], HW = 0;
    long total[3];
    size_t poff[3], off = 0;
    int max_nblk = 0;
    TxTensor DV[3];
    for (int b = 0; b < 3; ++b) {
        DV[b] = t->T[dv[b]];
        HW = DV[b].H * DV[b].W;
        total[b] = (long)t->B * HEAD_C[b] * HW;
        nblk[b] = (int)((total[b] + 255) / 256);
        poff[b] = off; off += (size_t)nblk[b];
        max_nblk = std::max(max_nblk, nblk[b]);
    }
    need_partial(t, off);
    t->ops.push_back([=](hipStream_t s) {
        tx::Multi<tx::HeadArgs> h;
        tx::Multi<tx::LossFinArgs> f;
        memset(&h, 0, sizeof h); memset(&f, 0, sizeof f);
        for (int b = 0; b < 3; ++b) {
            tx::HeadArgs &a = h.a[b];
            a.out = t->head_out[stage][b]; a.target = t->target[b]; a.fg = b == 2 ? t->fg : nullptr;
            if (dcat >= 0) { const TxTensor D = t->T[dcat]; a.dextra = D.at(CAT_OFF[b]); a.de_cs = D.cs(); a.de_split = D.split(); }
            a.dv = DV[b].p; a.dv_cs = DV[b].cs(); a.dv_split = DV[b].split();
            a.kind = HEAD_KIND[b]; a.C = HEAD_C[b]; a.HW = HW; a.total = total[b]; a.inv_numel = (float)(1.0 / (double)total[b]); a.partial = t->partial + poff[b];
            f.a[b].partial = t->partial + poff[b]; f.a[b].nblocks = nblk[b]; f.a[b].numel = (double)total[b]; f.a[b].loss = t->loss + 3 * stage + b;
        }
        if (t->f32) hipLaunchKernelGGL(tx::head_kernel<float>, dim3(max_nblk, 3), dim3(256), 0, s, h);
        else hipLaunchKernelGGL(tx::head_kernel<bf>, dim3(max_nblk, 3), dim3(256), 0, s, h);
        hipLaunchKernelGGL(tx::loss_finish_kernel, dim3(3), dim3(256), 0, s, f);
        PN_HIP_CHECK(t->ctx, hipGetLastError());
        return (int)PN_OK;
    });
}

// fork: the side stream may start once everything issued so far on the step's stream has finished; join: the other way round
int op_fork(pn_trainer *t, int side = 0) {
    if (!t->two_streams || (t->nside == 1 && t->ops.size() == t->side_tail)) return PN_OK;
    hipEvent_t ev;
    PN_HIP_CHECK(t->ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    t->events.push_back(ev);
    t->ops.push_back([t, ev, side](hipStream_t s) {
        PN_HIP_CHECK(t->ctx, hipEventRecord(ev, s));
        PN_HIP_CHECK(t->ctx, hipStreamWaitEvent(t->side[side], ev, 0));
        return (int)PN_OK;
    });
    return PN_OK;
}
int op_join(pn_trainer *t) {
    if (!t->two_streams) return PN_OK;
    for (int k = 0; k < t->nside; ++k) {
        hipEvent_t ev;
        PN_HIP_CHECK(t->ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        t->events.push_back(ev);
        t->ops.push_back([t, ev, k](hipStream_t s) {
            PN_HIP_CHECK(t->ctx, hipEventRecord(ev, t->side[k]));
            PN_HIP_CHECK(t->ctx, hipStreamWaitEvent(s, ev, 0));
            return (int)PN_OK;
        });
    }
    return PN_OK;
}
// every op appended since `from` runs on side stream `side`
void ops_to_side(pn_trainer *t, size_t from, int side = 0) {
    if (!t->two_streams) return;
    for (size_t i = from; i < t->ops.size(); ++i) {
        auto f = t->ops[i];
        t->ops[i] = [t, f, side](hipStream_t) { return f(t->side[side]); };
    }
    t->side_tail = t->ops.size();
}

// weight gradient of layer l: x = its input tensor, dy = gradient of its output
int op_wgrad(pn_trainer *t, int l, int dy) {
    const TxLayer &L = t->layers[l];
    const TxTensor X = t->T[L.x], DY = t->T[dy];
    if (t->legacy_wgrad) {
        // stepping stone / cross-check (POPNET_TRAINX_WGRAD=legacy): hand both operands to train.hip's NCHW f32 weight-gradient kernels
        const size_t need = (size_t)t->B * X.H * X.W * std::max(L.cin, L.cout);
        t->nchw_elems = std::max(t->nchw_elems, need);
        t->ops.push_back([=](hipStream_t s) {
            const TxLayer &LL = t->layers[l];
            const int HW = X.H * X.W;
            const int *cmap = LL.cat ? (const int *)t->cat_ref_map : (const int *)nullptr;
            if (t->f32) {
                hipLaunchKernelGGL(tx::planes_to_nchw_kernel<float>, dim3((HW + 63) / 64, (LL.cin + 63) / 64, t->B), dim3(256), 0, s, (const float *)X.p, X.cs(), 0, t->nchw_a, LL.cin, HW, cmap);
                hipLaunchKernelGGL(tx::planes_to_nchw_kernel<float>, dim3((HW + 63) / 64, (LL.cout + 63) / 64, t->B), dim3(256), 0, s, (const float *)DY.p, DY.cs(), 0, t->nchw_b, LL.cout, HW, (const int *)nullptr);
            } else {
                hipLaunchKernelGGL(tx::planes_to_nchw_kernel<bf>, dim3((HW + 63) / 64, (LL.cin + 63) / 64, t->B), dim3(256), 0, s, (const bf *)X.p, X.cs(), X.plane, t->nchw_a, LL.cin, HW, cmap);
                hipLaunchKernelGGL(tx::planes_to_nchw_kernel<bf>, dim3((HW + 63) / 64, (LL.cout + 63) / 64, t->B), dim3(256), 0, s, (const bf *)DY.p, DY.cs(), DY.plane, t->nchw_b, LL.cout, HW, (const int *)nullptr);
            }
            PN_HIP_CHECK(t->ctx, hipGetLastError());
            return pn_conv2d_wgrad(t->ctx, t->nchw_a, t->nchw_b, LL.dw, LL.db, t->B, LL.cin, X.H, X.W, LL.cout, LL.ks, 1, LL.ks / 2, (void *)s);
        });
        return PN_OK;
    }
    if (L.b && !L.bn_follows) op_dbias(t, l, dy);      // (a bias in front of a BatchNorm: its gradient is identically zero and the flat gradient buffer already holds 0)
    const int side = t->two_streams ? t->next_side : 0;
    t->next_side = (t->next_side + 1) % t->nside;
    if (int rc = op_fork(t, side)) return rc;
    const size_t from = t->ops.size();
    if (t->f32) {
        if (int rc = tx::plan_wgrad_f32(t->ctx, t->B, X.H, X.W, (const float *)X.p, X.plane, (const float *)DY.p, DY.plane, L.cin, L.cout, L.ks, L.cat ? t->cat_k_map : nullptr, L.dw,
                                        &t->wg_partial[side], &t->wg_partial_floats, t->ops))
            return rc;
    } else if (int rc = tx::plan_wgrad(t->ctx, t->B, X.H, X.W, (const bf *)X.p, X.plane, (const bf *)DY.p, DY.plane, L.cin, L.cout, L.ks, L.cat ? t->cat_k_map : nullptr, L.dw,
                                       &t->wg_partial[side], &t->wg_partial_floats, t->ops))
        return rc;
    ops_to_side(t, from, side);
    return PN_OK;
}

int build(pn_trainer *t) {
    pn_ctx *ctx = t->ctx;
    const int B = t->B, H = t->H, W = t->W;
    if (H % 8 || W % 8) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "pn_trainer: input size must be a multiple of 8");
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    int rc;
    if ((rc = tx_alloc(t, (void **)&t->zero_bias, 1024 * 4, true))) return rc;
    {   // channel maps of the stage-2 input: mine [feat 0..127 | paf 128.. | heat 156.. | z 172.. | pad] <-> reference [paf, heat, z, feat] (rtpose_light3d.py:339)
        std::vector<int> k(CAT_PLANE, -1), r(187, -1);
        for (int i = 0; i < 128; ++i) k[i] = 59 + i;
        for (int i = 0; i < 28; ++i) k[128 + i] = i;
        for (int i = 0; i < 16; ++i) k[156 + i] = 28 + i;
        for (int i = 0; i < 15; ++i) k[172 + i] = 44 + i;
        for (int i = 0; i < CAT_PLANE; ++i) if (k[i] >= 0) r[k[i]] = i;
        if ((rc = tx_alloc(t, (void **)&t->cat_k_map, k.size() * 4, false))) return rc;
        if ((rc = tx_alloc(t, (void **)&t->cat_ref_map, r.size() * 4, false))) return rc;
        PN_HIP_CHECK(ctx, hipMemcpy(t->cat_k_map, k.data(), k.size() * 4, hipMemcpyHostToDevice));
        PN_HIP_CHECK(ctx, hipMemcpy(t->cat_ref_map, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    }
    auto TT = [&](int h, int w, int plane, int *id) { return new_tensor(t, h, w, plane, id); };
#define TX(expr) do { if ((rc = (expr))) return rc; } while (0)

    // ---------------- forward ----------------
    // step prologue: every weight pack and padded bias from the live parameters -- on the side stream, under the stem (which reads the raw parameters)
    TX(op_fork(t));
    const size_t pack_from = t->ops.size();
    t->ops.push_back([t](hipStream_t s) {
        if (t->pack_groups && t->pack_gather) hipLaunchKernelGGL(tx::pack_kernel, dim3((t->pack_groups + 255) / 256), dim3(256), 0, s, (const tx::PackDesc *)t->packs_dev, (int)t->packs.size(), t->pack_groups);
        else if (t->pack_units) hipLaunchKernelGGL(tx::pack_rows_kernel, dim3((t->pack_units + 255) / 256), dim3(256), 0, s, (const tx::PackDesc *)t->packs_dev, (int)t->packs.size(), t->pack_units);
        if (!t->biases.empty()) hipLaunchKernelGGL(tx::bias_kernel, dim3((unsigned)t->biases.size()), dim3(128), 0, s, (const tx::BiasDesc *)t->biases_dev, (int)t->biases.size());
        PN_HIP_CHECK(t->ctx, hipGetLastError());
        return (int)PN_OK;
    });
    ops_to_side(t, pack_from);
    // stem: model0.conv1 7x7 / 2 on the NCHW f32 image (train.hip), handed over as planes
    int C0, A0;
    TX(TT(H2, W2, 64, &C0)); TX(TT(H2, W2, 64, &A0));
    const float *w_stem = nullptr; float *dw_stem = nullptr;
    TX(find_param(t, "model0.conv1.weight", 64 * 49, &w_stem, &dw_stem));
    t->nchw_elems = std::max(t->nchw_elems, (size_t)B * 64 * H2 * W2);
    {
        const TxTensor c0 = t->T[C0];
        const bool handover = getenv("POPNET_TRAINX_STEM_HANDOVER") != nullptr;      // the NCHW f32 hand-over of the round's first builds (bit-identical; A/B)
        const char *esf = getenv("POPNET_TRAINX_STEM_FWD");
        const int stem_gather = esf && !strcmp(esf, "gather");                        // A/B: tconv_fwd_kernel<7, PL> instead of tstem_fwd_kernel (bit-identical)
        t->ops.push_back([=](hipStream_t s) {
            if (!handover) return pn_stem_forward_planes(t->ctx, t->img, w_stem, c0.p, c0.cs(), c0.split(), t->f32, B, 1, H, W, 64, 7, 2, 3, stem_gather, s);
            if (int r = pn_conv2d_forward(t->ctx, t->img, w_stem, nullptr, t->nchw_a, B, 1, H, W, 64, 7, 2, 3, 0, (void *)s)) return r;
            const int HW = H2 * W2;
            if (t->f32) hipLaunchKernelGGL(tx::nchw_to_planes_kernel<float>, dim3((HW + 63) / 64, 1, B), dim3(256), 0, s, (const float *)t->nchw_a, (float *)c0.p, 64, HW, c0.cs(), 0);
            else hipLaunchKernelGGL(tx::nchw_to_planes_kernel<bf>, dim3((HW + 63) / 64, 1, B), dim3(256), 0, s, (const float *)t->nchw_a, (bf *)c0.p, 64, HW, c0.cs(), c0.plane);
            PN_HIP_CHECK(t->ctx, hipGetLastError());
            return (int)PN_OK;
        });
    }
    int bn_stem;
    TX(new_bn(t, "model0.bn1", 64, &bn_stem));
    op_bn_fwd(t, bn_stem, C0, -1, A0, 1);
    TX(op_join(t));                               // the packs are ready before the first planes convolution

    struct Block { int l1, l2, lds, bn1, bn2, bnds, C1, A1, C2, CD, D, in, out; };
    auto basic_block = [&](const std::string &p, int xin, int cin, int cout, int h, int w, bool down, Block *bk) -> int {
        int rc2;
        Block b;
        memset(&b, 0, sizeof b);
        b.in = xin; b.lds = b.bnds = b.CD = b.D = -1;
        if ((rc2 = TT(h, w, cout, &b.C1)) || (rc2 = TT(h, w, cout, &b.A1)) || (rc2 = TT(h, w, cout, &b.C2)) || (rc2 = TT(h, w, cout, &b.out))) return rc2;
        if ((rc2 = new_layer(t, p + ".conv1", 3, cin, cout, false, xin, false, &b.l1))) return rc2;
        if ((rc2 = new_bn(t, p + ".bn1", cout, &b.bn1))) return rc2;
        if ((rc2 = new_layer(t, p + ".conv2", 3, cout, cout, false, b.A1, false, &b.l2))) return rc2;
        if ((rc2 = new_bn(t, p + ".bn2", cout, &b.bn2))) return rc2;
        std::vector<ConvUse> lv = {{b.l1, false, xin, b.C1, 0, -1, PN_ACT_NONE, nullptr}};
        if (down) {
            if ((rc2 = TT(h, w, cout, &b.CD)) || (rc2 = TT(h, w, cout, &b.D))) return rc2;
            if ((rc2 = new_layer(t, p + ".downsample.0", 1, cin, cout, false, xin, false, &b.lds))) return rc2;
            if ((rc2 = new_bn(t, p + ".downsample.1", cout, &b.bnds))) return rc2;
            lv.push_back({b.lds, false, xin, b.CD, 0, -1, PN_ACT_NONE, nullptr});
        }
        if ((rc2 = add_conv_group(t, lv))) return rc2;
        op_bn_fwd(t, b.bn1, b.C1, -1, b.A1, 1);
        if (down) op_bn_fwd(t, b.bnds, b.CD, -1, b.D, 0);
        if ((rc2 = add_conv_group(t, {{b.l2, false, b.A1, b.C2, 0, -1, PN_ACT_NONE, nullptr}}))) return rc2;
        op_bn_fwd(t, b.bn2, b.C2, down ? b.D : xin, b.out, 1);
        *bk = b;
        return PN_OK;
    };
    Block b10, b11, b20;
    TX(basic_block("model0.layer1.0", A0, 64, 64, H2, W2, false, &b10));
    TX(basic_block("model0.layer1.1", b10.out, 64, 64, H2, W2, false, &b11));
    int P1;
    TX(TT(H4, W4, 64, &P1));
    op_pool_fwd(t, b11.out, P1, 0);
    TX(basic_block("model0.layer2.0", P1, 64, 128, H4, W4, true, &b20));
    int l_c2, bn_c2, C7, A7, FEAT, CAT;
    TX(TT(H4, W4, 128, &C7)); TX(TT(H4, W4, 128, &A7)); TX(TT(H8, W8, 128, &FEAT)); TX(TT(H8, W8, CAT_PLANE, &CAT));
    TX(new_layer(t, "model0.conv2", 1, 128, 128, false, b20.out, false, &l_c2));
    TX(new_bn(t, "model0.bn2", 128, &bn_c2));
    TX(add_conv_group(t, {{l_c2, false, b20.out, C7, 0, -1, PN_ACT_NONE, nullptr}}));
    op_bn_fwd(t, bn_c2, C7, -1, A7, 1);
    op_pool_fwd(t, A7, FEAT, 0);              // feat: the stage-1 input ...
    op_pool_fwd(t, A7, CAT, 0);               // ... and channels 0..127 of the stage-2 input (torch.cat is never executed)

    // stages (make_stages, rtpose_light3d.py:222-246, 263-309): conv BN LeakyReLU x 4, bare conv; three branches per level share a launch
    const int BR_C[3][4] = {{256, 256, 256, 128}, {128, 128, 128, 128}, {128, 64, 64, 64}};
    const int BR_KS[3][5] = {{3, 3, 3, 1, 1}, {3, 3, 3, 3, 3}, {3, 3, 3, 3, 3}};
    struct Branch { int l[5], bn[4], C[4], A[4]; };
    Branch br[2][3];
    for (int st = 0; st < 2; ++st) {
        const int xin = st == 0 ? FEAT : CAT;
        for (int b = 0; b < 3; ++b)
            for (int lv = 0; lv < 4; ++lv) { TX(TT(H8, W8, BR_C[b][lv], &br[st][b].C[lv])); TX(TT(H8, W8, BR_C[b][lv], &br[st][b].A[lv])); }
        for (int lv = 0; lv < 5; ++lv) {
            std::vector<ConvUse> uses;
            for (int b = 0; b < 3; ++b) {
                char nm[48];
                snprintf(nm, sizeof nm, "model%d_%d.%d", st + 1, b + 1, 3 * lv);
                const int in = lv == 0 ? xin : br[st][b].A[lv - 1];
                const int cin = lv == 0 ? (st == 0 ? 128 : 187) : BR_C[b][lv - 1];
                const int cout = lv < 4 ? BR_C[b][lv] : HEAD_C[b];
                TX(new_layer(t, nm, BR_KS[b][lv], cin, cout, true, in, lv == 0 && st == 1, &br[st][b].l[lv]));
                if (lv < 4) {
                    t->layers[br[st][b].l[lv]].bn_follows = true;
                    snprintf(nm, sizeof nm, "model%d_%d.%d", st + 1, b + 1, 3 * lv + 1);
                    TX(new_bn(t, nm, cout, &br[st][b].bn[lv]));
                    uses.push_back({br[st][b].l[lv], false, in, br[st][b].C[lv], 0, -1, PN_ACT_NONE, nullptr});
                } else {
                    TX(tx_alloc(t, (void **)&t->head_out[st][b], (size_t)B * cout * H8 * W8 * 4, true));
                    // the head's sigmoid range cast runs in the convolution epilogue; stage 1 also writes its slice of the stage-2 input
                    uses.push_back({br[st][b].l[lv], false, in, st == 0 ? CAT : -1, st == 0 ? CAT_OFF[b] : 0, -1, HEAD_ACT[b], t->head_out[st][b]});
                }
            }
            TX(add_conv_group(t, uses));
            if (lv < 4) {
                std::vector<BnUse> bu;
                for (int b = 0; b < 3; ++b) bu.push_back({br[st][b].bn[lv], br[st][b].C[lv], -1, br[st][b].A[lv], 2});
                op_bn_fwd(t, bu);
            }
        }
    }

    // ---------------- backward ----------------
    auto stage_bwd = [&](int st, int dcat_in /* -1 for stage 2 */, int dx_plane, int *dx_out /* three tensors */) -> int {
        int rc2;
        int dy[3];
        for (int b = 0; b < 3; ++b)
            if ((rc2 = TT(H8, W8, 64, &dy[b]))) return rc2;
        op_heads(t, st, dcat_in, dy);
        for (int lv = 4; lv >= 0; --lv) {
            int dc[3];
            std::vector<BnBwdUse> bu;
            for (int b = 0; b < 3; ++b) {
                if (lv < 4) {
                    if ((rc2 = TT(H8, W8, BR_C[b][lv], &dc[b]))) return rc2;
                    bu.push_back({br[st][b].bn[lv], br[st][b].C[lv], dy[b], br[st][b].A[lv], dc[b], -1, 2, false});
                } else dc[b] = dy[b];
            }
            if (!bu.empty()) op_bn_bwd(t, bu);
            for (int b = 0; b < 3; ++b)
                if ((rc2 = op_wgrad(t, br[st][b].l[lv], dc[b]))) return rc2;
            std::vector<ConvUse> uses;
            for (int b = 0; b < 3; ++b) {
                const int plane = lv == 0 ? dx_plane : BR_C[b][lv - 1];
                if ((rc2 = TT(H8, W8, plane, &dy[b]))) return rc2;
                uses.push_back({br[st][b].l[lv], true, dc[b], dy[b], 0, -1, PN_ACT_NONE, nullptr});
            }
            if ((rc2 = add_conv_group(t, uses))) return rc2;
        }
        for (int b = 0; b < 3; ++b) dx_out[b] = dy[b];
        return PN_OK;
    };
    int dcat_b[3], dfeat_b[3], DCAT, DFEAT;
    TX(stage_bwd(1, -1, CAT_PLANE, dcat_b));
    TX(TT(H8, W8, CAT_PLANE, &DCAT));
    op_add(t, {{dcat_b[0], 0}, {dcat_b[1], 0}, {dcat_b[2], 0}}, CAT_PLANE, DCAT);
    TX(stage_bwd(0, DCAT, 128, dfeat_b));
    TX(TT(H8, W8, 128, &DFEAT));
    op_add(t, {{DCAT, 0}, {dfeat_b[0], 0}, {dfeat_b[1], 0}, {dfeat_b[2], 0}}, 128, DFEAT);
    int dA7, dC7, dA6;
    TX(TT(H4, W4, 128, &dA7)); TX(TT(H4, W4, 128, &dC7)); TX(TT(H4, W4, 128, &dA6));
    op_pool_bwd(t, DFEAT, dA7);
    op_bn_bwd(t, bn_c2, C7, dA7, A7, dC7, -1, 1, false);
    TX(op_wgrad(t, l_c2, dC7));
    TX(add_conv_group(t, {{l_c2, true, dC7, dA6, 0, -1, PN_ACT_NONE, nullptr}}));

    auto block_bwd = [&](const Block &b, int dout, int h, int w, int cin, int cout, int *din, int scratch[6]) -> int {
        int rc2;
        // scratch: dC2, g (identity gradient / dD), dA1, dC1, dCD, tmp  (reused between the two 112x112 blocks)
        int dC2 = scratch[0], g = scratch[1], dA1 = scratch[2], dC1 = scratch[3];
        op_bn_bwd(t, b.bn2, b.C2, dout, b.out, dC2, g, 1, true);
        if ((rc2 = op_wgrad(t, b.l2, dC2))) return rc2;
        if ((rc2 = add_conv_group(t, {{b.l2, true, dC2, dA1, 0, -1, PN_ACT_NONE, nullptr}}))) return rc2;
        op_bn_bwd(t, b.bn1, b.C1, dA1, b.A1, dC1, -1, 1, false);
        if ((rc2 = op_wgrad(t, b.l1, dC1))) return rc2;
        if (b.lds >= 0) {
            int dCD = scratch[4], tmp = scratch[5];
            op_bn_bwd(t, b.bnds, b.CD, g, b.D, dCD, -1, 0, false);
            if ((rc2 = op_wgrad(t, b.lds, dCD))) return rc2;
            if ((rc2 = add_conv_group(t, {{b.lds, true, dCD, tmp, 0, -1, PN_ACT_NONE, nullptr}}))) return rc2;
            if ((rc2 = add_conv_group(t, {{b.l1, true, dC1, *din, 0, tmp, PN_ACT_NONE, nullptr}}))) return rc2;     // dx = dgrad(conv1) + dgrad(shortcut)
        } else {
            if ((rc2 = add_conv_group(t, {{b.l1, true, dC1, *din, 0, g, PN_ACT_NONE, nullptr}}))) return rc2;       // dx = dgrad(conv1) + identity gradient
        }
        (void)h; (void)w; (void)cin; (void)cout;
        return PN_OK;
    };
    int s56[6], dP1;
    for (int i = 0; i < 4; ++i) TX(TT(H4, W4, 128, &s56[i]));
    TX(TT(H4, W4, 128, &s56[4])); TX(TT(H4, W4, 64, &s56[5]));
    TX(TT(H4, W4, 64, &dP1));
    TX(block_bwd(b20, dA6, H4, W4, 64, 128, &dP1, s56));
    // (own scratch per block: a weight gradient still running on the side stream reads dC2 / dC1 of its block)
    // the stem's BatchNorm backward is applied inside its weight-gradient kernel (train.hip::tstem_wgrad_kernel): dC0 never exists
    const bool stem_handover = getenv("POPNET_TRAINX_STEM_HANDOVER") != nullptr;      // A/B: the NCHW f32 hand-over of the round's first builds (bit-identical)
    const char *esb = getenv("POPNET_TRAINX_STEM_BN");
    const bool stem_bn_separate = stem_handover || (esb && !strcmp(esb, "separate"));  // A/B: bn_bwd_apply_kernel writes dC0, the weight gradient reads it (bit-identical)
    const char *esd = getenv("POPNET_TRAINX_STEM_DEPTH");
    const int stem_depth = esd ? atoi(esd) : 2;                                         // A/B: chunks of 32 pixels in flight per block (1, 2, 4; bit-identical): 102-163 / 83-91 / 105 us over six traces
    int s112[6], s112b[6], dA4, dA2, dA0, dC0 = -1;
    for (int i = 0; i < 4; ++i) { TX(TT(H2, W2, 64, &s112[i])); TX(TT(H2, W2, 64, &s112b[i])); }
    s112[4] = s112[5] = s112b[4] = s112b[5] = -1;
    TX(TT(H2, W2, 64, &dA4)); TX(TT(H2, W2, 64, &dA2));
    if (stem_bn_separate) TX(TT(H2, W2, 64, &dC0));
    op_pool_bwd(t, dP1, dA4);
    TX(block_bwd(b11, dA4, H2, W2, 64, 64, &dA2, s112));
    dA0 = dA4;                                   // free again: layer1.1's output gradient has been consumed (by launches of the step's own stream)
    TX(block_bwd(b10, dA2, H2, W2, 64, 64, &dA0, s112b));
    op_bn_bwd(t, bn_stem, C0, dA0, A0, stem_bn_separate ? dC0 : -1, -1, 1, false);
    {
        const TxTensor d0 = dC0 >= 0 ? t->T[dC0] : TxTensor(), a0 = t->T[dA0], c0 = t->T[C0];
        t->ops.push_back([=](hipStream_t s) {
            if (!stem_bn_separate) {
                const TxBn &b = t->bns[bn_stem];
                PnStemBn sb;
                sb.x = c0.p; sb.x_cs = c0.cs(); sb.x_split = c0.split();
                sb.mean = b.mean; sb.invstd = b.invstd; sb.k1 = b.k1; sb.k2 = b.k2; sb.k3 = b.k3; sb.scale = b.scale; sb.shift = b.shift; sb.act = 1;
                return pn_stem_wgrad_planes(t->ctx, t->img, a0.p, a0.cs(), a0.split(), t->f32, &sb, dw_stem, B, 1, H, W, 64, 7, 2, 3, stem_depth, s);
            }
            if (!stem_handover) return pn_stem_wgrad_planes(t->ctx, t->img, d0.p, d0.cs(), d0.split(), t->f32, nullptr, dw_stem, B, 1, H, W, 64, 7, 2, 3, stem_depth, s);
            const int HW = H2 * W2;
            if (t->f32) hipLaunchKernelGGL(tx::planes_to_nchw_kernel<float>, dim3((HW + 63) / 64, 1, B), dim3(256), 0, s, (const float *)d0.p, d0.cs(), 0, t->nchw_b, 64, HW, (const int *)nullptr);
            else hipLaunchKernelGGL(tx::planes_to_nchw_kernel<bf>, dim3((HW + 63) / 64, 1, B), dim3(256), 0, s, (const bf *)d0.p, d0.cs(), d0.plane, t->nchw_b, 64, HW, (const int *)nullptr);
            PN_HIP_CHECK(t->ctx, hipGetLastError());
            return pn_conv2d_wgrad(t->ctx, t->img, t->nchw_b, dw_stem, nullptr, B, 1, H, W, 64, 7, 2, 3, (void *)s);
        });
    }
    TX(op_join(t));                               // the step ends when the last weight gradient has landed
#undef TX
    // scratch and descriptor tables
    if ((rc = tx_alloc(t, (void **)&t->partial, std::max<size_t>(t->partial_doubles, 1) * 8, true))) return rc;
    if ((rc = tx_alloc(t, (void **)&t->nchw_a, std::max<size_t>(t->nchw_elems, 1) * 4, true))) return rc;
    if ((rc = tx_alloc(t, (void **)&t->nchw_b, std::max<size_t>(t->nchw_elems, 1) * 4, true))) return rc;
    if (t->wg_partial_floats)
        for (int k = 0; k < t->nside; ++k)
            if ((rc = tx_alloc(t, (void **)&t->wg_partial[k], t->wg_partial_floats * 4, true))) return rc;
    if (!t->packs.empty()) {
        if ((rc = tx_alloc(t, (void **)&t->packs_dev, t->packs.size() * sizeof(tx::PackDesc), false))) return rc;
        PN_HIP_CHECK(ctx, hipMemcpy(t->packs_dev, t->packs.data(), t->packs.size() * sizeof(tx::PackDesc), hipMemcpyHostToDevice));
    }
    if (!t->biases.empty()) {
        if ((rc = tx_alloc(t, (void **)&t->biases_dev, t->biases.size() * sizeof(tx::BiasDesc), false))) return rc;
        PN_HIP_CHECK(ctx, hipMemcpy(t->biases_dev, t->biases.data(), t->biases.size() * sizeof(tx::BiasDesc), hipMemcpyHostToDevice));
    }
    PN_HIP_CHECK(ctx, hipDeviceSynchronize());
    return PN_OK;
}

}  // namespace

extern "C" {

pn_trainer *pn_trainer_create(pn_ctx *ctx) {
    if (!ctx) return nullptr;
    pn_trainer *t = new pn_trainer();
    t->ctx = ctx;
    return t;
}

void pn_trainer_destroy(pn_trainer *t) {
    if (!t) return;
    for (void *p : t->allocs) (void)hipFree(p);
    for (hipEvent_t ev : t->events) (void)hipEventDestroy(ev);
    for (int k = 0; k < 4; ++k)
        if (t->side[k]) (void)hipStreamDestroy(t->side[k]);
    delete t;
}

int pn_trainer_set_param(pn_trainer *t, const char *name, size_t offset, size_t numel) {
    if (!t || !name) return PN_ERR_INVALID;
    if (t->finalized) return pn_set_error(t->ctx, PN_ERR_STATE, "pn_trainer: already finalized");
    t->params[name] = {offset, numel};
    return PN_OK;
}

int pn_trainer_set_stat(pn_trainer *t, const char *name, float *stat_dev) {
    if (!t || !name || !stat_dev) return PN_ERR_INVALID;
    if (t->finalized) return pn_set_error(t->ctx, PN_ERR_STATE, "pn_trainer: already finalized");
    t->stats[name] = stat_dev;
    return PN_OK;
}

int pn_trainer_set_precision(pn_trainer *t, int precision) {
    if (!t) return PN_ERR_INVALID;
    if (t->finalized) return pn_set_error(t->ctx, PN_ERR_STATE, "pn_trainer: already finalized");
    if (precision != PN_PREC_F32 && precision != PN_PREC_BF16X3) return pn_set_error(t->ctx, PN_ERR_INVALID, "pn_trainer_set_precision: PN_PREC_F32 or PN_PREC_BF16X3");
    t->f32 = precision == PN_PREC_F32;
    return PN_OK;
}

int pn_trainer_finalize(pn_trainer *t, float *flat_param_dev, float *flat_grad_dev, int B, int H, int W, float bn_momentum, float bn_eps) {
    if (!t) return PN_ERR_INVALID;
    pn_ctx *ctx = t->ctx;
    if (t->finalized) return pn_set_error(ctx, PN_ERR_STATE, "pn_trainer: already finalized");
    if (!flat_param_dev || !flat_grad_dev || B < 1 || H < 8 || W < 8) return pn_set_error(ctx, PN_ERR_INVALID, "pn_trainer_finalize: bad arguments");
    PN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    t->flat_p = flat_param_dev; t->flat_g = flat_grad_dev; t->B = B; t->H = H; t->W = W; t->momentum = bn_momentum; t->eps = bn_eps;
    const char *e = getenv("POPNET_TRAINX_WGRAD");
    t->legacy_wgrad = e && !strcmp(e, "legacy");
    const char *epk = getenv("POPNET_TRAINX_PACK");
    t->pack_gather = epk && !strcmp(epk, "gather");
    const char *e2 = getenv("POPNET_TRAINX_STREAMS");
    t->two_streams = !(e2 && atoi(e2) == 1) && !t->legacy_wgrad;
    if (const char *e3 = getenv("POPNET_TRAINX_SIDES")) t->nside = std::max(1, std::min(4, atoi(e3)));
    if (!t->two_streams) t->nside = 1;
    if (t->two_streams) {
        // LOWEST priority: the weight gradients fill what the step's own stream (the dependency chain that bounds the step) leaves idle, never the other way round
        int lo = 0, hi = 0;
        PN_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char *ep = getenv("POPNET_TRAINX_SIDE_PRIORITY");       // experiments: "default" = no priority
        for (int k = 0; k < t->nside; ++k) {
            if (ep && !strcmp(ep, "default")) PN_HIP_CHECK(ctx, hipStreamCreateWithFlags(&t->side[k], hipStreamNonBlocking));
            else PN_HIP_CHECK(ctx, hipStreamCreateWithPriority(&t->side[k], hipStreamNonBlocking, lo));
        }
    }
    if (int rc = build(t)) return rc;
    t->finalized = true;
    return PN_OK;
}

int pn_trainer_forward_backward(pn_trainer *t, const float *img_dev, const float *heat_gt_dev, const float *paf_gt_dev, const float *z_gt_dev, const float *fg_mask_dev,
                                float *loss_terms_dev, void *hip_stream) {
    if (!t) return PN_ERR_INVALID;
    pn_ctx *ctx = t->ctx;
    if (!t->finalized) return pn_set_error(ctx, PN_ERR_STATE, "pn_trainer_forward_backward: pn_trainer_finalize has not been called");
    if (!img_dev || !heat_gt_dev || !paf_gt_dev || !z_gt_dev || !fg_mask_dev || !loss_terms_dev) return pn_set_error(ctx, PN_ERR_INVALID, "pn_trainer_forward_backward: null device pointer");
    t->img = img_dev; t->target[0] = paf_gt_dev; t->target[1] = heat_gt_dev; t->target[2] = z_gt_dev; t->fg = fg_mask_dev; t->loss = loss_terms_dev;
    hipStream_t s = (hipStream_t)hip_stream;
    for (auto &op : t->ops)
        if (int rc = op(s)) return rc;
    return PN_OK;
}

double pn_trainer_conv_flops(pn_trainer *t) { return t ? t->flops_conv : 0.0; }

}  // extern "C"
