// Network objects: reference state_dict in, NHWC activation plan + MFMA-packed folded weights out.
//
// Mirrors (structure only; kernels are in conv_mfma.hip / conv_misc.hip):
//   rtpose_light3d.__init__/forward   tpm/lib/network/rtpose_light3d.py:249-356
//   ResPreprocessNet                  tpm/lib/network/rtpose_light3d.py:124-219
//   YoloPoseNet / ResNetBackBone      tpm/lib/network/yolo_posenet.py:26-56,87-158
// BatchNorm (eval, eps 1e-5) is folded into the preceding convolution at finalize time:
//   w' = w * g / sqrt(var + eps),  b' = (b - mean) * g / sqrt(var + eps) + beta.
// The stage-2 input "torch.cat([paf, heat, z, feat], 1)" (rtpose_light3d.py:339) is never copied:
// the stem and the three stage-1 heads write straight into one 192-channel NHWC buffer laid out
// [feat 0..127 | paf 128..155 | heat 156..171 | z 172..186 | 0-pad], and the stage-2 weights'
// input channels are permuted to that order when they are packed.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <vector>
#include "pn_internal.h"
#include "conv_plan.h"
#include "preproc_pixel.h"

namespace {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    size_t numel() const { return data.size(); }
};

struct Buf {
    void *p = nullptr;
    int H = 0, W = 0, C = 0;   // C = channel stride
    int plane = 0;             // bf16x3: channels of one of the two stored planes [hi | lo] (C = 2 * plane); otherwise = C
};

struct ConvSpec {
    std::string w;            // "<prefix>" of "<prefix>.weight" (and ".bias" when present)
    std::string bn;           // BN prefix or ""
    int ks = 3, stride = 1;
    int in_buf = -1, in_coff = 0, cin = 0;
    std::vector<int> cin_map; // my input channel -> reference input channel (-1 = zero); empty = identity
    int out_buf = -1, out_coff = 0;
    int res_buf = -1, res_coff = 0;
    int act = PN_ACT_NONE;
    int nchw_slot = -1;       // index into pn_net::nchw_ptr or -1
    int cout = 0;
    // derived
    int cfg = 0, pitch = 0, R = 0, Wt = 0, cin_chunks = 0;
    int kern = 0, wc = 0, wp = 0, nbuf = 0, pt = 7, rpg = 4;   // kern 3: conv3_kernel<ks, wc, wp, nbuf, pt, rpg> (bf16, stride 1, strip tiles)
    int wc_min = 0, nbuf_min = 0;             // set by harmonize_level: share the launch of a wider sibling conv
    int k4_level = 0;                         // set by harmonize_level: a 3x3 conv of this level has Cin >= 128 and >= 64 couts (conv4_kernel)
    void *wpack = nullptr;
    float *bias = nullptr;
    double flops = 0;
    int tail_conv = -1;       // fuse_1x1_tails: index of the 1x1 conv (<= 32 couts) that runs inside this conv's launch on its output tile
    bool fused_away = false;  // this conv runs as the tail of another one: it has no launch of its own
    void *tail_wpack = nullptr;
    float *tail_bias = nullptr;
    bool embed3 = false;      // a 1x1 convolution run as the centre tap of a 3x3 one (zero weights elsewhere: the same sums, exact zeros added), so that it shares
                              // the launch of a 3x3 sibling of the same stride: YoloPoseNet's layer2.0 stride-2 shortcut (resnet.py:59-77, build_yolo)
    bool pool_tail = false;   // fuse_pool_tails: the AvgPool2d(3, 2, 1) that follows runs inside this conv's launch (conv3_kernel<1, 4, 1, 1, 7, 8, 2>)
    int pool_out_buf = -1, pool_out_coff = 0;
};

struct Step {
    enum Type { STEM, POOL, CONV, BBLOCK } type;
    // BBLOCK: fused BasicBlock(64) = convs[bb_a] (3x3 + ReLU) -> convs[bb_b] (3x3 + residual + ReLU), bb64_kernel.h
    int bb_a = -1, bb_b = -1, bb_wt = 0;
    void *bb_wpack = nullptr;
    float *bb_bias1 = nullptr, *bb_bias2 = nullptr;
    int *bb_tickets = nullptr;       // two zeroed ints: bb64_kernel's tile tickets (BBProblem::tickets)
    bool bb_static = false;          // POPNET_BB64_STATIC=1 when the net was compiled: tiles by position
    int bb_halves = 1;               // POPNET_BB64_HALVES=2 when the net was compiled: bb64_kernel<2>
    // STEM
    int out_buf = -1;
    int stem_pool_buf = -1;          // >= 0: the MaxPool2d(3, 2, 1) that follows runs inside the stem launch and writes this buffer (build_yolo)
    float *stem_w = nullptr, *stem_b = nullptr;
    void *stem_wfrag = nullptr;      // bf16 MFMA fragments of the same weights
    int stem_cin = 1;                // > 1: multi-channel input (input_dim != 1): stem_w = folded [64][Cin][7][7] f32 for pn_conv2d_forward, stem_scratch = its NCHW f32 map
    float *stem_scratch = nullptr;
    // POOL
    int mode = 0, in_buf = -1, C = 0, out_coff = 0;
    // CONV: indices into pn_net::convs, all sharing one kernel instantiation
    std::vector<int> conv_ids;
    ConvLaunch launch;
    std::vector<ConvProblem> host_probs;
    ConvProblem *dev_probs = nullptr;
};

uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace

struct pn_net {
    pn_ctx *ctx = nullptr;
    int kind = 0, num_parts = 15, a = 14, input_dim = 1;
    std::map<std::string, HostTensor> tensors;
    bool finalized = false;
    int prec = PN_PREC_F32, max_batch = 0, in_h = 0, in_w = 0;
    std::vector<Buf> bufs;
    std::vector<ConvSpec> convs;
    std::vector<Step> steps;
    std::vector<void *> dev_allocs;
    float *nchw_ptr[4] = {nullptr, nullptr, nullptr, nullptr};
    int last_B = -1;
    bool x3 = false;                 // PN_PREC_BF16X3: tensors stored as [hi | lo] bf16 planes, read as [hi | lo | hi] against weights [W_hi | W_hi | W_lo]
    bool locked = false;             // pn_net_lock: descriptors frozen (a captured hipGraph reads them at replay time)
    bool tickets_suspect = false;    // a forward of this net failed part-way: bb64_kernel's tile tickets may be non-zero -- re-zeroed before the next launch
    float *last_nchw[4] = {nullptr, nullptr, nullptr, nullptr};
    std::map<std::string, std::pair<int, std::pair<int, int>>> named;   // name -> (buf, (coff, C))
    double flops_per_frame = 0;
    int out_h = 0, out_w = 0;
    // optional per-launch HIP-event timing (bench.py's live roofline measurement)
    bool profiling = false;
    struct ProfRec { hipEvent_t a, b; int kind; double flops; std::string label; };
    std::map<std::string, std::pair<std::pair<double, int64_t>, double>> prof_by_kernel;   // label -> ((ms, launches), flops), filled by profile_end
    std::vector<ProfRec> prof;
    size_t prof_used = 0;
    // pn_*_forward_frames: the stem reads the raw depth frames (frame_src.frames != nullptr during that call)
    PnFrameSrc frame_src = {};
    const void *last_frames = nullptr;

    size_t esize() const { return prec == PN_PREC_BF16 ? 2 : 4; }
};

int pn_set_error(pn_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

namespace {

const HostTensor *find_t(pn_net *n, const std::string &name) {
    auto it = n->tensors.find(name);
    return it == n->tensors.end() ? nullptr : &it->second;
}

int dev_alloc(pn_net *n, void **p, size_t bytes, bool zero) {
    PN_HIP_CHECK(n->ctx, hipMalloc(p, bytes));
    n->dev_allocs.push_back(*p);
    if (zero) PN_HIP_CHECK(n->ctx, hipMemset(*p, 0, bytes));
    return PN_OK;
}

int new_buf(pn_net *n, int H, int W, int C) {
    Buf b;
    b.H = H; b.W = W; b.plane = C; b.C = n->x3 ? 2 * C : C;       // bf16x3: planes [hi | lo] (the third plane pair of the K loop re-reads hi: ConvProblem::in_wrap)
    n->bufs.push_back(b);
    return (int)n->bufs.size() - 1;
}

// Fold BN, permute/pad input channels, pack into MFMA A-fragment order (see conv_mfma.hip).
int prepare_conv(pn_net *n, ConvSpec &cs) {
    pn_ctx *ctx = n->ctx;
    const HostTensor *w = find_t(n, cs.w + ".weight");
    if (!w || w->shape.size() != 4)
        return pn_set_error(ctx, PN_ERR_INVALID, "missing conv weight %s.weight", cs.w.c_str());
    const int cout = (int)w->shape[0], cin_ref = (int)w->shape[1], wks = (int)w->shape[2];
    const int ks = cs.embed3 ? 3 : wks;
    if ((cs.embed3 ? wks != 1 : wks != cs.ks) || (int)w->shape[3] != wks || ks != cs.ks)
        return pn_set_error(ctx, PN_ERR_INVALID, "%s.weight: kernel %dx%d, expected %d", cs.w.c_str(), wks, (int)w->shape[3], cs.embed3 ? 1 : cs.ks);
    cs.cout = cout;
    std::vector<int> map = cs.cin_map;
    if (map.empty()) {
        map.resize(cin_ref);
        for (int i = 0; i < cin_ref; ++i) map[i] = i;
    }
    for (int m : map)
        if (m >= cin_ref)
            return pn_set_error(ctx, PN_ERR_INVALID, "%s: input-channel map exceeds Cin=%d", cs.w.c_str(), cin_ref);
    std::vector<int> wsel;                        // bf16x3: which half of the split weight multiplies this input channel (0 = hi, 1 = lo)
    // experiment switch (per-layer mixed precision inside a bf16x3 net): POPNET_X3_BF16_CONVS = comma-separated
    // substrings of conv names that run as PLAIN bf16 -- they read only the hi plane of their input (one MFMA pass instead of
    // three) and still write all three planes, so every other layer is unchanged
    bool x3_low = false;
#ifdef PN_EXPERIMENTS      // lab builds only (POPNET_EXTRA_HIPCC_FLAGS=-DPN_EXPERIMENTS): the shipped library has no result-changing environment switch
    if (n->x3)
        if (const char *e = getenv("POPNET_X3_BF16_CONVS")) {
            std::string pats(e);
            size_t pos = 0;
            while (pos <= pats.size()) {
                size_t c = pats.find(',', pos);
                if (c == std::string::npos) c = pats.size();
                const std::string pat = pats.substr(pos, c - pos);
                if (!pat.empty() && cs.w.find(pat) != std::string::npos) x3_low = true;
                pos = c + 1;
            }
        }
#endif
    if (n->x3 && x3_low) {
        const int pc = n->bufs[cs.in_buf].plane;
        if (cs.in_coff != 0 || (int)map.size() > pc)
            return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "%s: bf16x3 convolutions read whole buffers", cs.w.c_str());
        map.resize(pc, -1);
        wsel.assign(pc, 0);
    } else if (n->x3) {
        // the input is the WHOLE buffer, three planes of `plane` channels: [x_hi | x_lo | x_hi] against [W_hi | W_hi | W_lo]
        // = x_hi W_hi + x_lo W_hi + x_hi W_lo (the dropped x_lo W_lo term is 2^-16 of the product)
        const int pc = n->bufs[cs.in_buf].plane;
        if (cs.in_coff != 0 || (int)map.size() > pc)
            return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "%s: bf16x3 convolutions read whole buffers", cs.w.c_str());
        map.resize(pc, -1);
        std::vector<int> m3;
        for (int pl = 0; pl < 3; ++pl)
            for (int i = 0; i < pc; ++i) { m3.push_back(map[i]); wsel.push_back(pl == 2 ? 1 : 0); }
        map.swap(m3);
    }
    const int cin_pad = ((int)map.size() + 63) / 64 * 64;
    map.resize(cin_pad, -1);
    wsel.resize(cin_pad, 0);
    cs.cin_chunks = cin_pad / 64;
    cs.cin = cin_ref;

    std::vector<double> scale(cout, 1.0), shift(cout, 0.0);
    const HostTensor *bias = find_t(n, cs.w + ".bias");
    if (bias && (int)bias->numel() != cout) return pn_set_error(ctx, PN_ERR_INVALID, "%s.bias: bad size", cs.w.c_str());
    for (int o = 0; o < cout; ++o) shift[o] = bias ? bias->data[o] : 0.0;
    if (!cs.bn.empty()) {
        const HostTensor *g = find_t(n, cs.bn + ".weight"), *be = find_t(n, cs.bn + ".bias");
        const HostTensor *mu = find_t(n, cs.bn + ".running_mean"), *var = find_t(n, cs.bn + ".running_var");
        if (!g || !be || !mu || !var || (int)g->numel() != cout || (int)be->numel() != cout ||
            (int)mu->numel() != cout || (int)var->numel() != cout)
            return pn_set_error(ctx, PN_ERR_INVALID, "missing/ill-shaped BatchNorm tensors %s.*", cs.bn.c_str());
        for (int o = 0; o < cout; ++o) {
            double s = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
            scale[o] = s;
            shift[o] = (shift[o] - (double)mu->data[o]) * s + (double)be->data[o];
        }
    }

    auto wval = [&](int co, int idx, int tap) -> float {      // folded weight of packed input channel idx
        const int ci = map[idx];
        if (co >= cout || ci < 0) return 0.f;
        if (cs.embed3 && tap != 4) return 0.f;
        const float v = (float)((double)w->data[cs.embed3 ? (size_t)co * cin_ref + ci : ((size_t)co * cin_ref + ci) * (ks * ks) + tap] * scale[co]);
        if (!n->x3) return v;
        uint32_t hb = (uint32_t)f32_to_bf16(v) << 16;
        float hi;
        memcpy(&hi, &hb, 4);
        return wsel[idx] ? v - hi : hi;
    };

    ConvGeom geo;                                  // which kernel, on what tiles: conv_plan.h (shared with the training engine)
    geo.pt = cs.pt; geo.rpg = cs.rpg;
    {
        const Buf &ib0 = n->bufs[cs.in_buf];
        pn_plan_conv_kernel(n->prec, n->max_batch, n->ctx->num_cus, ib0.H, ib0.W, cout, ks, cs.stride, cs.cin_chunks, cs.wc_min, cs.nbuf_min, cs.k4_level, geo);
    }
    cs.cfg = geo.cfg; cs.kern = geo.kern; cs.wc = geo.wc; cs.wp = geo.wp; cs.nbuf = geo.nbuf; cs.pt = geo.pt; cs.rpg = geo.rpg; cs.Wt = geo.Wt; cs.R = geo.R;
    const int BC = cs.kern == 4 ? 128 : (cs.kern == 3 ? cs.wc * 32 : pn_cfg_couts(cs.cfg));
    const int cout_pad = (cout + BC - 1) / BC * BC;
    const int ctiles = cout_pad / 16;
    const int KK = ks * ks;
    const int ksteps = cs.cin_chunks * KK * 2;
    const size_t fragb = n->prec == PN_PREC_BF16 ? 1024 : 2048;
    const size_t bytes = cs.kern == 4 ? (size_t)(cout_pad / 128) * (ksteps + 3) * 8192      // conv4: [cout block][k-step (+3 spare)][8 tiles][lane][8]
                                      : (size_t)ctiles * ksteps * fragb + 5 * fragb;   // + spare fragments (the weight queue prefetches up to 5 k-steps ahead)
    std::vector<unsigned char> host(bytes, 0);
    uint16_t *h16 = reinterpret_cast<uint16_t *>(host.data());
    float *h32 = reinterpret_cast<float *>(host.data());
    if (cs.kern == 4) {
        for (int cbk = 0; cbk < cout_pad / 128; ++cbk)
            for (int hh = 0; hh < cs.cin_chunks * 2; ++hh)
                for (int tap = 0; tap < KK; ++tap)
                    for (int t = 0; t < 8; ++t)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = cbk * 128 + pn_conv_row_channel(t, lane & 15, 4), q = lane >> 4;
                            const size_t base = ((((size_t)cbk * (ksteps + 3) + (size_t)hh * KK + tap) * 8 + t) * 64 + lane) * 8;
                            for (int j = 0; j < 8; ++j) h16[base + j] = f32_to_bf16(wval(co, hh * 32 + 8 * q + j, tap));
                        }
    } else
    for (int ct = 0; ct < ctiles; ++ct)
        for (int chunk = 0; chunk < cs.cin_chunks; ++chunk)
            for (int sub = 0; sub < 2; ++sub)
                for (int tap = 0; tap < KK; ++tap) {
                    const size_t kstep = (size_t)(chunk * 2 + sub) * KK + tap;   // kernel k order: (chunk, half, tap)
                    const size_t frag = (size_t)ct * ksteps + kstep;
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = pn_conv_row_channel(ct, lane & 15, cs.kern == 3 ? 2 : pn_cfg_ct(cs.cfg)), q = lane >> 4;
                        for (int j = 0; j < 8; ++j) {
                            const float v = wval(co, chunk * 64 + sub * 32 + 8 * q + j, tap);
                            if (n->prec == PN_PREC_BF16) h16[(frag * 64 + lane) * 8 + j] = f32_to_bf16(v);
                            else h32[frag * 512 + (size_t)(j >> 2) * 256 + lane * 4 + (j & 3)] = v;
                        }
                    }
                }
    if (int rc = dev_alloc(n, &cs.wpack, bytes, false)) return rc;
    PN_HIP_CHECK(ctx, hipMemcpy(cs.wpack, host.data(), bytes, hipMemcpyHostToDevice));
    std::vector<float> hb(cout_pad, 0.f);
    for (int o = 0; o < cout; ++o) hb[o] = (float)shift[o];
    if (int rc = dev_alloc(n, (void **)&cs.bias, hb.size() * 4, false)) return rc;
    PN_HIP_CHECK(ctx, hipMemcpy(cs.bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));

    // geometry
    const Buf &ib = n->bufs[cs.in_buf];
    const int Ho = (ib.H + 2 * (ks / 2) - ks) / cs.stride + 1, Wo = (ib.W + 2 * (ks / 2) - ks) / cs.stride + 1;
    {
        const char *why = "";
        if (int rc = pn_plan_conv_tiles(n->prec, ib.H, ib.W, ks, cs.stride, geo, &why)) return pn_set_error(ctx, rc, "%s: %s", cs.w.c_str(), why);
        cs.cfg = geo.cfg; cs.pitch = geo.pitch; cs.Wt = geo.Wt; cs.R = geo.R;
    }
    cs.flops = 2.0 * Ho * Wo * (double)cout * cin_ref * (cs.embed3 ? 1 : KK);
    if (cs.out_buf >= 0) {
        const Buf &ob = n->bufs[cs.out_buf];
        if (ob.H != Ho || ob.W != Wo) return pn_set_error(ctx, PN_ERR_INVALID, "%s: output buffer is %dx%d, conv gives %dx%d", cs.w.c_str(), ob.H, ob.W, Ho, Wo);
    }
    return PN_OK;
}

int add_conv(pn_net *n, const std::string &w, const std::string &bn, int ks, int stride, int in_buf, int in_coff,
             int out_buf, int out_coff, int act, int res_buf = -1, int nchw_slot = -1,
             std::vector<int> cin_map = std::vector<int>()) {
    ConvSpec cs;
    cs.w = w; cs.bn = bn; cs.ks = ks; cs.stride = stride;
    cs.in_buf = in_buf; cs.in_coff = in_coff;
    cs.out_buf = out_buf; cs.out_coff = out_coff;
    cs.res_buf = res_buf; cs.act = act; cs.nchw_slot = nchw_slot;
    cs.cin_map = std::move(cin_map);
    n->convs.push_back(cs);
    return (int)n->convs.size() - 1;
}

// Independent convs of one level that could share a launch but for their block shape: a 33..64-cout conv next
// to a wider sibling of the same kernel size takes the sibling's 128-cout block (its two surplus waves only
// help with the halo DMA, conv3_kernel.h) and the double-buffered variant -- one launch instead of two.
void harmonize_level(pn_net *n, const std::vector<int> &ids) {
    {
        bool k4 = false;
        for (int id : ids) {
            const ConvSpec &c = n->convs[id];
            const HostTensor *w = find_t(n, c.w + ".weight");
            if (w && w->shape.size() == 4 && c.ks == 3 && c.stride == 1 && w->shape[0] >= 64 && (n->x3 || std::max<int64_t>(w->shape[1], (int64_t)c.cin_map.size()) > 64)) k4 = true;
        }
        // conv4's 128-cout x 224-pixel blocks need >= 2 per CU to pay (profiles/README.md r02: level of 448 blocks 37.5 vs
        // 40.4 us, level of 224 blocks 18.5 vs 12.5 us against conv3_kernel): POPNET_CONV4 = 0 never, 1 whenever eligible,
        // unset = when the level's 3x3 convs make at least 448 such blocks at max_batch
        long blocks = 0;
        for (int id : ids) {
            const ConvSpec &c = n->convs[id];
            const HostTensor *w = find_t(n, c.w + ".weight");
            if (!w || w->shape.size() != 4 || c.ks != 3 || c.stride != 1 || w->shape[0] < 64) continue;
            const Buf &ib = n->bufs[c.in_buf];
            const long strips = (long)n->max_batch * ((ib.H + 3) / 4) * ((ib.W + 29) / 30);
            blocks += ((strips + 1) / 2) * ((w->shape[0] + 127) / 128);
        }
        const char *e = getenv("POPNET_CONV4");
        if (e ? atoi(e) == 0 : blocks < 448) k4 = false;
        if (k4)
            for (int id : ids) n->convs[id].k4_level = 1;
    }
    for (int ks : {1, 3}) {
        bool wide = false, multi = false;
        for (int id : ids) {
            const ConvSpec &c = n->convs[id];
            const HostTensor *w = find_t(n, c.w + ".weight");
            if (!w || w->shape.size() != 4 || c.ks != ks || c.stride != 1) continue;
            if (w->shape[0] > 64) wide = true;
            if (w->shape[1] > 64) multi = true;
        }
        if (!wide) continue;
        for (int id : ids) {
            ConvSpec &c = n->convs[id];
            const HostTensor *w = find_t(n, c.w + ".weight");
            if (!w || w->shape.size() != 4 || c.ks != ks || c.stride != 1 || w->shape[0] <= 32) continue;
            c.wc_min = 4;
            if (multi) c.nbuf_min = 2;
        }
    }
}

void add_conv_level(pn_net *n, const std::vector<int> &ids) {
    // split a set of independent convs into launches that share one kernel instantiation
    std::vector<bool> used(ids.size(), false);
    for (size_t i = 0; i < ids.size(); ++i) {
        if (used[i]) continue;
        Step st;
        st.type = Step::CONV;
        const ConvSpec &a = n->convs[ids[i]];
        // conv3_mix_kernel: the 128-cout 3x3 blocks and the fused-tail 1x1 blocks of a level share one launch
        static const bool no_mix = getenv("POPNET_NO_MIX") != nullptr;
        auto mixable = [](const ConvSpec &c) {
            return c.kern == 3 && c.stride == 1 && c.wc == 4 && c.wp == 1 && c.nbuf == 1 && c.pt == 7 && c.rpg == 4 && !c.pool_tail && (c.ks == 3 ? c.tail_conv < 0 : c.ks == 1);
        };
        for (size_t j = i; j < ids.size(); ++j) {
            const ConvSpec &b = n->convs[ids[j]];
            if (used[j]) continue;
            const bool same = (b.tail_conv >= 0) == (a.tail_conv >= 0) && b.pool_tail == a.pool_tail && b.ks == a.ks && b.stride == a.stride && b.pitch == a.pitch && b.R == a.R && b.Wt == a.Wt && b.kern == a.kern &&
                (a.kern == 4 || (a.kern == 3 ? (b.wc == a.wc && b.wp == a.wp && b.nbuf == a.nbuf && b.pt == a.pt && b.rpg == a.rpg) : b.cfg == a.cfg));
            const bool mixed = !no_mix && mixable(a) && mixable(b) && b.R == a.R && b.Wt == a.Wt && (a.ks == 3 || b.ks == 3 || (a.tail_conv >= 0) == (b.tail_conv >= 0));
            if (same || mixed) {
                st.conv_ids.push_back(ids[j]);
                used[j] = true;
            }
        }
        n->steps.push_back(st);
    }
}

void add_pool(pn_net *n, int mode, int in_buf, int out_buf, int C, int out_coff) {
    Step st;
    st.type = Step::POOL;
    st.mode = mode; st.in_buf = in_buf; st.out_buf = out_buf; st.C = C; st.out_coff = out_coff;
    n->steps.push_back(st);
}

int add_stem(pn_net *n, int out_buf) {
    pn_ctx *ctx = n->ctx;
    const HostTensor *w = find_t(n, "model0.conv1.weight");
    if (!w || w->shape.size() != 4 || w->shape[0] != 64 || w->shape[1] != n->input_dim || w->shape[2] != 7 || w->shape[3] != 7)
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "model0.conv1.weight must be [64,%d,7,7] (input_dim = %d)", n->input_dim, n->input_dim);
    const HostTensor *g = find_t(n, "model0.bn1.weight"), *be = find_t(n, "model0.bn1.bias");
    const HostTensor *mu = find_t(n, "model0.bn1.running_mean"), *var = find_t(n, "model0.bn1.running_var");
    if (!g || !be || !mu || !var) return pn_set_error(ctx, PN_ERR_INVALID, "missing model0.bn1.*");
    if (n->input_dim != 1) {
        // multi-channel input (the reference constructors default to input_dim = 3, rtpose_light3d.py:250 / yolo_posenet.py:88): the folded
        // 7x7 convolution runs on the generic fp32 primitive (pn_conv2d_forward, any Cin), its NCHW map is handed to the NHWC layers by
        // nchw_relu_to_nhwc_kernel.  The depth path (input_dim = 1) keeps its fused matrix-core stem.
        const int cin = n->input_dim;
        std::vector<float> hw((size_t)64 * cin * 49), hb(64);
        for (int o = 0; o < 64; ++o) {
            const double s = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
            for (int i = 0; i < cin * 49; ++i) hw[(size_t)o * cin * 49 + i] = (float)((double)w->data[(size_t)o * cin * 49 + i] * s);
            hb[o] = (float)((0.0 - (double)mu->data[o]) * s + (double)be->data[o]);
        }
        Step st;
        st.type = Step::STEM;
        st.out_buf = out_buf;
        st.stem_cin = cin;
        if (int rc = dev_alloc(n, (void **)&st.stem_w, hw.size() * 4, false)) return rc;
        if (int rc = dev_alloc(n, (void **)&st.stem_b, hb.size() * 4, false)) return rc;
        PN_HIP_CHECK(ctx, hipMemcpy(st.stem_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
        PN_HIP_CHECK(ctx, hipMemcpy(st.stem_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        const Buf &ob = n->bufs[out_buf];
        if (int rc = dev_alloc(n, (void **)&st.stem_scratch, (size_t)n->max_batch * 64 * ob.H * ob.W * 4, false)) return rc;
        n->steps.push_back(st);
        n->flops_per_frame += 2.0 * ob.H * ob.W * 64.0 * 49.0 * cin;
        return PN_OK;
    }
    std::vector<float> hw(49 * 64), hb(64);
    for (int o = 0; o < 64; ++o) {
        double s = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
        for (int t = 0; t < 49; ++t) hw[t * 64 + o] = (float)((double)w->data[o * 49 + t] * s);
        hb[o] = (float)((0.0 - (double)mu->data[o]) * s + (double)be->data[o]);
    }
    Step st;
    st.type = Step::STEM;
    st.out_buf = out_buf;
    if (int rc = dev_alloc(n, (void **)&st.stem_w, hw.size() * 4, false)) return rc;
    if (int rc = dev_alloc(n, (void **)&st.stem_b, hb.size() * 4, false)) return rc;
    PN_HIP_CHECK(ctx, hipMemcpy(st.stem_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    PN_HIP_CHECK(ctx, hipMemcpy(st.stem_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    {   // 8 A-fragments [cout tile t][k-step s][lane][8]: tile row 4q'+r' <-> cout 16q'+4t+r', k = ky*8 + kx+1
        std::vector<uint16_t> hf(16 * 64 * 8, 0);     // fragments 8..15: the lo halves of the split weights (bf16x3 mode)
        for (int t = 0; t < 4; ++t)
            for (int s2 = 0; s2 < 2; ++s2)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r16 = lane & 15, qa = lane >> 4;
                    const int co = 16 * (r16 >> 2) + 4 * t + (r16 & 3);
                    const int ky = 4 * s2 + qa;
                    for (int j = 0; j < 8; ++j) {
                        float v = 0.f;
                        if (ky < 7 && j >= 1) v = hw[(ky * 7 + (j - 1)) * 64 + co];
                        const uint16_t hb = f32_to_bf16(v);
                        hf[((t * 2 + s2) * 64 + lane) * 8 + j] = hb;
                        uint32_t hu = (uint32_t)hb << 16;
                        float hi;
                        memcpy(&hi, &hu, 4);
                        hf[((8 + t * 2 + s2) * 64 + lane) * 8 + j] = f32_to_bf16(v - hi);
                    }
                }
        if (int rc = dev_alloc(n, &st.stem_wfrag, hf.size() * 2, false)) return rc;
        PN_HIP_CHECK(ctx, hipMemcpy(st.stem_wfrag, hf.data(), hf.size() * 2, hipMemcpyHostToDevice));
    }
    n->steps.push_back(st);
    n->flops_per_frame += 2.0 * n->bufs[out_buf].H * n->bufs[out_buf].W * 64.0 * 49.0;
    return PN_OK;
}


// Folded (BatchNorm -> weight scale / bias shift) parameters of one conv, as prepare_conv computes them.
int fold_bn(pn_net *n, const ConvSpec &cs, int cout, std::vector<double> &scale, std::vector<double> &shift) {
    scale.assign(cout, 1.0); shift.assign(cout, 0.0);
    const HostTensor *bias = find_t(n, cs.w + ".bias");
    for (int o = 0; o < cout; ++o) shift[o] = bias ? bias->data[o] : 0.0;
    if (!cs.bn.empty()) {
        const HostTensor *g = find_t(n, cs.bn + ".weight"), *be = find_t(n, cs.bn + ".bias");
        const HostTensor *mu = find_t(n, cs.bn + ".running_mean"), *var = find_t(n, cs.bn + ".running_var");
        if (!g || !be || !mu || !var) return pn_set_error(n->ctx, PN_ERR_INVALID, "missing BatchNorm tensors %s.*", cs.bn.c_str());
        for (int o = 0; o < cout; ++o) {
            const double s = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
            scale[o] = s;
            shift[o] = (shift[o] - (double)mu->data[o]) * s + (double)be->data[o];
        }
    }
    return PN_OK;
}

// BasicBlock(64) pairs (conv a: 3x3 64->64 + ReLU into a scratch buffer; conv b: 3x3 64->64 on it + residual = a's input +
// ReLU) become ONE level {-2, a, b} run by bb64_kernel (bf16) or bb64x3_kernel (bf16x3: two-plane LDS images, 6-row tiles).
void fuse_basic_blocks(pn_net *n, std::vector<std::vector<int>> &levels) {
    if (n->prec != PN_PREC_BF16 || getenv("POPNET_NO_BBLOCK") || getenv("POPNET_NO_CONV3")) return;
    // bf16x3: the fused kernel is built, bit-identical and faster ALONE (2 x 227 us against 4 x 100 us of conv3_kernel<3, 2, 2, 1, 7, 4> per step
    // once tensors are stored as two planes), but a persistent 160-KB-of-LDS workgroup per CU shuts the other in-flight batches out of
    // every CU for its whole life: through the bench's pipelined region the two-launch plan wins (25.7 k against 24.0 k frames/s on the
    // same box, profiles/r04_bb64x3_ab.txt).  POPNET_BBLOCK_X3=1 selects the fused kernel (tests, one-stream latency runs).
    if (n->x3 && !(getenv("POPNET_BBLOCK_X3") && atoi(getenv("POPNET_BBLOCK_X3")))) return;
#ifdef PN_EXPERIMENTS
    if (n->x3 && getenv("POPNET_X3_BF16_CONVS")) return;      // the mixed-precision experiment runs single-pass convs the fused kernel does not know
#endif
    for (size_t i = 0; i + 1 < levels.size(); ++i) {
        if (levels[i].size() != 1 || levels[i + 1].size() != 1 || levels[i][0] < 0 || levels[i + 1][0] < 0) continue;
        const ConvSpec &a = n->convs[levels[i][0]], &b = n->convs[levels[i + 1][0]];
        const HostTensor *wa = find_t(n, a.w + ".weight"), *wb = find_t(n, b.w + ".weight");
        if (!wa || !wb || wa->shape.size() != 4 || wb->shape.size() != 4) continue;
        auto is64 = [](const HostTensor *w) { return w->shape[0] == 64 && w->shape[1] == 64 && w->shape[2] == 3 && w->shape[3] == 3; };
        if (!is64(wa) || !is64(wb) || a.stride != 1 || b.stride != 1 || a.kern != 3 || b.kern != 3) continue;
        if (a.act != PN_ACT_RELU || b.act != PN_ACT_RELU || a.res_buf >= 0 || b.res_buf != a.in_buf || b.in_buf != a.out_buf) continue;
        if (a.in_coff || a.out_coff || b.in_coff || b.res_coff || a.nchw_slot >= 0 || b.nchw_slot >= 0 || b.out_buf < 0 || !a.cin_map.empty() || !b.cin_map.empty()) continue;
        const Buf &ib = n->bufs[a.in_buf];
        if (ib.W < 8 || ib.H < 8) continue;
        const int ids[2] = {levels[i][0], levels[i + 1][0]};
        levels[i] = {-2, ids[0], ids[1]};
        levels.erase(levels.begin() + i + 1);
    }
}

int add_bblock(pn_net *n, int ia, int ib) {
    pn_ctx *ctx = n->ctx;
    Step st;
    st.type = Step::BBLOCK;
    st.bb_a = ia; st.bb_b = ib;
    const Buf &inb = n->bufs[n->convs[ia].in_buf];
    const int segs = (inb.W + 27) / 28;
    st.bb_wt = (inb.W + segs - 1) / segs;                    // <= 28 columns: the 32-pixel halo row holds Wt + 4
    // bf16: [conv][18 k-steps = (half, tap)]; bf16x3: [conv][W_hi, W_lo][18 k-steps] -- the split of prepare_conv's wval()
    const int nsel = n->x3 ? 2 : 1;
    std::vector<uint16_t> pk((size_t)36 * nsel * 4 * 64 * 8, 0);
    std::vector<float> hb[2];
    for (int cv = 0; cv < 2; ++cv) {
        const ConvSpec &cs = n->convs[cv ? ib : ia];
        const HostTensor *w = find_t(n, cs.w + ".weight");
        std::vector<double> scale, shift;
        if (int rc = fold_bn(n, cs, 64, scale, shift)) return rc;
        hb[cv].resize(64);
        for (int o = 0; o < 64; ++o) hb[cv][o] = (float)shift[o];
        for (int sel = 0; sel < nsel; ++sel)
        for (int hh = 0; hh < 2; ++hh)
            for (int tap = 0; tap < 9; ++tap)
                for (int t = 0; t < 4; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = pn_conv_row_channel(t, lane & 15, 4), q = lane >> 4;
                        const size_t base = (((((size_t)cv * nsel + sel) * 18 + hh * 9 + tap) * 4 + t) * 64 + lane) * 8;
                        for (int j = 0; j < 8; ++j) {
                            const int ci = hh * 32 + 8 * q + j;
                            const float v = (float)((double)w->data[((size_t)co * 64 + ci) * 9 + tap] * scale[co]);
                            if (!n->x3) { pk[base + j] = f32_to_bf16(v); continue; }
                            const uint16_t hbits = f32_to_bf16(v);
                            uint32_t hu = (uint32_t)hbits << 16;
                            float hi;
                            memcpy(&hi, &hu, 4);
                            pk[base + j] = sel ? f32_to_bf16(v - hi) : hbits;
                        }
                    }
    }
    if (int rc = dev_alloc(n, &st.bb_wpack, pk.size() * 2, false)) return rc;
    PN_HIP_CHECK(ctx, hipMemcpy(st.bb_wpack, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    if (int rc = dev_alloc(n, (void **)&st.bb_bias1, 64 * 4, false)) return rc;
    if (int rc = dev_alloc(n, (void **)&st.bb_bias2, 64 * 4, false)) return rc;
    PN_HIP_CHECK(ctx, hipMemcpy(st.bb_bias1, hb[0].data(), 64 * 4, hipMemcpyHostToDevice));
    PN_HIP_CHECK(ctx, hipMemcpy(st.bb_bias2, hb[1].data(), 64 * 4, hipMemcpyHostToDevice));
    if (int rc = dev_alloc(n, (void **)&st.bb_tickets, 256, true)) return rc;
    st.bb_static = getenv("POPNET_BB64_STATIC") != nullptr;          // experiment switches, read when the net is compiled (A/B runs, bit-identity tests)
    if (const char *e = getenv("POPNET_BB64_HALVES")) st.bb_halves = atoi(e) == 2 ? 2 : 1;
    n->steps.push_back(st);
    return PN_OK;
}

// The PAF branch ends in `1x1 256 -> 128 + BN + LeakyReLU` followed by `1x1 128 -> 28 + bias` (rtpose_light3d.py:266-267): per pixel
// independent, so the second convolution can run on the first one's output tile before it ever leaves the CU.  A conv A
// (1x1, exactly 128 couts = one conv3 block of 4 waves) whose output buffer is read by exactly one conv B (1x1, 128 -> <= 32
// channels, next level) takes B as its tail: B disappears from its level, A's launch writes B's output.  bf16 only (the
// three-plane bf16x3 form keeps the two-launch path); POPNET_NO_TAILFUSE=1 keeps the two launches (bit-identity tests).
void fuse_1x1_tails(pn_net *n, std::vector<std::vector<int>> &levels) {
    if (n->prec != PN_PREC_BF16 || n->x3 || getenv("POPNET_NO_TAILFUSE") || getenv("POPNET_NO_CONV3")) return;
    for (size_t li = 0; li + 1 < levels.size(); ++li) {
        if (levels[li].empty() || levels[li][0] < 0 || levels[li + 1].empty() || levels[li + 1][0] < 0) continue;
        for (int ia : levels[li]) {
            ConvSpec &a = n->convs[ia];
            if (a.kern != 3 || a.ks != 1 || a.stride != 1 || a.cout != 128 || a.wc != 4 || a.wp != 1 || a.res_buf >= 0 || a.nchw_slot >= 0 || a.out_buf < 0 || a.out_coff != 0 ||
                (a.act != PN_ACT_NONE && a.act != PN_ACT_RELU && a.act != PN_ACT_LEAKY))
                continue;
            // the only reader of A's output before the buffer is written again must be ONE conv of the next level
            int readers = 0, ib = -1;
            bool rewritten = false, other_reader = false;
            for (size_t lj = li + 1; lj < levels.size() && !rewritten; ++lj) {
                if (levels[lj].empty()) continue;
                if (levels[lj][0] == -1) {                               // pool marker {-1, mode, in, out, C, coff}
                    if (levels[lj][2] == a.out_buf) other_reader = true;
                    if (levels[lj][3] == a.out_buf) rewritten = true;
                    continue;
                }
                for (size_t k = levels[lj][0] == -2 ? 1 : 0; k < levels[lj].size(); ++k) {
                    const ConvSpec &c = n->convs[levels[lj][k]];
                    if (c.in_buf == a.out_buf || c.res_buf == a.out_buf) {
                        if (lj == li + 1 && levels[lj][0] >= 0) { ++readers; ib = levels[lj][k]; }
                        else other_reader = true;
                    }
                }
                for (size_t k = levels[lj][0] == -2 ? 1 : 0; k < levels[lj].size(); ++k)
                    if (n->convs[levels[lj][k]].out_buf == a.out_buf) rewritten = true;
            }
            if (readers != 1 || other_reader) continue;
            ConvSpec &b = n->convs[ib];
            if (std::find(levels[li + 1].begin(), levels[li + 1].end(), ib) == levels[li + 1].end()) continue;
            const HostTensor *wb = find_t(n, b.w + ".weight");
            if (!wb || wb->shape.size() != 4 || b.ks != 1 || b.stride != 1 || wb->shape[1] != 128 || wb->shape[0] > 32 || !b.bn.empty() || b.res_buf >= 0 || b.in_coff != 0 ||
                (b.act != PN_ACT_NONE && b.act != PN_ACT_SIG && b.act != PN_ACT_SIG_PM2) ||
                !b.cin_map.empty())
                continue;
            a.tail_conv = ib;
            b.fused_away = true;
            a.flops += b.flops;
            levels[li + 1].erase(std::find(levels[li + 1].begin(), levels[li + 1].end(), ib));
        }
    }
    levels.erase(std::remove_if(levels.begin(), levels.end(), [](const std::vector<int> &l) { return l.empty(); }), levels.end());
}


// `1x1 128 -> 128 + BN + ReLU` followed by AvgPool2d(3, 2, 1) (model0.conv2 / bn2 / relu -> avgpool, rtpose_light3d.py:154-158): the
// convolution is per pixel, so a block can compute exactly the 7 x 15 patch that 3 x 7 pooled pixels need and sum the windows
// from LDS (conv3_kernel.h, TAIL == 2).  The pool launch and the full-resolution map (25.7 MB written and read back at B = 32)
// disappear; blocks recompute one shared row and column (25 % more matrix work on a bandwidth-bound layer).  Same values, same
// summation order as the two launches: bit-identical (POPNET_NO_POOLFUSE=1 keeps them).  bf16 and (round 5) bf16x3: the parked tile as two planes.
void fuse_pool_tails(pn_net *n, std::vector<std::vector<int>> &levels) {
    if (n->prec != PN_PREC_BF16 || getenv("POPNET_NO_POOLFUSE") || getenv("POPNET_NO_CONV3")) return;       // (bf16x3 since round 5: two parked planes)
    for (size_t li = 0; li + 1 < levels.size(); ++li) {
        if (levels[li].size() != 1 || levels[li][0] < 0) continue;
        const std::vector<int> &pl = levels[li + 1];
        if (pl.empty() || pl[0] != -1 || pl[1] != 0) continue;                 // {-1, mode 0 = average 3x3 s2, in, out, C, coff}
        ConvSpec &a = n->convs[levels[li][0]];
        if (a.kern != 3 || a.ks != 1 || a.stride != 1 || a.cout != 128 || a.wc != 4 || a.wp != 1 || a.res_buf >= 0 || a.nchw_slot >= 0 || a.out_buf < 0 || a.out_coff != 0 ||
            a.tail_conv >= 0 || (a.act != PN_ACT_NONE && a.act != PN_ACT_RELU && a.act != PN_ACT_LEAKY))
            continue;
        if (pl[2] != a.out_buf || pl[4] != 128 || (pl[5] & 7)) continue;
        bool other_reader = false, rewritten = false;                          // nothing else may read the full-resolution map
        for (size_t lj = li + 2; lj < levels.size() && !rewritten; ++lj) {
            if (levels[lj].empty()) continue;
            if (levels[lj][0] == -1) {
                if (levels[lj][2] == a.out_buf) other_reader = true;
                if (levels[lj][3] == a.out_buf) rewritten = true;
                continue;
            }
            for (size_t k = levels[lj][0] == -2 ? 1 : 0; k < levels[lj].size(); ++k) {
                const ConvSpec &c = n->convs[levels[lj][k]];
                if (c.in_buf == a.out_buf || c.res_buf == a.out_buf) other_reader = true;
            }
            for (size_t k = levels[lj][0] == -2 ? 1 : 0; k < levels[lj].size(); ++k)
                if (n->convs[levels[lj][k]].out_buf == a.out_buf) rewritten = true;
        }
        if (other_reader) continue;
        a.pool_tail = true;
        a.pool_out_buf = pl[3]; a.pool_out_coff = pl[5];
        a.rpg = 8; a.nbuf = 1;                                                 // 8-row image: the 7 patch rows
        levels.erase(levels.begin() + li + 1);
    }
}

int pack_tail(pn_net *n, ConvSpec &a) {
    pn_ctx *ctx = n->ctx;
    const ConvSpec &b = n->convs[a.tail_conv];
    const HostTensor *w = find_t(n, b.w + ".weight"), *bias = find_t(n, b.w + ".bias");
    const int cout = (int)w->shape[0];
    std::vector<uint16_t> pk((size_t)2 * 4 * 64 * 8, 0);
    for (int t = 0; t < 2; ++t)
        for (int ks = 0; ks < 4; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = pn_conv_row_channel(t, lane & 15, 2), q = lane >> 4;
                for (int j = 0; j < 8; ++j) {
                    const int ci = ks * 32 + 8 * q + j;
                    pk[(((size_t)t * 4 + ks) * 64 + lane) * 8 + j] = f32_to_bf16(co < cout ? w->data[(size_t)co * 128 + ci] : 0.f);
                }
            }
    if (int rc = dev_alloc(n, &a.tail_wpack, pk.size() * 2, false)) return rc;
    PN_HIP_CHECK(ctx, hipMemcpy(a.tail_wpack, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(32, 0.f);
    for (int o = 0; o < cout; ++o) hb[o] = bias ? bias->data[o] : 0.f;
    if (int rc = dev_alloc(n, (void **)&a.tail_bias, 32 * 4, false)) return rc;
    PN_HIP_CHECK(ctx, hipMemcpy(a.tail_bias, hb.data(), 32 * 4, hipMemcpyHostToDevice));
    return PN_OK;
}

// ---- graph builders -------------------------------------------------------------------------
int build_rtpose(pn_net *n) {
    const int H = n->in_h, W = n->in_w;
    if (H % 8 || W % 8) return pn_set_error(n->ctx, PN_ERR_UNSUPPORTED, "input size must be a multiple of 8");
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8;
    const int J1 = n->num_parts + 1, L2 = 2 * n->a, LZ = n->a + 1;      // heat / paf / z channels
    // the stage-1 heads write channel slices of the concat buffer with vector stores: every slice starts on a multiple of 4 channels (the
    // reference's default topology, 18 parts / 19 limbs = 38 + 19 + 20 channels, leaves 2 + 1 pad channels between the slices; they are
    // never written, the buffer starts as zeros and the stage-2 weights of a pad channel are zero: rtpose_light3d.py:250,339)
    auto up4 = [](int v) { return (v + 3) / 4 * 4; };
    const int off_paf = 128, off_heat = up4(off_paf + L2), off_z = up4(off_heat + J1);
    const int cat_c = (off_z + LZ + 63) / 64 * 64;
    n->out_h = H8; n->out_w = W8;

    int A1 = new_buf(n, H2, W2, 64), A2 = new_buf(n, H2, W2, 64), T1 = new_buf(n, H2, W2, 64);
    int P1 = new_buf(n, H4, W4, 64), T2 = new_buf(n, H4, W4, 128), D2 = new_buf(n, H4, W4, 128);
    int A3 = new_buf(n, H4, W4, 128), A4 = new_buf(n, H4, W4, 128);
    int CAT = new_buf(n, H8, W8, cat_c);
    int BLa = new_buf(n, H8, W8, 256), BLb = new_buf(n, H8, W8, 256), BLc = new_buf(n, H8, W8, 128);
    int BSa = new_buf(n, H8, W8, 128), BSb = new_buf(n, H8, W8, 128);
    int BDa = new_buf(n, H8, W8, 128), BDb = new_buf(n, H8, W8, 64), BDc = new_buf(n, H8, W8, 64);
    n->named["feat"] = {CAT, {0, 128}};
    n->named["paf1"] = {CAT, {off_paf, L2}};
    n->named["heat1"] = {CAT, {off_heat, J1}};
    n->named["z1"] = {CAT, {off_z, LZ}};

    if (int rc = add_stem(n, A1)) return rc;
    std::vector<std::vector<int>> levels;
    auto level = [&](std::vector<int> ids) { levels.push_back(ids); return 0; };
    // layer1: two BasicBlocks(64) @ H/2
    level({add_conv(n, "model0.layer1.0.conv1", "model0.layer1.0.bn1", 3, 1, A1, 0, T1, 0, PN_ACT_RELU)});
    level({add_conv(n, "model0.layer1.0.conv2", "model0.layer1.0.bn2", 3, 1, T1, 0, A2, 0, PN_ACT_RELU, A1)});
    level({add_conv(n, "model0.layer1.1.conv1", "model0.layer1.1.bn1", 3, 1, A2, 0, T1, 0, PN_ACT_RELU)});
    level({add_conv(n, "model0.layer1.1.conv2", "model0.layer1.1.bn2", 3, 1, T1, 0, A1, 0, PN_ACT_RELU, A2)});
    levels.push_back({-1, 0, A1, P1, 64, 0});    // avgpool1 marker: {-1, mode, in, out, C, out_coff}
    // layer2: BasicBlock(64->128) with 1x1 shortcut @ H/4
    level({add_conv(n, "model0.layer2.0.conv1", "model0.layer2.0.bn1", 3, 1, P1, 0, T2, 0, PN_ACT_RELU),
           add_conv(n, "model0.layer2.0.downsample.0", "model0.layer2.0.downsample.1", 1, 1, P1, 0, D2, 0, PN_ACT_NONE)});
    level({add_conv(n, "model0.layer2.0.conv2", "model0.layer2.0.bn2", 3, 1, T2, 0, A3, 0, PN_ACT_RELU, D2)});
    level({add_conv(n, "model0.conv2", "model0.bn2", 1, 1, A3, 0, A4, 0, PN_ACT_RELU)});
    levels.push_back({-1, 0, A4, CAT, 128, 0});  // avgpool2 -> feat slice of the concat buffer

    // stage-2 input-channel map: my [feat | paf | heat | z | pad] -> reference [paf, heat, z, feat]
    std::vector<int> map2(cat_c, -1);
    for (int i = 0; i < 128; ++i) map2[i] = L2 + J1 + LZ + i;
    for (int i = 0; i < L2; ++i) map2[off_paf + i] = i;
    for (int i = 0; i < J1; ++i) map2[off_heat + i] = L2 + i;
    for (int i = 0; i < LZ; ++i) map2[off_z + i] = L2 + J1 + i;

    for (int stage = 1; stage <= 2; ++stage) {
        char p1[32], p2[32], p3[32];
        snprintf(p1, sizeof p1, "model%d_1", stage);
        snprintf(p2, sizeof p2, "model%d_2", stage);
        snprintf(p3, sizeof p3, "model%d_3", stage);
        auto nm = [](const char *p, int i) { return std::string(p) + "." + std::to_string(i); };
        std::vector<int> m = stage == 2 ? map2 : std::vector<int>();
        const bool last = stage == 2;
        level({add_conv(n, nm(p1, 0), nm(p1, 1), 3, 1, CAT, 0, BLa, 0, PN_ACT_LEAKY, -1, -1, m),
               add_conv(n, nm(p2, 0), nm(p2, 1), 3, 1, CAT, 0, BSa, 0, PN_ACT_LEAKY, -1, -1, m),
               add_conv(n, nm(p3, 0), nm(p3, 1), 3, 1, CAT, 0, BDa, 0, PN_ACT_LEAKY, -1, -1, m)});
        level({add_conv(n, nm(p1, 3), nm(p1, 4), 3, 1, BLa, 0, BLb, 0, PN_ACT_LEAKY),
               add_conv(n, nm(p2, 3), nm(p2, 4), 3, 1, BSa, 0, BSb, 0, PN_ACT_LEAKY),
               add_conv(n, nm(p3, 3), nm(p3, 4), 3, 1, BDa, 0, BDb, 0, PN_ACT_LEAKY)});
        level({add_conv(n, nm(p1, 6), nm(p1, 7), 3, 1, BLb, 0, BLa, 0, PN_ACT_LEAKY),
               add_conv(n, nm(p2, 6), nm(p2, 7), 3, 1, BSb, 0, BSa, 0, PN_ACT_LEAKY),
               add_conv(n, nm(p3, 6), nm(p3, 7), 3, 1, BDb, 0, BDc, 0, PN_ACT_LEAKY)});
        level({add_conv(n, nm(p1, 9), nm(p1, 10), 1, 1, BLa, 0, BLc, 0, PN_ACT_LEAKY),
               add_conv(n, nm(p2, 9), nm(p2, 10), 3, 1, BSa, 0, BSb, 0, PN_ACT_LEAKY),
               add_conv(n, nm(p3, 9), nm(p3, 10), 3, 1, BDc, 0, BDb, 0, PN_ACT_LEAKY)});
        level({add_conv(n, nm(p1, 12), "", 1, 1, BLc, 0, last ? -1 : CAT, off_paf, PN_ACT_SIG_PM2, -1, last ? 0 : -1),
               add_conv(n, nm(p2, 12), "", 3, 1, BSb, 0, last ? -1 : CAT, off_heat, PN_ACT_SIG, -1, last ? 1 : -1),
               add_conv(n, nm(p3, 12), "", 3, 1, BDb, 0, last ? -1 : CAT, off_z, PN_ACT_SIG_PM2, -1, last ? 2 : -1)});
    }

    for (auto &lv : levels)
        if (lv[0] >= 0) harmonize_level(n, lv);
    for (auto &cs : n->convs)
        if (int rc = prepare_conv(n, cs)) return rc;
    for (auto &cs : n->convs) n->flops_per_frame += cs.flops;
    fuse_basic_blocks(n, levels);
    fuse_1x1_tails(n, levels);
    fuse_pool_tails(n, levels);
    for (auto &cs : n->convs)
        if (cs.tail_conv >= 0)
            if (int rc = pack_tail(n, cs)) return rc;
    for (auto &lv : levels) {
        if (lv[0] == -1) add_pool(n, lv[1], lv[2], lv[3], lv[4], lv[5]);
        else if (lv[0] == -2) { if (int rc = add_bblock(n, lv[1], lv[2])) return rc; }
        else add_conv_level(n, lv);
    }
    return PN_OK;
}

int build_yolo(pn_net *n) {
    const int H = n->in_h, W = n->in_w;
    if (H % 16 || W % 16) return pn_set_error(n->ctx, PN_ERR_UNSUPPORTED, "input size must be a multiple of 16");
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16;
    n->out_h = H16; n->out_w = W16;
    int A1 = new_buf(n, H2, W2, 64);
    int X0 = new_buf(n, H4, W4, 64), X1 = new_buf(n, H4, W4, 64), XT = new_buf(n, H4, W4, 64);
    int Y0 = new_buf(n, H8, W8, 128), Y1 = new_buf(n, H8, W8, 128), YT = new_buf(n, H8, W8, 128), YD = new_buf(n, H8, W8, 128);
    int Na = new_buf(n, H8, W8, 256), Nb = new_buf(n, H8, W8, 256);
    int Ha = new_buf(n, H16, W16, 256), Hb = new_buf(n, H16, W16, 256), Hc = new_buf(n, H16, W16, 128);

    if (int rc = add_stem(n, A1)) return rc;
    std::vector<std::vector<int>> levels;
    auto level = [&](std::vector<int> ids) { levels.push_back(ids); };
    // conv1 - bn1 - relu - maxpool (yolo_posenet.py:101-108): one launch in bf16 (conv_misc.hip::stem7x7_pool_kernel; the 112 x 112 x 64
    // map is never stored); POPNET_NO_STEMPOOL=1, fp32 and bf16x3 keep the stem and the pool as two launches
    if (n->prec == PN_PREC_BF16 && !n->x3 && n->input_dim == 1 && !getenv("POPNET_NO_STEMPOOL")) n->steps.back().stem_pool_buf = X0;
    else levels.push_back({-1, 1, A1, X0, 64, 0});   // maxpool 3x3 s2
    int cur = X0, other = X1;
    for (int i = 0; i < 3; ++i) {
        std::string p = "model0.layer1." + std::to_string(i);
        level({add_conv(n, p + ".conv1", p + ".bn1", 3, 1, cur, 0, XT, 0, PN_ACT_RELU)});
        level({add_conv(n, p + ".conv2", p + ".bn2", 3, 1, XT, 0, other, 0, PN_ACT_RELU, cur)});
        std::swap(cur, other);
    }
    {   // layer2.0: the 3x3 stride-2 conv and its 1x1 stride-2 shortcut (resnet.py:59-77) read the same map at the same stride.  bf16 / bf16x3: the
        // shortcut runs as the centre tap of a 3x3 problem in the SAME launch (one launch and 12 us of pure latency less per step; every added
        // product is an exact zero, so the sums are the 1x1 convolution's bit for bit: POPNET_NO_EMBED3=1 keeps the two launches, tested)
        const bool embed = n->prec == PN_PREC_BF16 && !getenv("POPNET_NO_EMBED3");
        const int c1 = add_conv(n, "model0.layer2.0.conv1", "model0.layer2.0.bn1", 3, 2, cur, 0, YT, 0, PN_ACT_RELU);
        const int cd = add_conv(n, "model0.layer2.0.downsample.0", "model0.layer2.0.downsample.1", embed ? 3 : 1, 2, cur, 0, YD, 0, PN_ACT_NONE);
        n->convs[cd].embed3 = embed;
        level({c1, cd});
    }
    level({add_conv(n, "model0.layer2.0.conv2", "model0.layer2.0.bn2", 3, 1, YT, 0, Y0, 0, PN_ACT_RELU, YD)});
    int yc = Y0, yo = Y1;
    for (int i = 1; i < 4; ++i) {
        std::string p = "model0.layer2." + std::to_string(i);
        level({add_conv(n, p + ".conv1", p + ".bn1", 3, 1, yc, 0, YT, 0, PN_ACT_RELU)});
        level({add_conv(n, p + ".conv2", p + ".bn2", 3, 1, YT, 0, yo, 0, PN_ACT_RELU, yc)});
        std::swap(yc, yo);
    }
    n->named["feat"] = {yc, {0, 128}};
    level({add_conv(n, "model1.0", "model1.1", 3, 1, yc, 0, Na, 0, PN_ACT_LEAKY)});
    level({add_conv(n, "model1.3", "model1.4", 3, 1, Na, 0, Nb, 0, PN_ACT_LEAKY)});
    level({add_conv(n, "model1.6", "model1.7", 3, 1, Nb, 0, Na, 0, PN_ACT_LEAKY)});
    level({add_conv(n, "model1.9", "model1.10", 3, 1, Na, 0, Nb, 0, PN_ACT_LEAKY)});
    level({add_conv(n, "model1.12", "", 3, 1, Nb, 0, Na, 0, PN_ACT_NONE)});
    level({add_conv(n, "model2_1.0", "model2_1.1", 3, 1, Na, 0, Nb, 0, PN_ACT_LEAKY)});
    levels.push_back({-1, 2, Nb, Ha, 256, 0});  // maxpool 2x2
    level({add_conv(n, "model2_2.0", "model2_2.1", 3, 1, Ha, 0, Hb, 0, PN_ACT_LEAKY)});
    level({add_conv(n, "model2_3.0", "model2_3.1", 3, 1, Hb, 0, Hc, 0, PN_ACT_LEAKY)});
    level({add_conv(n, "model2_4.0", "", 3, 1, Hc, 0, -1, 0, PN_ACT_YOLO, -1, 3)});

    for (auto &cs : n->convs)
        if (int rc = prepare_conv(n, cs)) return rc;
    for (auto &cs : n->convs) n->flops_per_frame += cs.flops;
    fuse_basic_blocks(n, levels);
    for (auto &lv : levels) {
        if (lv[0] == -1) add_pool(n, lv[1], lv[2], lv[3], lv[4], lv[5]);
        else if (lv[0] == -2) { if (int rc = add_bblock(n, lv[1], lv[2])) return rc; }
        else add_conv_level(n, lv);
    }
    return PN_OK;
}

int refresh_problems(pn_net *n, int B, hipStream_t stream) {
    for (auto &st : n->steps) {
        if (st.type != Step::CONV) continue;
        const ConvSpec &c0 = n->convs[st.conv_ids[0]];
        const int BC = c0.kern == 4 ? 128 : (c0.kern == 3 ? c0.wc * 32 : pn_cfg_couts(c0.cfg));
        st.host_probs.clear();
        int max_blocks = 0;
        bool two_bufs = false, has_tail = false, pool_tail = false;
        for (int id : st.conv_ids) {
            const ConvSpec &cs = n->convs[id];
            const Buf &ib = n->bufs[cs.in_buf];
            ConvProblem P;
            memset(&P, 0, sizeof P);
            const size_t es = n->esize();
            (void)es;
            P.in = ib.p;
            P.wpack = cs.wpack;
            P.bias = cs.bias;
            P.B = B; P.H = ib.H; P.W = ib.W;
            P.Ho = (ib.H + 2 * (cs.ks / 2) - cs.ks) / cs.stride + 1;
            P.Wo = (ib.W + 2 * (cs.ks / 2) - cs.ks) / cs.stride + 1;
            P.cin_chunks = cs.cin_chunks;
            P.in_cs = ib.C; P.in_coff = cs.in_coff;
            if (n->x3 && ib.plane % 64)
                return pn_set_error(n->ctx, PN_ERR_UNSUPPORTED, "bf16x3: input plane of %d channels is not a multiple of 64 (the K loop wraps to the hi plane per 64-channel chunk)", ib.plane);
            P.in_wrap = n->x3 ? 2 * (ib.plane / 64) : (1 << 20);      // (conv4_kernel doubles it: halves)
            P.cout = cs.cout;
            if (cs.out_buf >= 0) { P.out = n->bufs[cs.out_buf].p; P.out_cs = n->bufs[cs.out_buf].C; P.out_coff = cs.out_coff; P.split = n->x3 ? n->bufs[cs.out_buf].plane : 0; }
            if (cs.res_buf >= 0) { P.res = n->bufs[cs.res_buf].p; P.res_cs = n->bufs[cs.res_buf].C; P.res_coff = cs.res_coff; P.res_split = n->x3 ? n->bufs[cs.res_buf].plane : 0; }
            if (cs.nchw_slot >= 0) P.out_nchw = n->nchw_ptr[cs.nchw_slot];
            P.act = cs.act;
            P.yolo_naf = 5 + 3 * n->num_parts;
            P.R = cs.R;
            P.Wt = cs.Wt;
            P.tiles_x = (P.Wo + cs.Wt - 1) / cs.Wt;
            P.tiles_per_img = ((P.Ho + cs.R - 1) / cs.R) * P.tiles_x;
            P.cout_blocks = (cs.cout + BC - 1) / BC;
            P.nblocks = B * P.tiles_per_img * P.cout_blocks;
            if (cs.kern == 4) P.nblocks = ((B * P.tiles_per_img + 1) / 2) * P.cout_blocks;      // a block = two strips x 128 couts
            P.ksteps = cs.cin_chunks * cs.ks * cs.ks * 2;
            P.ks = cs.ks;
            P.lds_buf_bytes = (int)pn_conv_lds_bytes(n->prec, cs.ks, cs.stride, cs.pitch, cs.R);
            P.lds_two = (cs.cin_chunks > 1 && 2 * (size_t)P.lds_buf_bytes <= 160 * 1024) ? 1 : 0;
            P.in_zero_off = (unsigned)((size_t)n->max_batch * ib.H * ib.W * ib.C * es);      // zero page behind every activation buffer
            if (cs.pool_tail) {                        // 3 x 7 pooled pixels per block, the pooled map as the only output
                const Buf &pb = n->bufs[cs.pool_out_buf];
                P.tiles_x = (pb.W + 6) / 7;
                P.tiles_per_img = ((pb.H + 2) / 3) * P.tiles_x;
                P.nblocks = B * P.tiles_per_img * P.cout_blocks;
                P.tail_out = pb.p; P.tail_out_cs = pb.C; P.tail_out_coff = cs.pool_out_coff;
                P.tail_split = n->x3 ? pb.plane : 0;
                P.out = nullptr;
                pool_tail = true;
            }
            if (cs.tail_conv >= 0) {
                const ConvSpec &tb = n->convs[cs.tail_conv];
                P.tail_w = cs.tail_wpack; P.tail_bias = cs.tail_bias; P.tail_cout = tb.cout; P.tail_act = tb.act;
                if (tb.out_buf >= 0) { P.tail_out = n->bufs[tb.out_buf].p; P.tail_out_cs = n->bufs[tb.out_buf].C; P.tail_out_coff = tb.out_coff; }
                if (tb.nchw_slot >= 0) P.tail_nchw = n->nchw_ptr[tb.nchw_slot];
                P.out = nullptr;                       // the 128-channel tile never leaves the CU
                has_tail = true;
            }
            if (P.lds_two) two_bufs = true;
            max_blocks = std::max(max_blocks, P.nblocks);
            st.host_probs.push_back(P);
        }
        st.launch.prec = n->prec;
        st.launch.ks = c0.ks; st.launch.stride = c0.stride; st.launch.pitch = c0.pitch; st.launch.cfg = c0.cfg;
        st.launch.nprob = (int)st.host_probs.size();
        st.launch.max_blocks = max_blocks;
        st.launch.lds_bytes = pn_conv_lds_bytes(n->prec, c0.ks, c0.stride, c0.pitch, c0.R) * (two_bufs ? 2 : 1);
        st.launch.kern = c0.kern; st.launch.wc = c0.wc; st.launch.wp = c0.wp; st.launch.nbuf = c0.nbuf; st.launch.pt = c0.pt; st.launch.rpg = c0.rpg;
        if (c0.kern == 3) st.launch.lds_bytes = pn_conv3_lds_bytes(c0.ks, c0.wp, c0.nbuf, c0.rpg);
        st.launch.tail = (c0.kern == 3 && has_tail) ? 1 : (c0.kern == 3 && pool_tail) ? 2 : 0;
        st.launch.mix = 0;
        if (c0.kern == 3) {
            for (int id : st.conv_ids) {
                const ConvSpec &cs = n->convs[id];
                if (cs.ks != c0.ks || (cs.tail_conv >= 0) != (c0.tail_conv >= 0)) st.launch.mix = 1;   // different bodies in one launch: conv3_mix_kernel
                st.launch.lds_bytes = std::max(st.launch.lds_bytes, pn_conv3_lds_bytes(cs.ks, cs.wp, cs.nbuf, cs.rpg));
            }
            if (st.launch.tail == 1) st.launch.lds_bytes = std::max<size_t>(st.launch.lds_bytes, 4 * 7 * 1024 + 1024);   // the tail's fragment image
            if (st.launch.tail == 2 && n->x3) st.launch.lds_bytes = std::max<size_t>(st.launch.lds_bytes, 2 * 4 * 7 * 1024);   // fused pool, bf16x3: hi and lo tiles parked
        }
        if (c0.kern == 4) st.launch.lds_bytes = 0;                        // conv4_launch knows its own size
        st.launch.probs_dev = st.dev_probs;
        PN_HIP_CHECK(n->ctx, hipMemcpyAsync(st.dev_probs, st.host_probs.data(), st.host_probs.size() * sizeof(ConvProblem),
                                            hipMemcpyHostToDevice, stream));
    }
    n->last_B = B;
    for (int i = 0; i < 4; ++i) n->last_nchw[i] = n->nchw_ptr[i];
    return PN_OK;
}

int run_forward(pn_net *n, const float *x, int B, hipStream_t stream) {
    pn_ctx *ctx = n->ctx;
    if (!n->finalized) return pn_set_error(ctx, PN_ERR_STATE, "pn_net_finalize has not been called");
    if (B < 1 || B > n->max_batch) return pn_set_error(ctx, PN_ERR_INVALID, "batch %d outside [1, %d]", B, n->max_batch);
    bool dirty = B != n->last_B;
    for (int i = 0; i < 4; ++i) dirty |= n->nchw_ptr[i] != n->last_nchw[i];
    if (dirty) {
        if (n->locked)
            return pn_set_error(ctx, PN_ERR_STATE, "net is locked at batch %d (pn_net_lock): a forward with another batch size or other output buffers would rewrite descriptors a captured graph still reads", n->last_B);
        if (int rc = refresh_problems(n, B, stream)) return rc;
    }
    // bb64_kernel's tile tickets are persistent device state that only a COMPLETED launch returns to zero; one net must not run two forwards at once (one
    // stream per net: the engines keep it so), and after a forward that failed part-way the counters are re-zeroed here, in stream order, before anything
    // can read them (a non-zero ticket would make every later launch -- graph replays included -- skip tiles silently)
    if (n->tickets_suspect) {
        for (auto &st : n->steps)
            if (st.type == Step::BBLOCK && st.bb_tickets) PN_HIP_CHECK(ctx, hipMemsetAsync(st.bb_tickets, 0, 256, stream));
        n->tickets_suspect = false;
    }
    for (auto &st : n->steps) {
        int rc = PN_OK;
        pn_net::ProfRec *pr = nullptr;
        if (n->profiling) {
            if (n->prof_used == n->prof.size()) {
                pn_net::ProfRec r;
                PN_HIP_CHECK(ctx, hipEventCreate(&r.a));
                PN_HIP_CHECK(ctx, hipEventCreate(&r.b));
                n->prof.push_back(r);
            }
            pr = &n->prof[n->prof_used++];
            pr->kind = (int)st.type;
            pr->flops = 0;
            pr->label.clear();
            if (st.type == Step::CONV) {
                for (int id : st.conv_ids) pr->flops += n->convs[id].flops * B;
                char lb[96];
                const ConvLaunch &cl = st.launch;
                if (cl.kern == 4) snprintf(lb, sizeof lb, "conv4_kernel");
                else if (cl.kern == 3 && cl.mix) snprintf(lb, sizeof lb, "conv3_mix_kernel");
                else if (cl.kern == 3) snprintf(lb, sizeof lb, "conv3_kernel<%d, %d, %d, %d, %d, %d>", cl.ks, cl.wc, cl.wp, cl.nbuf, cl.pt, cl.rpg);
                else snprintf(lb, sizeof lb, "conv_mfma_kernel<%d, %d, %d, %d, %d>", cl.prec, cl.ks, cl.stride, cl.pitch, cl.cfg);
                pr->label = lb;
            }
            if (st.type == Step::BBLOCK) {       // counted with the convolutions: both convs' algorithmic FLOPs
                pr->kind = (int)Step::CONV;
                pr->flops = (n->convs[st.bb_a].flops + n->convs[st.bb_b].flops) * B;
                pr->label = n->x3 ? "bb64x3_kernel" : "bb64_kernel";
            }
            PN_HIP_CHECK(ctx, hipEventRecord(pr->a, stream));
        }
        // timing-only ablation (wrong results): POPNET_ABLATE_SKIP = comma-separated classes of launches to skip --
        // "pool", "head" (<= 32-cout generic launches), "c1x1" (conv3 1x1), "c3" (conv3 3x3), "stem"; what would the pipelined
        // throughput be if these launches cost nothing?  (docs/lab-archive/tail_ablation.sh)
#ifdef PN_EXPERIMENTS      // lab builds only: a timed region of the shipped library cannot be made to skip launches by the environment
        static const char *skip = getenv("POPNET_ABLATE_SKIP");
        if (skip) {
            const char *cls = st.type == Step::POOL ? "pool" : st.type == Step::STEM ? "stem" : st.type == Step::BBLOCK ? "bb64" :
                              (st.launch.kern == 4 ? "conv4" : st.launch.kern == 3 ? (st.launch.ks == 1 ? "c1x1" : "c3") : "head");
            if (strstr(skip, cls)) { if (pr) PN_HIP_CHECK(ctx, hipEventRecord(pr->b, stream)); continue; }
        }
#endif
        if (st.type == Step::STEM && st.stem_pool_buf >= 0) {
            const Buf &ob = n->bufs[st.out_buf], &pb = n->bufs[st.stem_pool_buf];
            rc = pn_launch_stem_pool(ctx, x, st.stem_wfrag, st.stem_b, pb.p, B, n->in_h, n->in_w, ob.H, ob.W, pb.C, stream, n->frame_src.frames ? &n->frame_src : nullptr);
        } else if (st.type == Step::STEM && st.stem_cin > 1) {
            const Buf &ob = n->bufs[st.out_buf];
            if (n->frame_src.frames) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "the frames-in forward takes single-channel depth frames (input_dim = 1)");
            rc = pn_conv2d_forward(ctx, x, st.stem_w, st.stem_b, st.stem_scratch, B, st.stem_cin, n->in_h, n->in_w, 64, 7, 2, 3, 0, (void *)stream);
            if (!rc) rc = pn_launch_nchw_relu_to_nhwc(ctx, n->prec, st.stem_scratch, ob.p, B, ob.H, ob.W, 64, ob.C, n->x3 ? ob.plane : 0, stream);
        } else if (st.type == Step::STEM) {
            const Buf &ob = n->bufs[st.out_buf];
            rc = pn_launch_stem(ctx, n->x3 ? PN_PREC_BF16X3 : n->prec, x, st.stem_w, st.stem_wfrag, st.stem_b, ob.p, B, n->in_h, n->in_w, ob.H, ob.W, ob.C, n->x3 ? ob.plane : 0, stream,
                                n->frame_src.frames ? &n->frame_src : nullptr);
        } else if (st.type == Step::POOL) {
            const Buf &ib = n->bufs[st.in_buf], &ob = n->bufs[st.out_buf];
            rc = pn_launch_pool(ctx, n->prec, st.mode, ib.p, ob.p, B, ib.H, ib.W, st.C, ib.C, ob.C, st.out_coff, n->x3 ? ib.plane : 0, n->x3 ? ob.plane : 0, stream);
        } else if (st.type == Step::BBLOCK) {
            const ConvSpec &ca = n->convs[st.bb_a], &cb = n->convs[st.bb_b];
            const Buf &ib = n->bufs[ca.in_buf], &ob = n->bufs[cb.out_buf];
            BBProblem P;
            memset(&P, 0, sizeof P);
            P.in = ib.p; P.out = ob.p; P.wpack = st.bb_wpack; P.bias1 = st.bb_bias1; P.bias2 = st.bb_bias2;
            P.B = B; P.H = ib.H; P.W = ib.W;
            P.in_cs = ib.C; P.in_coff = ca.in_coff; P.out_cs = ob.C; P.out_coff = cb.out_coff;
            P.Wt = st.bb_wt; P.tiles_x = (ib.W + st.bb_wt - 1) / st.bb_wt;
            P.tiles_per_img = ((ib.H + (n->x3 ? 5 : 7)) / (n->x3 ? 6 : 8)) * P.tiles_x;      // bb64x3_kernel: 6-row tiles
            P.ntiles = B * P.tiles_per_img;
            P.in_zero_off = (unsigned)((size_t)n->max_batch * ib.H * ib.W * ib.C * n->esize());
            P.in_split = n->x3 ? ib.plane : 0; P.out_split = n->x3 ? ob.plane : 0;
            P.tickets = (n->x3 || st.bb_static || P.ntiles < 4 * ctx->num_cus) ? nullptr : st.bb_tickets;      // nullptr: blockIdx.x + k * gridDim.x
            P.halves = st.bb_halves;
            rc = n->x3 ? pn_launch_bb64x3(ctx, P, stream) : pn_launch_bb64(ctx, P, stream);
        } else {
            rc = pn_launch_conv(ctx, st.launch, stream);
        }
        if (rc) { n->tickets_suspect = true; return rc; }
        if (pr) PN_HIP_CHECK(ctx, hipEventRecord(pr->b, stream));
    }
    return PN_OK;
}

}  // namespace

// ---- C ABI ----------------------------------------------------------------------------------
extern "C" {

pn_net *pn_net_create(pn_ctx *ctx, int kind, int num_parts, int a, int input_dim) {
    if (!ctx) return nullptr;
    if (kind != PN_NET_RTPOSE_LIGHT3D && kind != PN_NET_YOLO_POSENET) {
        pn_set_error(ctx, PN_ERR_INVALID, "unknown net kind %d", kind);
        return nullptr;
    }
    if (input_dim < 1 || input_dim > 16) {
        pn_set_error(ctx, PN_ERR_UNSUPPORTED, "input_dim %d outside [1, 16]", input_dim);
        return nullptr;
    }
    pn_net *n = new pn_net();
    n->ctx = ctx; n->kind = kind; n->num_parts = num_parts; n->a = a; n->input_dim = input_dim;
    return n;
}

void pn_net_destroy(pn_net *n) {
    if (!n) return;
    for (void *p : n->dev_allocs) (void)hipFree(p);
    for (auto &r : n->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    delete n;
}

int pn_net_set_tensor(pn_net *n, const char *name, const float *host_data, const int64_t *shape, int ndim) {
    if (!n || !name || (!host_data && ndim > 0)) return PN_ERR_INVALID;
    if (n->finalized) return pn_set_error(n->ctx, PN_ERR_STATE, "net already finalized");
    std::string s(name);
    if (s.rfind("module.", 0) == 0) s = s.substr(7);
    HostTensor t;
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); numel *= (size_t)shape[i]; }
    t.data.assign(host_data, host_data + numel);
    n->tensors[s] = std::move(t);
    return PN_OK;
}

int pn_net_finalize(pn_net *n, int precision, int max_batch, int in_h, int in_w) {
    if (!n) return PN_ERR_INVALID;
    pn_ctx *ctx = n->ctx;
    if (n->finalized) return pn_set_error(ctx, PN_ERR_STATE, "net already finalized");
    if (precision != PN_PREC_F32 && precision != PN_PREC_BF16 && precision != PN_PREC_BF16X3) return pn_set_error(ctx, PN_ERR_INVALID, "bad precision %d", precision);
    if (max_batch < 1) return pn_set_error(ctx, PN_ERR_INVALID, "max_batch must be >= 1");
    PN_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    n->x3 = precision == PN_PREC_BF16X3;                      // kernels and packing run as bf16; tensors carry three planes
    n->prec = n->x3 ? PN_PREC_BF16 : precision; n->max_batch = max_batch; n->in_h = in_h; n->in_w = in_w;
    int rc = n->kind == PN_NET_RTPOSE_LIGHT3D ? build_rtpose(n) : build_yolo(n);
    if (rc) return rc;
    for (auto &b : n->bufs) {
        size_t bytes = (size_t)max_batch * b.H * b.W * b.C * n->esize() + 2048;   // + zero page (halo padding source of conv3_kernel / conv4_kernel: 64 B per 32-channel half + 16)
        if (bytes >= ((size_t)1 << 32))     // the kernels address a buffer with 32-bit byte offsets (zero page, epilogue stores)
            return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "activation buffer %dx%dx%d x batch %d = %zu B exceeds the 4 GiB the kernels' 32-bit offsets address; lower max_batch", b.H, b.W, b.C, max_batch, bytes);
        if (int r = dev_alloc(n, &b.p, bytes, true)) return r;   // zero: pad channels must read as 0
    }
    for (auto &st : n->steps)
        if (st.type == Step::CONV)
            if (int r = dev_alloc(n, (void **)&st.dev_probs, st.conv_ids.size() * sizeof(ConvProblem), true)) return r;
    n->tensors.clear();
    PN_HIP_CHECK(ctx, hipDeviceSynchronize());      // the zero fills above ran on the null stream: a forward on a non-blocking stream must not overtake them
    n->finalized = true;
    return PN_OK;
}

int pn_rtpose_forward(pn_net *n, const float *x_dev, int B, float *paf_dev, float *heat_dev, float *z_dev, void *hip_stream) {
    if (!n) return PN_ERR_INVALID;
    if (n->kind != PN_NET_RTPOSE_LIGHT3D) return pn_set_error(n->ctx, PN_ERR_INVALID, "not an rtpose_light3d net");
    if (!x_dev || !paf_dev || !heat_dev || !z_dev) return pn_set_error(n->ctx, PN_ERR_INVALID, "null device pointer");
    n->nchw_ptr[0] = paf_dev; n->nchw_ptr[1] = heat_dev; n->nchw_ptr[2] = z_dev;
    return run_forward(n, x_dev, B, (hipStream_t)hip_stream);
}

int pn_yolo_forward(pn_net *n, const float *x_dev, int B, float *out_dev, void *hip_stream) {
    if (!n) return PN_ERR_INVALID;
    if (n->kind != PN_NET_YOLO_POSENET) return pn_set_error(n->ctx, PN_ERR_INVALID, "not a YoloPoseNet net");
    if (!x_dev || !out_dev) return pn_set_error(n->ctx, PN_ERR_INVALID, "null device pointer");
    n->nchw_ptr[3] = out_dev;
    return run_forward(n, x_dev, B, (hipStream_t)hip_stream);
}

// frames in: the 7x7 stem computes its input tile from the raw depth frames with pn_preprocess's arithmetic (preproc_pixel.h)
static int forward_frames(pn_net *n, const void *depth_dev, int depth_dtype, int B, int H, int W, float depth_max, float depth_mean, float depth_std,
                          hipStream_t stream) {
    pn_ctx *ctx = n->ctx;
    if (!depth_dev || H < 2 || W < 2) return pn_set_error(ctx, PN_ERR_INVALID, "forward_frames: bad arguments");
    if (depth_dtype != PN_DEPTH_F16 && depth_dtype != PN_DEPTH_F32) return pn_set_error(ctx, PN_ERR_INVALID, "forward_frames: unknown depth dtype %d", depth_dtype);
    if (!n->finalized) return pn_set_error(ctx, PN_ERR_STATE, "pn_net_finalize has not been called");
    if (n->prec != PN_PREC_BF16) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "the frames-in forward is built for the bf16 and bf16x3 nets (fp32: pn_preprocess + pn_*_forward)");
    if (n->in_h != n->in_w) return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "the frames-in forward resizes to a square input (net finalized for %dx%d)", n->in_h, n->in_w);
    const int S = n->in_h;
    if (W == 2 * S && H == 2 * S)   // as pn_preprocess: cv::resize switches to INTER_AREA at exactly 2x decimation
        return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "forward_frames: %dx%d -> %d is an exact 2x decimation, where cv2.resize(INTER_LINEAR) runs INTER_AREA instead: not built", W, H, S);
    PnFrameSrc fs;
    fs.frames = depth_dev; fs.dtype = depth_dtype; fs.H = H; fs.W = W;
    const double inv_x = (double)S / (double)W, inv_y = (double)S / (double)H;
    fs.scale_x = 1.0 / inv_x; fs.scale_y = 1.0 / inv_y;                    // as cv::resize computes them (pn_preprocess)
    fs.dmax = depth_max; fs.mean = depth_mean; fs.stdv = depth_std;
    n->frame_src = fs;
    const int rc = run_forward(n, nullptr, B, stream);
    n->frame_src.frames = nullptr;
    return rc;
}

int pn_rtpose_forward_frames(pn_net *n, const void *depth_dev, int depth_dtype, int B, int H, int W, float depth_max, float depth_mean, float depth_std,
                             float *paf_dev, float *heat_dev, float *z_dev, void *hip_stream) {
    if (!n) return PN_ERR_INVALID;
    if (n->kind != PN_NET_RTPOSE_LIGHT3D) return pn_set_error(n->ctx, PN_ERR_INVALID, "not an rtpose_light3d net");
    if (!paf_dev || !heat_dev || !z_dev) return pn_set_error(n->ctx, PN_ERR_INVALID, "null device pointer");
    n->nchw_ptr[0] = paf_dev; n->nchw_ptr[1] = heat_dev; n->nchw_ptr[2] = z_dev;
    return forward_frames(n, depth_dev, depth_dtype, B, H, W, depth_max, depth_mean, depth_std, (hipStream_t)hip_stream);
}

int pn_yolo_forward_frames(pn_net *n, const void *depth_dev, int depth_dtype, int B, int H, int W, float depth_max, float depth_mean, float depth_std,
                           float *out_dev, void *hip_stream) {
    if (!n) return PN_ERR_INVALID;
    if (n->kind != PN_NET_YOLO_POSENET) return pn_set_error(n->ctx, PN_ERR_INVALID, "not a YoloPoseNet net");
    if (!out_dev) return pn_set_error(n->ctx, PN_ERR_INVALID, "null device pointer");
    n->nchw_ptr[3] = out_dev;
    return forward_frames(n, depth_dev, depth_dtype, B, H, W, depth_max, depth_mean, depth_std, (hipStream_t)hip_stream);
}

int pn_net_copy_activation(pn_net *n, const char *name, int B, float *dev_out, void *hip_stream) {
    if (!n || !name || !dev_out) return PN_ERR_INVALID;
    pn_ctx *ctx = n->ctx;
    if (!n->finalized) return pn_set_error(ctx, PN_ERR_STATE, "net not finalized");
    auto it = n->named.find(name);
    if (it == n->named.end()) return pn_set_error(ctx, PN_ERR_INVALID, "unknown activation '%s'", name);
    const Buf &b = n->bufs[it->second.first];
    return pn_launch_nhwc_to_nchw(ctx, n->prec, b.p, dev_out, B, b.H, b.W, it->second.second.second, b.C,
                                  it->second.second.first, n->x3 ? b.plane : 0, (hipStream_t)hip_stream);
}

int pn_net_read_activation(pn_net *n, const char *name, int B, float *host_out, size_t host_elems, void *hip_stream) {
    if (!n || !name || !host_out) return PN_ERR_INVALID;
    pn_ctx *ctx = n->ctx;
    auto it = n->named.find(name);
    if (it == n->named.end()) return pn_set_error(ctx, PN_ERR_INVALID, "unknown activation '%s'", name);
    const Buf &b = n->bufs[it->second.first];
    const size_t elems = (size_t)B * it->second.second.second * b.H * b.W;
    if (host_elems < elems) return pn_set_error(ctx, PN_ERR_INVALID, "host buffer too small: %zu < %zu", host_elems, elems);
    float *tmp = nullptr;
    PN_HIP_CHECK(ctx, hipMalloc((void **)&tmp, elems * 4));
    hipStream_t s = (hipStream_t)hip_stream;
    int rc = pn_net_copy_activation(n, name, B, tmp, hip_stream);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(host_out, tmp, elems * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = pn_set_error(ctx, PN_ERR_HIP, "read_activation copy: %s", hipGetErrorString(e));
    }
    (void)hipFree(tmp);
    return rc;
}

double pn_net_flops_per_frame(pn_net *n) { return n ? n->flops_per_frame : 0.0; }

int pn_net_lock(pn_net *n, int locked) {
    if (!n) return PN_ERR_INVALID;
    if (locked && n->last_B < 0) return pn_set_error(n->ctx, PN_ERR_STATE, "pn_net_lock: run one forward first");
    n->locked = locked != 0;
    return PN_OK;
}

int pn_net_profile_begin(pn_net *n) {
    if (n) n->prof_by_kernel.clear();
    if (!n) return PN_ERR_INVALID;
    n->profiling = true;
    n->prof_used = 0;
    return PN_OK;
}

int pn_net_profile_end(pn_net *n, double *conv_ms, int64_t *conv_launches, double *conv_flops, double *other_ms,
                       int64_t *other_launches) {
    if (!n) return PN_ERR_INVALID;
    pn_ctx *ctx = n->ctx;
    n->profiling = false;
    double cm = 0, om = 0, cf = 0;
    int64_t cl = 0, ol = 0;
    for (size_t i = 0; i < n->prof_used; ++i) {
        PN_HIP_CHECK(ctx, hipEventSynchronize(n->prof[i].b));
        float ms = 0.f;
        PN_HIP_CHECK(ctx, hipEventElapsedTime(&ms, n->prof[i].a, n->prof[i].b));
        if (n->prof[i].kind == (int)Step::CONV) {
            cm += ms; cf += n->prof[i].flops; ++cl;
            auto &e = n->prof_by_kernel[n->prof[i].label];
            e.first.first += ms; e.first.second += 1; e.second += n->prof[i].flops;
        } else { om += ms; ++ol; }
    }
    n->prof_used = 0;
    if (conv_ms) *conv_ms = cm;
    if (conv_launches) *conv_launches = cl;
    if (conv_flops) *conv_flops = cf;
    if (other_ms) *other_ms = om;
    if (other_launches) *other_launches = ol;
    return PN_OK;
}

int pn_net_profile_kernel(pn_net *n, int rank, char *name, size_t name_cap, double *ms, int64_t *launches, double *flops) {
    if (!n || rank < 0) return PN_ERR_INVALID;
    std::vector<std::pair<double, std::string>> order;
    for (auto &kv : n->prof_by_kernel) order.push_back({-kv.second.first.first, kv.first});
    std::sort(order.begin(), order.end());
    if ((size_t)rank >= order.size()) return PN_ERR_INVALID;
    const auto &e = n->prof_by_kernel[order[rank].second];
    if (name && name_cap) { strncpy(name, order[rank].second.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (ms) *ms = e.first.first;
    if (launches) *launches = e.first.second;
    if (flops) *flops = e.second;
    return PN_OK;
}

}  // extern "C"
