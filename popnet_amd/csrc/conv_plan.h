// Which kernel runs a convolution and on what tile geometry: ONE rule set for the inference nets (net.hip::prepare_conv) and the training
// engine (trainx.hip), so a training-mode convolution runs exactly the kernel / tiling the inference plan would pick for the same shape
// (and is covered by the same bit-identity tests).  Moved here from net.hip in round 6, unchanged.
#pragma once
#include <algorithm>
#include <cstdlib>
#include "pn_internal.h"

struct ConvGeom {
    int kern = 0, cfg = 0, pitch = 0, R = 0, Wt = 0;
    int wc = 0, wp = 0, nbuf = 0, pt = 7, rpg = 4;      // kern 3: conv3_kernel<ks, wc, wp, nbuf, pt, rpg>
};

inline int pn_pick_pitch(int cols) {
    const int classes[4] = {16, 32, 64, 120};
    for (int c : classes)
        if (cols <= c) return c;
    return -1;
}

inline int pn_pick_cfg(int cout) {
    if (cout % 128 == 0) return PN_CFG_C128;
    if (cout >= 64) return PN_CFG_C64;
    if (cout <= 16) return PN_CFG_C16;
    return PN_CFG_C32;
}

// prec: PN_PREC_BF16 for the bf16 / bf16x3 nets, PN_PREC_F32 otherwise.  H, W: input map; the caller has set g.pt / g.rpg defaults (7 / 4).
// wc_min / nbuf_min / k4_level: what net.hip::harmonize_level decided for the level this convolution belongs to.
inline void pn_plan_conv_kernel(int prec, int max_batch, int num_cus, int H, int W, int cout, int ks, int stride, int cin_chunks, int wc_min, int nbuf_min,
                                int k4_level, ConvGeom &g) {
    g.cfg = pn_pick_cfg(cout);
    {   // bf16 stride-1 layers run conv3_kernel (conv3_kernel.h) when the map splits into column strips (<= 30 wide: the
        // halo row is 32 pixels) whose 4-row tiles fill >= 75 % of a wave group's 112 pixel slots
        int segs = 0, wt = 0, rows = 0, rpg = 4;
        double best = 0;
        for (int sg = (W + 29) / 30; sg <= (W + 15) / 16; ++sg) {
            const int w = (W + sg - 1) / sg;
            for (int gg : {4, 8}) {                      // rows per wave group kept in LDS (8: narrow maps only, 3x3 / 128-cout blocks)
                if (gg == 8 && !(w <= 14 && ks == 3 && cout > 64 && getenv("POPNET_CONV3_RPG8"))) continue;   // measured slower than the generic kernel on 14x14 maps (profiles/README.md v15)
                const int r = std::min(std::min(H, gg), 112 / w);
                const double util = r * ((double)W / sg) / 112.0;
                if (util > best + 1e-9) { best = util; segs = sg; wt = w; rows = r; rpg = gg; }
            }
        }
        // (cout > 32: the <= 32-cout heads stay on the generic kernel -- on conv3_kernel<3, 1, 1, 1> the two head launches took 64 us per step against 25, round 6)
        if (prec == PN_PREC_BF16 && stride == 1 && (ks == 3 || ks == 1) && best >= 0.75 && cout > 32 && !getenv("POPNET_NO_CONV3")) {
            g.kern = 3;
            g.wc = std::max(cout > 64 ? 4 : (cout > 32 ? 2 : 1), wc_min);
            const long tiles112 = (long)max_batch * ((H + rows - 1) / rows) * segs;   // strip tiles of one wave group
            g.wp = (g.wc == 2 && rows == 4 && rpg == 4 && tiles112 * ((cout + 63) / 64) >= 1536) ? 2 : 1;   // big maps: 8-row tiles, 256 threads
            g.rpg = rpg;
            if (g.wp == 2 && ks == 3 && cin_chunks == 1 && getenv("POPNET_CONV3_PT14") && atoi(getenv("POPNET_CONV3_PT14")) == 1) { g.wp = 1; g.pt = 14; g.rpg = 8; }   // 8 rows per WAVE: half the weight bytes
            if (const char *e = getenv("POPNET_CONV3_PT14"))          // =2: 128-cout blocks of 224-pixel wave tiles on every 28-column 3x3 level as well
                if (atoi(e) == 2 && ks == 3 && g.wc == 4 && g.wp == 1 && rpg == 4 && rows == 4 && H >= 8) { g.pt = 14; g.rpg = 8; }
            const int hr = rpg * g.wp + ks - 1, ngw = (8 * (hr / 2) + g.wc * g.wp - 1) / (g.wc * g.wp);
            // single halo image (4 waves / SIMD) beats the double-buffered variant (3 waves / SIMD) on every level
            // of both networks (profiles/README.md, r01 v8); POPNET_CONV3_NBUF2=1 selects the latter for experiments
            g.nbuf = ((cin_chunks > 1 || nbuf_min == 2) && ks == 3 && ngw <= 18 && getenv("POPNET_CONV3_NBUF2")) ? 2 : 1;
            g.Wt = wt;
            g.R = std::min(H, rows * g.wp * (g.pt / 7));          // rows * Wt <= 112 (224) pixel slots per wave group
            // conv4_kernel (both operands through LDS, 64-cout x 112-pixel wave tiles): the 3x3 layers with Cin >= 128 and
            // >= 64 couts on 4-row strip tiles; a Cin = 64 conv joins only as the sibling of such a layer (one launch per level)
            if (ks == 3 && g.wp == 1 && g.pt == 7 && g.rpg == 4 && cout >= 64 && k4_level)
                g.kern = 4;
        }
    }
    // Small maps on the generic kernel (YoloPoseNet's 14 x 14 levels: 2 tiles per frame): 128-cout blocks give fewer blocks than the chip
    // has CUs (128 at B = 32) -- 64-cout x 128-pixel blocks double them.  POPNET_GENERIC_C64=0 keeps the 128-cout blocks.
    if (g.kern == 0 && prec == PN_PREC_BF16 && g.cfg == PN_CFG_C128 && stride == 1) {
        const long blocks128 = (long)max_batch * ((H * W + 111) / 112) * (cout / 128);
        const char *e = getenv("POPNET_GENERIC_C64");
        if (blocks128 < num_cus && !(e && atoi(e) == 0)) g.cfg = PN_CFG_C64;
    }
}

// Tile geometry of the chosen kernel (kern 3 / 4: the strip tiles are already in g; the generic kernel: rows x segments of its pixel tile).
// Returns PN_OK, or a negative status with *why set (the caller formats the message).
inline int pn_plan_conv_tiles(int prec, int H, int W, int ks, int stride, ConvGeom &g, const char **why) {
    const int Ho = (H + 2 * (ks / 2) - ks) / stride + 1, Wo = (W + 2 * (ks / 2) - ks) / stride + 1;
    if (g.kern == 3 || g.kern == 4) { g.pitch = 32; return PN_OK; }
    if (g.cfg == PN_CFG_C64 && ks == 3 && stride == 1 && Wo >= 48 && (long)Ho * Wo >= 2048) g.cfg = PN_CFG_C64W;   // wide maps: 224-pixel tiles
    const int BP = pn_cfg_pixels(g.cfg);
    // a block owns R full rows when they fit its pixel tile, else one row cut into equal segments
    const int wt_cap = std::min(BP, (120 - ks) / stride + 1);     // widest segment the largest pitch class holds
    const int segs = (Wo + wt_cap - 1) / wt_cap;
    g.Wt = (Wo + segs - 1) / segs;
    g.R = std::max(1, std::min(Ho, BP / g.Wt));
    {   // the register-prefetched staging path holds at most this many halo pixels
        const int maxpx = pn_conv_stage_maxpx(prec, ks, stride, pn_pick_pitch((g.Wt - 1) * stride + ks), g.cfg);
        while (maxpx > 0 && g.R > 1 && ((g.R - 1) * stride + ks) * ((g.Wt - 1) * stride + ks) > maxpx) --g.R;
        if (maxpx > 0 && ((g.R - 1) * stride + ks) * ((g.Wt - 1) * stride + ks) > maxpx) { *why = "halo tile exceeds the staging capacity"; return PN_ERR_UNSUPPORTED; }
    }
    g.pitch = pn_pick_pitch((g.Wt - 1) * stride + ks);
    if (g.pitch < 0) { *why = "halo width has no pitch class"; return PN_ERR_UNSUPPORTED; }
    return PN_OK;
}
