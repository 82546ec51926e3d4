// Launch dispatch of the MFMA convolution (kernel template: conv_mfma_kernel.h).  The
// instantiations -- {3x3, 1x1} x {stride 1, 2} x LDS pitch class {16, 32, 64, 120 pixels} x tile
// configuration x {bf16, f32} -- are spread over conv_inst_*.hip so they compile in parallel.
#include "pn_internal.h"

int pn_cfg_couts(int cfg) {
    switch (cfg) {
        case PN_CFG_C128: return 128;
        case PN_CFG_C64: return 64;
        case PN_CFG_C64W: return 64;
        case PN_CFG_C32: return 32;
        default: return 16;
    }
}


int pn_cfg_pixels(int cfg) { return cfg == PN_CFG_C128 ? 112 : (cfg == PN_CFG_C64W ? 224 : 128); }

size_t pn_conv_lds_bytes(int prec, int ks, int stride, int pitch, int R) {
    size_t pixb = prec == PN_PREC_BF16 ? 128 : 256;
    size_t rows = (size_t)(R - 1) * stride + ks;
    return rows * pitch * pixb;
}

int pn_launch_conv_part0(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv_part1(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv_part2(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);
int pn_launch_conv_part3(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream);

int pn_launch_conv(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    int rc;
    if (L.kern == 4) return pn_launch_conv4(ctx, L, stream);
    if (L.kern == 3) return pn_launch_conv3(ctx, L, stream);
    if ((rc = pn_launch_conv_part0(ctx, L, stream)) != 1) return rc;
    if ((rc = pn_launch_conv_part1(ctx, L, stream)) != 1) return rc;
    if ((rc = pn_launch_conv_part2(ctx, L, stream)) != 1) return rc;
    if ((rc = pn_launch_conv_part3(ctx, L, stream)) != 1) return rc;
    return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "no conv kernel for prec=%d ks=%d stride=%d pitch=%d cfg=%d", L.prec, L.ks,
                        L.stride, L.pitch, L.cfg);
}

// host mirror of StageCfg (conv_mfma_kernel.h): largest halo tile (pixels) the register-prefetched
// staging path of this instantiation can hold; 0 = that instantiation stages with the plain loop.
int pn_conv_stage_maxpx(int prec, int ks, int stride, int pitch, int cfg) {
    if (cfg == PN_CFG_C64W) return 0;
    const int maxpx = stride != 1 ? 0 : (ks == 1 ? 128 : (pitch <= 32 ? 192 : (pitch <= 64 ? 288 : 360)));
    const int nch = prec == PN_PREC_BF16 ? 8 : 16;
    const int raw = (maxpx * nch + 255) / 256;
    return raw <= 12 ? maxpx : 0;
}
