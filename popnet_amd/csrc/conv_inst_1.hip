// MFMA convolution instantiations, share 1 of 4 (see conv_mfma.hip).
#include "conv_mfma_kernel.h"

int pn_launch_conv_part1(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    PN_CASES_ALLCFG(3, 1, 64)
    PN_CASES_ALLCFG(1, 1, 64)
    PN_CASES_PREC(3, 1, 64, PN_CFG_C64W)
    return 1;
}
