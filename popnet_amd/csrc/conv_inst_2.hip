// MFMA convolution instantiations, share 2 of 4 (see conv_mfma.hip).
#include "conv_mfma_kernel.h"

int pn_launch_conv_part2(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    PN_CASES_ALLCFG(3, 1, 120)
    PN_CASES_ALLCFG(1, 1, 120)
    PN_CASES_PREC(3, 1, 120, PN_CFG_C64W)
    return 1;
}
