// MFMA convolution instantiations, share 3 of 4 (see conv_mfma.hip).
#include "conv_mfma_kernel.h"

int pn_launch_conv_part3(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    PN_CASES_ALLCFG(3, 1, 16)
    PN_CASES_PREC(3, 2, 64, PN_CFG_C128)
    PN_CASES_PREC(3, 2, 64, PN_CFG_C64)
    PN_CASES_PREC(1, 2, 64, PN_CFG_C128)
    PN_CASES_PREC(1, 2, 64, PN_CFG_C64)
    PN_CASES_PREC(3, 2, 120, PN_CFG_C128)
    PN_CASES_PREC(3, 2, 120, PN_CFG_C64)
    PN_CASES_PREC(1, 2, 120, PN_CFG_C128)
    PN_CASES_PREC(1, 2, 120, PN_CFG_C64)
    return 1;
}
