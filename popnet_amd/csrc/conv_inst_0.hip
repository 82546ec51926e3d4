// MFMA convolution instantiations, share 0 of 4 (see conv_mfma.hip).
#include "conv_mfma_kernel.h"

int pn_launch_conv_part0(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    PN_CASES_ALLCFG(3, 1, 32)
    PN_CASES_ALLCFG(1, 1, 32)
    return 1;
}
