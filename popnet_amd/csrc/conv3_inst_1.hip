// conv3_kernel instantiations, share 1 of 3 (64-cout blocks).
#include "conv3_kernel.h"

int pn_launch_conv3_part1(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    PN3_CASE(3, 2, 1, 1) PN3_CASE(3, 2, 1, 2) PN3_CASE(3, 2, 2, 1) PN3_CASE(3, 2, 2, 2)
    PN3_CASE_PT(3, 2, 1, 1, 14)
    return 1;
}
