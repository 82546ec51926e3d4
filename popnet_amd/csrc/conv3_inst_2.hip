// conv3_kernel instantiations, share 2 of 3 (32-cout blocks, 1x1 layers) + the dispatcher.
#include "conv3_kernel.h"

int pn_launch_conv3_part2(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    PN3_CASE(3, 1, 1, 1) PN3_CASE(1, 2, 1, 1) PN3_CASE(1, 2, 2, 1) PN3_CASE(1, 1, 1, 1)
    return 1;
}

// LDS bytes of one block: NBUF piece-major halo images + the dump slot of the branch-free DMA
size_t pn_conv3_lds_bytes(int ks, int WP, int nbuf, int rpg) {
    const int hr = rpg * WP + ks - 1 + (((ks - 1) & 1) ? 1 : 0);
    return (size_t)8 * hr * 32 * 16 * nbuf + 1024;
}

int pn_launch_conv3(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    int rc;
    if ((rc = pn_launch_conv3_part0(ctx, L, stream)) != 1) return rc;
    if ((rc = pn_launch_conv3_part1(ctx, L, stream)) != 1) return rc;
    if ((rc = pn_launch_conv3_part2(ctx, L, stream)) != 1) return rc;
    return pn_set_error(ctx, PN_ERR_UNSUPPORTED, "no conv3 kernel for ks=%d wc=%d wp=%d nbuf=%d", L.ks, L.wc, L.wp, L.nbuf);
}
