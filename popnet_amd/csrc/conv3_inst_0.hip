// conv3_kernel instantiations, share 0 of 3 (128-cout blocks).
#include "conv3_kernel.h"

int pn_launch_conv3_part0(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) {
    if (L.mix) {                         // 3x3 blocks + fused-tail 1x1 blocks in one launch (conv3_mix_kernel)
        hipLaunchKernelGGL(conv3_mix_kernel<0>, dim3(L.max_blocks, L.nprob), dim3(256), L.lds_bytes, stream, L.probs_dev);
        PN_HIP_CHECK(ctx, hipGetLastError());
        return PN_OK;
    }
    PN3_CASE(3, 4, 1, 1) PN3_CASE(3, 4, 1, 2) PN3_CASE(1, 4, 1, 1) PN3_CASE_TAIL(1, 4, 1, 1) PN3_CASE_POOLTAIL(1, 4, 1, 1)
    PN3_CASE_RPG(3, 4, 1, 1, 8)          // 14-column maps: 8 rows x 14
    PN3_CASE_PT(3, 4, 1, 1, 14)          // 224-pixel wave tiles (8 rows x 28), 2 waves / SIMD: POPNET_CONV3_PT14=2
    return 1;
}
