// bb64x3_kernel (fused BasicBlock(64) of the bf16x3 nets: two-plane LDS images, conv1 -> LDS -> conv2 + residual; bb64x3_kernel.h) and its launcher.
#include "bb64x3_kernel.h"

int pn_launch_bb64x3(pn_ctx *ctx, const BBProblem &P, hipStream_t stream) { return bb64x3_launch(ctx, P, ctx->num_cus, stream); }
