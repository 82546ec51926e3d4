// bb64_kernel (fused BasicBlock(64): conv1 -> LDS -> conv2 + residual; bb64_kernel.h) and its launcher.
#include "bb64_kernel.h"

int pn_launch_bb64(pn_ctx *ctx, const BBProblem &P, hipStream_t stream) { return bb64_launch(ctx, P, ctx->num_cus, stream); }
