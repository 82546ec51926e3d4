// conv4_kernel (both operands through LDS; conv4_kernel.h) and its launcher.
#include "conv4_kernel.h"

int pn_launch_conv4(pn_ctx *ctx, const ConvLaunch &L, hipStream_t stream) { return conv4_launch(ctx, L, stream); }
