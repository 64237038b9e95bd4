// float build of the implicit-GEMM convolution (see conv_igemm_impl.h)
#include "conv_igemm_impl.h"

int pssr_conv::launch_f32(const ConvArgs& a, hipStream_t s) { return launch_bn<float>(a, s); }
