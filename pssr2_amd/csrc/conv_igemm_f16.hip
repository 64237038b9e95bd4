// f16_t build of the implicit-GEMM convolution (see conv_igemm_impl.h)
#include "conv_igemm_impl.h"

int pssr_conv::launch_f16(const ConvArgs& a, hipStream_t s) { return launch_bn<f16_t>(a, s); }
