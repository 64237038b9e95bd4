// Launch arguments of the implicit-GEMM convolution kernels, shared by the C ABI (conv_igemm.hip) and the per-type builds.
#pragma once
#include "common.h"
#include "tunables.h"

namespace pssr_conv {

struct ConvArgs {
    int N, H, W;
    int tiles_x, tiles_y, tiles_n;
    const void* in[2]; int in_cs[2]; int in_co[2]; int nchunks[2]; int taps[2]; const void* w[2];
    int prologue; const float* pro_scale; const float* pro_shift;
    void* out; int out_cs, out_co, cout, n_pad;
    const float* bias;
    int epi, flags;
    const void* aux; int aux_cs, aux_co;
    const float* aux_scale; const float* aux_shift; const float* aux_mean; const float* aux_invstd;
    double* stats;
    int in0_blk, out_blk, aux_blk;
    float out_scale, out_shift;
    int epi8;                   // 16-bit, FINAL excluded, cout / strides / offsets multiples of 8: straight-line 8-channel epilogue
    unsigned* stamps;           // diagnostic build (-DPSSR_V3_STAMPS) only: per-(workgroup, wave) segment cycle sums
    int dbg;                    // diagnostic bits (tunable IGEMM_DBG; 0 in production): 1 skip the epilogue, 2 skip the multiply
    int ksplit; float* ws;      // split-K: blockIdx.y owns a chunk range, raw accumulators go to ws (single-source convs only)
    const float* head_w; float* head_q;                        // EPI_HEADQ (conv_v3_kernel<T, 128> only)
};

int launch_bf16(const ConvArgs& a, hipStream_t s);
int launch_f16(const ConvArgs& a, hipStream_t s);
int launch_f32(const ConvArgs& a, hipStream_t s);

}  // namespace pssr_conv
