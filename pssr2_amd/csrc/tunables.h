// Kernel-selection tunables of libpssr_mi355.so: one table, initialised once from PSSR_<NAME> environment variables, changed
// through pssr_set_option() (include/pssr_mi355.h).  Launch paths read the table, never the environment.
#pragma once
#include "../../include/pssr_mi355.h"

struct PssrTunables {
    int igemm_flat;         // 1x1 convolutions through the stage-of-chunks kernel
    int igemm_big;          // 256-pixel x 64-channel tiles for 3x3 layers >= 16x16 (0 off, 1 Cout <= 64, 2 all)
    int igemm_v3;           // LDS-DMA / counted-wait 3x3 loop for 16-bit layers with > 64 output channels and >= 16x16 images
    int igemm_v3_64;        // the same loop in 16x32-pixel x 64-channel tiles for layers with 33..64 output channels (0 off, 1 on)
    int v3_lds_pad;         // experiment: KiB of unused LDS per v3 workgroup (1 workgroup per CU from ~25)
    int igemm_dbg;          // diagnostic bits of the v3 loop (0 in production)
    int igemm_n64;          // 128 x 64 tiles for the 3x3 layers whose 128 x 128 tiles would leave one workgroup per CU
    int igemm_ksplit;       // workgroups a split-K launch of the 128-pixel loop aims for
    int conv_epi8;          // straight-line 8-channel epilogue
    int wgrad_lean;         // lean-loader weight-gradient kernel
    int wgrad_x2;           // 3x3 weight gradients of 16x8-pixel tiles: 512-thread workgroups of two phase-shifted wave groups (0 off)
    int wgrad_dma;          // all-DMA 3x3 weight-gradient kernel for prologue-free inputs (conv_wgrad16d_kernel)
    int wgrad_blocks;       // partial slabs of a 3x3 weight gradient
    int wgrad_blocks_1x1;   // ... of a 1x1 weight gradient
    int dwconv_tile;        // LDS-tiled depthwise 7x7
    int dwwg_blocks;        // slabs of the depthwise weight gradient
    int ln_bwd_blocks;      // most workgroups of a LayerNorm2d backward launch (each ends with 2c f64 atomic pairs: 512 -> 256 = -5..-10 us on the 64^2 / 32^2 maps)
    int ln_dbg;             // diagnostic bits of the LayerNorm2d backward kernel (0 in production)
};
PssrTunables& pssr_tunables();
