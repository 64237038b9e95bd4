// Implicit-GEMM 3x3 / 1x1 convolution for gfx950 (MI355X): forward and input-gradient.
//
//   M = output pixels (128 per workgroup: a TH x TW patch of NI images), N = output channels
//   (BN per workgroup), K = taps x input channels, walked in 32-byte channel chunks.
//
// Per K-chunk the workgroup stages ONE halo tile of the input ((TH+2)x(TW+2) pixels x 32 B) and
// the 9 tap slices of the packed weights into LDS; the 9 taps are then 9 shifted reads of the same
// LDS halo (so HBM/L2 -> LDS input traffic is 1.4x the tile instead of 9x), each feeding 32x32
// MFMA tiles (v_mfma_f32_32x32x16_bf16, or 4 x v_mfma_f32_32x32x2_f32 in the exact-f32 build).
// Staging is register-mediated: BatchNorm+ReLU of the producing layer and the zero padding are
// applied between the global load and the LDS write, and the loads of chunk i+1 are in flight
// while chunk i is multiplied.  The accumulators leave through LDS so that every epilogue
// (bias, ReLU, residual tail, ReLU mask, f64 BatchNorm statistics) runs on pixel-major rows and
// the stores are coalesced along channels.
//
// This header holds the kernels and their launch logic; it is compiled once per storage type (conv_igemm_bf16.hip,
// conv_igemm_f16.hip, conv_igemm_f32.hip) so that the three builds run in parallel; conv_igemm.hip holds the C ABI.
#pragma once
#include <type_traits>

#include "conv_igemm_args.h"

using pssr_conv::ConvArgs;

namespace {



constexpr int PRO_LDS_MAX = 4096 * 8;     // prologue table: scale + shift (f32) of up to 4096 input channels

template <int GEO> struct Geo;
template <> struct Geo<0> { static constexpr int TWL = 4, THL = 3; };   // 8 x 16, 1 image
template <> struct Geo<1> { static constexpr int TWL = 3, THL = 3; };   // 8 x 8,  2 images
template <> struct Geo<2> { static constexpr int TWL = 2, THL = 2; };   // 4 x 4,  8 images
template <> struct Geo<3> { static constexpr int TWL = 1, THL = 1; };   // 2 x 2, 32 images
template <> struct Geo<4> { static constexpr int TWL = 0, THL = 0; };   // 1 x 1, 128 images
template <> struct Geo<5> { static constexpr int TWL = 4, THL = 4; };   // 16 x 16 = 256 pixels x 64 channels: the four waves stacked along M

template <int BN, int GEO> struct Cfg {
    static constexpr int TWL = Geo<GEO>::TWL, THL = Geo<GEO>::THL;
    // GEO 5 (BN = 64 only): 256 pixels x 64 channels per workgroup.  Per MFMA it stages 40 % fewer LDS bytes than the
    // 128 x 128 tile (the weight slices, 3/4 of a chunk's LDS writes, are shared by twice the pixels), which is what
    // bounds the loop: LDS write + read time of a chunk ~ its MFMA time at 128 x 128
    static constexpr bool BIG = GEO == 5;
    static_assert(!BIG || BN == 64, "the 256-pixel tile is built for 64 output channels");
    static constexpr int WM = BIG ? 4 : ((BN == 32) ? 4 : 2), WN = 4 / WM;
    static constexpr int MI = BIG ? 2 : 4 / WM, NJ = BN / (32 * WN);
    static constexpr int MP = WM * MI * 32;                                   // output pixels per workgroup
    static constexpr int TW = 1 << TWL, TH = 1 << THL, NI = MP >> (TWL + THL);
    static constexpr int HW2 = TW + 2, HPI = (TH + 2) * (TW + 2), HP = NI * HPI;
    // pitch of a halo row in LDS, in pixels.  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (and
    // the same + 32): with the 16-byte halves swizzled by pixel bit 3 a group is conflict-free iff its 16 pixel indices differ mod 16,
    // and the rows of a tile that share a group are TW apart in x -- with the natural pitch TW + 2 the second row lands two slots into
    // the first one's (measured 24-38 % conflict cycles on these reads); 32 (16-wide tiles) / 24 (8-wide: rows 8 apart mod 16) separate them
    static constexpr int HWP = (TW == 16 && !BIG) ? 32 : HW2;
    static constexpr int HPIP = (TH + 2) * HWP;
    static constexpr int A_ITEMS = (2 * HP + 255) / 256;
    static constexpr int A_BYTES = NI * HPIP * 32;
    static constexpr int B_BYTES = 9 * BN * 32;
    static constexpr int B_ITEMS = (9 * BN * 2 + 255) / 256;
    static constexpr int E_BYTES = WM * 32 * BN * 4;
    static constexpr int RED_BYTES = 4 * BN * 2 * 4;
    static constexpr int MAIN_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = (MAIN_BYTES > E_BYTES + RED_BYTES) ? MAIN_BYTES : (E_BYTES + RED_BYTES);
    static constexpr int PERM = 0;          // MFMA row r of a 32-row tile is pixel r of the tile's pixel order
};

// output pixel (within the workgroup's tile) of accumulator row `row` (0 .. WM*32-1) of row tile mi.  C::PERM (conv_v3.h): the
// two image rows of a 32-row tile are interleaved so that every ds_read_b128 lane group reads one image row
template <class C> __device__ __forceinline__ int epi_pixel(int row, int mi) {
    int r = row & 31;
    if constexpr (C::PERM != 0) {
        const int rx = r & 15, ry = (r >> 4) ^ ((rx >= 4 && rx < 12) ? 1 : 0);
        if constexpr (C::PERM == 2)          // 16 rows x 32 columns, wave = 4 image rows, row tile = (row pair mi >> 1, column half mi & 1)
            return (((row >> 5) * 4 + (mi >> 1) * 2 + ry) << 5) | ((mi & 1) << 4) | rx;
        r = (ry << 4) | rx;
    }
    return (row >> 5) * C::MI * 32 + mi * 32 + r;
}

// Epilogue shared by both main loops: the accumulators leave through LDS so that bias / ReLU / residual tail / ReLU mask /
// GELU derivative / f64 BatchNorm statistics run on pixel-major rows and the stores are coalesced along channels.
// CFG provides WM, WN, MI, NJ, TW, TH, TWL, THL, E_BYTES; BN the channel tile.
template <typename T, int BN, class C>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[C::MI][C::NJ], char* smem, int tid, int x0, int y0, int img0, int n0) {
    using X = TT<T>;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int h = lane >> 5, r = lane & 31;
    float* Es = (float*)smem;                       // [WM*32][BN] f32
    float* Red = (float*)(smem + C::E_BYTES);       // [4 waves][BN][2]
    constexpr int CG = BN / 4;                      // 4-channel groups per row
    constexpr int PASSES = C::WM * 32 * CG / 256;
    // row tile 0 leaves its registers before the per-channel constants are loaded (with 128 accumulators per lane the
    // epilogue otherwise spills)
#pragma unroll
    for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const int col = wn * C::NJ * 32 + nj * 32 + r;
            Es[row * BN + col] = acc[0][nj][e];
        }
    asm volatile("" ::: "memory");
    const int c4 = tid % CG;
    const int n_base = n0 + c4 * 4;
    const bool n_ok = n_base < p.cout;
    float bias[4] = {0, 0, 0, 0}, xs[4] = {0, 0, 0, 0}, xh[4] = {0, 0, 0, 0}, xm[4] = {0, 0, 0, 0}, xi[4] = {0, 0, 0, 0};
    if (n_ok) {
        if (p.bias) {
            if (p.epi == PSSR_EPI_FINAL) { for (int e = 0; e < 4; ++e) if (n_base + e < p.cout) bias[e] = p.bias[n_base + e]; }
            else load4(p.bias + n_base, bias);
        }
        if (p.epi == PSSR_EPI_TAIL || p.epi == PSSR_EPI_DGRAD_MASK) { load4(p.aux_scale + n_base, xs); load4(p.aux_shift + n_base, xh); }
        if (p.epi == PSSR_EPI_DGRAD_MASK && (p.flags & PSSR_FLAG_STATS)) {
            load4(p.aux_mean + n_base, xm); load4(p.aux_invstd + n_base, xi);
        }
    }
    float s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    T* outp = (T*)p.out;
    const T* auxp = (const T*)p.aux;

#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) {
        if (mi) {
            __syncthreads();
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    const int col = wn * C::NJ * 32 + nj * 32 + r;
                    Es[row * BN + col] = acc[mi][nj][e];
                }
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int piece = tid + ps * 256;
            const int row = piece / CG;
            const int m = epi_pixel<C>(row, mi);
            const int tx = m & (C::TW - 1), ty = (m >> C::TWL) & (C::TH - 1), img = m >> (C::TWL + C::THL);
            const int gy = y0 + ty, gx = x0 + tx, gi = img0 + img;
            if (!(n_ok && gi < p.N && gy < p.H && gx < p.W)) continue;
            float v[4];
            load4(Es + row * BN + c4 * 4, v);
            if (p.epi == PSSR_EPI_FINAL) {
                float* of = (float*)p.out;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n_base + e < p.cout)
                        of[(((long)gi * p.cout + n_base + e) * p.H + gy) * p.W + gx] = fmaf(v[e] + bias[e], p.out_scale, p.out_shift);
                continue;
            }
            const long pix = pix_index(gi, gy, gx, p.H, p.W, p.out_blk);
            if (p.epi == PSSR_EPI_STORE) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] += bias[e];
                    if (p.flags & PSSR_FLAG_RELU) v[e] = fmaxf(v[e], 0.f);
                }
            } else {
                float a[4];
                load4(auxp + pix_index(gi, gy, gx, p.H, p.W, p.aux_blk) * p.aux_cs + p.aux_co + n_base, a);
                if (p.epi == PSSR_EPI_TAIL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + bias[e] + fmaf(a[e], xs[e], xh[e]), 0.f);
                } else if (p.epi == PSSR_EPI_DGRAD_GELU) {
                    gelu_grad_mul_vec<T, 4>(a, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (fmaf(a[e], xs[e], xh[e]) > 0.f) ? v[e] : 0.f;
                        a[e] = (a[e] - xm[e]) * xi[e];       // xhat
                    }
                }
                if (p.flags & PSSR_FLAG_STATS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float g = X::round(v[e]); s1[e] += g; s2[e] += g * a[e]; }
                }
            }
            if (p.epi == PSSR_EPI_STORE && (p.flags & PSSR_FLAG_STATS)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float g = X::round(v[e]); s1[e] += g; s2[e] += g * g; }
            }
            store4(outp + pix * p.out_cs + p.out_co + n_base, v);
        }
    }
    if (p.flags & PSSR_FLAG_STATS) {
#pragma unroll
        for (int off = 32; off >= CG; off >>= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], off); s2[e] += __shfl_xor(s2[e], off); }
        if (lane < CG) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Red[(wave * BN + lane * 4 + e) * 2 + 0] = s1[e];
                Red[(wave * BN + lane * 4 + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.cout) {
            float t1 = 0, t2 = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { t1 += Red[(w * BN + tid) * 2]; t2 += Red[(w * BN + tid) * 2 + 1]; }
            double* st = p.stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * p.cout;
            const long lo = (long)PSSR_STAT_STRIPES * 2 * p.cout;
            stat_add(st + n0 + tid, lo, t1);
            stat_add(st + p.cout + n0 + tid, lo, t2);
        }
    }
}

// 16-bit outputs whose channel counts / strides / offsets are multiples of 8 (every layer of the models): 8-channel
// pieces = one 16-byte store per piece, and one straight-line body per epilogue kind (the generic epilogue above
// branches on p.epi per piece and unrolls 16 passes: ~12k instructions, most of a K=576 layer's time).
// Es columns are permuted so that both float4 halves of a piece are lane-contiguous: logical 8c+4u+e -> u*BN/2 + 4c + e.
// TR: the accumulators are those of the TRANSPOSED product (weights as the MFMA's A operand, conv_v3.h): a lane holds, for ONE pixel
// (lane & 31), channels (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of a 32-channel tile, i.e. runs of 4 consecutive channels = one
// ds_write_b128 each (a quarter of the LDS write instructions of the plain layout, whose 128 scalar writes per lane were most of a
// K = 576 layer's epilogue).  Rows are then BN + 4 floats apart (the 16 lanes of a write group hit 16 different bank quads).
template <typename T, int BN, class C, int EPI, bool STATS, bool TR = false, bool AFF = false>
__device__ __forceinline__ void conv_epilogue8(const ConvArgs& p, f32x16 (&acc)[C::MI][C::NJ], char* smem, int tid, int x0, int y0, int img0, int n0) {
    static_assert(!AFF || (EPI == PSSR_EPI_STORE && !STATS), "FLAG_AFFINE: EPI_STORE without statistics");
    using X = TT<T>;
    static_assert(sizeof(T) == 2, "8-channel pieces are 16 bytes of a 16-bit type");
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int h = lane >> 5, r = lane & 31;
    float* Es = (float*)smem;
    float* Red = (float*)(smem + C::E_BYTES);       // [4 waves][BN][2]
    constexpr int CG = BN / 8, RP = 256 / CG;       // pieces per row, rows per pass
    constexpr int EP = TR ? BN + 4 : BN;            // floats between rows of Es
    constexpr int ROWS = C::WM * 32, PASSES = ROWS / RP;
    static_assert(PASSES >= 1 && ROWS % RP == 0, "pass geometry");
    int wcol[C::NJ];
#pragma unroll
    for (int nj = 0; nj < C::NJ; ++nj) {
        const int col = wn * C::NJ * 32 + nj * 32 + r;
        wcol[nj] = ((col >> 2) & 1) * (BN / 2) + ((col >> 3) << 2) + (col & 3);
    }
    // row tile 0 leaves its registers before the per-channel constants are loaded (with 128 accumulators per lane the
    // epilogue otherwise spills)
    auto es_write = [&](int mi) {
        if constexpr (TR) {
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {       // channels 8 (c) + 4 h + (0..3), c = (wn NJ + nj) 4 + q  ->  column h BN/2 + 4 c
                    const int c = (wn * C::NJ + nj) * 4 + q;
                    *(float4*)(Es + (wm * 32 + r) * EP + h * (BN / 2) + 4 * c) =
                        make_float4(acc[mi][nj][4 * q], acc[mi][nj][4 * q + 1], acc[mi][nj][4 * q + 2], acc[mi][nj][4 * q + 3]);
                }
        } else {
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                    Es[row * EP + wcol[nj]] = acc[mi][nj][e];
                }
        }
    };
    es_write(0);
    asm volatile("" ::: "memory");
    // TR, 8 pieces per row: the two rows of a 16-lane read group are 8 apart (8 x (BN + 4) floats = 128 bytes mod 256: no shared bank)
    const int c8 = tid % CG, rs = tid / CG;
    const int row0 = (TR && CG == 8) ? ((rs >> 1) & 7) + ((rs & 1) << 3) + ((rs >> 4) << 4) : rs;
    const int n_base = n0 + c8 * 8;
    const bool n_ok = n_base < p.cout;
    float bias[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xs[8], xh[8], xm[8], xi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { xs[e] = 0.f; xh[e] = 0.f; xm[e] = 0.f; xi[e] = 0.f; }
    if (n_ok) {
        if ((EPI == PSSR_EPI_STORE || EPI == PSSR_EPI_TAIL) && p.bias) { load4(p.bias + n_base, bias); load4(p.bias + n_base + 4, bias + 4); }
        if (EPI == PSSR_EPI_TAIL || EPI == PSSR_EPI_DGRAD_MASK || AFF) {
            load4(p.aux_scale + n_base, xs); load4(p.aux_scale + n_base + 4, xs + 4);
            load4(p.aux_shift + n_base, xh); load4(p.aux_shift + n_base + 4, xh + 4);
        }
        if (EPI == PSSR_EPI_DGRAD_MASK && STATS) {
            load4(p.aux_mean + n_base, xm); load4(p.aux_mean + n_base + 4, xm + 4);
            load4(p.aux_invstd + n_base, xi); load4(p.aux_invstd + n_base + 4, xi + 4);
        }
    }
    const float relu_lo = (EPI == PSSR_EPI_STORE && (p.flags & PSSR_FLAG_RELU)) ? 0.f : -__builtin_inff();
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    T* outp = (T*)p.out + p.out_co + n_base;
    const T* auxp = (const T*)p.aux + p.aux_co + n_base;
    // FLAG_SHUF2: the piece belongs to sub-pixel shuf_s = (i, j) of its pixel and to channel n_base % (cout / 4) of the shuffled map
    const bool shuf = EPI == PSSR_EPI_STORE && !STATS && (p.flags & PSSR_FLAG_SHUF2);
    int shuf_i = 0, shuf_j = 0;
    if (shuf) {
        const int q4 = p.cout >> 2, s = n_base / q4;
        shuf_i = s >> 1; shuf_j = s & 1;
        outp = (T*)p.out + p.out_co + (n_base - s * q4);
    }

#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) {
        if (mi) {
            __syncthreads();
            es_write(mi);
        }
        __syncthreads();
#pragma unroll 1
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = row0 + ps * RP;
            const int m = epi_pixel<C>(row, mi);
            const int tx = m & (C::TW - 1), ty = (m >> C::TWL) & (C::TH - 1), img = m >> (C::TWL + C::THL);
            const int gy = y0 + ty, gx = x0 + tx, gi = img0 + img;
            if (!(n_ok && gi < p.N && gy < p.H && gx < p.W)) continue;
            float v[8];
            load4(Es + row * EP + c8 * 4, v);
            load4(Es + row * EP + BN / 2 + c8 * 4, v + 4);
            float a[8];
            if (EPI != PSSR_EPI_STORE) {
                const u32x4 araw = *(const u32x4*)(auxp + pix_index(gi, gy, gx, p.H, p.W, p.aux_blk) * p.aux_cs);
                X::unpack(araw, a);
            }
            if constexpr (EPI == PSSR_EPI_DGRAD_GELU) gelu_grad_mul_vec<T, 8>(a, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (EPI == PSSR_EPI_STORE) v[e] = fmaxf(AFF ? fmaf(v[e] + bias[e], xs[e], xh[e]) : v[e] + bias[e], relu_lo);
                else if (EPI == PSSR_EPI_TAIL) v[e] = fmaxf(v[e] + bias[e] + fmaf(a[e], xs[e], xh[e]), 0.f);
                else if (EPI == PSSR_EPI_DGRAD_GELU) {}
                else {
                    v[e] = (fmaf(a[e], xs[e], xh[e]) > 0.f) ? v[e] : 0.f;
                    a[e] = (a[e] - xm[e]) * xi[e];       // xhat
                }
            }
            const u32x4 packed = X::pack(v);
            if (STATS) {
                float g[8];
                X::unpack(packed, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // GELU-derivative kind: the first sum is added with explicit v_add_f32.  hipcc packs these adds into
                    // "v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]" (halves crossed) between the v_rcp / v_exp of the next channel
                    // pair, and with that instruction the sum came out different by one workgroup's worth in a few launches out of 40
                    // (outputs and second sum identical, tools/diag/det_gelu_stats.py, tests/test_gpu_determinism.py); the other
                    // kinds have no transcendental instructions here and get an uncrossed v_pk_add_f32.
                    if (EPI == PSSR_EPI_DGRAD_GELU) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s1[e]) : "v"(g[e]));
                    else s1[e] += g[e];
                    s2[e] += g[e] * (EPI == PSSR_EPI_STORE ? g[e] : a[e]);
                }
            }
            const long opix = shuf ? ((long)gi * 2 * p.H + 2 * gy + shuf_i) * (2 * p.W) + 2 * gx + shuf_j : pix_index(gi, gy, gx, p.H, p.W, p.out_blk);
            *(u32x4*)(outp + opix * p.out_cs) = packed;
        }
    }
    if (STATS) {
#pragma unroll
        for (int off = 32; off >= CG; off >>= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += __shfl_xor(s1[e], off); s2[e] += __shfl_xor(s2[e], off); }
        if (lane < CG) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                Red[(wave * BN + lane * 8 + e) * 2 + 0] = s1[e];
                Red[(wave * BN + lane * 8 + e) * 2 + 1] = s2[e];
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < p.cout) {
            float t1 = 0, t2 = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { t1 += Red[(w * BN + tid) * 2]; t2 += Red[(w * BN + tid) * 2 + 1]; }
            double* st = p.stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * p.cout;
            const long lo = (long)PSSR_STAT_STRIPES * 2 * p.cout;
            stat_add(st + n0 + tid, lo, t1);
            stat_add(st + p.cout + n0 + tid, lo, t2);
        }
    }
}

// PSSR_EPI_HEADQ (round 4): the inference form of Reconstruction.  In eval mode nothing needs `pre`'s 64-channel high-resolution
// activation again, so instead of writing it (1.07 GB at batch 32) for the head kernel to read back, the epilogue multiplies it with
// Reconstruction.conv's weights right out of the accumulators.  TR layout (conv_v3.h): a lane holds, for pixel lane & 31 of row tile mi,
// channels (e & 3) + 8 (e >> 2) + 4 (lane >> 5) of each 32-channel tile nj -- which IS the B-operand layout of v_mfma_f32_32x32x16 for
// the k slots 8 (lane >> 5) + i when element e = 8 ks + i of tile nj is taken as slot i of k-step (nj, ks).  So relu(acc + bias) is
// rounded to the storage type (exactly the value the store would have written), packed in place, and four MFMAs per row tile against
// the head weights arranged in the matching slot order (A operand: row = tap) leave q[tap][pixel] = sum over the wave's 64 channels
// = one sub-pixel: rows 0-3 and 8 with lanes 0-31, rows 4-7 with lanes 32-63.  36 bytes per high-resolution pixel leave instead of 128.
// (`hq`: [0, 576) Reconstruction.conv's weights [c][tap], [576, 704) pre's bias of the workgroup's 128 channels -- staged in LDS by
// conv_headq_stage at kernel entry: fetched here, 64 dependent-latency loads per lane stood in front of every workgroup's epilogue)
constexpr int HEADQ_LDS = (64 * 9 + 128) * 4;
__device__ __forceinline__ void conv_headq_stage(const ConvArgs& p, float* hq, int tid, int n0) {
    for (int i = tid; i < 64 * 9; i += 256) hq[i] = p.head_w[i];
    if (tid < 128) hq[576 + tid] = (p.bias && n0 + tid < p.cout) ? p.bias[n0 + tid] : 0.f;
}

template <typename T, class C>
__device__ __forceinline__ void conv_headq_epilogue(const ConvArgs& p, f32x16 (&acc)[C::MI][C::NJ], const float* hq, char* smem, int tid, int x0, int y0, int img0,
                                                    int n0) {
    using X = TT<T>;
    static_assert(sizeof(T) == 2 && C::NJ == 2 && C::MI == 4, "conv_v3 tiles of 128 channels");
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int h = lane >> 5, r = lane & 31;
    const int cbase = n0 + wn * 64;                  // first stored channel of this wave's 64 = sub * 64
    const int sub = cbase >> 6;                     // sub-pixel (i, j) = (sub >> 2, sub & 3) of the 4 x 4 block
    float bias[2][16];
    u32x4 wa[2][2];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
#pragma unroll
        for (int e = 0; e < 16; ++e) bias[nj][e] = hq[576 + wn * 64 + nj * 32 + (e & 3) + 8 * (e >> 2) + 4 * h];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = 8 * ks + i;
                const int c = nj * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;              // hidden channel of k slot (h, i)
                wv[i] = r < 9 ? hq[c * 9 + r] : 0.f;
            }
            wa[nj][ks] = X::pack(wv);
        }
    }
    // Every product goes to the plane of its (tap, sub-pixel) at LOW resolution, head_q[tap][sub][n][h][w]: the 16 lanes that hold one
    // image row of the tile write 64 contiguous bytes.  (The first versions stored at the high-resolution output position -- every
    // lane its own 4- or 8-byte piece of a different 32-byte sector: +150 us on c2's `pre`, an L2 write transaction per piece.)
    const long lrplane = (long)p.N * p.H * p.W;
    float* const qsub = p.head_q + (long)sub * lrplane + (long)img0 * p.H * p.W;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        f32x16 q;
#pragma unroll
        for (int e = 0; e < 16; ++e) q[e] = 0.f;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaxf(acc[mi][nj][8 * ks + i] + bias[nj][8 * ks + i], 0.f);
                X::mma(q, wa[nj][ks], X::pack(v));
            }
        const int m = epi_pixel<C>(wm * 32 + r, mi);
        const int tx = m & (C::TW - 1), ty = (m >> C::TWL) & (C::TH - 1);
        const int gy = y0 + ty, gx = x0 + tx;
        if (gy < p.H && gx < p.W) {
            float* const d = qsub + (long)gy * p.W + gx;
#pragma unroll
            for (int u = 0; u < 4; ++u) d[(long)(4 * h + u) * 16 * lrplane] = q[u];
            if (h == 0) d[(long)8 * 16 * lrplane] = q[4];
        }
    }
    (void)smem;
}

// picks the straight-line 8-channel epilogue when the layout allows (p.epi8, set by the host), else the generic one
// the straight-line 8-channel epilogue of p.epi (16-bit storage, p.epi8 layouts only)
template <typename T, int BN, class C, bool TR = false>
__device__ __forceinline__ void conv_epilogue8_any(const ConvArgs& p, f32x16 (&acc)[C::MI][C::NJ], char* smem, int tid, int x0, int y0, int img0, int n0) {
    const bool st = p.flags & PSSR_FLAG_STATS;
    switch (p.epi) {
    case PSSR_EPI_STORE:
        if (st) conv_epilogue8<T, BN, C, PSSR_EPI_STORE, true, TR>(p, acc, smem, tid, x0, y0, img0, n0);
        else if (p.flags & PSSR_FLAG_AFFINE) conv_epilogue8<T, BN, C, PSSR_EPI_STORE, false, TR, true>(p, acc, smem, tid, x0, y0, img0, n0);
        else conv_epilogue8<T, BN, C, PSSR_EPI_STORE, false, TR>(p, acc, smem, tid, x0, y0, img0, n0);
        return;
    case PSSR_EPI_TAIL: conv_epilogue8<T, BN, C, PSSR_EPI_TAIL, false, TR>(p, acc, smem, tid, x0, y0, img0, n0); return;
    case PSSR_EPI_DGRAD_MASK:
        if (st) conv_epilogue8<T, BN, C, PSSR_EPI_DGRAD_MASK, true, TR>(p, acc, smem, tid, x0, y0, img0, n0);
        else conv_epilogue8<T, BN, C, PSSR_EPI_DGRAD_MASK, false, TR>(p, acc, smem, tid, x0, y0, img0, n0);
        return;
    default:
        if (st) conv_epilogue8<T, BN, C, PSSR_EPI_DGRAD_GELU, true, TR>(p, acc, smem, tid, x0, y0, img0, n0);
        else conv_epilogue8<T, BN, C, PSSR_EPI_DGRAD_GELU, false, TR>(p, acc, smem, tid, x0, y0, img0, n0);
        return;
    }
}

template <typename T, int BN, class C>
__device__ __forceinline__ void conv_epilogue_any(const ConvArgs& p, f32x16 (&acc)[C::MI][C::NJ], char* smem, int tid, int x0, int y0, int img0, int n0) {
    if constexpr (sizeof(T) == 2) {
        if (p.epi8) { conv_epilogue8_any<T, BN, C>(p, acc, smem, tid, x0, y0, img0, n0); return; }
    }
    conv_epilogue<T, BN, C>(p, acc, smem, tid, x0, y0, img0, n0);
}

template <typename T, int BN, int GEO, int TAPS0>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    using C = Cfg<BN, GEO>;
    using X = TT<T>;
    constexpr int EPS = X::EPS, KCH = X::KCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + C::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int h = lane >> 5, r = lane & 31;

    // workgroups go round-robin to the 8 XCDs: renumber them so that consecutive tiles -- the channel tiles of one pixel tile first, then
    // its neighbour along x -- run on ONE XCD, whose L2 then serves the shared input tile (with bid % tiles_n on different XCDs every
    // L2 fetched it again: up to 8 x for the 1024-channel layers) and the shared halo columns (round 3; the 1x1 kernel below had it)
    int bid = blockIdx.x;
    if (p.ksplit <= 1 && !(p.dbg & 64)) {
        const int per = gridDim.x >> 3;
        if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
    }
    const int tn = bid % p.tiles_n;
    int tmi = bid / p.tiles_n;
    const int tile_x = tmi % p.tiles_x; tmi /= p.tiles_x;
    const int tile_y = tmi % p.tiles_y;
    const int tile_i = tmi / p.tiles_y;
    const int x0 = tile_x << C::TWL, y0 = tile_y << C::THL, img0 = tile_i * C::NI, n0 = tn * BN;

    // ---- per-thread staging descriptors for the input halo (same pixels for every chunk).  All addressing that does
    // not change from chunk to chunk lives in 32-bit buffer offsets computed once: the per-chunk part of every load
    // is a scalar offset, so the main loop issues no address arithmetic on the vector ALU (measured before this:
    // 7.5 vector instructions per MFMA, i.e. the VALU port, not the MFMA pipe, set the pace).
    const long img_base = (long)img0 * p.H * p.W;                  // first pixel of this tile's first image
    constexpr int ESZ = (int)sizeof(T);
    unsigned a_off0[C::A_ITEMS], a_off1[C::A_ITEMS];               // byte offsets from that pixel (source 0 may be in blocked order)
    int a_lds[C::A_ITEMS];
    bool a_ok[C::A_ITEMS];
    const int half = tid & 1;                                      // (tid + it * 256) & 1: the same 16-byte half for every item
#pragma unroll
    for (int it = 0; it < C::A_ITEMS; ++it) {
        const int idx = tid + it * 256;
        const int pp = idx >> 1;
        const int img = pp / C::HPI, rem = pp % C::HPI;
        const int hy = rem / C::HW2, hx = rem % C::HW2;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1, gi = img0 + img;
        const bool inb = (idx < 2 * C::HP) && gi < p.N && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_ok[it] = inb;
        a_off0[it] = inb ? (unsigned)((pix_index(gi, gy, gx, p.H, p.W, p.in0_blk) - img_base) * p.in_cs[0] * ESZ + half * 16) : 0u;
        a_off1[it] = inb ? (unsigned)(((long)img * p.H + gy) * p.W + gx) * (unsigned)(p.in_cs[1] * ESZ) + half * 16 : 0u;
        const int pl = img * C::HPIP + hy * C::HWP + hx;          // LDS slot of the pixel (rows padded to HWP)
        a_lds[it] = (idx < 2 * C::HP) ? (pl * 32 + ((half ^ ((pl >> 3) & 1)) << 4)) : -1;
    }
    // buffer resources (wave-uniform by construction: kernel arguments and blockIdx only)
    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.in[0] + (img_base * p.in_cs[0] + p.in_co[0]) * ESZ), 0, (int)0xfffffff0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.in[1] + (img_base * p.in_cs[1] + p.in_co[1]) * ESZ), 0, (int)0xfffffff0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w[0] + (long)n0 * 32), 0, (int)0xfffffff0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w[1] + (long)n0 * 32), 0, (int)0xfffffff0u, 0x00020000);
    // weights: piece idx = tid + it * 256 of the [taps][BN][32 B] slice of a chunk; PT pieces per tap, TPI taps per item
    constexpr int PT = BN * 2, TPI = 256 / PT;
    const int tap_stride = p.n_pad * 32;
    const unsigned b_voff = (unsigned)((tid / PT) * tap_stride + (tid % PT) * 16);
    const unsigned b_voff_tail = (unsigned)((tid % PT) * 16);      // partial last item: lanes past the slice re-read its first tap (never written)
    // ---- fragment read addresses (loop-invariant: one per accumulator row tile and tap)
    constexpr int CT = (TAPS0 == 9) ? 4 : 0;                       // index of the centre tap
    int a_rd[C::MI][TAPS0];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) {
        const int m = wm * C::MI * 32 + mi * 32 + r;
        const int tx = m & (C::TW - 1), ty = (m >> C::TWL) & (C::TH - 1), img = m >> (C::TWL + C::THL);
        const int p0 = img * C::HPIP + ty * C::HWP + tx;
#pragma unroll
        for (int t = 0; t < TAPS0; ++t) {
            const int ky = (TAPS0 == 9) ? t / 3 : 1, kx = (TAPS0 == 9) ? t % 3 : 1;
            const int pa = p0 + ky * C::HWP + kx;
            a_rd[mi][t] = pa * 32 + ((h ^ ((pa >> 3) & 1)) << 4);
        }
    }
    int b_off[C::NJ];
#pragma unroll
    for (int nj = 0; nj < C::NJ; ++nj) {
        const int n = wn * C::NJ * 32 + nj * 32 + r;
        b_off[nj] = n * 32 + ((h ^ ((n >> 3) & 1)) << 4);
    }

    f32x16 acc[C::MI][C::NJ];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

    const int total = p.nchunks[0] + p.nchunks[1];
    u32x4 a_reg[C::A_ITEMS];
    u32x4 b_reg[C::B_ITEMS];

    // BatchNorm prologue coefficients of source 0 live in LDS behind the tiles, one 32-byte-chunk half after the other:
    // [chunk][half][scale x EPS | shift x EPS], so that a thread's 2*EPS coefficients of a chunk are 4 (bf16) / 2 (f32)
    // 16-byte LDS reads at constant offsets from one address
    float* const tab = (float*)(smem + C::LDS_BYTES);
    if (p.prologue == PSSR_PRO_BN_RELU) {
        const int cin0 = p.nchunks[0] * KCH;
        for (int i = tid; i < cin0; i += 256) {
            const int c = i / KCH, w_ = i % KCH;
            float* q = tab + ((c * 2 + w_ / EPS) * 2) * EPS + (w_ % EPS);
            q[0] = p.pro_scale[i]; q[EPS] = p.pro_shift[i];
        }
        __syncthreads();
    }
    const float* const tab_t = tab + half * 2 * EPS;

    // NB: the three phases are macros, not lambdas: with lambdas hipcc keeps a_reg/b_reg in scratch.
    // S = source (0: TAPS taps, 1: the 1x1 second source), both compile-time
#define PSSR_ISSUE(S, CHUNK, TAPS)                                                                                \
    {                                                                                                             \
        /* unconditional loads (out-of-image items read pixel 0 and are zeroed at commit) */                      \
        const int asoff_ = (CHUNK) * 32;                                                                          \
        _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it)                                                 \
            a_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128((S) ? ra1 : ra0, (int)((S) ? a_off1[it] : a_off0[it]), asoff_, 0)); \
        const int wsoff_ = (CHUNK) * (TAPS) * tap_stride;                                                         \
        constexpr int NB_ = ((TAPS) * PT + 255) / 256;                                                            \
        _Pragma("unroll") for (int it = 0; it < NB_; ++it) {                                                      \
            const bool full_ = (it + 1) * 256 <= (TAPS) * PT;                                                     \
            b_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128((S) ? rw1 : rw0, (int)(full_ ? b_voff : b_voff_tail), wsoff_ + it * TPI * tap_stride, 0)); \
        }                                                                                                         \
    }
#define PSSR_COMMIT(S, CHUNK, TAPS)                                                                               \
    {                                                                                                             \
        const bool pro_ = ((S) == 0) && (p.prologue == PSSR_PRO_BN_RELU);                                         \
        const bool gelu_ = ((S) == 0) && (p.prologue == PSSR_PRO_GELU);                                           \
        if (pro_) {                                                                                               \
            float tsc_[EPS], tsh_[EPS];                                                                           \
            const float* tq_ = tab_t + (CHUNK) * (4 * EPS);                                                       \
            _Pragma("unroll") for (int e = 0; e < EPS; e += 4) { load4(tq_ + e, tsc_ + e); load4(tq_ + EPS + e, tsh_ + e); } \
            _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it) {                                           \
                if (a_lds[it] >= 0) {                                                                             \
                    u32x4 v = X::bn_relu(a_reg[it], tsc_, tsh_);                                                  \
                    if (!a_ok[it]) v = u32x4{0u, 0u, 0u, 0u};                                                     \
                    *(u32x4*)(As + a_lds[it]) = v;                                                                \
                }                                                                                                 \
            }                                                                                                     \
        } else {                                                                                                  \
            _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it) {                                           \
                if (a_lds[it] >= 0) {                                                                             \
                    u32x4 v = a_reg[it];                                                                          \
                    if (!a_ok[it]) v = u32x4{0u, 0u, 0u, 0u};                                                     \
                    if (gelu_) {      /* gelu(0) == 0: padding stays zero */                                      \
                        float f[EPS];                                                                             \
                        X::unpack(v, f);                                                                          \
                        gelu_vec<T, EPS>(f);                      \
                        v = X::pack(f);                                                                           \
                    }                                                                                             \
                    *(u32x4*)(As + a_lds[it]) = v;                                                                \
                }                                                                                                 \
            }                                                                                                     \
        }                                                                                                         \
        constexpr int NB_ = ((TAPS) * PT + 255) / 256;                                                            \
        _Pragma("unroll") for (int it = 0; it < NB_; ++it) {                                                      \
            const bool full_ = (it + 1) * 256 <= (TAPS) * PT;                                                     \
            if (full_ || tid < (TAPS) * PT - it * 256) *(u32x4*)(Bs + tid * 16 + it * 4096) = b_reg[it];          \
        }                                                                                                         \
    }
#define PSSR_FRAGS(T_, BUF)                                                                                       \
    {                                                                                                             \
        _Pragma("unroll") for (int mi = 0; mi < C::MI; ++mi)                                                      \
            af[BUF][mi] = *(const u32x4*)(As + a_rd[mi][(TAPS_) == TAPS0 ? (T_) : CT]);                           \
        _Pragma("unroll") for (int nj = 0; nj < C::NJ; ++nj) bf[BUF][nj] = *(const u32x4*)(Bs + (T_) * BN * 32 + b_off[nj]); \
    }
    // the fragments of tap t + 1 are requested before the MFMAs of tap t (two register sets): left to itself hipcc keeps
    // one or two reads of lookahead and every tap waits ~64 cycles for LDS
#define PSSR_COMPUTE(TAPS)                                                                                        \
    {                                                                                                             \
        constexpr int TAPS_ = (TAPS);                                                                             \
        u32x4 af[2][C::MI], bf[2][C::NJ];                                                                         \
        __builtin_amdgcn_s_setprio(2);      /* the multiplying wave goes first: the other workgroup is staging */ \
        PSSR_FRAGS(0, 0)                                                                                          \
        if (sizeof(T) == 2) __builtin_amdgcn_sched_group_barrier(0x100, C::MI + C::NJ, 0);                        \
        _Pragma("unroll") for (int t = 0; t < TAPS_; ++t) {                                                       \
            if (t + 1 < TAPS_) PSSR_FRAGS(t + 1, (t + 1) & 1)                                                     \
            _Pragma("unroll") for (int mi = 0; mi < C::MI; ++mi)                                                  \
                _Pragma("unroll") for (int nj = 0; nj < C::NJ; ++nj) X::mma(acc[mi][nj], af[t & 1][mi], bf[t & 1][nj]); \
            if (sizeof(T) == 2) {           /* pin the order: the next tap's LDS reads, then this tap's MFMAs */     \
                if (t + 1 < TAPS_) __builtin_amdgcn_sched_group_barrier(0x100, C::MI + C::NJ, 0);                 \
                __builtin_amdgcn_sched_group_barrier(0x008, C::MI * C::NJ, 0);                                    \
            }                                                                                                     \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    }

    // source 0 (TAPS0 taps) then the optional 1x1 source 1; the loads of the next chunk fly during the MFMAs
    // split-K (p.ksplit > 1): blockIdx.y multiplies chunks [cb, n0c) of source 0 only
    int cb = 0, n0c = p.nchunks[0];
    if (p.ksplit > 1) {
        const int per = (n0c + p.ksplit - 1) / p.ksplit;
        cb = blockIdx.y * per;
        n0c = cb + per < n0c ? cb + per : n0c;
    }
    const bool second = p.nchunks[1] > 0 && (p.ksplit <= 1 || (int)blockIdx.y == p.ksplit - 1);   // the 1x1 source rides with the last (shortest) slice
    if (cb < n0c) PSSR_ISSUE(0, cb, TAPS0)
    else if (second) PSSR_ISSUE(1, 0, 1)
    for (int i = cb; i < n0c; ++i) {
        PSSR_COMMIT(0, i, TAPS0)
        __syncthreads();
        if (i + 1 < n0c) PSSR_ISSUE(0, i + 1, TAPS0)
        else if (second) PSSR_ISSUE(1, 0, 1)
        PSSR_COMPUTE(TAPS0)
        __syncthreads();
    }
    if (second) {
        for (int i = 0; i < p.nchunks[1]; ++i) {
            PSSR_COMMIT(1, i, 1)
            __syncthreads();
            if (i + 1 < p.nchunks[1]) PSSR_ISSUE(1, i + 1, 1)
            PSSR_COMPUTE(1)
            __syncthreads();
        }
    }
#undef PSSR_ISSUE
#undef PSSR_COMMIT
#undef PSSR_COMPUTE
#undef PSSR_FRAGS

    if (p.ksplit > 1) {
        // raw accumulators of this K slice: ws[((block * ksplit + slice) * 16*MI*NJ + j) * 256 + tid] as float4 (coalesced)
        float4* dst = (float4*)p.ws + ((long)blockIdx.x * p.ksplit + blockIdx.y) * (4 * C::MI * C::NJ) * 256 + tid;
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[((mi * C::NJ + nj) * 4 + q) * 256] = make_float4(acc[mi][nj][4 * q], acc[mi][nj][4 * q + 1], acc[mi][nj][4 * q + 2], acc[mi][nj][4 * q + 3]);
        return;
    }
    conv_epilogue_any<T, BN, C>(p, acc, smem, tid, x0, y0, img0, n0);
}

// second pass of a split-K convolution: sum the K slices of a tile and run the ordinary epilogue
template <typename T, int BN, int GEO>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const ConvArgs p) {
    using C = Cfg<BN, GEO>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int tn = bid % p.tiles_n;
    int tmi = bid / p.tiles_n;
    const int tile_x = tmi % p.tiles_x; tmi /= p.tiles_x;
    const int tile_y = tmi % p.tiles_y;
    const int tile_i = tmi / p.tiles_y;
    const int x0 = tile_x << C::TWL, y0 = tile_y << C::THL, img0 = tile_i * C::NI, n0 = tn * BN;
    f32x16 acc[C::MI][C::NJ];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;
    for (int kz = 0; kz < p.ksplit; ++kz) {
        const float4* src = (const float4*)p.ws + ((long)bid * p.ksplit + kz) * (4 * C::MI * C::NJ) * 256 + tid;
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = src[((mi * C::NJ + nj) * 4 + q) * 256];
                    acc[mi][nj][4 * q] += v.x; acc[mi][nj][4 * q + 1] += v.y; acc[mi][nj][4 * q + 2] += v.z; acc[mi][nj][4 * q + 3] += v.w;
                }
    }
    conv_epilogue_any<T, BN, C>(p, acc, smem, tid, x0, y0, img0, n0);
}

// =================================================================================================================
// 1x1 convolutions (single source, no BatchNorm prologue).  With one tap the loop above multiplies only 4 MFMAs per wave
// between two barriers and a full staging round (measured 50-110 TFLOP/s on RDNet's 1x1 layers).  Here a stage is KC
// consecutive 32-byte channel chunks: a thread stages the same (pixel, 16-byte half) for all KC chunks (constant offsets
// from one address), the weight slices of KC chunks are one contiguous run of the 1-tap packed layout, and a wave
// multiplies 4 * KC MFMAs per stage -- the 9-tap loop's density without a halo.
template <typename T, int BN, int GEO, int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(KC <= 4 ? 3 : 2))) void conv_flat_kernel(const ConvArgs p) {
    using C = Cfg<BN, GEO>;
    using X = TT<T>;
    constexpr int EPS = X::EPS;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int A_BYTES = KC * 128 * 32;
    constexpr int PT = BN * 2, TPI = 256 / PT, NB = (KC * PT + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                        // [KC][128 px][32 B]
    char* Bs = smem + A_BYTES;              // [KC][BN][32 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int h = lane >> 5, r = lane & 31;
    // workgroups go round-robin to the 8 XCDs: renumber them so that consecutive tiles (the channel tiles of one pixel tile first)
    // run back to back on ONE XCD -- its L2 then serves the shared input tile and merges the 256-byte output segments of a
    // pixel row before they go to HBM (with bid % tiles_n on 8 different XCDs every L2 wrote its own fragment of each row)
    int bid = blockIdx.x;
    if (p.ksplit <= 1) {
        const int per = gridDim.x >> 3;
        if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
    }
    const int tn = bid % p.tiles_n;
    int tmi = bid / p.tiles_n;
    const int tile_x = tmi % p.tiles_x; tmi /= p.tiles_x;
    const int tile_y = tmi % p.tiles_y;
    const int tile_i = tmi / p.tiles_y;
    const int x0 = tile_x << C::TWL, y0 = tile_y << C::THL, img0 = tile_i * C::NI, n0 = tn * BN;

    // staging: thread <-> (pixel tid >> 1, half tid & 1), item <-> chunk of the stage
    const long img_base = (long)img0 * p.H * p.W;
    const int pm = tid >> 1, half = tid & 1;
    bool a_ok;
    unsigned a_off;
    {
        const int tx = pm & (C::TW - 1), ty = (pm >> C::TWL) & (C::TH - 1), img = pm >> (C::TWL + C::THL);
        const int gy = y0 + ty, gx = x0 + tx, gi = img0 + img;
        a_ok = gi < p.N && gy < p.H && gx < p.W;
        a_off = a_ok ? (unsigned)((pix_index(gi, gy, gx, p.H, p.W, p.in0_blk) - img_base) * p.in_cs[0] * ESZ + half * 16) : 0u;
    }
    const int a_lds = pm * 32 + ((half ^ ((pm >> 3) & 1)) << 4);
    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)p.in[0] + (img_base * p.in_cs[0] + p.in_co[0]) * ESZ), 0, (int)0xfffffff0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.w[0] + (long)n0 * 32), 0, (int)0xfffffff0u, 0x00020000);
    const int tap_stride = p.n_pad * 32;                            // one chunk of the 1-tap packed weights
    const unsigned b_voff = (unsigned)((tid / PT) * tap_stride + (tid % PT) * 16);
    const unsigned b_voff_tail = (unsigned)((tid % PT) * 16);
    int a_rd[C::MI];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) {
        const int m = wm * C::MI * 32 + mi * 32 + r;
        a_rd[mi] = m * 32 + ((h ^ ((m >> 3) & 1)) << 4);
    }
    int b_off[C::NJ];
#pragma unroll
    for (int nj = 0; nj < C::NJ; ++nj) {
        const int n = wn * C::NJ * 32 + nj * 32 + r;
        b_off[nj] = n * 32 + ((h ^ ((n >> 3) & 1)) << 4);
    }
    f32x16 acc[C::MI][C::NJ];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

    u32x4 a_reg[KC];
    u32x4 b_reg[NB];
    const int nch = p.nchunks[0];
#define PF_ISSUE(U)                                                                                               \
    {                                                                                                             \
        const int rem_ = nch - (U) * KC;          /* chunks left: the last stage may be short */                  \
        const int asoff_ = (U) * KC * 32, wsoff_ = (U) * KC * tap_stride;                                         \
        _Pragma("unroll") for (int it = 0; it < KC; ++it) {                                                       \
            if (it < rem_) a_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra0, (int)a_off, asoff_ + it * 32, 0)); \
            else a_reg[it] = u32x4{0u, 0u, 0u, 0u};                                                               \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < NB; ++it) {                                                       \
            const bool full_ = (it + 1) * 256 <= KC * PT;                                                         \
            unsigned vo_ = full_ ? b_voff : b_voff_tail;                                                          \
            if (rem_ < KC && it * TPI + (full_ ? tid / PT : 0) >= rem_) vo_ = 0xffffffffu;   /* zero-fill */      \
            b_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rw0, (int)vo_, wsoff_ + it * TPI * tap_stride, 0)); \
        }                                                                                                         \
    }
#define PF_COMMIT()                                                                                               \
    {                                                                                                             \
        _Pragma("unroll") for (int it = 0; it < KC; ++it) {                                                       \
            u32x4 v = a_reg[it];                                                                                  \
            if (!a_ok) v = u32x4{0u, 0u, 0u, 0u};                                                                 \
            if (p.prologue == PSSR_PRO_GELU) {      /* gelu(0) == 0 */                                            \
                float f[EPS];                                                                                     \
                X::unpack(v, f);                                                                                  \
                gelu_vec<T, EPS>(f);                              \
                v = X::pack(f);                                                                                   \
            }                                                                                                     \
            *(u32x4*)(As + it * 4096 + a_lds) = v;                                                                \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < NB; ++it) {                                                       \
            const bool full_ = (it + 1) * 256 <= KC * PT;                                                         \
            if (full_ || tid < KC * PT - it * 256) *(u32x4*)(Bs + tid * 16 + it * 4096) = b_reg[it];              \
        }                                                                                                         \
    }
    int cb = 0, n0c = (nch + KC - 1) / KC;        // stages; split-K: blockIdx.y owns a range of them
    if (p.ksplit > 1) {
        const int per = (n0c + p.ksplit - 1) / p.ksplit;
        cb = blockIdx.y * per;
        n0c = cb + per < n0c ? cb + per : n0c;
    }
    if (cb < n0c) PF_ISSUE(cb)
    for (int u = cb; u < n0c; ++u) {
        PF_COMMIT()
        __syncthreads();
        if (u + 1 < n0c) PF_ISSUE(u + 1)
#pragma unroll
        for (int t = 0; t < KC; ++t) {
            u32x4 af[C::MI], bf[C::NJ];
#pragma unroll
            for (int mi = 0; mi < C::MI; ++mi) af[mi] = *(const u32x4*)(As + t * 4096 + a_rd[mi]);
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj) bf[nj] = *(const u32x4*)(Bs + t * BN * 32 + b_off[nj]);
            if (!(p.dbg & 2)) {          // (diagnostic bit 2: no multiplies)
#pragma unroll
                for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < C::NJ; ++nj) X::mma(acc[mi][nj], af[mi], bf[nj]);
            }
        }
        __syncthreads();
    }
#undef PF_ISSUE
#undef PF_COMMIT
    if (p.dbg & 1) return;               // (diagnostic bit 1: no epilogue)
    if (p.ksplit > 1) {
        float4* dst = (float4*)p.ws + ((long)blockIdx.x * p.ksplit + blockIdx.y) * (4 * C::MI * C::NJ) * 256 + tid;
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    dst[((mi * C::NJ + nj) * 4 + q) * 256] = make_float4(acc[mi][nj][4 * q], acc[mi][nj][4 * q + 1], acc[mi][nj][4 * q + 2], acc[mi][nj][4 * q + 3]);
        return;
    }
    conv_epilogue_any<T, BN, C>(p, acc, smem, tid, x0, y0, img0, n0);
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

}  // namespace
#include "conv_v3.h"
namespace {

// second pass of a split-K v3 launch
template <typename T, int BN>
__global__ __launch_bounds__(256) void conv_splitk_finish3_kernel(const ConvArgs p) {
    using C = Cfg3<BN>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int bid = blockIdx.x;
    const int tn = bid % p.tiles_n;
    int tmi = bid / p.tiles_n;
    const int tile_x = tmi % p.tiles_x; tmi /= p.tiles_x;
    const int tile_y = tmi % p.tiles_y;
    const int img0 = tmi / p.tiles_y;
    const int x0 = tile_x << C::TWL, y0 = tile_y << C::THL, n0 = tn * BN;
    f32x16 acc[C::MI][C::NJ];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;
    for (int kz = 0; kz < p.ksplit; ++kz) {
        const float4* src = (const float4*)p.ws + ((long)bid * p.ksplit + kz) * (4 * C::MI * C::NJ) * 256 + tid;
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
            for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = src[((mi * C::NJ + nj) * 4 + q) * 256];
                    acc[mi][nj][4 * q] += v.x; acc[mi][nj][4 * q + 1] += v.y; acc[mi][nj][4 * q + 2] += v.z; acc[mi][nj][4 * q + 3] += v.w;
                }
    }
    conv_epilogue8_any<T, BN, C, true>(p, acc, smem, tid, x0, y0, img0, n0);
}

// v3 launch: 16x16-pixel x 128-channel tiles; split-K when the tiles leave more than half of the 512 workgroup slots empty
template <typename T, int BN>
int launch3_t(const ConvArgs& a, hipStream_t stream) {
    using C = Cfg3<BN>;
    ConvArgs p = a;
    p.tiles_x = cdiv(a.W, C::TW);
    p.tiles_y = cdiv(a.H, C::TH);
    p.tiles_n = cdiv(a.cout, BN);
    const long blocks = (long)p.tiles_x * p.tiles_y * a.N * p.tiles_n;
    PSSR_CHECK(blocks > 0 && blocks < (1L << 31), PSSR_ERR_ARG, "conv2d: bad grid %ld", blocks);
    int ksplit = 1;
    if (a.epi != PSSR_EPI_FINAL && a.epi != PSSR_EPI_HEADQ && !(a.flags & PSSR_FLAG_HEADQ) && blocks < 192 && a.nchunks[0] >= 8) {
        ksplit = (int)((256 + blocks - 1) / blocks);
        if (ksplit > a.nchunks[0] / 4) ksplit = a.nchunks[0] / 4;
        if (ksplit > 8) ksplit = 8;
    }
    const int pro_lds = a.prologue == PSSR_PRO_BN_RELU ? a.nchunks[0] * TT<T>::KCH * 8 : 0;
    PSSR_CHECK(pro_lds <= PRO_LDS_MAX, PSSR_ERR_UNSUPPORTED, "conv2d: BatchNorm prologue over %d input channels", a.nchunks[0] * TT<T>::KCH);
    const long ws_bytes = ksplit > 1 ? blocks * ksplit * (long)(64 * C::MI * C::NJ) * 256 : 0;
    if (a.ksplit < 0) { *(long*)a.ws = ws_bytes; return PSSR_OK; }
    if (ksplit > 1 && (a.ws == nullptr || (long)a.ksplit * 1024 < ws_bytes)) ksplit = 1;
    p.ksplit = ksplit;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_v3_kernel<T, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_splitk_finish3_kernel<T, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_done = true;
    }
    // V3_LDS_PAD (KiB, experiment): unused LDS that keeps a second workgroup off the CU -- halves the input lines an XCD's L2 has to
    // hold between the K chunks of a tile
    int pad = pssr_tunables().v3_lds_pad * 1024;
    if (C::LDS_BYTES + pro_lds + pad + HEADQ_LDS > 160 * 1024) pad = 160 * 1024 - C::LDS_BYTES - pro_lds - HEADQ_LDS;
    const int hq_lds = (a.epi == PSSR_EPI_HEADQ || (a.flags & PSSR_FLAG_HEADQ)) ? HEADQ_LDS : 0;          // (no BatchNorm prologue table with it: pre reads activated inputs)
    PSSR_CHECK(!(hq_lds && pro_lds), PSSR_ERR_UNSUPPORTED, "conv2d: EPI_HEADQ with a BatchNorm prologue");
    hipLaunchKernelGGL((conv_v3_kernel<T, BN>), dim3((unsigned)blocks, ksplit), dim3(256), C::LDS_BYTES + pro_lds + pad + hq_lds, stream, p);
    if (ksplit > 1)
        hipLaunchKernelGGL((conv_splitk_finish3_kernel<T, BN>), dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, stream, p);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

// v3 takes the 3x3 layers (16-bit storage) whose images hold a 16x16 tile; g_v3_mode 0 disables it (tests run both loops)
template <typename T, int BN>
bool use_v3(const ConvArgs& a) {
    if constexpr (sizeof(T) != 2) return false;
    const int mode = pssr_tunables().igemm_v3;
    constexpr int TW = BN == 128 ? 16 : 32;
    if (BN == 64 && !pssr_tunables().igemm_v3_64) return false;
    if (!mode || a.taps[0] != 9 || a.W < TW || a.H < 16 || a.prologue == PSSR_PRO_GELU || a.epi == PSSR_EPI_FINAL || !a.epi8) return false;
    if (mode == 2) return true;             // tests: whenever the shape allows
    // two workgroups per CU cover each other's barriers, pipeline fill and epilogue: with fewer than ~1.5 tiles per CU the
    // 128-pixel loop (twice the workgroups) measured faster (32^2 x 256 and 16^2 x 512 layers at batch 32: 828 vs 757, 713 vs 643 TFLOP/s)
    const long blocks = (long)cdiv(a.W, TW) * cdiv(a.H, 16) * a.N * cdiv(a.cout, BN);
    return blocks >= 384;
}

// workgroups a split-K launch aims for (PSSR_IGEMM_KSPLIT overrides)
static inline int ksplit_target() { return pssr_tunables().igemm_ksplit; }

template <typename T, int BN, int GEO, int TAPS0>
int launch_t(const ConvArgs& a, hipStream_t stream) {
    using C = Cfg<BN, GEO>;
    ConvArgs p = a;
    p.tiles_x = cdiv(a.W, C::TW);
    p.tiles_y = cdiv(a.H, C::TH);
    p.tiles_n = cdiv(a.cout, BN);
    const long blocks = (long)p.tiles_x * p.tiles_y * cdiv(a.N, C::NI) * p.tiles_n;
    PSSR_CHECK(blocks > 0 && blocks < (1L << 31), PSSR_ERR_ARG, "conv2d: bad grid %ld", blocks);
    // split-K when the tiles do not fill the chip and K is long: ~384 workgroups, >= 4 chunks per slice, <= 8 slices
    int ksplit = 1;
    if (BN >= 64 && a.epi != PSSR_EPI_FINAL && blocks < 192 && a.nchunks[0] >= 8) {
        ksplit = (int)((ksplit_target() + blocks - 1) / blocks);
        if (ksplit > a.nchunks[0] / 4) ksplit = a.nchunks[0] / 4;
        if (ksplit > 8) ksplit = 8;
    }
    const int pro_lds = a.prologue == PSSR_PRO_BN_RELU ? a.nchunks[0] * TT<T>::KCH * 8 : 0;     // prologue table behind the tiles
    PSSR_CHECK(pro_lds <= PRO_LDS_MAX, PSSR_ERR_UNSUPPORTED, "conv2d: BatchNorm prologue over %d input channels", a.nchunks[0] * TT<T>::KCH);
    const long ws_bytes = ksplit > 1 ? blocks * ksplit * (long)(64 * C::MI * C::NJ) * 256 : 0;
    if (a.ksplit < 0) { *(long*)a.ws = ws_bytes; return PSSR_OK; }          // workspace-size query
    if (ksplit > 1 && (a.ws == nullptr || (long)a.ksplit * 1024 < ws_bytes)) ksplit = 1;   // caller gave no (or too small a) workspace
    p.ksplit = ksplit;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BN, GEO, TAPS0>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES + PRO_LDS_MAX);
        (void)hipFuncSetAttribute((const void*)conv_splitk_finish_kernel<T, BN, GEO>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, BN, GEO, TAPS0>), dim3((unsigned)blocks, ksplit), dim3(256), C::LDS_BYTES + pro_lds, stream, p);
    if (ksplit > 1)
        hipLaunchKernelGGL((conv_splitk_finish_kernel<T, BN, GEO>), dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, stream, p);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}


template <typename T, int BN, int GEO, int KC>
int launch_flat(const ConvArgs& a, hipStream_t stream) {
    using C = Cfg<BN, GEO>;
    constexpr int MAIN = KC * 128 * 32 + KC * BN * 32;
    constexpr int LDS = MAIN > C::E_BYTES + C::RED_BYTES ? MAIN : C::E_BYTES + C::RED_BYTES;
    ConvArgs p = a;
    p.tiles_x = cdiv(a.W, C::TW);
    p.tiles_y = cdiv(a.H, C::TH);
    p.tiles_n = cdiv(a.cout, BN);
    const long blocks = (long)p.tiles_x * p.tiles_y * cdiv(a.N, C::NI) * p.tiles_n;
    PSSR_CHECK(blocks > 0 && blocks < (1L << 31), PSSR_ERR_ARG, "conv2d: bad grid %ld", blocks);
    const int stages = cdiv(a.nchunks[0], KC);
    int ksplit = 1;      // same policy as the 9-tap loop, in stages (a stage ~ a 9-tap chunk)
    if (BN >= 64 && a.epi != PSSR_EPI_FINAL && blocks < 192 && stages >= 4) {
        ksplit = (int)((ksplit_target() + blocks - 1) / blocks);
        if (ksplit > stages / 2) ksplit = stages / 2;
        if (ksplit > 8) ksplit = 8;
    }
    const long ws_bytes = ksplit > 1 ? blocks * ksplit * (long)(64 * C::MI * C::NJ) * 256 : 0;
    if (a.ksplit < 0) { *(long*)a.ws = ws_bytes; return PSSR_OK; }
    if (ksplit > 1 && (a.ws == nullptr || (long)a.ksplit * 1024 < ws_bytes)) ksplit = 1;
    p.ksplit = ksplit;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_flat_kernel<T, BN, GEO, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void*)conv_splitk_finish_kernel<T, BN, GEO>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_flat_kernel<T, BN, GEO, KC>), dim3((unsigned)blocks, ksplit), dim3(256), LDS, stream, p);
    if (ksplit > 1)
        hipLaunchKernelGGL((conv_splitk_finish_kernel<T, BN, GEO>), dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, stream, p);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

template <typename T, int BN, int GEO>
int launch(const ConvArgs& a, hipStream_t s) {
    if (a.taps[0] == 1 && a.nchunks[1] == 0 && a.prologue != PSSR_PRO_BN_RELU && pssr_tunables().igemm_flat) {
        // stage length: 9 chunks unless 4 wastes clearly less of the last stage (K = 64: 4 chunks)
        const int n = a.nchunks[0];
        const bool nine = n > 8 && cdiv(n, 9) * 9 * 100 <= cdiv(n, 4) * 4 * 115;
        return nine ? launch_flat<T, BN, GEO, 9>(a, s) : launch_flat<T, BN, GEO, 4>(a, s);
    }
    return a.taps[0] == 9 ? launch_t<T, BN, GEO, 9>(a, s) : launch_t<T, BN, GEO, 1>(a, s);
}

template <typename T, int BN>
int launch_geo(const ConvArgs& a, hipStream_t s) {
    const int w = a.W;
    if constexpr (BN == 64 && sizeof(T) == 2) {
        if (pssr_tunables().igemm_big && a.taps[0] == 9 && w >= 16 && a.H >= 16) return launch_t<T, 64, 5, 9>(a, s);
    }
    if (w > 8) return launch<T, BN, 0>(a, s);
    if (w > 4) return launch<T, BN, 1>(a, s);
    if (w > 2) return launch<T, BN, 2>(a, s);
    if (w > 1) return launch<T, BN, 3>(a, s);
    return launch<T, BN, 4>(a, s);
}

template <typename T>
int launch_bn(const ConvArgs& a, hipStream_t s) {
    if (a.cout > 64) {
        // (16x32-pixel x 64-channel v3 tiles for these wider layers too -- twice the workgroups, two rounds that drift apart instead of
        // one in lockstep -- measured 49.3 vs 47.7 us on 128 -> 128 @64^2: no)
        if constexpr (sizeof(T) == 2) {
            if (a.epi == PSSR_EPI_HEADQ || (a.flags & PSSR_FLAG_HEADQ)) {            // the tap-product epilogue exists in the conv_v3 tiles only (shape checked by the entry)
                PSSR_CHECK(a.epi8 && pssr_tunables().igemm_v3, PSSR_ERR_UNSUPPORTED, "conv2d: EPI_HEADQ needs the conv_v3 loop");
                return launch3_t<T, 128>(a, s);
            }
        }
        PSSR_CHECK(a.epi != PSSR_EPI_HEADQ && !(a.flags & PSSR_FLAG_HEADQ), PSSR_ERR_UNSUPPORTED, "conv2d: EPI_HEADQ needs 16-bit storage");
        if constexpr (sizeof(T) == 2) { if (use_v3<T, 128>(a)) return launch3_t<T, 128>(a, s); }
        if constexpr (sizeof(T) == 2) {
            // layers whose 128 x 128 tiles give about ONE workgroup per CU (512 channels on the 16 x 16 maps at batch 32) take 128 x 64 tiles:
            // two co-resident workgroups cover each other's barriers and staging latency (512 -> 512 @16^2 57.0 -> 45.2 us, 768 -> 512
            // 72.3 -> 62.8).  Only for launches that have the chip to themselves (FLAG_SOLO: forward passes): in the backward pass the wider
            // launches took from the weight-gradient stream what they gained (kernel trace: forward -33 us, backward +40 us)
            if (pssr_tunables().igemm_n64 && (a.flags & PSSR_FLAG_SOLO) && a.taps[0] == 9 && a.W > 8) {
                const long b128 = (long)cdiv(a.W, 16) * cdiv(a.H, 8) * a.N * cdiv(a.cout, 128);
                if (b128 >= 192 && b128 <= 320) return launch<T, 64, 0>(a, s);
            }
        }
        if constexpr (sizeof(T) == 2) {
            if (pssr_tunables().igemm_big == 2 && a.taps[0] == 9 && a.W >= 16 && a.H >= 16) return launch_geo<T, 64>(a, s);   // 256 x 64 tiles for wide layers too
        }
        return launch_geo<T, 128>(a, s);
    }
    PSSR_CHECK(a.epi != PSSR_EPI_HEADQ && !(a.flags & PSSR_FLAG_HEADQ), PSSR_ERR_UNSUPPORTED, "conv2d: EPI_HEADQ needs cout > 64");
    if (a.cout > 32) {
        if constexpr (sizeof(T) == 2) { if (use_v3<T, 64>(a)) return launch3_t<T, 64>(a, s); }
        return launch_geo<T, 64>(a, s);
    }
    return launch_geo<T, 32>(a, s);
}

}  // namespace
