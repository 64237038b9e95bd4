// Reconstruction.conv (pssr/models/_blocks.py:11,17): the last 3x3 convolution, hidden -> C_out with C_out = 1..3, on the
// 4x-upsampled image.  It touches the largest tensor of the network (hidden channels at HR resolution, 1.07 GB in bf16 at
// batch 32) and has almost no arithmetic, so the generic implicit-GEMM tiles (>= 32 output channels) waste the matrix
// cores and, worse, the memory system.  These three kernels stream that tensor exactly once each:
//
//   forward   Z[q][(co,tap)] = sum_ci P[q][ci] * W[co][ci][tap]   (one 16x16x32 MFMA chain per 16 pixels, taps in GEMM-N),
//             out[p][co] = sum_tap Z[p + off(tap)][(co,tap)]      (9-point gather from LDS)
//   dgrad     dP[q][ci] = relu'(P[q][ci]) * sum_(co,tap) g[q - off(tap)][co] * W[co][ci][tap]   (gathered g in GEMM-K)
//   wgrad     dW[co][ci][tap] = sum_q g[q - off(tap)][co] * P[q][ci]   (f32 FMAs, g tile in LDS, P read coalesced once)
//
// P / dP are NHWC in the "blocked" pixel order of the pixel-shuffle (see pssr_conv_desc), g is the incoming gradient
// d(out)/d(x*scale+shift) taken straight from the f32 NCHW tensor autograd hands over.  16-bit storage (bf16 / fp16) only:
// the exact-f32 parity build keeps the generic kernels.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <typename H> struct HV;
template <> struct HV<bf16_t> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x4_t mma(v8 a, v8 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct HV<f16_t> {
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x4_t mma(v8 a, v8 b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

constexpr int TS = 16;                 // output tile edge
constexpr int HS = TS + 2;             // halo tile edge
constexpr int HPIX = HS * HS;          // 324
constexpr int HGROUPS = (HPIX + 15) / 16;   // 21 groups of 16 halo pixels

static unsigned* g_head_stamps = nullptr;      // diagnostic build (-DPSSR_WG_STAMPS) only

struct HeadArgs {
    const void* P; int p_cs, p_co, blk;            // activations (blocked NHWC, bf16 or fp16), cin channels
    void* dP; int dp_cs, dp_co;                     // dgrad output (same layout)
    const float* w;                                 // OIHW f32 [cout][cin][3][3]
    const float* bias;
    float* out;                                     // forward: f32 NCHW [n][cout][H][W]
    const float* g;                                 // backward: f32 NCHW [n][cout][H][W]
    float* dw;                                      // wgrad: f32 OIHW, atomically accumulated
    int N, H, W, cin, cout, tiles_x, tiles_y;
    float out_scale, out_shift, g_scale;
};

template <typename H> __device__ __forceinline__ typename HV<H>::v8 zero_frag() { typename HV<H>::v8 z; for (int j = 0; j < 8; ++j) z[j] = (H)0.f; return z; }

// B fragment of the (co,tap)-major weight matrix: forward B[k = ci][n = co*9+tap]
template <typename H> __device__ __forceinline__ typename HV<H>::v8 wfrag_fwd(const HeadArgs& p, int ks, int nt, int lane) {
    typename HV<H>::v8 b;
    const int m = nt * 16 + (lane & 15);
    const int co = m / 9, tap = m % 9;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ks * 32 + 8 * (lane >> 4) + j;
        b[j] = (H)((co < p.cout && ci < p.cin) ? p.w[((long)co * p.cin + ci) * 9 + tap] : 0.f);
    }
    return b;
}
// dgrad B[k = co*9+tap][n = ci]
template <typename H> __device__ __forceinline__ typename HV<H>::v8 wfrag_bwd(const HeadArgs& p, int nt, int lane) {
    typename HV<H>::v8 b;
    const int ci = nt * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = 8 * (lane >> 4) + j;
        const int co = m / 9, tap = m % 9;
        b[j] = (H)((co < p.cout && ci < p.cin) ? p.w[((long)co * p.cin + ci) * 9 + tap] : 0.f);
    }
    return b;
}

// ------------------------------------------------------------------------------------------------ forward
template <typename H, int NT, int KS>     // NT = ceil(cout*9/16) N tiles, KS = cin/32 K steps
__global__ __launch_bounds__(256) void head_fwd_kernel(const HeadArgs p) {
    typedef typename HV<H>::v8 v8;
    constexpr int ZS = NT * 16 + 1;                       // padded row of the tap-product image
    __shared__ float Z[HGROUPS * 16 * ZS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx0 = (t % p.tiles_x) * TS; t /= p.tiles_x;
    const int ty0 = (t % p.tiles_y) * TS;
    const int img = t / p.tiles_y;
    v8 bw[KS][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[ks][nt] = wfrag_fwd<H>(p, ks, nt, lane);
    // a wave owns halo groups wave, wave+4, ...: all of its loads are issued before the first MFMA (a constant trip count
    // so that the loop unrolls; the activation stream is the only HBM traffic of this kernel)
    constexpr int GPW = (HGROUPS + 3) / 4;
    v8 av[GPW][KS];
    bool okv[GPW];
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        const int hp = (wave + 4 * j) * 16 + (lane & 15);
        const int gy = ty0 + hp / HS - 1, gx = tx0 + hp % HS - 1;
        okv[j] = hp < HPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const H* src = (const H*)p.P + pix_index(img, okv[j] ? gy : 0, okv[j] ? gx : 0, p.H, p.W, p.blk) * p.p_cs + p.p_co + 8 * (lane >> 4);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) av[j][ks] = *(const v8*)(src + ks * 32);
    }
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        const int g = wave + 4 * j;
        if (g >= HGROUPS) continue;
        f32x4_t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const v8 a = okv[j] ? av[j][ks] : zero_frag<H>();
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = HV<H>::mma(a, bw[ks][nt], acc[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Z[(g * 16 + (lane >> 4) * 4 + r) * ZS + nt * 16 + (lane & 15)] = acc[nt][r];
    }
    __syncthreads();
    const int ty = tid / TS, tx = tid % TS;
    const int gy = ty0 + ty, gx = tx0 + tx;
    if (gy < p.H && gx < p.W) {
        for (int co = 0; co < p.cout; ++co) {
            float s = p.bias ? p.bias[co] : 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) s += Z[((ty + tap / 3) * HS + tx + tap % 3) * ZS + co * 9 + tap];
            p.out[(((long)img * p.cout + co) * p.H + gy) * p.W + gx] = fmaf(s, p.out_scale, p.out_shift);
        }
    }
}

// ------------------------------------------------------------------------------------------------ sum of the tap products
// Second half of PSSR_EPI_HEADQ / PSSR_FLAG_HEADQ (conv_igemm_impl.h: conv_headq_epilogue): q[tap][sub][n][h][w] holds, for the
// high-resolution pixel (4 y + i, 4 x + j) = (low-resolution pixel (y, x), sub-pixel sub = 4 i + j), its product with tap (ky, kx) of
// Reconstruction.conv.  out[P] = bias + sum_tap q[tap][P + (ky - 1, kx - 1)] (zero outside the image).  A thread makes the four outputs
// (4 y + i, 4 x .. 4 x + 3) of one low-resolution pixel row position; lanes walk x, so each of its 36 reads is a unit-stride wave access
// into one (tap, sub-pixel) plane and the store is one float4.
__global__ __launch_bounds__(256) void head_q_gather_kernel(const float* __restrict__ q, const float* __restrict__ bias, float* __restrict__ out,
                                                            int n, int h, int w, float out_scale, float out_shift) {
    const long lrplane = (long)n * h * w;
    const float b = bias ? bias[0] : 0.f;
    const long total = lrplane * 4;                     // (image, y, i, x)
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int x = (int)(t % w);
        long r = t / w;
        const int i = (int)(r & 3); r >>= 2;
        const int y = (int)(r % h);
        const long img = r / h;
        float s[4] = {b, b, b, b};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int Y = 4 * y + i + ky - 1;
            if (Y < 0 || Y >= 4 * h) continue;
            const int yy = Y >> 2, ii = Y & 3;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* const qt = q + ((long)(ky * 3 + kx) * 16 + ii * 4) * lrplane + (img * h + yy) * (long)w;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int X = 4 * x + j + kx - 1;           // = 4 xx + jj
                    if (X < 0 || X >= 4 * w) continue;
                    s[j] += qt[(long)(X & 3) * lrplane + (X >> 2)];
                }
            }
        }
        *(float4*)(out + ((img * h + y) * 4 + i) * (4L * w) + 4 * x) =
            make_float4(fmaf(s[0], out_scale, out_shift), fmaf(s[1], out_scale, out_shift), fmaf(s[2], out_scale, out_shift), fmaf(s[3], out_scale, out_shift));
    }
}

// The same sums for widths that are multiples of 4 (every tile size the drivers use): a thread makes the 16 outputs (4 y + i, 16 x4 .. 16 x4 + 15)
// from 36 16-byte loads (a (tap, sub-pixel) plane row at 4 x4 .. 4 x4 + 3; the six planes that feed from the neighbouring low-resolution
// column add one 4-byte load each) and stores 64 contiguous bytes.  4-byte loads stream at 3.5 TB/s here, 16-byte ones at the rate of a copy.
// Per output the order of the nine additions is that of the kernel above: the two are bit-identical.
__global__ __launch_bounds__(256) void head_q_gather4_kernel(const float* __restrict__ q, const float* __restrict__ bias, float* __restrict__ out,
                                                             int n, int h, int w, float out_scale, float out_shift) {
    const long lrplane = (long)n * h * w;
    const int w4 = w >> 2;
    const float b = bias ? bias[0] : 0.f;
    const long total = (long)n * h * 4 * w4;            // (image, y, i, x4)
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int x4 = (int)(t % w4);
        long r = t / w4;
        const int i = (int)(r & 3); r >>= 2;
        const int y = (int)(r % h);
        const long img = r / h;
        float s[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[a][j] = b;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int Y = 4 * y + i + ky - 1;
            if (Y < 0 || Y >= 4 * h) continue;
            const int yy = Y >> 2, ii = Y & 3;
            const float* const row = q + ((long)(ky * 3) * 16 + ii * 4) * lrplane + (img * h + yy) * (long)w + 4 * x4;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = j + kx - 1;                   // high-resolution column 4 x + d of low-resolution column x
                    const float* const pl = row + ((long)kx * 16 + (d & 3)) * lrplane;
                    const float4 v = *(const float4*)pl;
                    if (d < 0) {
                        const float prev = x4 > 0 ? pl[-1] : 0.f;
                        s[0][j] += prev; s[1][j] += v.x; s[2][j] += v.y; s[3][j] += v.z;
                    } else if (d > 3) {
                        const float next = x4 < w4 - 1 ? pl[4] : 0.f;
                        s[0][j] += v.y; s[1][j] += v.z; s[2][j] += v.w; s[3][j] += next;
                    } else {
                        s[0][j] += v.x; s[1][j] += v.y; s[2][j] += v.z; s[3][j] += v.w;
                    }
                }
            }
        }
        float* const o = out + ((img * h + y) * 4 + i) * (4L * w) + 16 * x4;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            *(float4*)(o + 4 * a) = make_float4(fmaf(s[a][0], out_scale, out_shift), fmaf(s[a][1], out_scale, out_shift),
                                                fmaf(s[a][2], out_scale, out_shift), fmaf(s[a][3], out_scale, out_shift));
    }
}

// ------------------------------------------------------------------------------------------------ dgrad (+ ReLU mask)
template <typename H, int NT>     // NT = cin/16 output-channel tiles; cout*9 <= 32 (one K step)
__global__ __launch_bounds__(256) void head_dgrad_kernel(const HeadArgs p) {
    typedef typename HV<H>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* G = (float*)smem;                               // [cout][HS][HS] incoming gradient tile (+halo), scaled
    H* O = (H*)(smem + ((3 * HPIX * 4 + 15) / 16) * 16);   // [256 px][NT*16] result tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int t = blockIdx.x;
    const int tx0 = (t % p.tiles_x) * TS; t /= p.tiles_x;
    const int ty0 = (t % p.tiles_y) * TS;
    const int img = t / p.tiles_y;
    for (int i = tid; i < p.cout * HPIX; i += 256) {
        const int co = i / HPIX, hp = i % HPIX;
        const int gy = ty0 + hp / HS - 1, gx = tx0 + hp % HS - 1;
        G[i] = (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? p.g[(((long)img * p.cout + co) * p.H + gy) * p.W + gx] * p.g_scale : 0.f;
    }
    v8 bw[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[nt] = wfrag_bwd<H>(p, nt, lane);
    __syncthreads();
    constexpr int C = NT * 16;
    for (int g = wave; g < 16; g += 4) {                   // 16 groups of 16 output pixels = tile rows
        const int px = lane & 15, py = g;                  // group g = tile row g
        v8 a;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = 8 * (lane >> 4) + j;
            const int co = m / 9, tap = m % 9;
            // dP[q] += g[q - off(tap)] * W[tap], off = (ky-1, kx-1): halo coordinates (py + 1 - (ky-1), px + 1 - (kx-1))
            const float v = co < p.cout ? G[co * HPIX + (py + 2 - tap / 3) * HS + px + 2 - tap % 3] : 0.f;
            a[j] = (H)v;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4_t acc = HV<H>::mma(a, bw[nt], f32x4_t{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) O[(g * 16 + (lane >> 4) * 4 + r) * C + nt * 16 + (lane & 15)] = (H)acc[r];
        }
    }
    __syncthreads();
    // coalesced write-out with the ReLU mask of the forward activation: 16-byte pieces, whole pixel rows per wave
    constexpr int PPP = C / 8;                             // pieces per pixel
    v8 actv[PPP];
    long qv[PPP];
#pragma unroll
    for (int u = 0; u < PPP; ++u) {                        // PPP trips of 256 pieces: all activation loads first
        const int i = tid + u * 256;
        const int pix = i / PPP, pc = i % PPP;
        const int gy = ty0 + pix / TS, gx = tx0 + pix % TS;
        const bool ok = gy < p.H && gx < p.W;
        qv[u] = ok ? pix_index(img, gy, gx, p.H, p.W, p.blk) : -1;
        actv[u] = *(const v8*)((const H*)p.P + (ok ? qv[u] : pix_index(img, ty0, tx0, p.H, p.W, p.blk)) * p.p_cs + p.p_co + pc * 8);
    }
#pragma unroll
    for (int u = 0; u < PPP; ++u) {
        if (qv[u] < 0) continue;
        const int i = tid + u * 256;
        const int pix = i / PPP, pc = i % PPP;
        v8 v = *(const v8*)(O + pix * C + pc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (!((float)actv[u][j] > 0.f)) v[j] = (H)0.f;
        *(v8*)((H*)p.dP + qv[u] * p.dp_cs + p.dp_co + pc * 8) = v;
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
template <typename H, int COUT>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const HeadArgs p, int n_tiles) {
    __shared__ float G[COUT * HPIX];
    __shared__ float R[256 * 4];
    const int tid = threadIdx.x;
    const int cgc = p.cin / 4, ppb = 256 / cgc;            // channel groups, pixel lanes
    const int cg = tid % cgc, pl = tid / cgc;
    const bool active = pl < ppb;
    float acc[COUT][9][4];
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[co][tap][e] = 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int t = tile;
        const int tx0 = (t % p.tiles_x) * TS; t /= p.tiles_x;
        const int ty0 = (t % p.tiles_y) * TS;
        const int img = t / p.tiles_y;
        __syncthreads();
        for (int i = tid; i < COUT * HPIX; i += 256) {
            const int co = i / HPIX, hp = i % HPIX;
            const int gy = ty0 + hp / HS - 1, gx = tx0 + hp % HS - 1;
            G[i] = (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) ? p.g[(((long)img * COUT + co) * p.H + gy) * p.W + gx] * p.g_scale : 0.f;
        }
        __syncthreads();
        if (active) {
            // U pixels per trip: the U (independent) global loads are issued before any arithmetic, so every thread keeps
            // U x 8 bytes in flight instead of one (the kernel is a latency-bound stream otherwise)
            constexpr int U = 4;                     // pixels in flight per thread
            for (int pix0 = pl; pix0 < 256; pix0 += U * ppb) {
                float v[U][4];
                int pyv[U], pxv[U];
                bool okv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pix = pix0 + u * ppb;
                    pyv[u] = pix / TS; pxv[u] = pix % TS;
                    const int gy = ty0 + pyv[u], gx = tx0 + pxv[u];
                    okv[u] = pix < 256 && gy < p.H && gx < p.W;
                    const long q = okv[u] ? pix_index(img, gy, gx, p.H, p.W, p.blk) : pix_index(img, ty0, tx0, p.H, p.W, p.blk);
                    load4((const H*)p.P + q * p.p_cs + p.p_co + cg * 4, v[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (!okv[u]) continue;
#pragma unroll
                    for (int co = 0; co < COUT; ++co)
#pragma unroll
                        for (int tap = 0; tap < 9; ++tap) {
                            // dW[tap] += g[q - off(tap)] * P[q]
                            const float gv = G[co * HPIX + (pyv[u] + 2 - tap / 3) * HS + pxv[u] + 2 - tap % 3];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[co][tap][e] = fmaf(gv, v[u][e], acc[co][tap][e]);
                        }
                }
            }
        }
    }
    // combine the pixel lanes, one atomic per weight per workgroup (unrolled: a runtime index would push acc to scratch)
#pragma unroll
    for (int co = 0; co < COUT; ++co)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) R[tid * 4 + e] = active ? acc[co][tap][e] : 0.f;
            __syncthreads();
            if (tid < cgc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float s = 0.f;
                    for (int q = 0; q < ppb; ++q) s += R[(q * cgc + tid) * 4 + e];
                    atomicAdd(p.dw + ((long)co * p.cin + tid * 4 + e) * 9 + tap, s);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ dgrad + wgrad in one pass
// The two backward kernels above each stream the 1 GB activation tensor; this one reads every activation tile once and
// uses it three times from LDS: as the ReLU mask of dP, as the MFMA operand of dW (transposing LDS reads, as in
// conv_wgrad.hip) and -- optionally -- to keep per-channel sums of dP (the bias gradient of the preceding pixel-shuffle
// convolution, whose channels are (sub-pixel, ci) in the blocked order).  Workgroups are persistent over tiles: the
// activation pieces of the next tile are in flight while this one is multiplied; dW stays in accumulators.
template <typename H> struct HV32;
template <> struct HV32<bf16_t> {
    static __device__ __forceinline__ f32x16 mma(const u32x4& a, const u32x4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct HV32<f16_t> {
    static __device__ __forceinline__ f32x16 mma(const u32x4& a, const u32x4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ u32x4 tr_frag(const char* a0, const char* a1) {      // two ds_read_b64_tr_b16 (see conv_wgrad.hip)
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0));
    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1));
    u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
    return r;
}

template <typename H, int NT, bool BS, bool FULL>     // NT = cin/16; cout*9 <= 27; BS: keep the bias sums (NT = 2, 4 or 8); FULL: H, W multiples of 16
__global__ __launch_bounds__(256, NT <= 4 ? 2 : 1) void head_bwd_kernel(const HeadArgs p, int n_tiles, float* bsum, unsigned* stamps, double* dw_rows, double* bs_rows) {
#ifdef PSSR_WG_STAMPS
    unsigned long long st_prev = 0;
    unsigned st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned st_n = 0;
#define HB_STAMP(I)                                                                                               \
    {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        unsigned long long t_;                                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((I) >= 0) st_sum[(I) < 0 ? 0 : (I)] += (unsigned)(t_ - st_prev);                                      \
        st_prev = t_;                                                                                             \
    }
#else
#define HB_STAMP(I)
#endif
    typedef typename HV<H>::v8 v8;
    typedef __attribute__((ext_vector_type(4))) H v4;
    constexpr int C = NT * 16, CI_S = NT / 2, PPP = C / 8;          // channels, 32-channel sub-tiles, 16-byte pieces per pixel
    constexpr int OROW = C * 2 + 16;                                // dP row: 16 bytes of padding spread the 8-byte MFMA stores over the banks
    constexpr int G_BYTES = ((3 * HPIX * 4 + 15) / 16) * 16 + 16;    // + one slot that stays zero: the target of unused (co, tap) operand slots
    constexpr int GZ = 3 * HPIX;
    constexpr int GI = (3 * HPIX + 255) / 256;                      // gradient-tile elements per thread (cout <= 3)
    constexpr int UNR = NT <= 4 ? 2 : 1;                            // tile rows in flight per wave in the two MFMA phases (register budget: 2 workgroups per CU)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* G = (float*)smem;                                        // [cout][HS][HS] incoming gradient tile (+halo), scaled
    char* Pt = smem + G_BYTES;                                      // [CI_S][256 px][64 B] activation tile
    char* O = Pt + 256 * C * 2;                                     // [256 px][OROW] dP before the mask
    float* Bs = (float*)(O + 256 * OROW);                           // [16 sub-pixels][C] bias sums of this workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bshift = p.blk, bmask = (1 << p.blk) - 1;

    v8 bw[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bw[nt] = wfrag_bwd<H>(p, nt, lane);
    f32x16 accw[CI_S];
#pragma unroll
    for (int s_ = 0; s_ < CI_S; ++s_)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[s_][e] = 0.f;
    // wgrad A operand: row m = (co, tap) of this lane, offset of its g value for tile pixel (0, 0)
    const int wm_ = lane & 31, wh_ = lane >> 5;
    const int w_co = wm_ / 9, w_tap = wm_ % 9;
    const int gbase = (w_co < p.cout) ? w_co * HPIX + (2 - w_tap / 3) * HS + (2 - w_tap % 3) + 8 * wh_ : -1;
    // wgrad B operand (transposing read): lane 4q+pc of each 16-lane group addresses row q, columns 4pc..4pc+3
    const int tg = lane >> 4, tq = (lane & 15) >> 2, tpc = lane & 3;
    const int tr_off = ((tg >> 1) * 8 + tq) * 64 + ((tg & 1) * 16 + tpc * 4) * 2;
    for (int i = tid; i < 16 * C; i += 256) Bs[i] = 0.f;
    if (tid < 4) G[GZ + tid] = 0.f;
    // dgrad: k = (co, tap) slots 8 * (lane >> 4) + j of this lane, as offsets into G for tile pixel (0, 0)
    int doff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int m = 8 * (lane >> 4) + j;
        const int co = m / 9, tap = m % 9;
        doff[j] = co < p.cout ? co * HPIX + (2 - tap / 3) * HS + 2 - tap % 3 : -1;
    }
    // bias sums: piece u of a thread is pixel tid / PPP + u * PXU of the tile, i.e. (for PXU a multiple of the tile width)
    // always the same column and a row that moves by PXU / TS per u: NS distinct sub-pixel rows
    constexpr int PXU = 256 / PPP, NS = BS ? (PXU >= 4 * TS ? 1 : 4 * TS / PXU) : 1;
    static_assert(!BS || (256 % PPP == 0 && PXU % TS == 0), "bias sums need a power-of-two piece count");
    float bacc[NS][8];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) bacc[k][j] = 0.f;

    // ---- tile-independent piece descriptors.  Full tiles: the offset of a piece from the tile origin does not depend on the
    // tile (origins are multiples of 16 >= the blocked layout's 2^blk), so a tile costs one scalar origin and PPP adds.
    int rel_p[PPP], rel_d[PPP];       // element offsets of piece u from the origin pixel, activation / gradient tensor
#pragma unroll
    for (int u = 0; u < PPP; ++u) {
        const int i = tid + u * 256;
        const int pix = i / PPP, pc = i % PPP;
        const int dy = pix / TS, dx = pix % TS;
        const int rp = (((dy >> bshift) * (p.W >> bshift) + (dx >> bshift)) << (2 * bshift)) + ((dy & bmask) << bshift) + (dx & bmask);
        rel_p[u] = rp * p.p_cs + pc * 8;
        rel_d[u] = rp * p.dp_cs + pc * 8;
    }
    // gradient-tile elements of this thread: halo position -> offset from the tile's (-1, -1) corner in the NCHW plane
    int g_rel[GI], g_hy[GI], g_hx[GI];
#pragma unroll
    for (int u = 0; u < GI; ++u) {
        const int i = tid + u * 256;
        const int co = i / HPIX, hp = i % HPIX;
        g_hy[u] = hp / HS - 1; g_hx[u] = hp % HS - 1;
        g_rel[u] = i < p.cout * HPIX ? co * p.H * p.W + g_hy[u] * p.W + g_hx[u] : INT_MIN;      // (plane offset fits: cout <= 3, one image)
    }

    u32x4 pre[PPP];
    float gpre[GI];
    unsigned okm = 0, okm_cur = 0;    // partial tiles: bit u = piece u lies inside the image
#define HB_ISSUE(TILE)                                                                                                  \
    {                                                                                                                   \
        int t_ = (TILE);                                                                                                \
        const int tx0_ = (t_ % p.tiles_x) * TS; t_ /= p.tiles_x;                                                        \
        const int ty0_ = (t_ % p.tiles_y) * TS;                                                                         \
        const int img_ = t_ / p.tiles_y;                                                                                \
        const long org_ = pix_index(img_, ty0_, tx0_, p.H, p.W, p.blk);                                                 \
        const H* pb_ = (const H*)p.P + org_ * p.p_cs + p.p_co;                                                          \
        okm = 0;                                                                                                        \
        _Pragma("unroll") for (int u = 0; u < PPP; ++u) {                                                               \
            bool ok = true;                                                                                             \
            if (!FULL) {                                                                                                \
                const int pix = (tid + u * 256) / PPP;                                                                  \
                ok = ty0_ + pix / TS < p.H && tx0_ + pix % TS < p.W;                                                    \
            }                                                                                                           \
            okm |= (ok ? 1u : 0u) << u;                                                                                 \
            pre[u] = *(const u32x4*)(pb_ + (ok ? rel_p[u] : 0));                                                        \
        }                                                                                                               \
        const float* gb_ = p.g + (long)img_ * p.cout * p.H * p.W + (long)ty0_ * p.W + tx0_;                             \
        _Pragma("unroll") for (int u = 0; u < GI; ++u) {                                                                \
            const int gy = ty0_ + g_hy[u], gx = tx0_ + g_hx[u];                                                         \
            const bool ok = g_rel[u] != INT_MIN && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;                               \
            gpre[u] = ok ? gb_[g_rel[u]] : 0.f;                                                             \
        }                                                                                                               \
    }
    int tile = blockIdx.x;
    if (tile < n_tiles) HB_ISSUE(tile)
    HB_STAMP(-1)
    for (; tile < n_tiles; tile += gridDim.x) {
#ifdef PSSR_WG_STAMPS
        ++st_n;
#endif
        int t = tile;
        const int tx0 = (t % p.tiles_x) * TS; t /= p.tiles_x;
        const int ty0 = (t % p.tiles_y) * TS;
        const int img = t / p.tiles_y;
        H* const db = (H*)p.dP + pix_index(img, ty0, tx0, p.H, p.W, p.blk) * p.dp_cs + p.dp_co;
        // ---- stage: activation pieces and gradient tile (both prefetched one tile ahead)
        okm_cur = okm;
#pragma unroll
        for (int u = 0; u < PPP; ++u) {
            const int i = tid + u * 256;
            const int pix = i / PPP, pc = i % PPP;
            u32x4 v = pre[u];
            if (!FULL && !((okm >> u) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            *(u32x4*)(Pt + (pc / 4) * (256 * 64) + pix * 64 + (pc % 4) * 16) = v;
        }
#pragma unroll
        for (int u = 0; u < GI; ++u) {
            const int i = tid + u * 256;
            if (i < p.cout * HPIX) G[i] = gpre[u] * p.g_scale;
        }
        HB_STAMP(0)
        __syncthreads();
        HB_STAMP(1)
        if (tile + (int)gridDim.x < n_tiles) HB_ISSUE(tile + gridDim.x)
        HB_STAMP(2)

        // ---- dP^T = W^T (*) g: one tile row of 16 pixels per MFMA chain; a lane ends up with 4 consecutive channels of one
        // pixel, stored as one 8-byte piece (the row-major product would need 16 two-byte stores per row instead of 4)
#pragma unroll (UNR)
        for (int gi = 0; gi < 4; ++gi) {
            const int g = wave + 4 * gi;
            const int px = lane & 15, py = g;
            v8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (H)G[doff[j] >= 0 ? doff[j] + py * HS + px : GZ];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4_t acc = HV<H>::mma(bw[nt], a, f32x4_t{0.f, 0.f, 0.f, 0.f});
                v4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (H)acc[r];
                *(v4*)(O + (g * 16 + px) * OROW + (nt * 16 + (lane >> 4) * 4) * 2) = o;
            }
        }
        HB_STAMP(3)
        // ---- dW[(co,tap)][ci] += sum over the tile's pixels of g[q - off(tap)] * P[q]: k-steps of 16 pixels = tile rows
#pragma unroll (UNR)
        for (int ki = 0; ki < 4; ++ki) {
            const int ks = wave + 4 * ki;
            float gv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = G[gbase >= 0 ? gbase + ks * HS + j : GZ];
            u32x4 af;
            {
                v8 a8;
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = (H)gv[j];
                af = __builtin_bit_cast(u32x4, a8);
            }
#pragma unroll
            for (int s_ = 0; s_ < CI_S; ++s_) {
                const char* b0 = Pt + s_ * (256 * 64) + ks * 16 * 64 + tr_off;
                accw[s_] = HV32<H>::mma(af, tr_frag(b0, b0 + 4 * 64), accw[s_]);
            }
        }
        HB_STAMP(4)
        __syncthreads();
        HB_STAMP(5)
        // ---- masked write-out (16-byte pieces, whole pixel rows per wave) + bias sums per (sub-pixel, channel).  The ReLU mask
        // is a packed 16-bit test: a bf16 / fp16 value is > 0 exactly when its bit pattern is > 0 as a signed 16-bit integer
#pragma unroll
        for (int u = 0; u < PPP; ++u) {
            if (!FULL && !((okm_cur >> u) & 1u)) continue;
            const int i = tid + u * 256;
            const int pix = i / PPP, pc = i % PPP;
            typedef __attribute__((ext_vector_type(8))) short s16x8;
            typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
            const u16x8 raw = *(const u16x8*)(O + pix * OROW + pc * 16);
            const s16x8 actv = *(const s16x8*)(Pt + (pc / 4) * (256 * 64) + pix * 64 + (pc % 4) * 16);
            const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            const u16x8 one = {1, 1, 1, 1, 1, 1, 1, 1}, ones = {0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff, 0xffff};
            const u16x8 pos = __builtin_elementwise_min(__builtin_bit_cast(u16x8, __builtin_elementwise_max(actv, zero)), one);
            const u16x8 mv = raw & (pos * ones);
            const v8 v = __builtin_bit_cast(v8, mv);
            *(v8*)(db + rel_d[u]) = v;
            if (BS) {       // this thread's piece u always lies on sub-pixel row set u % NS and one sub-pixel column
#pragma unroll
                for (int j = 0; j < 8; ++j) bacc[u % NS][j] += (float)v[j];
            }
        }
        HB_STAMP(6)
        __syncthreads();
        HB_STAMP(7)
    }
#ifdef PSSR_WG_STAMPS
    if (stamps && lane == 0) {
        unsigned* q = stamps + ((long)blockIdx.x * 4 + wave) * 12;
        for (int i = 0; i < 8; ++i) q[i] = st_sum[i];
        q[8] = st_n;
    }
#endif
#undef HB_STAMP
#undef HB_ISSUE
    // ---- dW: one atomic per weight and wave -- f32 onto the gradient itself, or (dw_rows) the order-independent pieces of
    // stat_add onto a [PSSR_STAT_ROWS][cout * cin * 9] buffer that pssr_f64_to_f32 folds afterwards (bit-reproducible)
    const int kcol = lane & 31;
    const int stripe = blockIdx.x % PSSR_STAT_STRIPES;
    const long ndw = (long)p.cout * p.cin * 9;
#pragma unroll
    for (int s_ = 0; s_ < CI_S; ++s_)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (m < p.cout * 9) {
                const long idx = ((long)(m / 9) * p.cin + s_ * 32 + kcol) * 9 + m % 9;
                if (dw_rows) stat_add(dw_rows + stripe * ndw + idx, PSSR_STAT_STRIPES * ndw, accw[s_][e]);
                else atomicAdd(p.dw + idx, accw[s_][e]);
            }
        }
    if (BS && bs_rows) {
        // the bias sums of this workgroup in a fixed order (the tile buffers are free now): every thread parks its sums, an output
        // (sub-pixel, channel) then adds its contributors by increasing pixel
        constexpr int NPX = 256 / PPP;
        float* bl = (float*)Pt;                  // [256][NS * 8]
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) bl[tid * (NS * 8) + k * 8 + j] = bacc[k][j];
        __syncthreads();
        const int nsub = 1 << (2 * p.blk);
        const long nbs = (long)nsub * C;
        for (int o = tid; o < nsub * C; o += 256) {
            const int sb = o / C, ch = o % C, pcq = ch / 8, j = ch % 8;
            float t = 0.f;
            for (int px = 0; px < NPX; ++px)
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int sub = (((px / TS + k * (PXU / TS)) & bmask) << bshift) + ((px % TS) & bmask);
                    if (sub == sb) t += bl[(px * PPP + pcq) * (NS * 8) + k * 8 + j];
                }
            stat_add(bs_rows + stripe * nbs + o, PSSR_STAT_STRIPES * nbs, t);
        }
    } else if (BS) {
        const int pix0 = tid / PPP, pc = tid % PPP;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int sub = (((pix0 / TS + k * (PXU / TS)) & bmask) << bshift) + ((pix0 % TS) & bmask);
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(Bs + sub * C + pc * 8 + j, bacc[k][j]);
        }
        __syncthreads();
        const int nsub = 1 << (2 * p.blk);
        for (int i = tid; i < nsub * C; i += 256) atomicAdd(bsum + i, Bs[i]);
    }
}

int check_common(const void* P, int cs, int co, int blk, int n, int h, int w, int cin, int cout, int dtype, const char* what) {
    PSSR_CHECK(dtype == PSSR_BF16 || dtype == PSSR_F16, PSSR_ERR_UNSUPPORTED, "%s: 16-bit storage only (use pssr_conv2d for the exact-f32 build)", what);
    PSSR_CHECK(P && n > 0 && h > 0 && w > 0, PSSR_ERR_ARG, "%s: bad shape", what);
    PSSR_CHECK(cout >= 1 && cout <= 3, PSSR_ERR_ARG, "%s: cout=%d (1..3)", what, cout);
    PSSR_CHECK(cin % 32 == 0 && cin >= 32 && cin <= 128, PSSR_ERR_ARG, "%s: cin=%d (32, 64, 96 or 128)", what, cin);
    PSSR_CHECK(cs % 8 == 0 && co % 8 == 0 && co + cin <= cs, PSSR_ERR_ARG, "%s: channel stride/offset", what);
    PSSR_CHECK(blk >= 0 && blk <= 3 && h % (1 << blk) == 0 && w % (1 << blk) == 0, PSSR_ERR_ARG, "%s: blocked order", what);
    return PSSR_OK;
}

}  // namespace

extern "C" {

int pssr_head_conv_fwd(const void* in, int in_cs, int in_co, int in_blk, const float* w_oihw, const float* bias, float* out_nchw,
                       int n, int h, int w, int cin, int cout, float out_scale, float out_shift, int dtype, pssr_stream_t s) {
    int rc = check_common(in, in_cs, in_co, in_blk, n, h, w, cin, cout, dtype, "head_conv_fwd");
    if (rc != PSSR_OK) return rc;
    PSSR_CHECK(w_oihw && out_nchw, PSSR_ERR_ARG, "head_conv_fwd: null pointer");
    HeadArgs a{};
    a.P = in; a.p_cs = in_cs; a.p_co = in_co; a.blk = in_blk; a.w = w_oihw; a.bias = bias; a.out = out_nchw;
    a.N = n; a.H = h; a.W = w; a.cin = cin; a.cout = cout; a.tiles_x = cdiv(w, TS); a.tiles_y = cdiv(h, TS);
    a.out_scale = out_scale; a.out_shift = out_shift;
    const long blocks = (long)a.tiles_x * a.tiles_y * n;
    PSSR_CHECK(blocks < (1L << 31), PSSR_ERR_ARG, "head_conv_fwd: grid");
    const int nt = cdiv(cout * 9, 16), ks = cin / 32;
#define HF(NT_, KS_)                                                                                                    \
    do {                                                                                                                \
        if (dtype == PSSR_BF16) hipLaunchKernelGGL((head_fwd_kernel<bf16_t, NT_, KS_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, a); \
        else hipLaunchKernelGGL((head_fwd_kernel<f16_t, NT_, KS_>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, a); \
    } while (0)
    if (nt == 1) { if (ks == 1) HF(1, 1); else if (ks == 2) HF(1, 2); else if (ks == 3) HF(1, 3); else HF(1, 4); }
    else { if (ks == 1) HF(2, 1); else if (ks == 2) HF(2, 2); else if (ks == 3) HF(2, 3); else HF(2, 4); }
#undef HF
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_head_q_gather(const float* q, const float* bias, float* out_nchw, int n, int h, int w, float out_scale, float out_shift,
                       pssr_stream_t s) {
    PSSR_CHECK(q && out_nchw && n > 0 && h > 0 && w > 0, PSSR_ERR_ARG, "head_q_gather: bad args");
    if (w % 4 == 0 && ((uintptr_t)q | (uintptr_t)out_nchw) % 16 == 0) {
        const long blocks4 = ((long)n * h * w + 255) / 256;
        hipLaunchKernelGGL(head_q_gather4_kernel, dim3((unsigned)(blocks4 < 16384 ? blocks4 : 16384)), dim3(256), 0, (hipStream_t)s, q, bias, out_nchw,
                           n, h, w, out_scale, out_shift);
        PSSR_LAUNCH_CHECK();
        return PSSR_OK;
    }
    const long total = (long)n * h * w * 4;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(head_q_gather_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)s, q, bias, out_nchw, n, h, w,
                       out_scale, out_shift);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_head_conv_dgrad(const float* g_nchw, float g_scale, const float* w_oihw, const void* act, int act_cs, int act_co, void* dact,
                         int d_cs, int d_co, int blk, int n, int h, int w, int cin, int cout, int dtype, pssr_stream_t s) {
    int rc = check_common(act, act_cs, act_co, blk, n, h, w, cin, cout, dtype, "head_conv_dgrad");
    if (rc != PSSR_OK) return rc;
    PSSR_CHECK(g_nchw && w_oihw && dact && d_cs % 8 == 0 && d_co % 8 == 0 && d_co + cin <= d_cs, PSSR_ERR_ARG, "head_conv_dgrad: bad args");
    HeadArgs a{};
    a.P = act; a.p_cs = act_cs; a.p_co = act_co; a.blk = blk; a.dP = dact; a.dp_cs = d_cs; a.dp_co = d_co;
    a.w = w_oihw; a.g = g_nchw; a.g_scale = g_scale;
    a.N = n; a.H = h; a.W = w; a.cin = cin; a.cout = cout; a.tiles_x = cdiv(w, TS); a.tiles_y = cdiv(h, TS);
    const long blocks = (long)a.tiles_x * a.tiles_y * n;
    PSSR_CHECK(blocks < (1L << 31), PSSR_ERR_ARG, "head_conv_dgrad: grid");
    const int lds = ((3 * HPIX * 4 + 15) / 16) * 16 + 256 * cin * 2;
#define HD(NT_)                                                                                                         \
    do {                                                                                                                \
        if (dtype == PSSR_BF16) {                                                                                       \
            (void)hipFuncSetAttribute((const void*)head_dgrad_kernel<bf16_t, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((head_dgrad_kernel<bf16_t, NT_>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)s, a); \
        } else {                                                                                                        \
            (void)hipFuncSetAttribute((const void*)head_dgrad_kernel<f16_t, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((head_dgrad_kernel<f16_t, NT_>), dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)s, a); \
        }                                                                                                               \
    } while (0)
    switch (cin / 16) { case 2: HD(2); break; case 4: HD(4); break; case 6: HD(6); break; default: HD(8); break; }
#undef HD
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_head_conv_wgrad(const float* g_nchw, float g_scale, const void* act, int act_cs, int act_co, int blk, float* dw_oihw,
                         int n, int h, int w, int cin, int cout, int dtype, pssr_stream_t s) {
    int rc = check_common(act, act_cs, act_co, blk, n, h, w, cin, cout, dtype, "head_conv_wgrad");
    if (rc != PSSR_OK) return rc;
    PSSR_CHECK(g_nchw && dw_oihw, PSSR_ERR_ARG, "head_conv_wgrad: null pointer");
    HeadArgs a{};
    a.P = act; a.p_cs = act_cs; a.p_co = act_co; a.blk = blk; a.g = g_nchw; a.g_scale = g_scale; a.dw = dw_oihw;
    a.N = n; a.H = h; a.W = w; a.cin = cin; a.cout = cout; a.tiles_x = cdiv(w, TS); a.tiles_y = cdiv(h, TS);
    const long tiles = (long)a.tiles_x * a.tiles_y * n;
    PSSR_CHECK(tiles < (1L << 31), PSSR_ERR_ARG, "head_conv_wgrad: grid");
    const int grid = tiles < 1024 ? (int)tiles : 1024;
#define HW(H_, CO_) hipLaunchKernelGGL((head_wgrad_kernel<H_, CO_>), dim3(grid), dim3(256), 0, (hipStream_t)s, a, (int)tiles)
    if (dtype == PSSR_BF16) { if (cout == 1) HW(bf16_t, 1); else if (cout == 2) HW(bf16_t, 2); else HW(bf16_t, 3); }
    else { if (cout == 1) HW(f16_t, 1); else if (cout == 2) HW(f16_t, 2); else HW(f16_t, 3); }
#undef HW
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

#ifdef PSSR_WG_STAMPS
void pssr_debug_head_stamp_buffer(void* p) { g_head_stamps = (unsigned*)p; }
#endif

static int head_conv_bwd_impl(const float* g_nchw, float g_scale, const float* w_oihw, const void* act, int act_cs, int act_co, void* dact,
                              int d_cs, int d_co, int blk, float* dw_oihw, float* bias_sum, double* dw_rows, double* bias_rows, int n, int h,
                              int w, int cin, int cout, int dtype, pssr_stream_t s) {
    int rc = check_common(act, act_cs, act_co, blk, n, h, w, cin, cout, dtype, "head_conv_bwd");
    if (rc != PSSR_OK) return rc;
    PSSR_CHECK(g_nchw && w_oihw && dact && (dw_oihw || dw_rows) && d_cs % 8 == 0 && d_co % 8 == 0 && d_co + cin <= d_cs, PSSR_ERR_ARG, "head_conv_bwd: bad args");
    PSSR_CHECK(blk <= 2, PSSR_ERR_ARG, "head_conv_bwd: blocked order up to 4x4 sub-pixels");
    HeadArgs a{};
    a.P = act; a.p_cs = act_cs; a.p_co = act_co; a.blk = blk; a.dP = dact; a.dp_cs = d_cs; a.dp_co = d_co;
    a.w = w_oihw; a.g = g_nchw; a.g_scale = g_scale; a.dw = dw_oihw;
    a.N = n; a.H = h; a.W = w; a.cin = cin; a.cout = cout; a.tiles_x = cdiv(w, TS); a.tiles_y = cdiv(h, TS);
    const long tiles = (long)a.tiles_x * a.tiles_y * n;
    PSSR_CHECK(tiles < (1L << 31), PSSR_ERR_ARG, "head_conv_bwd: grid");
    const int lds = ((3 * HPIX * 4 + 15) / 16) * 16 + 16 + 256 * cin * 2 + 256 * (cin * 2 + 16) + 16 * cin * 4;
    const bool full = h % TS == 0 && w % TS == 0;
    const int grid = tiles < 512 ? (int)tiles : 512;
#define HB2(NT_, BS_, FULL_)                                                                                                \
    do {                                                                                                                \
        if (dtype == PSSR_BF16) {                                                                                       \
            (void)hipFuncSetAttribute((const void*)head_bwd_kernel<bf16_t, NT_, BS_, FULL_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((head_bwd_kernel<bf16_t, NT_, BS_, FULL_>), dim3(grid), dim3(256), lds, (hipStream_t)s, a, (int)tiles, bias_sum, g_head_stamps, dw_rows, bias_rows); \
        } else {                                                                                                        \
            (void)hipFuncSetAttribute((const void*)head_bwd_kernel<f16_t, NT_, BS_, FULL_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
            hipLaunchKernelGGL((head_bwd_kernel<f16_t, NT_, BS_, FULL_>), dim3(grid), dim3(256), lds, (hipStream_t)s, a, (int)tiles, bias_sum, g_head_stamps, dw_rows, bias_rows); \
        }                                                                                                               \
    } while (0)
#define HB(NT_, BS_) do { if (full) HB2(NT_, BS_, true); else HB2(NT_, BS_, false); } while (0)
    PSSR_CHECK(!(bias_sum || bias_rows) || cin == 32 || cin == 64 || cin == 128, PSSR_ERR_ARG, "head_conv_bwd: bias sums need cin = 32, 64 or 128");
    if (bias_sum || bias_rows) { switch (cin / 16) { case 2: HB(2, true); break; case 4: HB(4, true); break; default: HB(8, true); break; } }
    else { switch (cin / 16) { case 2: HB(2, false); break; case 4: HB(4, false); break; case 6: HB(6, false); break; default: HB(8, false); break; } }
#undef HB
#undef HB2
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_head_conv_bwd(const float* g_nchw, float g_scale, const float* w_oihw, const void* act, int act_cs, int act_co, void* dact,
                       int d_cs, int d_co, int blk, float* dw_oihw, float* bias_sum, int n, int h, int w, int cin, int cout, int dtype,
                       pssr_stream_t s) {
    return head_conv_bwd_impl(g_nchw, g_scale, w_oihw, act, act_cs, act_co, dact, d_cs, d_co, blk, dw_oihw, bias_sum, nullptr, nullptr, n, h, w,
                              cin, cout, dtype, s);
}

int pssr_head_conv_bwd_rows(const float* g_nchw, float g_scale, const float* w_oihw, const void* act, int act_cs, int act_co, void* dact,
                            int d_cs, int d_co, int blk, double* dw_rows, double* bias_rows, int n, int h, int w, int cin, int cout,
                            int dtype, pssr_stream_t s) {
    PSSR_CHECK(dw_rows != nullptr, PSSR_ERR_ARG, "head_conv_bwd_rows: null dw_rows");
    return head_conv_bwd_impl(g_nchw, g_scale, w_oihw, act, act_cs, act_co, dact, d_cs, d_co, blk, nullptr, nullptr, dw_rows, bias_rows, n, h, w,
                              cin, cout, dtype, s);
}

}  // extern "C"
