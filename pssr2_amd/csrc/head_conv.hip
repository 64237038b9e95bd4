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

}  // extern "C"
