// HBM-bound kernels of the RDNet encoder (pssr/models/_rdnet.py): patchify stem input, depthwise 7x7
// convolution (forward / input gradient / weight gradient), LayerNorm2d over the channels of every
// pixel (forward / backward, optionally writing the 2x2 space-to-depth layout that turns the
// following stride-2 transition conv into a 1x1 conv), the Effective-SE gate and the layer-scale.
// All tensors are NHWC slices (pointer, channel stride, channel offset): the dense-concatenation of
// RDNet (torch.cat of all previous features, _rdnet.py:132-138) is elided by writing every new
// feature at its channel offset of one buffer per stage.
#include "common.h"
#include "tunables.h"
#include <type_traits>

namespace {

constexpr int TPB = 256;

struct Ref { const void* p; int cs, co; };
struct MRef { void* p; int cs, co; };
template <typename T> __device__ __forceinline__ const T* at(const Ref& r, long pix, int c) { return (const T*)r.p + pix * r.cs + r.co + c; }
template <typename T> __device__ __forceinline__ T* at(const MRef& r, long pix, int c) { return (T*)r.p + pix * r.cs + r.co + c; }

static inline int grid1d(long total, int cap = 8192) { long b = (total + TPB - 1) / TPB; return (int)(b < cap ? (b > 0 ? b : 1) : cap); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------
// xpatch[n, oy, ox, ci*ps*ps + dy*ps + dx] = bn(x[n, ci, oy*ps+dy, ox*ps+dx]*pre_scale + pre_shift)
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ x, T* __restrict__ xp, int n, int c, int h, int w, int ps, int pc,
                                float prs, float prb, const float* scale, const float* shift) {
    const int ho = h / ps, wo = w / ps, kk = ps * ps;
    const long total = (long)n * ho * wo * pc;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i % pc;
        long pix = i / pc;
        const int ox = pix % wo; pix /= wo;
        const int oy = pix % ho;
        const int img = pix / ho;
        float v = 0.f;
        if (k < c * kk) {
            const int ch = k / kk, tap = k % kk;
            const int sy = oy * ps + tap / ps, sx = ox * ps + tap % ps;
            v = fmaf(fmaf(x[(((long)img * c + ch) * h + sy) * w + sx], prs, prb), scale[ch], shift[ch]);
        }
        xp[i] = (T)v;
    }
}

// packed depthwise weight: wp[tap][c] = w[c][flip ? 48 - tap : tap]
__global__ void dw_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int c, int flip) {
    const int total = 49 * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = i % c, tap = i / c;
        wp[i] = w[ch * 49 + (flip ? 48 - tap : tap)];
    }
}

// every depthwise weight of a model (both orientations) by one launch, blockIdx.y = item: RDNet packed 42 of them per training step,
// each a launch of its own at the head of a dependent chain
__global__ void dw_pack_batch_kernel(pssr_dwpack_batch items) {
    const int it = blockIdx.y;
    const float* __restrict__ w = items.w[it];
    float* __restrict__ wp = items.packed[it];
    const int c = items.c[it], flip = items.flip[it];
    const int total = 49 * c;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ch = i % c, tap = i / c;
        wp[i] = w[ch * 49 + (flip ? 48 - tap : tap)];
    }
}

// depthwise 7x7, stride 1, zero padding 3: out[p, c] (+)= bias[c] + sum_tap in[p + tap - 3, c] * wp[tap][c]
template <typename T>
__global__ void dwconv7_kernel(Ref in, const float* __restrict__ wp, const float* __restrict__ bias, MRef out, int n, int h, int w, int c,
                               int accumulate) {
    const int cg = c / 4;
    const long total = (long)n * h * w * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (i % cg) * 4;
        long pix = i / cg;
        const int px = pix % w;
        const int py = (pix / w) % h;
        const long img_base = (pix / ((long)w * h)) * h * w;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) load4(bias + c0, acc);
        for (int ky = 0; ky < 7; ++ky) {
            const int sy = py + ky - 3;
            if (sy < 0 || sy >= h) continue;
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int sx = px + kx - 3;
                if (sx < 0 || sx >= w) continue;
                float v[4];
                load4(at<T>(in, img_base + (long)sy * w + sx, c0), v);
                const float4 wv = *(const float4*)(wp + (ky * 7 + kx) * c + c0);
                acc[0] = fmaf(v[0], wv.x, acc[0]); acc[1] = fmaf(v[1], wv.y, acc[1]);
                acc[2] = fmaf(v[2], wv.z, acc[2]); acc[3] = fmaf(v[3], wv.w, acc[3]);
            }
        }
        if (accumulate) {
            float o[4];
            load4(at<T>(out, pix, c0), o);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += o[e];
        }
        store4(at<T>(out, pix, c0), acc);
    }
}

// dW[c][ky*7+kx] += sum_p dy[p, c] * x[p + (ky-3, kx-3), c]; blockIdx.y = ky, a thread keeps one channel group
template <typename T>
__global__ void dwconv7_wgrad_kernel(Ref dy, Ref x, float* __restrict__ dw, int n, int h, int w, int c) {
    __shared__ float lds[TPB * 28];
    const int cgc = c / 4;
    const int ky = blockIdx.y;
    const long npix = (long)n * h * w;
    const int ppb = cgc <= TPB ? TPB / cgc : 1;
    for (int cg0 = 0; cg0 < cgc; cg0 += TPB) {       // one pass when c <= 1024
        const int cg = cg0 + (cgc <= TPB ? (int)threadIdx.x % cgc : (int)threadIdx.x);
        const int pl = cgc <= TPB ? (int)threadIdx.x / cgc : 0;
        const bool active = cg < cgc && pl < ppb;
        float acc[7][4];
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
        if (active) {
            for (long pix = (long)blockIdx.x * ppb + pl; pix < npix; pix += (long)gridDim.x * ppb) {
                const int px = pix % w;
                const int py = (pix / w) % h;
                const int sy = py + ky - 3;
                if (sy < 0 || sy >= h) continue;
                float g[4];
                load4(at<T>(dy, pix, cg * 4), g);
                const long row = pix - px + (long)(ky - 3) * w;
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const int sx = px + kx - 3;
                    if (sx < 0 || sx >= w) continue;
                    float v[4];
                    load4(at<T>(x, row + sx, cg * 4), v);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[kx][e] = fmaf(g[e], v[e], acc[kx][e]);
                }
            }
        }
        // combine the pixel lanes of the block, then one atomic per (channel, tap) per block
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) lds[threadIdx.x * 28 + k * 4 + e] = active ? acc[k][e] : 0.f;
        __syncthreads();
        if (cgc <= TPB) {
            if ((int)threadIdx.x < cgc) {
                for (int k = 0; k < 7; ++k)
                    for (int e = 0; e < 4; ++e) {
                        float t = 0.f;
                        for (int q = 0; q < ppb; ++q) t += lds[(q * cgc + threadIdx.x) * 28 + k * 4 + e];
                        atomicAdd(dw + (long)(threadIdx.x * 4 + e) * 49 + ky * 7 + k, t);
                    }
            }
        } else if (active) {
            for (int k = 0; k < 7; ++k)
                for (int e = 0; e < 4; ++e) atomicAdd(dw + (long)(cg * 4 + e) * 49 + ky * 7 + k, acc[k][e]);
        }
        __syncthreads();
    }
}

// Row-segment variants (W % 8 == 0): a thread produces 8 consecutive x positions of 4 channels from a sliding window of
// 14 inputs per kernel row held in registers: 98 + 49 loads per 8 outputs instead of 98 per output.
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_seg_kernel(Ref in, const float* __restrict__ wp, const float* __restrict__ bias, MRef out,
                                                          int n, int h, int w, int c, int accumulate) {
    const int cg = c / 4, ws = w / 8;
    const long total = (long)n * h * ws * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (i % cg) * 4;
        long t = i / cg;
        const int xs = (t % ws) * 8; t /= ws;
        const int y = t % h;
        const long img_base = (t / h) * h * w;
        float acc[8][4];
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) load4(bias + c0, b4);
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] = b4[e];
        for (int ky = 0; ky < 7; ++ky) {
            const int sy = y + ky - 3;
            if (sy < 0 || sy >= h) continue;
            float v[14][4];
            const T* xrow = at<T>(in, img_base + (long)sy * w + xs, c0);      // one base per row, constant strides, edge padding only
            const bool left = xs == 0, right = xs + 8 == w;
#pragma unroll
            for (int j = 0; j < 14; ++j) {
                const bool pad = (j < 3 && left) || (j >= 11 && right);
                if (!pad) load4(xrow + (long)(j - 3) * in.cs, v[j]);
                else { v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f; }
            }
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {      // packed f32 FMAs (v_pk_fma_f32): the kernel is bound by the vector ALU
                const float4 wv = *(const float4*)(wp + (ky * 7 + kx) * c + c0);
                const f32x2 w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const f32x2 a01 = __builtin_elementwise_fma(f32x2{v[o + kx][0], v[o + kx][1]}, w01, f32x2{acc[o][0], acc[o][1]});
                    const f32x2 a23 = __builtin_elementwise_fma(f32x2{v[o + kx][2], v[o + kx][3]}, w23, f32x2{acc[o][2], acc[o][3]});
                    acc[o][0] = a01[0]; acc[o][1] = a01[1]; acc[o][2] = a23[0]; acc[o][3] = a23[1];
                }
            }
        }
        const long prow = img_base + (long)y * w + xs;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            if (accumulate) {
                float prev[4];
                load4(at<T>(out, prow + o, c0), prev);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[o][e] += prev[e];
            }
            store4(at<T>(out, prow + o, c0), acc[o]);
        }
    }
}

// LDS-tiled version of the row-segment kernel: a workgroup owns TY rows x 16 columns x 64 channels and stages their
// (TY+6) x 22 input halo once (1.9x / 2.4x the tile instead of the 12x each thread of dwconv7_seg_kernel asks of L1 / L2:
// consecutive workgroups of that kernel land on different XCDs, so most of those re-reads cross the fabric -- it ran at the
// fabric's ~6 TB/s, not at the vector ALU's rate).  A thread is (channel group of 4, 8-pixel column segment, row).
template <typename T> struct DwTile {
    static constexpr int TX = 16, CT = 64, HX = TX + 6;
    // LDS pixel stride = the 64 channels, row pitch 23 pixels (odd): the two rows that share a 32-lane LDS pass sit 128 B apart
    // modulo the 256 B bank width, so the 8-byte reads of 16 channel groups x 2 rows are conflict-free without padding bytes
    static constexpr int PS = CT * (int)sizeof(T), HXP = HX + 1;
    static constexpr int W_BYTES = 49 * CT * 4;
    static constexpr int lds_bytes(int ty) { return (ty + 6) * HXP * PS + W_BYTES; }
};

template <typename T, int TY>
__global__ __launch_bounds__(32 * TY) void dwconv7_tile_kernel(Ref in, const float* __restrict__ wp, const float* __restrict__ bias, MRef out,
                                                               int n, int h, int w, int c, int accumulate, int tiles_x, int tiles_y, int ctiles) {
    using D = DwTile<T>;
    using V = typename std::conditional<sizeof(T) == 2, uint2, uint4>::type;       // 4 channels of T
    constexpr int NT = 32 * TY, HY = TY + 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* halo = smem;                                                   // [HY][23] pixels of PS bytes (22 used)
    float* wl = (float*)(smem + HY * D::HXP * D::PS);                    // [49][64]
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    {   // consecutive tiles on ONE XCD (workgroups go round-robin to the 8 XCDs): neighbouring tiles share their halo in its L2
        const int per = gridDim.x >> 3;
        if (b < per * 8) b = (b & 7) * per + (b >> 3);
    }
    const int ct = b % ctiles; b /= ctiles;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int img = b / tiles_y;
    const int x0 = tx * D::TX, y0 = ty * TY, cb = ct * D::CT;
    const long img_base = (long)img * h * w;

    // all of a thread's halo loads are issued before the first LDS write (a rolled loop waits for each load in turn)
    constexpr int ITEMS = (HY * D::HX * 16 + NT - 1) / NT;
    V stage[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int it = tid + k * NT;
        const int cgi = it & 15, px = it >> 4;
        const int col = px % D::HX, row = px / D::HX;
        const int gy = y0 + row - 3, gx = x0 + col - 3;
        V v = {};
        if (it < HY * D::HX * 16 && gy >= 0 && gy < h && gx >= 0 && gx < w && cb + cgi * 4 < c)
            v = *(const V*)at<T>(in, img_base + (long)gy * w + gx, cb + cgi * 4);
        stage[k] = v;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int it = tid + k * NT;
        const int cgi = it & 15, px = it >> 4;
        const int col = px % D::HX, row = px / D::HX;
        if (it < HY * D::HX * 16) *(V*)(halo + (row * D::HXP + col) * D::PS + cgi * 4 * (int)sizeof(T)) = stage[k];
    }
    for (int it = tid; it < 49 * 16; it += NT) {
        const int cgi = it & 15, tap = it >> 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cb + cgi * 4 < c) v = *(const float4*)(wp + (long)tap * c + cb + cgi * 4);
        *(float4*)(wl + tap * D::CT + cgi * 4) = v;
    }
    __syncthreads();

    const int cgi = tid & 15, xseg = (tid >> 5) & 1, row = (tid >> 6) * 2 + ((tid >> 4) & 1);
    const int c0 = cb + cgi * 4, y = y0 + row, xs = x0 + xseg * 8;
    if (c0 >= c || y >= h || xs >= w) return;
    float acc[8][4];
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) load4(bias + c0, b4);
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[o][e] = b4[e];
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        float v[14][4];
        const char* hrow = halo + ((row + ky) * D::HXP + xseg * 8) * D::PS + cgi * 4 * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < 14; ++j) load4((const T*)(hrow + j * D::PS), v[j]);
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float4 wv = *(const float4*)(wl + (ky * 7 + kx) * D::CT + cgi * 4);
            const f32x2 w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const f32x2 a01 = __builtin_elementwise_fma(f32x2{v[o + kx][0], v[o + kx][1]}, w01, f32x2{acc[o][0], acc[o][1]});
                const f32x2 a23 = __builtin_elementwise_fma(f32x2{v[o + kx][2], v[o + kx][3]}, w23, f32x2{acc[o][2], acc[o][3]});
                acc[o][0] = a01[0]; acc[o][1] = a01[1]; acc[o][2] = a23[0]; acc[o][3] = a23[1];
            }
        }
    }
    const long prow = img_base + (long)y * w + xs;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        if (accumulate) {
            float prev[4];
            load4(at<T>(out, prow + o, c0), prev);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[o][e] += prev[e];
        }
        store4(at<T>(out, prow + o, c0), acc[o]);
    }
}

template <typename T, int TY>
static void launch_dw_tile(const void* in, int in_cs, int in_co, const float* wp, const float* bias, void* out, int out_cs, int out_co, int n, int h,
                           int w, int c, int accumulate, hipStream_t s) {
    using D = DwTile<T>;
    static bool attr_set = false;
    constexpr int LDS = D::lds_bytes(TY);
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)dwconv7_tile_kernel<T, TY>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr_set = true; }
    const int tiles_x = (w + D::TX - 1) / D::TX, tiles_y = (h + TY - 1) / TY, ctiles = (c + D::CT - 1) / D::CT;
    hipLaunchKernelGGL((dwconv7_tile_kernel<T, TY>), dim3((unsigned)((long)n * tiles_y * tiles_x * ctiles)), dim3(32 * TY), LDS, s, Ref{in, in_cs, in_co}, wp, bias,
                       MRef{out, out_cs, out_co}, n, h, w, c, accumulate, tiles_x, tiles_y, ctiles);
}

// blockIdx.y = ky; a thread keeps one channel group and walks row segments of 8 pixels
template <typename T>
__global__ __launch_bounds__(256) void dwconv7_wgrad_seg_kernel(Ref dy, Ref x, float* __restrict__ dw, int n, int h, int w, int c) {
    __shared__ float lds[TPB * 28];
    const int cgc = c / 4, ws = w / 8;
    const int ky = blockIdx.y;
    const long nseg = (long)n * h * ws;
    const int ppb = cgc <= TPB ? TPB / cgc : 1;
    for (int cg0 = 0; cg0 < cgc; cg0 += TPB) {
        const int cg = cg0 + (cgc <= TPB ? (int)threadIdx.x % cgc : (int)threadIdx.x);
        const int pl = cgc <= TPB ? (int)threadIdx.x / cgc : 0;
        const bool active = cg < cgc && pl < ppb;
        float acc[7][4];
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
        if (active) {
            for (long sgi = (long)blockIdx.x * ppb + pl; sgi < nseg; sgi += (long)gridDim.x * ppb) {
                long t = sgi;
                const int xs = (t % ws) * 8; t /= ws;
                const int y = t % h;
                const long img_base = (t / h) * h * w;
                const int sy = y + ky - 3;
                if (sy < 0 || sy >= h) continue;
                float g[8][4], v[14][4];
#pragma unroll
                for (int o = 0; o < 8; ++o) load4(at<T>(dy, img_base + (long)y * w + xs + o, cg * 4), g[o]);
#pragma unroll
                for (int j = 0; j < 14; ++j) {
                    const int sx = xs - 3 + j;
                    if (sx >= 0 && sx < w) load4(at<T>(x, img_base + (long)sy * w + sx, cg * 4), v[j]);
                    else { v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f; }
                }
#pragma unroll
                for (int kx = 0; kx < 7; ++kx)
#pragma unroll
                    for (int o = 0; o < 8; ++o)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[kx][e] = fmaf(g[o][e], v[o + kx][e], acc[kx][e]);
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) lds[threadIdx.x * 28 + k * 4 + e] = active ? acc[k][e] : 0.f;
        __syncthreads();
        if (cgc <= TPB) {
            if ((int)threadIdx.x < cgc) {
                for (int k = 0; k < 7; ++k)
                    for (int e = 0; e < 4; ++e) {
                        float t = 0.f;
                        for (int q = 0; q < ppb; ++q) t += lds[(q * cgc + threadIdx.x) * 28 + k * 4 + e];
                        atomicAdd(dw + (long)(threadIdx.x * 4 + e) * 49 + ky * 7 + k, t);
                    }
            }
        } else if (active) {
            for (int k = 0; k < 7; ++k)
                for (int e = 0; e < 4; ++e) atomicAdd(dw + (long)(cg * 4 + e) * 49 + ky * 7 + k, acc[k][e]);
        }
        __syncthreads();
    }
}

// Same sums with the 7 kernel rows spread over the THREADS of a workgroup instead of blockIdx.y: a thread is (kernel row ky,
// channel group) and the workgroup walks consecutive output rows of one 8-pixel column, so the dy segment is fetched once
// per workgroup (the 7 ky threads hit the same lines) and an input row serves the 7 kernel rows on consecutive iterations
// from L1/L2 -- the blockIdx.y version above streams both tensors 7 times from HBM.
template <typename T>
__global__ __launch_bounds__(224) void dwconv7_wgrad_rows_kernel(Ref dy, Ref x, float* __restrict__ dw, int n, int h, int w, int c, long per_block,
                                                                 float* __restrict__ part) {
    const int cgc = c / 4, ws = w / 8;
    const int ky = threadIdx.x / 32, cg = blockIdx.y * 32 + threadIdx.x % 32;
    if (cg >= cgc) return;
    const long nseg = (long)n * ws * h;                        // segment index = (img * ws + xseg) * h + y: y fastest
    int bx = blockIdx.x;                                        // consecutive segment runs on ONE XCD: they share input rows in its L2
    {
        const int per = gridDim.x >> 3;
        if (bx < per * 8) bx = (bx & 7) * per + (bx >> 3);
    }
    const long s0 = (long)bx * per_block, s1 = s0 + per_block < nseg ? s0 + per_block : nseg;
    float acc[7][4];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
    for (int sgi = (int)s0; sgi < (int)s1; ++sgi) {             // nseg < 2^31 (checked by the launcher): 32-bit divisions
        const int y = sgi % h;
        const int t = sgi / h;
        const int xs = (t % ws) * 8;
        const long img_base = (long)(t / ws) * h * w;
        const int sy = y + ky - 3;
        if (sy < 0 || sy >= h) continue;
        float g[8][4], v[14][4];
        // one base address per row segment, constant strides from it; only the first / last segment of a row needs the
        // left / right zero padding (the per-load 64-bit index arithmetic and range tests were most of this kernel's VALU work)
        const T* grow = at<T>(dy, img_base + (long)y * w + xs, cg * 4);
        const T* xrow = at<T>(x, img_base + (long)sy * w + xs, cg * 4);
#pragma unroll
        for (int o = 0; o < 8; ++o) load4(grow + (long)o * dy.cs, g[o]);
        const bool left = xs == 0, right = xs + 8 == w;
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const bool pad = (j < 3 && left) || (j >= 11 && right);
            if (!pad) load4(xrow + (long)(j - 3) * x.cs, v[j]);
            else { v[j][0] = v[j][1] = v[j][2] = v[j][3] = 0.f; }
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)       // packed f32 FMAs (v_pk_fma_f32): the kernel is bound by the vector ALU
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const f32x2 a01 = __builtin_elementwise_fma(f32x2{g[o][0], g[o][1]}, f32x2{v[o + kx][0], v[o + kx][1]}, f32x2{acc[kx][0], acc[kx][1]});
                const f32x2 a23 = __builtin_elementwise_fma(f32x2{g[o][2], g[o][3]}, f32x2{v[o + kx][2], v[o + kx][3]}, f32x2{acc[kx][2], acc[kx][3]});
                acc[kx][0] = a01[0]; acc[kx][1] = a01[1]; acc[kx][2] = a23[0]; acc[kx][3] = a23[1];
            }
    }
    if (part) {      // one slab [gridDim.y * 128 channels][49] per workgroup, summed in a fixed order by dwconv7_wgrad_reduce_kernel
        float* slab = part + (long)bx * gridDim.y * 128 * 49;
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) slab[(long)(cg * 4 + e) * 49 + ky * 7 + k] = acc[k][e];
        return;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dw + (long)(cg * 4 + e) * 49 + ky * 7 + k, acc[k][e]);
}

// dw[i] += sum over the workgroup slabs in a fixed order (the depthwise weight gradient is reproducible bit for bit):
// a workgroup owns 64 consecutive elements; 16 slab lanes each sum every 16th slab, then lane 0 adds the 16 sums in order.
__global__ __launch_bounds__(1024) void dwconv7_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int elems, long slab, int slabs) {
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, lane = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + col;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < elems) {
        int b = lane;
        for (; b + 48 < slabs; b += 64) {
            s0 += part[(long)b * slab + i]; s1 += part[(long)(b + 16) * slab + i];
            s2 += part[(long)(b + 32) * slab + i]; s3 += part[(long)(b + 48) * slab + i];
        }
        for (; b < slabs; b += 16) s0 += part[(long)b * slab + i];
    }
    red[lane][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (lane == 0 && i < elems) {
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < 16; ++l) s += red[l][col];
        dw[i] += s;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm2d: one wave per pixel, lane l holds channel groups l, l+64, ...  (C <= 256*LN_MAXIT)
constexpr int LN_MAXIT = 8;

__device__ __forceinline__ void ln_out_pos(long pix, int h, int w, int s2d, int cpad, long& opix, int& cbase) {
    if (!s2d) { opix = pix; cbase = 0; return; }
    const int px = pix % w;
    const int py = (pix / w) % h;
    const long img = pix / ((long)w * h);
    opix = (img * (h / 2) + py / 2) * (w / 2) + px / 2;
    cbase = ((py & 1) * 2 + (px & 1)) * cpad;
}

// NIT = channel slices of 256 a lane walks, PP = pixels a wave keeps in flight: a pixel is a load -> two wave reductions -> store
// chain of ~3 us, so with one pixel per wave the kernel ran at the rate of that chain (1.2 TB/s on 64 x 64 x 128 channels)
template <typename T, int NIT, int PP>
__global__ __launch_bounds__(TPB) void ln_fwd_kernel(Ref in, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, MRef out,
                                                     int s2d, int cpad, long npix, int h, int w, int c, float* __restrict__ mean,
                                                     float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6), nw = (long)gridDim.x * (TPB / 64);
    const float inv_c = 1.f / (float)c;
    for (long pix0 = wid * PP; pix0 < npix; pix0 += nw * PP) {
        float v[PP][NIT][4];
        float s[PP], mu[PP], q[PP], rs[PP];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            s[u] = 0.f;
            const long pix = pix0 + u < npix ? pix0 + u : npix - 1;          // the tail repeats the last pixel (stores are guarded)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c0 = (lane + 64 * it) * 4;
                if (c0 < c) {
                    load4(at<T>(in, pix, c0), v[u][it]);
                    s[u] += (v[u][it][0] + v[u][it][1]) + (v[u][it][2] + v[u][it][3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            mu[u] = wave_sum(s[u]) * inv_c;
            q[u] = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c0 = (lane + 64 * it) * 4;
                if (c0 < c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[u][it][e] - mu[u]; q[u] = fmaf(d, d, q[u]); }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PP; ++u) rs[u] = rsqrtf(wave_sum(q[u]) * inv_c + eps);
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const long pix = pix0 + u;
            if (pix >= npix) break;
            long opix; int cbase;
            ln_out_pos(pix, h, w, s2d, cpad, opix, cbase);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c0 = (lane + 64 * it) * 4;
                if (c0 < c) {
                    float g[4], b[4], o[4];
                    load4(gamma + c0, g); load4(beta + c0, b);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaf((v[u][it][e] - mu[u]) * rs[u], g[e], b[e]);
                    store4(at<T>(out, opix, cbase + c0), o);
                } else if (c0 < cpad) {
                    const float z[4] = {0.f, 0.f, 0.f, 0.f};
                    store4(at<T>(out, opix, cbase + c0), z);
                }
            }
            if (lane == 0 && mean) { mean[pix] = mu[u]; rstd[pix] = rs[u]; }
        }
    }
}

// dx (+)= rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat));  stats += [sum_p g*xhat | sum_p g]
// NIT = channel slices of 256 a lane walks (registers scale with it), NT = threads: few slices leave room for 16 waves per
// workgroup, which share one flush of the statistics (the f64 atomics bound how many workgroups a launch can afford)
// PP = pixels a wave has in flight per trip (their loads issued together, their reductions independent).  Measured
// (tools/diag/microbench_ln.py): PP = 2 / 4 change nothing -- what a launch pays besides its bytes is the statistics flush at the end,
// 2c f64 atomic pairs per workgroup (10-20 us for 256-512 workgroups: PSSR_LN_DBG=1 leaves it out) -- so only PP = 1 is built
template <typename T, int NIT, int NT, int PP>
__global__ __launch_bounds__(NT) void ln_bwd_kernel(Ref g, int s2d, int cpad, Ref x, const float* __restrict__ gamma, const float* __restrict__ mean,
                              const float* __restrict__ rstd, MRef dx, int accumulate, long npix, int h, int w, int c, double* stats, int dbg) {
    __shared__ float lds[(NT / 64) * 64 * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * (NT / 64) + wv, nw = (long)gridDim.x * (NT / 64);
    const float inv_c = 1.f / (float)c;
    if (dbg & 2) npix = 0;          // (diagnostic: no pixels)
    float dgam[NIT][4], dbet[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) dgam[it][e] = dbet[it][e] = 0.f;
    float gm[NIT][4];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int c0 = (lane + 64 * it) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) gm[it][e] = 0.f;
        if (c0 < c) load4(gamma + c0, gm[it]);
    }
    for (long pix0 = wid; pix0 < npix; pix0 += nw * PP) {
        float xh[PP][NIT][4], gg[PP][NIT][4], gv[PP][NIT][4];
        float mu[PP], rs[PP], s1[PP], s2[PP];
        bool ok[PP];
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const long pix = pix0 + u * nw;
            ok[u] = pix < npix;
            mu[u] = ok[u] ? mean[pix] : 0.f;
            rs[u] = ok[u] ? rstd[pix] : 0.f;
            long opix = 0; int cbase = 0;
            if (ok[u]) ln_out_pos(pix, h, w, s2d, cpad, opix, cbase);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c0 = (lane + 64 * it) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { xh[u][it][e] = 0.f; gv[u][it][e] = 0.f; }
                if (ok[u] && c0 < c) { load4(at<T>(x, pix, c0), xh[u][it]); load4(at<T>(g, opix, cbase + c0), gv[u][it]); }
            }
        }
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c0 = (lane + 64 * it) * 4;
                if (c0 < c) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xh[u][it][e] = (xh[u][it][e] - mu[u]) * rs[u];
                        gg[u][it][e] = gv[u][it][e] * gm[it][e];
                        s1[u] += gg[u][it][e];
                        s2[u] = fmaf(gg[u][it][e], xh[u][it][e], s2[u]);
                        if (ok[u]) {
                            dgam[it][e] = fmaf(gv[u][it][e], xh[u][it][e], dgam[it][e]);
                            dbet[it][e] += gv[u][it][e];
                        }
                    }
                }
            }
        }
        float m1[PP], m2[PP];
#pragma unroll
        for (int u = 0; u < PP; ++u) { m1[u] = wave_sum(s1[u]) * inv_c; m2[u] = wave_sum(s2[u]) * inv_c; }
#pragma unroll
        for (int u = 0; u < PP; ++u) {
            const long pix = pix0 + u * nw;
            if (!ok[u]) continue;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c0 = (lane + 64 * it) * 4;
                if (c0 < c) {
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = rs[u] * (gg[u][it][e] - m1[u] - xh[u][it][e] * m2[u]);
                    if (accumulate) {
                        float prev[4];
                        load4(at<T>(dx, pix, c0), prev);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] += prev[e];
                    }
                    store4(at<T>(dx, pix, c0), o);
                }
            }
        }
    }
    // block combine (4 waves) per iteration slice, then one f64 atomic per channel per block into a stripe
    if (dbg & 1) return;            // (diagnostic: no statistics)
    double* st = stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * c;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (64 * it * 4 >= c) break;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) { lds[(wv * 64 + lane) * 8 + e] = dgam[it][e]; lds[(wv * 64 + lane) * 8 + 4 + e] = dbet[it][e]; }
        __syncthreads();
        if (wv == 0) {
            const int c0 = (lane + 64 * it) * 4;
            if (c0 < c) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int q = 0; q < NT / 64; ++q) { a += lds[(q * 64 + lane) * 8 + e]; b += lds[(q * 64 + lane) * 8 + 4 + e]; }
                    stat_add(st + c0 + e, (long)PSSR_STAT_STRIPES * 2 * c, a);
                    stat_add(st + c + c0 + e, (long)PSSR_STAT_STRIPES * 2 * c, b);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// out[img][c] += scale * sum_{p in img} a[p, c] * (b ? b[p, c] : 1);  grid = (32-channel slabs, images), 1024 threads
// A workgroup owns ALL pixels of one image for 32 channels (8 four-channel lanes x 128 pixel lanes, four pixels per trip in flight per
// thread) and sums its 128 pixel lanes through LDS in a fixed tree: no partial sums in memory, no atomics, the same bits on every run.
// (The version before split an image's pixels over up to 64 workgroups that met through a workspace, `__threadfence()` and a ticket:
// on this part an agent-scope release is an L2 write-back per workgroup, and the launch took ~2.5 us per workgroup of an image --
// 101 us for the 33 MB of a 64^2 x 64-channel layer.)
constexpr int ICD_T = 1024;
template <typename T>
__global__ __launch_bounds__(ICD_T) void image_channel_dot_kernel(Ref a, Ref b, int hw, int c, float scale, float* __restrict__ out) {
    __shared__ float red[ICD_T * 4];
    const int tid = threadIdx.x, q = tid & 7, pl = tid >> 3, img = blockIdx.y;
    const int c0 = blockIdx.x * 32 + q * 4;
    const bool active = c0 < c;
    const long base = (long)img * hw;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        constexpr int PL = ICD_T / 8;
        int p = pl;
        for (; p + 3 * PL < hw; p += 4 * PL) {
            float av[4][4], bv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) load4(at<T>(a, base + p + u * PL, c0), av[u]);
            if (b.p) {
#pragma unroll
                for (int u = 0; u < 4; ++u) load4(at<T>(b, base + p + u * PL, c0), bv[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(av[u][e], bv[u][e], acc[e]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += av[u][e];
            }
        }
        for (; p < hw; p += PL) {
            float av[4];
            load4(at<T>(a, base + p, c0), av);
            if (b.p) {
                float bv[4];
                load4(at<T>(b, base + p, c0), bv);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(av[e], bv[e], acc[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += av[e];
            }
        }
    }
    *(float4*)(red + tid * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
#pragma unroll
    for (int s = ICD_T / 16; s >= 1; s >>= 1) {         // pixel lanes pl and pl + s: thread tid + 8 s
        if (pl < s) {
            const float4 x = *(const float4*)(red + tid * 4), y = *(const float4*)(red + (tid + 8 * s) * 4);
            *(float4*)(red + tid * 4) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
        }
        __syncthreads();
    }
    if (pl == 0 && active) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c0 + e < c) out[(long)img * c + c0 + e] += red[q * 4 + e] * scale;
    }
}

// Effective-SE gate: u[n][co] = b[co] + sum_ci W[co][ci] * s[n][ci];  gate = relu6(u + 3) / 6
__global__ __launch_bounds__(256) void ese_gate_kernel(const float* __restrict__ s, const float* __restrict__ w, const float* __restrict__ b, int n, int c,
                                                       float* __restrict__ u, float* __restrict__ gate) {
    // one wave per output: the lanes run along ci, so the row of W is read coalesced (a thread per output walks W with stride c)
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n * c) return;
    const int img = i / c, co = i % c;
    float acc = 0.f;
    for (int ci = lane; ci < c; ci += 64) acc = fmaf(w[(long)co * c + ci], s[(long)img * c + ci], acc);
    for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) {
        acc += b[co];
        u[i] = acc;
        gate[i] = fminf(fmaxf(acc + 3.f, 0.f), 6.f) * (1.f / 6.f);
    }
}

// out[p, c] = t[p, c] * (gate ? gate[img][c] : 1) * gamma[c] + (add ? add[img][c] : 0)
template <typename T>
__global__ void scale_nc_kernel(Ref t, const float* __restrict__ gate, const float* __restrict__ gamma, const float* __restrict__ add, MRef out,
                                long npix, int hw, int c) {
    const int cg = c / 4;
    const long total = npix * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (i % cg) * 4;
        const long pix = i / cg;
        const long img = pix / hw;
        float v[4], gm[4], gt[4] = {1.f, 1.f, 1.f, 1.f}, ad[4] = {0.f, 0.f, 0.f, 0.f};
        load4(at<T>(t, pix, c0), v); load4(gamma + c0, gm);
        if (gate) load4(gate + img * c + c0, gt);
        if (add) load4(add + img * c + c0, ad);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], gt[e] * gm[e], ad[e]);
        store4(at<T>(out, pix, c0), v);
    }
}

// ESE / layer-scale backward on the [N, C] side tensors.  A[n][c] = sum_p dout*t.
//   stage 0: du[n][c] = (|u| < 3) ? A*gamma/6 : 0
//   stage 1: dgamma[c] = sum_n A*gate ; dbfc[c] = sum_n du
//   stage 2: dWfc[co][ci] = sum_n du[n][co]*s[n][ci]
//   stage 3: add[n][ci] = inv_hw * sum_co W[co][ci]*du[n][co]
__global__ void ese_bwd_kernel(int stage, const float* __restrict__ A, const float* __restrict__ gate, const float* __restrict__ u,
                               const float* __restrict__ gamma, const float* __restrict__ s, const float* __restrict__ w, int n, int c, float inv_hw,
                               float* __restrict__ du, float* __restrict__ dgamma, float* __restrict__ dbfc, float* __restrict__ dwfc,
                               float* __restrict__ add) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (stage == 0) {
        if (i >= (long)n * c) return;
        const float uv = u[i];
        du[i] = (uv > -3.f && uv < 3.f) ? A[i] * gamma[i % c] * (1.f / 6.f) : 0.f;
    } else if (stage == 1) {
        if (i >= c) return;
        float a = 0.f, b = 0.f;
        for (int k = 0; k < n; ++k) {
            a = fmaf(A[(long)k * c + i], gate ? gate[(long)k * c + i] : 1.f, a);
            if (du) b += du[(long)k * c + i];
        }
        dgamma[i] = a;
        if (dbfc) dbfc[i] = b;
    } else if (stage == 2) {
        if (i >= (long)c * c) return;
        const int co = i / c, ci = i % c;
        float a = 0.f;
        for (int k = 0; k < n; ++k) a = fmaf(du[(long)k * c + co], s[(long)k * c + ci], a);
        dwfc[i] = a;
    } else {
        if (i >= (long)n * c) return;
        const int img = i / c, ci = i % c;
        float a = 0.f;
        for (int co = 0; co < c; ++co) a = fmaf(w[(long)co * c + ci], du[(long)img * c + co], a);
        add[i] = a * inv_hw;
    }
}

// The gate path of ese_bwd in one launch (four dependent launches of a few workgroups each cost ~20 us apiece in a step):
// du is cheap to recompute, so the three reductions do not have to wait for it.  Workgroup roles by blockIdx.x:
//   [0, n*cdiv(c,64))       (image, 64 input channels): du row (to LDS and to `du`), then add[img][ci] = inv_hw * sum_co W[co][ci] * du[img][co]
//   next cdiv(c*c,256)      dWfc[co][ci] = sum_k du[k][co] * s[k][ci]
//   the rest                dgamma[c] = sum_k A*gate ; dbfc[c] = sum_k du
constexpr int ESE_MAX_C = 2048;
__global__ __launch_bounds__(256) void ese_bwd_fused_kernel(const float* __restrict__ A, const float* __restrict__ gate, const float* __restrict__ u,
                                                            const float* __restrict__ gamma, const float* __restrict__ s, const float* __restrict__ w,
                                                            int n, int c, float inv_hw, float* __restrict__ du, float* __restrict__ dgamma,
                                                            float* __restrict__ dbfc, float* __restrict__ dwfc, float* __restrict__ add) {
    __shared__ float dul[ESE_MAX_C];
    const int tid = threadIdx.x;
    auto du_at = [&](int k, int co) {
        const float uv = u[(long)k * c + co];
        return (uv > -3.f && uv < 3.f) ? A[(long)k * c + co] * gamma[co] * (1.f / 6.f) : 0.f;
    };
    const int nw = (c * c + 255) / 256;
    int b = blockIdx.x;
    const int nchunk = (c + 63) / 64, nimg = n * nchunk;
    if (b < nimg) {      // (image, chunk of 64 input channels): 4 thread groups share the sum over co
        __shared__ float red[4][64];
        const int img = b / nchunk, chunk = b % nchunk;
        for (int co = tid; co < c; co += 256) {
            const float d = du_at(img, co);
            dul[co] = d;
            if (chunk == 0) du[(long)img * c + co] = d;
        }
        __syncthreads();
        const int ci = chunk * 64 + (tid & 63), g = tid >> 6;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (ci < c) {
            int co = g;
            for (; co + 12 < c; co += 16) {
                a0 = fmaf(w[(long)co * c + ci], dul[co], a0); a1 = fmaf(w[(long)(co + 4) * c + ci], dul[co + 4], a1);
                a2 = fmaf(w[(long)(co + 8) * c + ci], dul[co + 8], a2); a3 = fmaf(w[(long)(co + 12) * c + ci], dul[co + 12], a3);
            }
            for (; co < c; co += 4) a0 = fmaf(w[(long)co * c + ci], dul[co], a0);
        }
        red[g][tid & 63] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (g == 0 && ci < c) add[(long)img * c + ci] = ((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid])) * inv_hw;
        return;
    }
    b -= nimg;
    if (b < nw) {        // 256 consecutive (co, ci): the few co rows of du they need are recomputed once into LDS
        const int i0 = b * 256, i = i0 + tid;
        const int co0 = i0 / c, co1 = min((i0 + 255) / c, c - 1), nco = co1 - co0 + 1;
        const bool cached = n * nco <= ESE_MAX_C;
        if (cached) {
            for (int q = tid; q < n * nco; q += 256) dul[q] = du_at(q / nco, co0 + q % nco);
            __syncthreads();
        }
        if (i >= c * c) return;
        const int co = i / c, ci = i % c;
        float a0 = 0.f, a1 = 0.f;
        if (cached) {
            int k = 0;
            for (; k + 2 <= n; k += 2) {
                a0 = fmaf(dul[k * nco + co - co0], s[(long)k * c + ci], a0);
                a1 = fmaf(dul[(k + 1) * nco + co - co0], s[(long)(k + 1) * c + ci], a1);
            }
            if (k < n) a0 = fmaf(dul[k * nco + co - co0], s[(long)k * c + ci], a0);
        } else {
            for (int k = 0; k < n; ++k) a0 = fmaf(du_at(k, co), s[(long)k * c + ci], a0);
        }
        dwfc[i] = a0 + a1;
        return;
    }
    b -= nw;             // 64 channels x 4 groups of images
    __shared__ float red2[2][4][64];
    const int ch = b * 64 + (tid & 63), g = tid >> 6;
    float a = 0.f, d = 0.f;
    if (ch < c)
        for (int k = g; k < n; k += 4) {
            a = fmaf(A[(long)k * c + ch], gate[(long)k * c + ch], a);
            d += du_at(k, ch);
        }
    red2[0][g][tid & 63] = a; red2[1][g][tid & 63] = d;
    __syncthreads();
    if (g == 0 && ch < c) {
        dgamma[ch] = (red2[0][0][tid] + red2[0][1][tid]) + (red2[0][2][tid] + red2[0][3][tid]);
        dbfc[ch] = (red2[1][0][tid] + red2[1][1][tid]) + (red2[1][2][tid] + red2[1][3][tid]);
    }
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                             \
    do {                                                                    \
        if ((dtype) == PSSR_BF16) { using T = bf16_t; CALL; }               \
        else if ((dtype) == PSSR_F16) { using T = f16_t; CALL; }            \
        else if ((dtype) == PSSR_F32) { using T = float; CALL; }            \
        else { pssr_set_error("bad dtype %d", (dtype)); return PSSR_ERR_ARG; } \
    } while (0)
#define CHECK_REF(name, cs, co, c)                                                                               \
    PSSR_CHECK((cs) % 4 == 0 && (co) % 4 == 0 && (co) + (c) <= (cs), PSSR_ERR_ARG, name ": bad channel stride/offset (%d,%d,%d)", cs, co, c)

extern "C" {

int pssr_input_patchify(const float* x, void* xpatch, int n, int c, int h, int w, int ps, int pc, float pre_scale, float pre_shift,
                        const float* scale, const float* shift, int dtype, pssr_stream_t s) {
    PSSR_CHECK(x && xpatch && scale && shift && n > 0 && c > 0 && ps > 0 && h % ps == 0 && w % ps == 0 && pc >= c * ps * ps && pc % 16 == 0,
               PSSR_ERR_ARG, "input_patchify: bad args");
    const long total = (long)n * (h / ps) * (w / ps) * pc;
    DISPATCH_T(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3(grid1d(total)), dim3(TPB), 0, (hipStream_t)s, x, (T*)xpatch, n, c, h, w, ps, pc,
                                         pre_scale, pre_shift, scale, shift));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_dwconv7_pack(const float* w, float* packed, int c, int flip, pssr_stream_t s) {
    PSSR_CHECK(w && packed && c > 0, PSSR_ERR_ARG, "dwconv7_pack: bad args");
    hipLaunchKernelGGL(dw_pack_kernel, dim3(grid1d(49L * c, 256)), dim3(TPB), 0, (hipStream_t)s, w, packed, c, flip);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_dwconv7_pack_batch(const pssr_dwpack_batch* items, int n_items, pssr_stream_t s) {
    PSSR_CHECK(items && n_items > 0 && n_items <= PSSR_DWPACK_BATCH_MAX, PSSR_ERR_ARG, "dwconv7_pack_batch: 1..%d items", PSSR_DWPACK_BATCH_MAX);
    int cmax = 0;
    for (int i = 0; i < n_items; ++i) {
        PSSR_CHECK(items->w[i] && items->packed[i] && items->c[i] > 0, PSSR_ERR_ARG, "dwconv7_pack_batch: item %d", i);
        if (items->c[i] > cmax) cmax = items->c[i];
    }
    hipLaunchKernelGGL(dw_pack_batch_kernel, dim3(grid1d(49L * cmax, 16), n_items), dim3(TPB), 0, (hipStream_t)s, *items);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_dwconv7(const void* in, int in_cs, int in_co, const float* w_packed, const float* bias, void* out, int out_cs, int out_co,
                 int n, int h, int w, int c, int accumulate, int dtype, pssr_stream_t s) {
    PSSR_CHECK(in && w_packed && out && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "dwconv7: bad args");
    CHECK_REF("dwconv7 in", in_cs, in_co, c); CHECK_REF("dwconv7 out", out_cs, out_co, c);
    const long total = (long)n * h * w * (c / 4);
    const int tile_mode = pssr_tunables().dwconv_tile;
    if (w % 8 == 0 && tile_mode && (long)n * ((h + 7) / 8) * ((w + 15) / 16) * ((c + 63) / 64) < (1L << 31)) {
        if (h > 8) { DISPATCH_T(dtype, (launch_dw_tile<T, 16>(in, in_cs, in_co, w_packed, bias, out, out_cs, out_co, n, h, w, c, accumulate, (hipStream_t)s))); }
        else { DISPATCH_T(dtype, (launch_dw_tile<T, 8>(in, in_cs, in_co, w_packed, bias, out, out_cs, out_co, n, h, w, c, accumulate, (hipStream_t)s))); }
    } else if (w % 8 == 0) {
        DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv7_seg_kernel<T>, dim3(grid1d(total / 8, 1 << 20)), dim3(TPB), 0, (hipStream_t)s, Ref{in, in_cs, in_co},
                                             w_packed, bias, MRef{out, out_cs, out_co}, n, h, w, c, accumulate));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv7_kernel<T>, dim3(grid1d(total, 1 << 20)), dim3(TPB), 0, (hipStream_t)s, Ref{in, in_cs, in_co}, w_packed,
                                             bias, MRef{out, out_cs, out_co}, n, h, w, c, accumulate));
    }
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

// Workgroups along the row segments for the slab version: >= 4 segments each, ~1024 workgroups in all (16 waves per CU)
static inline long dw_slab_blocks(int n, int h, int w, int c, long* per_block_out) {
    const int gy = (c / 4 + 31) / 32;
    const long nseg = (long)n * h * (w / 8);
    const int target = pssr_tunables().dwwg_blocks;
    long gs = target / gy;
    if (gs > nseg / 4) gs = nseg / 4;
    if (gs < 1) gs = 1;
    const long per_block = (nseg + gs - 1) / gs;
    if (per_block_out) *per_block_out = per_block;
    return (nseg + per_block - 1) / per_block;
}

int64_t pssr_dwconv7_wgrad_workspace_bytes(int n, int h, int w, int c) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 4 || w % 8) return 0;
    return dw_slab_blocks(n, h, w, c, nullptr) * ((c / 4 + 31) / 32) * 128 * 49 * (int64_t)sizeof(float);
}

int pssr_dwconv7_wgrad_ws(const void* dy, int dy_cs, int dy_co, const void* x, int x_cs, int x_co, float* dw, int n, int h, int w, int c,
                          int dtype, void* workspace, int64_t workspace_bytes, pssr_stream_t s) {
    PSSR_CHECK(dy && x && dw && workspace && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && w % 8 == 0 && (long)n * h * w < (1L << 31), PSSR_ERR_ARG,
               "dwconv7_wgrad_ws: bad args (w must be a multiple of 8)");
    CHECK_REF("dwconv7_wgrad dy", dy_cs, dy_co, c); CHECK_REF("dwconv7_wgrad x", x_cs, x_co, c);
    PSSR_CHECK(workspace_bytes >= pssr_dwconv7_wgrad_workspace_bytes(n, h, w, c), PSSR_ERR_ARG, "dwconv7_wgrad_ws: workspace of %ld bytes, %ld needed",
               (long)workspace_bytes, (long)pssr_dwconv7_wgrad_workspace_bytes(n, h, w, c));
    long per_block;
    const long blocks = dw_slab_blocks(n, h, w, c, &per_block);
    const int gy = (c / 4 + 31) / 32;
    DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv7_wgrad_rows_kernel<T>, dim3((unsigned)blocks, gy), dim3(224), 0, (hipStream_t)s, Ref{dy, dy_cs, dy_co},
                                         Ref{x, x_cs, x_co}, dw, n, h, w, c, per_block, (float*)workspace));
    PSSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(dwconv7_wgrad_reduce_kernel, dim3((c * 49 + 63) / 64), dim3(1024), 0, (hipStream_t)s, (const float*)workspace, dw, c * 49,
                       (long)gy * 128 * 49, (int)blocks);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_dwconv7_wgrad(const void* dy, int dy_cs, int dy_co, const void* x, int x_cs, int x_co, float* dw, int n, int h, int w, int c,
                       int dtype, pssr_stream_t s) {
    PSSR_CHECK(dy && x && dw && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && (long)n * h * w < (1L << 31), PSSR_ERR_ARG, "dwconv7_wgrad: bad args");
    CHECK_REF("dwconv7_wgrad dy", dy_cs, dy_co, c); CHECK_REF("dwconv7_wgrad x", x_cs, x_co, c);
    const int cgc = c / 4, ppb = cgc <= TPB ? TPB / cgc : 1;
    long gx = ((long)n * h * w + ppb - 1) / ppb / 16;       // >= 16 pixels per thread before the atomics
    if (gx < 1) gx = 1;
    if (gx > 96) gx = 96;
    if (w % 8 == 0) {
        // ~512 workgroups of (7 kernel rows x 32 channel groups) threads, each a contiguous run of row segments
        const int gy = (cgc + 31) / 32;
        const long nseg = (long)n * h * (w / 8);
        long gs = 512 / gy;
        if (gs < 1) gs = 1;
        if (gs > nseg) gs = nseg;
        const long per_block = (nseg + gs - 1) / gs;
        DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv7_wgrad_rows_kernel<T>, dim3((unsigned)((nseg + per_block - 1) / per_block), gy), dim3(224), 0, (hipStream_t)s,
                                             Ref{dy, dy_cs, dy_co}, Ref{x, x_cs, x_co}, dw, n, h, w, c, per_block, (float*)nullptr));
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL(dwconv7_wgrad_kernel<T>, dim3((unsigned)gx, 7), dim3(TPB), 0, (hipStream_t)s, Ref{dy, dy_cs, dy_co},
                                             Ref{x, x_cs, x_co}, dw, n, h, w, c));
    }
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_layernorm2d_fwd(const void* in, int in_cs, int in_co, const float* gamma, const float* beta, float eps, void* out, int out_cs, int out_co,
                         int s2d, int c_pad, int n, int h, int w, int c, float* mean, float* rstd, int dtype, pssr_stream_t s) {
    PSSR_CHECK(in && gamma && beta && out && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && c <= 256 * LN_MAXIT, PSSR_ERR_ARG,
               "layernorm2d_fwd: bad args (c=%d)", c);
    PSSR_CHECK(c_pad >= c && c_pad % 4 == 0 && (!s2d || (h % 2 == 0 && w % 2 == 0)), PSSR_ERR_ARG, "layernorm2d_fwd: c_pad / s2d");
    PSSR_CHECK((mean == nullptr) == (rstd == nullptr), PSSR_ERR_ARG, "layernorm2d_fwd: mean/rstd come in pairs");
    CHECK_REF("layernorm2d_fwd in", in_cs, in_co, c); CHECK_REF("layernorm2d_fwd out", out_cs, out_co, (s2d ? 4 : 1) * c_pad);
    const long npix = (long)n * h * w;
#define PSSR_LN_FWD(NIT_, PP_)                                                                                                            \
    do {                                                                                                                                    \
        long blocks = (npix + (TPB / 64) * (PP_) - 1) / ((TPB / 64) * (PP_));                                                               \
        if (blocks > 4096) blocks = 4096;                                                                                                   \
        DISPATCH_T(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T, NIT_, PP_>), dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)s, Ref{in, in_cs, in_co}, \
                                             gamma, beta, eps, MRef{out, out_cs, out_co}, s2d, c_pad, npix, h, w, c, mean, rstd));           \
    } while (0)
    // the zero fill of [c, c_pad) is walked with the same slices: size them for c_pad
    if (c_pad <= 256) PSSR_LN_FWD(1, 4);
    else if (c_pad <= 512) PSSR_LN_FWD(2, 2);
    else if (c_pad <= 1024) PSSR_LN_FWD(4, 1);
    else PSSR_LN_FWD(8, 1);
#undef PSSR_LN_FWD
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_layernorm2d_bwd(const void* g, int g_cs, int g_co, int s2d, int c_pad, const void* x, int x_cs, int x_co, const float* gamma,
                         const float* mean, const float* rstd, void* dx, int dx_cs, int dx_co, int accumulate, int n, int h, int w, int c,
                         double* stats, int dtype, pssr_stream_t s) {
    PSSR_CHECK(g && x && gamma && mean && rstd && dx && stats && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && c <= 256 * LN_MAXIT,
               PSSR_ERR_ARG, "layernorm2d_bwd: bad args (c=%d)", c);
    PSSR_CHECK(c_pad >= c && c_pad % 4 == 0 && (!s2d || (h % 2 == 0 && w % 2 == 0)), PSSR_ERR_ARG, "layernorm2d_bwd: c_pad / s2d");
    CHECK_REF("layernorm2d_bwd g", g_cs, g_co, (s2d ? 4 : 1) * c_pad); CHECK_REF("layernorm2d_bwd x", x_cs, x_co, c); CHECK_REF("layernorm2d_bwd dx", dx_cs, dx_co, c);
    const long npix = (long)n * h * w;
    // >= 4 pixels per wave before the statistics are flushed, <= 512 workgroups (2c f64 atomics each)
#define PSSR_LN_BWD(NIT_, NT_, PP_)                                                                                                         \
    do {                                                                                                                                    \
        long blocks = (npix + (NT_) / 64 - 1) / ((NT_) / 64) / 4;                                                                           \
        if (blocks < 1) blocks = 1;                                                                                                         \
        if (blocks > pssr_tunables().ln_bwd_blocks) blocks = pssr_tunables().ln_bwd_blocks;                                                 \
        DISPATCH_T(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T, NIT_, NT_, PP_>), dim3((unsigned)blocks), dim3(NT_), 0, (hipStream_t)s, Ref{g, g_cs, g_co}, s2d, \
                                             c_pad, Ref{x, x_cs, x_co}, gamma, mean, rstd, MRef{dx, dx_cs, dx_co}, accumulate, npix, h, w, c, stats, pssr_tunables().ln_dbg)); \
    } while (0)
    if (c <= 256) PSSR_LN_BWD(1, 1024, 1);
    else if (c <= 512) PSSR_LN_BWD(2, 1024, 1);
    else if (c <= 1024) PSSR_LN_BWD(4, 512, 1);
    else PSSR_LN_BWD(8, 256, 1);
#undef PSSR_LN_BWD
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int64_t pssr_image_channel_dot_workspace_bytes(int n, int hw, int c) {
    if (n <= 0 || hw <= 0 || c <= 0) return PSSR_ERR_ARG;
    return 0;           // kept for callers of the ticket version: the slab kernel needs no workspace
}

int pssr_image_channel_dot_ws(const void* a, int a_cs, int a_co, const void* b, int b_cs, int b_co, int n, int hw, int c, float scale, float* out,
                              int dtype, void* workspace, int64_t workspace_bytes, pssr_stream_t s) {
    (void)workspace; (void)workspace_bytes;
    PSSR_CHECK(a && out && n > 0 && hw > 0 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "image_channel_dot: bad args");
    CHECK_REF("image_channel_dot a", a_cs, a_co, c);
    if (b) CHECK_REF("image_channel_dot b", b_cs, b_co, c);
    DISPATCH_T(dtype, hipLaunchKernelGGL(image_channel_dot_kernel<T>, dim3(cdiv(c, 32), n), dim3(ICD_T), 0, (hipStream_t)s, Ref{a, a_cs, a_co},
                                         Ref{b, b_cs, b_co}, hw, c, scale, out));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_image_channel_dot(const void* a, int a_cs, int a_co, const void* b, int b_cs, int b_co, int n, int hw, int c, float scale, float* out,
                           int dtype, pssr_stream_t s) {
    return pssr_image_channel_dot_ws(a, a_cs, a_co, b, b_cs, b_co, n, hw, c, scale, out, dtype, nullptr, 0, s);
}

int pssr_ese_gate(const float* s_mean, const float* w_fc, const float* b_fc, int n, int c, float* u, float* gate, pssr_stream_t s) {
    PSSR_CHECK(s_mean && w_fc && b_fc && u && gate && n > 0 && c > 0, PSSR_ERR_ARG, "ese_gate: bad args");
    hipLaunchKernelGGL(ese_gate_kernel, dim3(cdiv(n * c, 4)), dim3(256), 0, (hipStream_t)s, s_mean, w_fc, b_fc, n, c, u, gate);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_scale_nc(const void* t, int t_cs, int t_co, const float* gate, const float* gamma, const float* add, void* out, int out_cs, int out_co,
                  int n, int hw, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(t && gamma && out && n > 0 && hw > 0 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "scale_nc: bad args");
    CHECK_REF("scale_nc t", t_cs, t_co, c); CHECK_REF("scale_nc out", out_cs, out_co, c);
    const long npix = (long)n * hw;
    DISPATCH_T(dtype, hipLaunchKernelGGL(scale_nc_kernel<T>, dim3(grid1d(npix * (c / 4))), dim3(TPB), 0, (hipStream_t)s, Ref{t, t_cs, t_co}, gate, gamma, add,
                                         MRef{out, out_cs, out_co}, npix, hw, c));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_ese_bwd(const float* A, const float* gate, const float* u, const float* gamma, const float* s_mean, const float* w_fc, int n, int c, int hw,
                 float* du, float* dgamma, float* db_fc, float* dw_fc, float* add, pssr_stream_t s) {
    PSSR_CHECK(A && gamma && dgamma && n > 0 && c > 0 && hw > 0, PSSR_ERR_ARG, "ese_bwd: bad args");
    hipStream_t st = (hipStream_t)s;
    const float inv_hw = 1.f / (float)hw;
    if (gate) {
        PSSR_CHECK(u && s_mean && w_fc && du && db_fc && dw_fc && add, PSSR_ERR_ARG, "ese_bwd: gate path needs u, s, W, du, db, dW, add");
        if (c <= ESE_MAX_C) {
            hipLaunchKernelGGL(ese_bwd_fused_kernel, dim3(n * cdiv(c, 64) + cdiv(c * c, 256) + cdiv(c, 64)), dim3(256), 0, st, A, gate, u, gamma, s_mean, w_fc, n, c, inv_hw,
                               du, dgamma, db_fc, dw_fc, add);
        } else {
            hipLaunchKernelGGL(ese_bwd_kernel, dim3(cdiv(n * c, TPB)), dim3(TPB), 0, st, 0, A, gate, u, gamma, s_mean, w_fc, n, c, inv_hw, du, dgamma, db_fc, dw_fc, add);
            hipLaunchKernelGGL(ese_bwd_kernel, dim3(cdiv(c, TPB)), dim3(TPB), 0, st, 1, A, gate, u, gamma, s_mean, w_fc, n, c, inv_hw, du, dgamma, db_fc, dw_fc, add);
            hipLaunchKernelGGL(ese_bwd_kernel, dim3(cdiv(c * c, TPB)), dim3(TPB), 0, st, 2, A, gate, u, gamma, s_mean, w_fc, n, c, inv_hw, du, dgamma, db_fc, dw_fc, add);
            hipLaunchKernelGGL(ese_bwd_kernel, dim3(cdiv(n * c, TPB)), dim3(TPB), 0, st, 3, A, gate, u, gamma, s_mean, w_fc, n, c, inv_hw, du, dgamma, db_fc, dw_fc, add);
        }
    } else {
        hipLaunchKernelGGL(ese_bwd_kernel, dim3(cdiv(c, TPB)), dim3(TPB), 0, st, 1, A, (const float*)nullptr, u, gamma, s_mean, w_fc, n, c, inv_hw,
                           (float*)nullptr, dgamma, (float*)nullptr, dw_fc, add);
    }
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

}  // extern "C"
