// HBM-bound pointwise / per-channel kernels around the convolutions: input normalisation + im2col,
// BatchNorm finalisation and backward, max-pool, pixel (un)shuffle, ReLU backward, layout changes.
// All NHWC kernels move 4 channels (8 or 16 bytes) per thread with consecutive lanes on consecutive
// channels, so every wave touches whole 128-byte lines; per-channel reductions are accumulated in
// registers (a thread keeps one channel group for its whole life), combined across the workgroup in
// LDS and added to the f64 result with one atomic per channel per workgroup.
#include "common.h"

namespace {

constexpr int TPB = 256;

struct Ref { const void* p; int cs, co; };   // NHWC tensor slice: base, channel stride, channel offset
struct MRef { void* p; int cs, co; };

template <typename T> __device__ __forceinline__ const T* at(const Ref& r, long pix, int c) { return (const T*)r.p + pix * r.cs + r.co + c; }
template <typename T> __device__ __forceinline__ T* at(const MRef& r, long pix, int c) { return (T*)r.p + pix * r.cs + r.co + c; }

// Thread -> (pixel lane, channel group) assignment shared by every per-channel kernel.
struct ChanMap {
    int cg_count, threads, ppb;   // channel groups, active threads per block, pixels per block-iteration
    __device__ __forceinline__ bool active() const { return (int)threadIdx.x < threads; }
};
static ChanMap make_map(int c) {
    ChanMap m;
    m.cg_count = c / 4;
    if (m.cg_count <= TPB) { m.ppb = TPB / m.cg_count; m.threads = m.ppb * m.cg_count; }
    else { m.ppb = 1; m.threads = TPB; }
    return m;
}
static int grid_for(long npix, const ChanMap& m) {
    long b = (npix + m.ppb - 1) / m.ppb;
    return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

// Block-level combine of NS per-thread sums of 4 channels each, then f64 atomics: dst[s*C + c]
template <int NS>
__device__ __forceinline__ void flush_sums(const ChanMap& m, int cg, float (*acc)[4], double* dst, int C, float* lds) {
    // lds: [TPB][NS*4]; dst is striped: [PSSR_STAT_STRIPES][NS*C]
    dst += (long)(blockIdx.x % PSSR_STAT_STRIPES) * NS * C;
    const long lo_off = (long)PSSR_STAT_STRIPES * NS * C;
    if (m.cg_count <= TPB) {
        const int tid = threadIdx.x;
        if (m.active())
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) lds[tid * NS * 4 + s * 4 + e] = acc[s][e];
        __syncthreads();
        if (tid < m.cg_count) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = 0.f;
                    for (int pl = 0; pl < m.ppb; ++pl) t += lds[(pl * m.cg_count + tid) * NS * 4 + s * 4 + e];
                    stat_add(dst + (long)s * C + tid * 4 + e, lo_off, t);
                }
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) stat_add(dst + (long)s * C + cg * 4 + e, lo_off, acc[s][e]);
    }
}

// Iterates this thread's (pixel, channel-group) items: F(pix, c0) ; calls G(cg) after each group is done.
template <class F, class G>
__device__ __forceinline__ void for_items(const ChanMap& m, long npix, F f, G done) {
    if (m.cg_count <= TPB) {
        const int cg = threadIdx.x % m.cg_count, pl = threadIdx.x / m.cg_count;
        if (m.active())
            for (long pix = (long)blockIdx.x * m.ppb + pl; pix < npix; pix += (long)gridDim.x * m.ppb) f(pix, cg * 4);
        done(cg);
    } else {
        for (int cg = threadIdx.x; cg < m.cg_count; cg += TPB) {
            for (long pix = blockIdx.x; pix < npix; pix += gridDim.x) f(pix, cg * 4);
            done(cg);
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void nchw_stats_kernel(const float* __restrict__ x, int n, int c, long hw, float ps, float pb, double* stats) {
    // one (image, channel) plane chunk per block column; grid = (chunks, n*c)
    const int plane = blockIdx.y, ch = plane % c;
    const float* src = x + (long)plane * hw;
    float s1 = 0.f, s2 = 0.f;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < hw; i += (long)gridDim.x * TPB) {
        const float v = fmaf(src[i], ps, pb);
        s1 += v; s2 += v * v;
    }
    __shared__ float r1[TPB], r2[TPB];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* st = stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * c;
        stat_add(st + ch, (long)PSSR_STAT_STRIPES * 2 * c, r1[0]); stat_add(st + c + ch, (long)PSSR_STAT_STRIPES * 2 * c, r2[0]);
    }
}

// Sum of the f64 statistic stripes for one channel with the stripes spread over SG thread groups of the workgroup (these kernels
// are a few microseconds of pure dependent latency on the step's critical path: 64 serial loads per thread otherwise).
// Launch with dim3(STRIPE_CH, STRIPE_SG) threads; returns the full sums in every thread of group 0 (others return partials).
constexpr int STRIPE_CH = 32, STRIPE_SG = 8;
template <int NV>
__device__ __forceinline__ void stripe_sums(const double* __restrict__ base, long stripe_stride, const long (&off)[NV], int stripes, bool active,
                                            double (&out)[NV]) {
    __shared__ double red[STRIPE_SG][STRIPE_CH][NV];
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    // a thread's rows eight at a time: all loads issued before the first add (a rolled loop waited an L2 round trip per row -- these
    // kernels are 5 us links of the dependent chain, 74 of them per c2 step); the order of the adds is unchanged
    if (active)
        for (int k0 = threadIdx.y; k0 < stripes; k0 += STRIPE_SG * 8) {
            double t[8][NV];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j * STRIPE_SG;
#pragma unroll
                for (int v = 0; v < NV; ++v) t[j][v] = k < stripes ? base[(long)k * stripe_stride + off[v]] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += t[j][v];
        }
#pragma unroll
    for (int v = 0; v < NV; ++v) red[threadIdx.y][threadIdx.x][v] = acc[v];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double t = 0.0;
        for (int g = 0; g < STRIPE_SG; ++g) t += red[g][threadIdx.x][v];
        out[v] = t;
    }
}

__global__ void bn_finalize_kernel(const double* stats, double count, const float* gamma, const float* beta, float eps, float momentum,
                                   float* rmean, float* rvar, float* scale, float* shift, float* mean, float* invstd, int c) {
    const int i = blockIdx.x * STRIPE_CH + threadIdx.x;
    const long off[2] = {i, (long)c + i};
    double sv[2];
    stripe_sums<2>(stats, 2L * c, off, PSSR_STAT_ROWS, i < c, sv);
    if (i >= c || threadIdx.y != 0) return;
    const double s1 = sv[0], s2 = sv[1];
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0) var = 0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[i] : 1.f, b = beta ? beta[i] : 0.f;
    scale[i] = g * is;
    shift[i] = b - (float)mu * g * is;
    if (mean) mean[i] = (float)mu;
    if (invstd) invstd[i] = is;
    if (rmean) {
        const double unbiased = count > 1 ? var * count / (count - 1) : var;
        rmean[i] = (1.f - momentum) * rmean[i] + momentum * (float)mu;
        rvar[i] = (1.f - momentum) * rvar[i] + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rmean, const float* rvar, float eps,
                               float* scale, float* shift, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float is = 1.f / sqrtf(rvar[i] + eps);
    scale[i] = gamma[i] * is;
    shift[i] = beta[i] - rmean[i] * gamma[i] * is;
}

__global__ void bn_bwd_coefs_kernel(const double* stats, double count, const float* gamma, const float* mean, const float* invstd,
                                    float* A, float* B, float* Cc, float* dgamma, float* dbeta, int c) {
    const int i = blockIdx.x * STRIPE_CH + threadIdx.x;
    const long off[2] = {i, (long)c + i};
    double sv[2];
    stripe_sums<2>(stats, 2L * c, off, PSSR_STAT_ROWS, i < c, sv);
    if (i >= c || threadIdx.y != 0) return;
    const double s1 = sv[0], s2 = sv[1];
    const double c1 = s1 / count, c2 = s2 / count;
    const double g = gamma[i], is = invstd[i], mu = mean[i];
    A[i] = (float)(g * is);
    B[i] = (float)(-g * is * is * c2);
    Cc[i] = (float)(g * is * (mu * is * c2 - c1));
    if (dgamma) dgamma[i] = (float)s2;
    if (dbeta) dbeta[i] = (float)s1;
}

// xcol[n,y,x, ch*9+tap] = bn(x[n,ch,y+ky-1,x+kx-1]/128-1) with zero padding; channels >= 9c are zero
template <typename T>
__global__ void input_im2col_kernel(const float* __restrict__ x, T* __restrict__ xcol, int n, int c, int h, int w, int xc,
                                    float ps, float pb, const float* scale, const float* shift) {
    const long total = (long)n * h * w * xc;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i % xc;
        long pix = i / xc;
        const int px = pix % w; pix /= w;
        const int py = pix % h;
        const int img = pix / h;
        float v = 0.f;
        if (k < 9 * c) {
            const int ch = k / 9, tap = k % 9;
            const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
            if (sy >= 0 && sy < h && sx >= 0 && sx < w)
                v = fmaf(fmaf(x[(((long)img * c + ch) * h + sy) * w + sx], ps, pb), scale[ch], shift[ch]);
        }
        xcol[i] = (T)v;
    }
}

// fold the gradient of xcol back onto the normalised input and reduce the input-BN parameter grads
template <typename T>
__global__ void input_norm_bwd_kernel(const T* __restrict__ da, const T* __restrict__ db, int xc, const T* __restrict__ dp, int pc, int pk,
                                      const float* __restrict__ x,
                                      float ps, float pb, const float* mean, const float* invstd, int n, int c, int h, int w, double* stats) {
    const int ch = blockIdx.y;
    const long hw = (long)h * w, total = (long)n * hw;
    float s1 = 0.f, s2 = 0.f;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long)gridDim.x * TPB) {
        const int img = i / hw;
        const int py = (i % hw) / w, px = i % w;
        float g = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // xcol[q][ch*9+tap] = x0[q + (ky-1,kx-1)]  =>  x0[p] receives dxcol[p - (ky-1,kx-1)][tap]
            const int qy = py - (tap / 3 - 1), qx = px - (tap % 3 - 1);
            if (qy >= 0 && qy < h && qx >= 0 && qx < w) {
                const long q = ((long)img * h + qy) * w + qx;
                if (da) g += (float)da[q * xc + ch * 9 + tap];
                if (db) g += (float)db[q * xc + ch * 9 + tap];
            }
        }
        if (dp) {   // patchify stem (pssr_input_patchify): pixel (py,px) is element ch*pk*pk + (py%pk)*pk + px%pk of patch (py/pk, px/pk)
            const long q = ((long)img * (h / pk) + py / pk) * (w / pk) + px / pk;
            g += (float)dp[q * pc + ch * pk * pk + (py % pk) * pk + px % pk];
        }
        const float xh = (fmaf(x[((long)img * c + ch) * hw + i % hw], ps, pb) - mean[ch]) * invstd[ch];
        s1 += g; s2 += g * xh;
    }
    __shared__ float r1[TPB], r2[TPB];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    for (int o = TPB / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* st = stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * c;
        stat_add(st + ch, (long)PSSR_STAT_STRIPES * 2 * c, r1[0]); stat_add(st + c + ch, (long)PSSR_STAT_STRIPES * 2 * c, r2[0]);
    }
}

template <typename T>
__global__ void maxpool2_kernel(Ref in, MRef out, int n, int h, int w, int c) {
    const int cg = c / 4, ho = h / 2, wo = w / 2;
    const long total = (long)n * ho * wo * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (i % cg) * 4;
        long pix = i / cg;
        const int ox = pix % wo; pix /= wo;
        const int oy = pix % ho;
        const int img = pix / ho;
        const long p00 = ((long)img * h + 2 * oy) * w + 2 * ox;
        float a[4], b[4], cc[4], d[4], m[4];
        load4(at<T>(in, p00, c0), a); load4(at<T>(in, p00 + 1, c0), b);
        load4(at<T>(in, p00 + w, c0), cc); load4(at<T>(in, p00 + w + 1, c0), d);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(cc[e], d[e]));
        store4(at<T>(out, ((long)img * ho + oy) * wo + ox, c0), m);
    }
}

// dout = dskip (+) route(dpool): the FIRST maximum in row-major window order receives the gradient (torch semantics)
template <typename T>
__global__ void maxpool2_bwd_kernel(Ref act, Ref dpool, Ref dskip, MRef dout, int n, int h, int w, int c) {
    const int cg = c / 4, ho = h / 2, wo = w / 2;
    const long total = (long)n * ho * wo * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (i % cg) * 4;
        long pix = i / cg;
        const int ox = pix % wo; pix /= wo;
        const int oy = pix % ho;
        const int img = pix / ho;
        const long p00 = ((long)img * h + 2 * oy) * w + 2 * ox;
        const long pp[4] = {p00, p00 + 1, p00 + w, p00 + w + 1};
        float v[4][4], g[4], o[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) load4(at<T>(act, pp[k], c0), v[k]);
        load4(at<T>(dpool, ((long)img * ho + oy) * wo + ox, c0), g);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (dskip.p) load4(at<T>(dskip, pp[k], c0), o[k]);
            else { o[k][0] = o[k][1] = o[k][2] = o[k][3] = 0.f; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int best = 0; float bv = v[0][e];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (v[k][e] > bv) { bv = v[k][e]; best = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k == best) o[k][e] += g[e];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) store4(at<T>(dout, pp[k], c0), o[k]);
    }
    // odd trailing rows/cols (H or W odd) take only the skip gradient
    if ((h & 1) || (w & 1)) {
        const long tot2 = (long)n * h * w * cg;
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < tot2; i += (long)gridDim.x * blockDim.x) {
            const int c0 = (i % cg) * 4;
            const long pix = i / cg;
            const int px = pix % w, py = (pix / w) % h;
            if (py < 2 * ho && px < 2 * wo) continue;
            float o[4] = {0, 0, 0, 0};
            if (dskip.p) load4(at<T>(dskip, pix, c0), o);
            store4(at<T>(dout, pix, c0), o);
        }
    }
}

// forward: out[n, r*y+i, r*x+j, c] = in[n, y, x, c*r*r + i*r + j]; `inverse` swaps the roles (gradient)
template <typename T>
__global__ void pixel_shuffle_kernel(Ref lo, MRef hi, int n, int h, int w, int c_hi, int r, int inverse) {
    const long total = (long)n * h * r * w * r * c_hi;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = i % c_hi;
        long pix = i / c_hi;
        const int X = pix % (w * r); pix /= (w * r);
        const int Y = pix % (h * r);
        const int img = pix / (h * r);
        const long phi = ((long)img * h * r + Y) * (w * r) + X;
        const long plo = ((long)img * h + Y / r) * w + X / r;
        const int clo = c * r * r + (Y % r) * r + (X % r);
        if (!inverse) *at<T>(hi, phi, c) = *at<T>(lo, plo, clo);
        else *((T*)lo.p + plo * lo.cs + lo.co + clo) = *((const T*)hi.p + phi * hi.cs + hi.co + c);
    }
}

// r = 2, 16-bit storage, channel counts / strides / offsets multiples of 8: a thread moves the 32 low-resolution channels
// 4c + k (c = 0..7 of its group, k = 2i + j) = 64 contiguous bytes <-> the same 8 high-resolution channels of the 4 pixels
// (2y + i, 2x + j), 16 bytes each, de-interleaving 16-bit halves in registers (the kernel above moves 2 bytes per thread).
template <typename T>
__global__ __launch_bounds__(TPB) void pixel_shuffle2_kernel(Ref lo, MRef hi, int n, int h, int w, int c_hi, int inverse) {
    static_assert(sizeof(T) == 2, "16-bit storage");
    const int cg = c_hi / 8;
    const long total = (long)n * h * w * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        long pix = i / cg;
        const int x = (int)(pix % w); pix /= w;
        const int y = (int)(pix % h);
        const long img = pix / h;
        const long plo = (img * h + y) * w + x;
        unsigned short* lop = (unsigned short*)lo.p + plo * lo.cs + lo.co + g * 32;
        unsigned short e[32];
        if (!inverse) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(u32x4*)(e + 8 * q) = *(const u32x4*)(lop + 8 * q);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long phi = (img * 2 * h + 2 * y + (k >> 1)) * (2L * w) + 2 * x + (k & 1);
            unsigned short* hip_ = (unsigned short*)hi.p + phi * hi.cs + hi.co + g * 8;
            unsigned short v[8];
            if (!inverse) {
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = e[4 * c + k];
                *(u32x4*)hip_ = *(const u32x4*)v;
            } else {
                *(u32x4*)v = *(const u32x4*)hip_;
#pragma unroll
                for (int c = 0; c < 8; ++c) e[4 * c + k] = v[c];
            }
        }
        if (inverse) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *(u32x4*)(lop + 8 * q) = *(const u32x4*)(e + 8 * q);
        }
    }
}

template <typename T>
__global__ void relu_bwd_stats_kernel(Ref dout, Ref out, Ref y, const float* mean, const float* invstd, MRef dz, double* stats,
                                      long npix, int c, ChanMap m) {
    __shared__ float lds[TPB * 8];
    float acc[2][4];
    auto reset = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][e] = acc[1][e] = 0.f;
    };
    reset();
    for_items(m, npix,
        [&](long pix, int c0) {
            float g[4], o[4], yv[4], mu[4], is[4];
            load4(at<T>(dout, pix, c0), g); load4(at<T>(out, pix, c0), o); load4(at<T>(y, pix, c0), yv);
            load4(mean + c0, mu); load4(invstd + c0, is);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g[e] = o[e] > 0.f ? g[e] : 0.f;
                acc[0][e] += g[e];
                acc[1][e] += g[e] * (yv[e] - mu[e]) * is[e];
            }
            store4(at<T>(dz, pix, c0), g);
        },
        [&](int cg) { flush_sums<2>(m, cg, acc, stats, c, lds); reset(); });
}

template <typename T>
__global__ void bn_bwd_apply_kernel(Ref g, Ref y, const float* A, const float* B, const float* Cc, MRef dy, long npix, int c) {
    const int cg = c / 4;
    const long total = npix * cg;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c0 = (i % cg) * 4;
        const long pix = i / cg;
        float gv[4], yv[4], a[4], b[4], cc[4], o[4];
        load4(at<T>(g, pix, c0), gv); load4(at<T>(y, pix, c0), yv);
        load4(A + c0, a); load4(B + c0, b); load4(Cc + c0, cc);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(a[e], gv[e], fmaf(b[e], yv[e], cc[e]));
        store4(at<T>(dy, pix, c0), o);
    }
}

// 16-bit storage, power-of-two channel counts (every BatchNorm of the models): 8 channels = one 16-byte access per thread and
// tensor, the thread's channels fixed for the whole launch (the grid stride is a multiple of the channel-group count), so the
// per-channel coefficients / statistics live in registers and the loop body is loads, a few FMAs and a store.
template <typename T>
__global__ __launch_bounds__(TPB) void bn_bwd_apply8_kernel(Ref g, Ref y, const float* __restrict__ A, const float* __restrict__ B,
                                                            const float* __restrict__ Cc, MRef dy, long npix, int c, int cg_log2) {
    using X = TT<T>;
    const int cg = 1 << cg_log2;
    const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c0 = (int)(t0 & (cg - 1)) * 8;
    float a[8], b[8], cc[8];
    load4(A + c0, a); load4(A + c0 + 4, a + 4); load4(B + c0, b); load4(B + c0 + 4, b + 4); load4(Cc + c0, cc); load4(Cc + c0 + 4, cc + 4);
    const long total = npix << cg_log2;
    for (long i = t0; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i >> cg_log2;
        float gv[8], yv[8], o[8];
        X::unpack(*(const u32x4*)at<T>(g, pix, c0), gv);
        X::unpack(*(const u32x4*)at<T>(y, pix, c0), yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fmaf(a[e], gv[e], fmaf(b[e], yv[e], cc[e]));
        *(u32x4*)at<T>(dy, pix, c0) = X::pack(o);
    }
}

// a = relu(scale * y + shift) materialised in the storage type with the SAME helper the convolution loaders use for their BatchNorm +
// ReLU prologue (X::bn_relu: f32 FMA, one rounding, 16-bit max) -- bit-identical to what a prologue would have staged.  The engine
// runs it on the second stream under the forward pass so that the weight-gradient kernel can take its input by LDS-DMA.
template <typename T>
__global__ __launch_bounds__(TPB) void bn_relu_apply8_kernel(Ref y, const float* __restrict__ scale, const float* __restrict__ shift, MRef a, long npix,
                                                             int cg_log2) {
    using X = TT<T>;
    const int cg = 1 << cg_log2;
    const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c0 = (int)(t0 & (cg - 1)) * 8;
    float sc[8], sh[8];
    load4(scale + c0, sc); load4(scale + c0 + 4, sc + 4); load4(shift + c0, sh); load4(shift + c0 + 4, sh + 4);
    const long total = npix << cg_log2;
    for (long i = t0; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i >> cg_log2;
        *(u32x4*)at<T>(a, pix, c0) = X::bn_relu(*(const u32x4*)at<T>(y, pix, c0), sc, sh);
    }
}

template <typename T>
__global__ __launch_bounds__(TPB) void relu_bwd_stats8_kernel(Ref dout, Ref out, Ref y, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              MRef dz, double* __restrict__ stats, long npix, int c, int cg_log2) {
    using X = TT<T>;
    __shared__ float lds[TPB * 16];
    const int cg = 1 << cg_log2;
    const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int cgi = (int)(t0 & (cg - 1)), c0 = cgi * 8;
    float mu[8], is[8], s1[8], s2[8];
    load4(mean + c0, mu); load4(mean + c0 + 4, mu + 4); load4(invstd + c0, is); load4(invstd + c0 + 4, is + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const long total = npix << cg_log2;
    for (long i = t0; i < total; i += (long)gridDim.x * blockDim.x) {
        const long pix = i >> cg_log2;
        float gv[8], ov[8], yv[8];
        X::unpack(*(const u32x4*)at<T>(dout, pix, c0), gv);
        X::unpack(*(const u32x4*)at<T>(out, pix, c0), ov);
        X::unpack(*(const u32x4*)at<T>(y, pix, c0), yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gv[e] = ov[e] > 0.f ? gv[e] : 0.f;
            s1[e] += gv[e];
            s2[e] += gv[e] * (yv[e] - mu[e]) * is[e];
        }
        *(u32x4*)at<T>(dz, pix, c0) = X::pack(gv);
    }
    // threads tid, tid + cg, ... of the workgroup hold the same channels (TPB is a multiple of cg): combine, then f64 atomics
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 8; ++e) { lds[tid * 16 + e] = s1[e]; lds[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    double* dst = stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * c;
    for (int j = tid; j < cg * 16; j += TPB) {
        const int g_ = j >> 4, q = j & 15;
        float t = 0.f;
        for (int k = g_; k < TPB; k += cg) t += lds[k * 16 + q];
        stat_add(dst + (long)(q >> 3) * c + g_ * 8 + (q & 7), (long)PSSR_STAT_STRIPES * 2 * c, t);
    }
}

// Combine the 8-channel partial sums s1 / s2 that the threads of a workgroup hold for their fixed channel group (threads tid, tid + cg,
// ... share channels; TPB is a multiple of cg) and add them to the striped f64 statistic rows: [sum][C] | [sum * xhat][C].
__device__ __forceinline__ void flush_stats8(const float (&s1)[8], const float (&s2)[8], int cg, int c, double* __restrict__ stats, float* lds) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int e = 0; e < 8; ++e) { lds[tid * 16 + e] = s1[e]; lds[tid * 16 + 8 + e] = s2[e]; }
    __syncthreads();
    double* dst = stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * c;
    for (int j = tid; j < cg * 16; j += TPB) {
        const int g_ = j >> 4, q = j & 15;
        float t = 0.f;
        for (int k = g_; k < TPB; k += cg) t += lds[k * 16 + q];
        stat_add(dst + (long)(q >> 3) * c + g_ * 8 + (q & 7), (long)PSSR_STAT_STRIPES * 2 * c, t);
    }
    __syncthreads();
}

// relu_bwd_stats8 with the max-pool backward folded into its loader (round 4: 4 launches and 3 tensor passes per encoder level less on the
// backward's dependent chain).  The block output `out` feeds max_pool2d AND the decoder's skip, so
//   d(out) = dskip + route(dpool)   (the FIRST maximum of a 2x2 window in row-major order takes the pooled gradient: torch semantics)
// is formed per window in registers -- rounded to the storage type exactly where maxpool2_bwd_kernel used to store it, so dz is bit for
// bit what the two kernels produced -- then masked by out > 0 and summed into the BatchNorm-backward statistics.  An item is one 2x2
// window x 8 channels: out / dskip / y are read once (4 x 16 bytes each), dpool once, dz is written once.  H and W even.
template <typename T>
__global__ __launch_bounds__(TPB) void relu_bwd_stats8_pool_kernel(Ref dpool, Ref dskip, Ref out, Ref y, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, MRef dz, double* __restrict__ stats,
                                                                   int n, int h, int w, int c, int cg_log2) {
    using X = TT<T>;
    __shared__ float lds[TPB * 16];
    const int cg = 1 << cg_log2;
    const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int c0 = (int)(t0 & (cg - 1)) * 8;
    float mu[8], is[8], s1[8], s2[8];
    load4(mean + c0, mu); load4(mean + c0 + 4, mu + 4); load4(invstd + c0, is); load4(invstd + c0 + 4, is + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    const int ho = h >> 1, wo = w >> 1;
    const long total = ((long)n * ho * wo) << cg_log2;
    for (long i = t0; i < total; i += (long)gridDim.x * blockDim.x) {
        long win = i >> cg_log2;
        const int ox = (int)(win % wo); win /= wo;
        const int oy = (int)(win % ho);
        const long img = win / ho;
        const long p00 = (img * h + 2 * oy) * w + 2 * ox;
        const long pp[4] = {p00, p00 + 1, p00 + w, p00 + w + 1};
        float ov[4][8], gv[4][8], yv[4][8], gp[8];
        X::unpack(*(const u32x4*)at<T>(dpool, (img * ho + oy) * wo + ox, c0), gp);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            X::unpack(*(const u32x4*)at<T>(out, pp[k], c0), ov[k]);
            X::unpack(*(const u32x4*)at<T>(dskip, pp[k], c0), gv[k]);
            X::unpack(*(const u32x4*)at<T>(y, pp[k], c0), yv[k]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int best = 0; float bv = ov[0][e];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (ov[k][e] > bv) { bv = ov[k][e]; best = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (k == best) gv[k][e] += gp[e];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r[8];
            X::unpack(X::pack(gv[k]), r);            // d(out) in the storage type, as the separate max-pool backward stored it
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                r[e] = ov[k][e] > 0.f ? r[e] : 0.f;
                s1[e] += r[e];
                s2[e] += r[e] * (yv[k][e] - mu[e]) * is[e];
            }
            *(u32x4*)at<T>(dz, pp[k], c0) = X::pack(r);
        }
    }
    flush_stats8(s1, s2, cg, c, stats, lds);
}

// relu_bwd_stats8 with the inverse pixel shuffle (r = 2) folded into its loader: the block output was shuffled into the first c / 4
// channels of the next decoder level's concat buffer, so d(out)[n, y, x, 4 ch + 2 i + j] = dcat[n, 2 y + i, 2 x + j, ch].  An item is one
// low-resolution pixel x 32 channels = 8 channels (16 bytes) of each of its 4 high-resolution pixels, de-interleaved in registers as
// pixel_shuffle2_kernel does; four 8-channel pieces of out / y / dz per item.  c a power of two >= 32.
template <typename T>
__global__ __launch_bounds__(TPB) void relu_bwd_stats8_unshuffle_kernel(Ref dhi, Ref out, Ref y, const float* __restrict__ mean,
                                                                        const float* __restrict__ invstd, MRef dz, double* __restrict__ stats,
                                                                        int n, int h, int w, int c, int cg_log2) {
    using X = TT<T>;
    static_assert(sizeof(T) == 2, "16-bit storage");
    __shared__ float lds[TPB * 16];
    const int cg = 1 << cg_log2;                        // 32-channel groups
    const long t0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
    const int g32 = (int)(t0 & (cg - 1)), c0 = g32 * 32;
    float s1[4][8], s2[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[q][e] = 0.f; s2[q][e] = 0.f; }
    const long total = ((long)n * h * w) << cg_log2;
    for (long i = t0; i < total; i += (long)gridDim.x * blockDim.x) {
        long pix = i >> cg_log2;
        const int x = (int)(pix % w); pix /= w;
        const int yy = (int)(pix % h);
        const long img = pix / h;
        const long plo = (img * h + yy) * w + x;
        unsigned short e16[32];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long phi = (img * 2 * h + 2 * yy + (k >> 1)) * (2L * w) + 2 * x + (k & 1);
            unsigned short v[8];
            *(u32x4*)v = *(const u32x4*)((const unsigned short*)dhi.p + phi * dhi.cs + dhi.co + g32 * 8);
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) e16[4 * ch + k] = v[ch];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float gv[8], ov[8], yv[8], mu[8], is[8];
            X::unpack(*(const u32x4*)(e16 + 8 * q), gv);
            X::unpack(*(const u32x4*)at<T>(out, plo, c0 + 8 * q), ov);
            X::unpack(*(const u32x4*)at<T>(y, plo, c0 + 8 * q), yv);
            load4(mean + c0 + 8 * q, mu); load4(mean + c0 + 8 * q + 4, mu + 4);
            load4(invstd + c0 + 8 * q, is); load4(invstd + c0 + 8 * q + 4, is + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                gv[e] = ov[e] > 0.f ? gv[e] : 0.f;
                s1[q][e] += gv[e];
                s2[q][e] += gv[e] * (yv[e] - mu[e]) * is[e];
            }
            *(u32x4*)at<T>(dz, plo, c0 + 8 * q) = X::pack(gv);
        }
    }
    // four rounds through the 16 KB combine buffer: round q carries channels c0 + 8 q .. + 7 of every 32-channel group
    const int tid = threadIdx.x;
    double* dst = stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * 2 * c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { lds[tid * 16 + e] = s1[q][e]; lds[tid * 16 + 8 + e] = s2[q][e]; }
        __syncthreads();
        for (int j = tid; j < cg * 16; j += TPB) {
            const int g_ = j >> 4, r = j & 15;
            float t = 0.f;
            for (int k = g_; k < TPB; k += cg) t += lds[k * 16 + r];
            stat_add(dst + (long)(r >> 3) * c + g_ * 32 + 8 * q + (r & 7), (long)PSSR_STAT_STRIPES * 2 * c, t);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void channel_sum_kernel(Ref x, double* out, long npix, int c, ChanMap m) {
    __shared__ float lds[TPB * 4];
    float acc[1][4] = {{0, 0, 0, 0}};
    for_items(m, npix,
        [&](long pix, int c0) {
            float v[4];
            load4(at<T>(x, pix, c0), v);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[0][e] += v[e];
        },
        [&](int cg) { flush_sums<1>(m, cg, acc, out, c, lds); acc[0][0] = acc[0][1] = acc[0][2] = acc[0][3] = 0.f; });
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int n, int c, long hw, int cs, float scale) {
    const long total = (long)n * hw * cs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i % cs;
        const long pix = i / cs;
        const long img = pix / hw, p = pix % hw;
        out[i] = (T)(k < c ? in[(img * c + k) * hw + p] * scale : 0.f);
    }
}

// Up to PSSR_COPY_BATCH_MAX small f32 copies in one launch (blockIdx.y = item): the engine moves ~10 tiny side results per
// residual block into their gradient slots, and as memcpy nodes of the step graph each cost ~10 us
__global__ __launch_bounds__(256) void copy_batch_kernel(pssr_copy_batch items) {
    const int it = blockIdx.y;
    float* __restrict__ dst = items.dst[it];
    const float* __restrict__ src = items.src[it];
    const long n = items.n[it];
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) dst[i] = src[i];
}

__global__ void clip_u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = fminf(fmaxf(in[i], 0.f), 255.f);
        out[i] = (uint8_t)v;   // truncation toward zero, as numpy astype(uint8) on a clipped array
    }
}

__global__ void f64_to_f32_kernel(const double* in, float* out, int n, int accumulate, int stripes) {
    const int i = blockIdx.x * STRIPE_CH + threadIdx.x;
    const long off[1] = {i};
    double sv[1];
    stripe_sums<1>(in, (long)n, off, stripes, i < n, sv);
    if (i >= n || threadIdx.y != 0) return;
    const double s = sv[0];
    out[i] = accumulate ? out[i] + (float)s : (float)s;
}

// several folds by one launch (blockIdx.y = item): RDNet's backward pass ends ~115 bias / LayerNorm gradient sums per step this way
__global__ void f64_to_f32_batch_kernel(pssr_fold_batch items, int stripes) {
    const int it = blockIdx.y;
    const int n = items.n[it];
    if ((int)(blockIdx.x * STRIPE_CH) >= n) return;          // (uniform per workgroup)
    const int i = blockIdx.x * STRIPE_CH + threadIdx.x;
    const long off[1] = {i};
    double sv[1];
    stripe_sums<1>(items.src[it], (long)n, off, stripes, i < n, sv);
    if (i >= n || threadIdx.y != 0) return;
    float* out = items.dst[it];
    out[i] = items.accumulate[it] ? out[i] + (float)sv[0] : (float)sv[0];
}

static inline int grid1d(long total) { long b = (total + TPB - 1) / TPB; return (int)(b < 8192 ? (b > 0 ? b : 1) : 8192); }

}  // namespace

#define DISPATCH_T(dtype, CALL)                                             \
    do {                                                                    \
        if ((dtype) == PSSR_BF16) { using T = bf16_t; CALL; }               \
        else if ((dtype) == PSSR_F16) { using T = f16_t; CALL; }            \
        else if ((dtype) == PSSR_F32) { using T = float; CALL; }            \
        else { pssr_set_error("bad dtype %d", (dtype)); return PSSR_ERR_ARG; } \
    } while (0)

extern "C" {

int pssr_channel_stats_nchw(const float* x, int n, int c, int64_t hw, float pre_scale, float pre_shift, double* stats, pssr_stream_t s) {
    PSSR_CHECK(x && stats && n > 0 && c > 0 && hw > 0, PSSR_ERR_ARG, "channel_stats_nchw: bad args");
    int chunks = (int)((hw + TPB * 8 - 1) / (TPB * 8));
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(nchw_stats_kernel, dim3(chunks, n * c), dim3(TPB), 0, (hipStream_t)s, x, n, c, (long)hw, pre_scale, pre_shift, stats);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_bn_finalize(const double* stats, double count, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd, int c, pssr_stream_t s) {
    PSSR_CHECK(stats && scale && shift && c > 0 && count > 0, PSSR_ERR_ARG, "bn_finalize: bad args");
    PSSR_CHECK((running_mean == nullptr) == (running_var == nullptr), PSSR_ERR_ARG, "bn_finalize: running stats come in pairs");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(c, STRIPE_CH)), dim3(STRIPE_CH, STRIPE_SG), 0, (hipStream_t)s, stats, count, gamma, beta, eps, momentum,
                       running_mean, running_var, scale, shift, mean, invstd, c);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                        float* scale, float* shift, int c, pssr_stream_t s) {
    PSSR_CHECK(gamma && beta && running_mean && running_var && scale && shift && c > 0, PSSR_ERR_ARG, "bn_eval_affine: bad args");
    hipLaunchKernelGGL(bn_eval_kernel, dim3(cdiv(c, TPB)), dim3(TPB), 0, (hipStream_t)s, gamma, beta, running_mean, running_var, eps, scale, shift, c);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_bn_bwd_coefs(const double* stats, double count, const float* gamma, const float* mean, const float* invstd,
                      float* coef_a, float* coef_b, float* coef_c, float* dgamma, float* dbeta, int c, pssr_stream_t s) {
    PSSR_CHECK(stats && gamma && mean && invstd && coef_a && coef_b && coef_c && c > 0 && count > 0, PSSR_ERR_ARG, "bn_bwd_coefs: bad args");
    hipLaunchKernelGGL(bn_bwd_coefs_kernel, dim3(cdiv(c, STRIPE_CH)), dim3(STRIPE_CH, STRIPE_SG), 0, (hipStream_t)s, stats, count, gamma, mean, invstd,
                       coef_a, coef_b, coef_c, dgamma, dbeta, c);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_input_im2col(const float* x, void* xcol, int n, int c, int h, int w, int xc, float pre_scale, float pre_shift,
                      const float* scale, const float* shift, int dtype, pssr_stream_t s) {
    PSSR_CHECK(x && xcol && scale && shift && n > 0 && c > 0 && h > 0 && w > 0 && xc >= 9 * c && xc % 16 == 0, PSSR_ERR_ARG, "input_im2col: bad args");
    const long total = (long)n * h * w * xc;
    DISPATCH_T(dtype, hipLaunchKernelGGL(input_im2col_kernel<T>, dim3(grid1d(total)), dim3(TPB), 0, (hipStream_t)s, x, (T*)xcol, n, c, h, w, xc,
                                         pre_scale, pre_shift, scale, shift));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_input_norm_bwd2(const void* dxcol_a, const void* dxcol_b, int xc, const void* dpatch, int pc, int patch, const float* x,
                         float pre_scale, float pre_shift, const float* mean, const float* invstd, int n, int c, int h, int w, double* stats,
                         int dtype, pssr_stream_t s) {
    PSSR_CHECK((dxcol_a || dxcol_b || dpatch) && x && mean && invstd && stats, PSSR_ERR_ARG, "input_norm_bwd: bad args");
    PSSR_CHECK(!(dxcol_a || dxcol_b) || xc >= 9 * c, PSSR_ERR_ARG, "input_norm_bwd: xc=%d", xc);
    PSSR_CHECK(!dpatch || (patch > 0 && pc >= c * patch * patch && h % patch == 0 && w % patch == 0), PSSR_ERR_ARG, "input_norm_bwd: patch gradient layout");
    const long total = (long)n * h * w;
    int gx = (int)((total + TPB * 4 - 1) / (TPB * 4));
    if (gx > 512) gx = 512;
    DISPATCH_T(dtype, hipLaunchKernelGGL(input_norm_bwd_kernel<T>, dim3(gx, c), dim3(TPB), 0, (hipStream_t)s, (const T*)dxcol_a, (const T*)dxcol_b, xc,
                                         (const T*)dpatch, pc, patch > 0 ? patch : 1, x, pre_scale, pre_shift, mean, invstd, n, c, h, w, stats));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_input_norm_bwd(const void* dxcol_a, const void* dxcol_b, int xc, const float* x, float pre_scale, float pre_shift,
                        const float* mean, const float* invstd, int n, int c, int h, int w, double* stats, int dtype, pssr_stream_t s) {
    PSSR_CHECK(dxcol_a != nullptr, PSSR_ERR_ARG, "input_norm_bwd: bad args");
    return pssr_input_norm_bwd2(dxcol_a, dxcol_b, xc, nullptr, 0, 0, x, pre_scale, pre_shift, mean, invstd, n, c, h, w, stats, dtype, s);
}

#define CHECK_REF(name, cs, co, c)                                                                               \
    PSSR_CHECK((cs) % 4 == 0 && (co) % 4 == 0 && (co) + (c) <= (cs), PSSR_ERR_ARG, name ": bad channel stride/offset (%d,%d,%d)", cs, co, c)

int pssr_maxpool2(const void* in, int in_cs, int in_co, void* out, int out_cs, int out_co, int n, int h, int w, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(in && out && n > 0 && h > 1 && w > 1 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "maxpool2: bad args");
    CHECK_REF("maxpool2 in", in_cs, in_co, c); CHECK_REF("maxpool2 out", out_cs, out_co, c);
    const long total = (long)n * (h / 2) * (w / 2) * (c / 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_kernel<T>, dim3(grid1d(total)), dim3(TPB), 0, (hipStream_t)s, Ref{in, in_cs, in_co}, MRef{out, out_cs, out_co}, n, h, w, c));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_maxpool2_bwd(const void* act, int act_cs, int act_co, const void* dpool, int dp_cs, int dp_co,
                      const void* dskip, int ds_cs, int ds_co, void* dout, int do_cs, int do_co,
                      int n, int h, int w, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(act && dpool && dout && n > 0 && h > 1 && w > 1 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "maxpool2_bwd: bad args");
    CHECK_REF("maxpool2_bwd act", act_cs, act_co, c); CHECK_REF("maxpool2_bwd dpool", dp_cs, dp_co, c); CHECK_REF("maxpool2_bwd dout", do_cs, do_co, c);
    if (dskip) CHECK_REF("maxpool2_bwd dskip", ds_cs, ds_co, c);
    const long total = (long)n * (h / 2) * (w / 2) * (c / 4);
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_bwd_kernel<T>, dim3(grid1d(total)), dim3(TPB), 0, (hipStream_t)s, Ref{act, act_cs, act_co},
                                         Ref{dpool, dp_cs, dp_co}, Ref{dskip, ds_cs, ds_co}, MRef{dout, do_cs, do_co}, n, h, w, c));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_pixel_shuffle(const void* lo, int lo_cs, int lo_co, void* hi, int hi_cs, int hi_co, int n, int h, int w, int c_hi, int r,
                       int inverse, int dtype, pssr_stream_t s) {
    PSSR_CHECK(lo && hi && n > 0 && h > 0 && w > 0 && c_hi > 0 && r > 0, PSSR_ERR_ARG, "pixel_shuffle: bad args");
    PSSR_CHECK(lo_co + c_hi * r * r <= lo_cs && hi_co + c_hi <= hi_cs, PSSR_ERR_ARG, "pixel_shuffle: slice exceeds stride");
    const long total = (long)n * h * r * w * r * c_hi;
    if (r == 2 && dtype != PSSR_F32 && c_hi % 8 == 0 && ((lo_cs | lo_co | hi_cs | hi_co) & 7) == 0) {
        const long threads = (long)n * h * w * (c_hi / 8);
        hipLaunchKernelGGL(pixel_shuffle2_kernel<bf16_t>, dim3(grid1d(threads)), dim3(TPB), 0, (hipStream_t)s, Ref{lo, lo_cs, lo_co},
                           MRef{hi, hi_cs, hi_co}, n, h, w, c_hi, inverse);      // a byte permutation: one build serves bf16 and fp16
        PSSR_LAUNCH_CHECK();
        return PSSR_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(pixel_shuffle_kernel<T>, dim3(grid1d(total)), dim3(TPB), 0, (hipStream_t)s, Ref{lo, lo_cs, lo_co},
                                         MRef{hi, hi_cs, hi_co}, n, h, w, c_hi, r, inverse));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_relu_bwd_stats_pool(const void* dpool, int dp_cs, int dp_co, const void* dskip, int ds_cs, int ds_co, const void* out, int o_cs, int o_co,
                             const void* y, int y_cs, int y_co, const float* mean, const float* invstd, void* dz, int dz_cs, int dz_co,
                             double* stats, int n, int h, int w, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(dpool && dskip && out && y && mean && invstd && dz && stats && n > 0 && h > 1 && w > 1 && c > 0, PSSR_ERR_ARG, "relu_bwd_stats_pool: bad args");
    PSSR_CHECK(dtype == PSSR_BF16 || dtype == PSSR_F16, PSSR_ERR_UNSUPPORTED, "relu_bwd_stats_pool: 16-bit storage only (dtype %d)", dtype);
    PSSR_CHECK(h % 2 == 0 && w % 2 == 0, PSSR_ERR_UNSUPPORTED, "relu_bwd_stats_pool: %dx%d is not even", h, w);
    PSSR_CHECK(c >= 8 && c <= 8 * TPB && (c & (c - 1)) == 0, PSSR_ERR_UNSUPPORTED, "relu_bwd_stats_pool: c=%d must be a power of two in [8, %d]", c, 8 * TPB);
    PSSR_CHECK(((dp_cs | dp_co | ds_cs | ds_co | o_cs | o_co | y_cs | y_co | dz_cs | dz_co) & 7) == 0, PSSR_ERR_ARG, "relu_bwd_stats_pool: strides / offsets must be multiples of 8");
    CHECK_REF("relu_bwd_stats_pool dpool", dp_cs, dp_co, c); CHECK_REF("relu_bwd_stats_pool dskip", ds_cs, ds_co, c);
    CHECK_REF("relu_bwd_stats_pool out", o_cs, o_co, c); CHECK_REF("relu_bwd_stats_pool y", y_cs, y_co, c); CHECK_REF("relu_bwd_stats_pool dz", dz_cs, dz_co, c);
    int lg = 0;
    while ((8 << lg) < c) ++lg;
    const long threads = ((long)n * (h / 2) * (w / 2)) << lg;
    const int grid = (int)((threads + TPB - 1) / TPB < 2048 ? (threads + TPB - 1) / TPB : 2048);
    if (dtype == PSSR_BF16)
        hipLaunchKernelGGL(relu_bwd_stats8_pool_kernel<bf16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{dpool, dp_cs, dp_co}, Ref{dskip, ds_cs, ds_co},
                           Ref{out, o_cs, o_co}, Ref{y, y_cs, y_co}, mean, invstd, MRef{dz, dz_cs, dz_co}, stats, n, h, w, c, lg);
    else
        hipLaunchKernelGGL(relu_bwd_stats8_pool_kernel<f16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{dpool, dp_cs, dp_co}, Ref{dskip, ds_cs, ds_co},
                           Ref{out, o_cs, o_co}, Ref{y, y_cs, y_co}, mean, invstd, MRef{dz, dz_cs, dz_co}, stats, n, h, w, c, lg);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_relu_bwd_stats_unshuffle(const void* dhi, int dh_cs, int dh_co, const void* out, int o_cs, int o_co, const void* y, int y_cs, int y_co,
                                  const float* mean, const float* invstd, void* dz, int dz_cs, int dz_co, double* stats,
                                  int n, int h, int w, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(dhi && out && y && mean && invstd && dz && stats && n > 0 && h > 0 && w > 0 && c > 0, PSSR_ERR_ARG, "relu_bwd_stats_unshuffle: bad args");
    PSSR_CHECK(dtype == PSSR_BF16 || dtype == PSSR_F16, PSSR_ERR_UNSUPPORTED, "relu_bwd_stats_unshuffle: 16-bit storage only (dtype %d)", dtype);
    PSSR_CHECK(c >= 32 && c <= 32 * TPB && (c & (c - 1)) == 0, PSSR_ERR_UNSUPPORTED, "relu_bwd_stats_unshuffle: c=%d must be a power of two in [32, %d]", c, 32 * TPB);
    PSSR_CHECK(((dh_cs | dh_co | o_cs | o_co | y_cs | y_co | dz_cs | dz_co) & 7) == 0, PSSR_ERR_ARG, "relu_bwd_stats_unshuffle: strides / offsets must be multiples of 8");
    PSSR_CHECK(dh_co + c / 4 <= dh_cs, PSSR_ERR_ARG, "relu_bwd_stats_unshuffle: high-resolution slice exceeds its stride");
    CHECK_REF("relu_bwd_stats_unshuffle out", o_cs, o_co, c); CHECK_REF("relu_bwd_stats_unshuffle y", y_cs, y_co, c);
    CHECK_REF("relu_bwd_stats_unshuffle dz", dz_cs, dz_co, c);
    int lg = 0;
    while ((32 << lg) < c) ++lg;
    const long threads = ((long)n * h * w) << lg;
    const int grid = (int)((threads + TPB - 1) / TPB < 2048 ? (threads + TPB - 1) / TPB : 2048);
    if (dtype == PSSR_BF16)
        hipLaunchKernelGGL(relu_bwd_stats8_unshuffle_kernel<bf16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{dhi, dh_cs, dh_co}, Ref{out, o_cs, o_co},
                           Ref{y, y_cs, y_co}, mean, invstd, MRef{dz, dz_cs, dz_co}, stats, n, h, w, c, lg);
    else
        hipLaunchKernelGGL(relu_bwd_stats8_unshuffle_kernel<f16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{dhi, dh_cs, dh_co}, Ref{out, o_cs, o_co},
                           Ref{y, y_cs, y_co}, mean, invstd, MRef{dz, dz_cs, dz_co}, stats, n, h, w, c, lg);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_relu_bwd_stats(const void* dout, int do_cs, int do_co, const void* out, int o_cs, int o_co, const void* y, int y_cs, int y_co,
                        const float* mean, const float* invstd, void* dz, int dz_cs, int dz_co, double* stats,
                        int64_t npix, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(dout && out && y && mean && invstd && dz && stats && npix > 0 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "relu_bwd_stats: bad args");
    CHECK_REF("relu_bwd_stats dout", do_cs, do_co, c); CHECK_REF("relu_bwd_stats out", o_cs, o_co, c);
    CHECK_REF("relu_bwd_stats y", y_cs, y_co, c); CHECK_REF("relu_bwd_stats dz", dz_cs, dz_co, c);
    if (dtype != PSSR_F32 && c >= 8 && c <= 8 * TPB && (c & (c - 1)) == 0 && ((do_cs | do_co | o_cs | o_co | y_cs | y_co | dz_cs | dz_co) & 7) == 0) {
        int lg = 0;
        while ((8 << lg) < c) ++lg;
        const long threads = npix << lg;
        const int grid = (int)((threads + TPB - 1) / TPB < 2048 ? (threads + TPB - 1) / TPB : 2048);
        if (dtype == PSSR_BF16)
            hipLaunchKernelGGL(relu_bwd_stats8_kernel<bf16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{dout, do_cs, do_co}, Ref{out, o_cs, o_co},
                               Ref{y, y_cs, y_co}, mean, invstd, MRef{dz, dz_cs, dz_co}, stats, (long)npix, c, lg);
        else
            hipLaunchKernelGGL(relu_bwd_stats8_kernel<f16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{dout, do_cs, do_co}, Ref{out, o_cs, o_co},
                               Ref{y, y_cs, y_co}, mean, invstd, MRef{dz, dz_cs, dz_co}, stats, (long)npix, c, lg);
        PSSR_LAUNCH_CHECK();
        return PSSR_OK;
    }
    const ChanMap m = make_map(c);
    DISPATCH_T(dtype, hipLaunchKernelGGL(relu_bwd_stats_kernel<T>, dim3(grid_for(npix, m)), dim3(TPB), 0, (hipStream_t)s, Ref{dout, do_cs, do_co},
                                         Ref{out, o_cs, o_co}, Ref{y, y_cs, y_co}, mean, invstd, MRef{dz, dz_cs, dz_co}, stats, (long)npix, c, m));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_bn_bwd_apply(const void* g, int g_cs, int g_co, const void* y, int y_cs, int y_co, const float* coef_a, const float* coef_b,
                      const float* coef_c, void* dy, int dy_cs, int dy_co, int64_t npix, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(g && y && coef_a && coef_b && coef_c && dy && npix > 0 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "bn_bwd_apply: bad args");
    CHECK_REF("bn_bwd_apply g", g_cs, g_co, c); CHECK_REF("bn_bwd_apply y", y_cs, y_co, c); CHECK_REF("bn_bwd_apply dy", dy_cs, dy_co, c);
    if (dtype != PSSR_F32 && c >= 8 && c <= 8 * TPB && (c & (c - 1)) == 0 && ((g_cs | g_co | y_cs | y_co | dy_cs | dy_co) & 7) == 0) {
        int lg = 0;
        while ((8 << lg) < c) ++lg;
        const long threads = (long)npix << lg;
        const int grid = (int)((threads + TPB - 1) / TPB < 4096 ? (threads + TPB - 1) / TPB : 4096);
        if (dtype == PSSR_BF16)
            hipLaunchKernelGGL(bn_bwd_apply8_kernel<bf16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{g, g_cs, g_co}, Ref{y, y_cs, y_co}, coef_a, coef_b,
                               coef_c, MRef{dy, dy_cs, dy_co}, (long)npix, c, lg);
        else
            hipLaunchKernelGGL(bn_bwd_apply8_kernel<f16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{g, g_cs, g_co}, Ref{y, y_cs, y_co}, coef_a, coef_b,
                               coef_c, MRef{dy, dy_cs, dy_co}, (long)npix, c, lg);
        PSSR_LAUNCH_CHECK();
        return PSSR_OK;
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3(grid1d(npix * (c / 4))), dim3(TPB), 0, (hipStream_t)s, Ref{g, g_cs, g_co},
                                         Ref{y, y_cs, y_co}, coef_a, coef_b, coef_c, MRef{dy, dy_cs, dy_co}, (long)npix, c));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_bn_relu_apply(const void* y, int y_cs, int y_co, const float* scale, const float* shift, void* a, int a_cs, int a_co, int64_t npix, int c, int dtype,
                       pssr_stream_t s) {
    PSSR_CHECK(y && scale && shift && a && npix > 0, PSSR_ERR_ARG, "bn_relu_apply: bad args");
    PSSR_CHECK(dtype != PSSR_F32 && c >= 8 && c <= 8 * TPB && (c & (c - 1)) == 0 && ((y_cs | y_co | a_cs | a_co) & 7) == 0, PSSR_ERR_UNSUPPORTED,
               "bn_relu_apply: 16-bit storage, power-of-two channel count >= 8, 16-byte aligned slices (c=%d)", c);
    CHECK_REF("bn_relu_apply y", y_cs, y_co, c); CHECK_REF("bn_relu_apply a", a_cs, a_co, c);
    int lg = 0;
    while ((8 << lg) < c) ++lg;
    const long threads = (long)npix << lg;
    const int grid = (int)((threads + TPB - 1) / TPB < 4096 ? (threads + TPB - 1) / TPB : 4096);
    if (dtype == PSSR_BF16)
        hipLaunchKernelGGL(bn_relu_apply8_kernel<bf16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{y, y_cs, y_co}, scale, shift, MRef{a, a_cs, a_co}, (long)npix, lg);
    else
        hipLaunchKernelGGL(bn_relu_apply8_kernel<f16_t>, dim3(grid), dim3(TPB), 0, (hipStream_t)s, Ref{y, y_cs, y_co}, scale, shift, MRef{a, a_cs, a_co}, (long)npix, lg);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_channel_sum_nhwc(const void* x, int cs, int co, int64_t npix, int c, double* out, int dtype, pssr_stream_t s) {
    PSSR_CHECK(x && out && npix > 0 && c > 0 && c % 4 == 0, PSSR_ERR_ARG, "channel_sum: bad args");
    CHECK_REF("channel_sum x", cs, co, c);
    const ChanMap m = make_map(c);
    // every workgroup ends with c f64 atomics: with wide tensors on small maps (8192 pixels x 600 channels) 2048 workgroups spent
    // 43 us on 1.2 M atomics for 10 MB of input -- keep the launch near 128 k atomics
    int blocks = grid_for(npix, m);
    const int cap = 131072 / c < 128 ? 128 : 131072 / c;
    if (blocks > cap) blocks = cap;
    DISPATCH_T(dtype, hipLaunchKernelGGL(channel_sum_kernel<T>, dim3(blocks), dim3(TPB), 0, (hipStream_t)s, Ref{x, cs, co}, out, (long)npix, c, m));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_nchw_to_nhwc(const float* in, void* out, int n, int c, int64_t hw, int out_cs, float scale, int dtype, pssr_stream_t s) {
    PSSR_CHECK(in && out && n > 0 && c > 0 && hw > 0 && out_cs >= c, PSSR_ERR_ARG, "nchw_to_nhwc: bad args");
    const long total = (long)n * hw * out_cs;
    DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<T>, dim3(grid1d(total)), dim3(TPB), 0, (hipStream_t)s, in, (T*)out, n, c, (long)hw, out_cs, scale));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_copy_f32_batch(const pssr_copy_batch* items, int n_items, pssr_stream_t s) {
    PSSR_CHECK(items && n_items > 0 && n_items <= PSSR_COPY_BATCH_MAX, PSSR_ERR_ARG, "copy_f32_batch: 1..%d items", PSSR_COPY_BATCH_MAX);
    int64_t longest = 0;
    for (int i = 0; i < n_items; ++i) {
        PSSR_CHECK(items->dst[i] && items->src[i] && items->n[i] > 0, PSSR_ERR_ARG, "copy_f32_batch: item %d is empty", i);
        if (items->n[i] > longest) longest = items->n[i];
    }
    long gx = (longest + 255) / 256;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(copy_batch_kernel, dim3((unsigned)gx, n_items), dim3(256), 0, (hipStream_t)s, *items);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_clip_u8(const float* in, uint8_t* out, int64_t n, pssr_stream_t s) {
    PSSR_CHECK(in && out && n > 0, PSSR_ERR_ARG, "clip_u8: bad args");
    hipLaunchKernelGGL(clip_u8_kernel, dim3(grid1d(n)), dim3(TPB), 0, (hipStream_t)s, in, out, (long)n);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_f64_to_f32(const double* in, float* out, int n, int accumulate, int stripes, pssr_stream_t s) {
    PSSR_CHECK(in && out && n > 0 && stripes > 0, PSSR_ERR_ARG, "f64_to_f32: bad args");
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3(cdiv(n, STRIPE_CH)), dim3(STRIPE_CH, STRIPE_SG), 0, (hipStream_t)s, in, out, n, accumulate, stripes);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_f64_to_f32_batch(const pssr_fold_batch* items, int n_items, int stripes, pssr_stream_t s) {
    PSSR_CHECK(items && n_items > 0 && n_items <= PSSR_COPY_BATCH_MAX && stripes > 0, PSSR_ERR_ARG, "f64_to_f32_batch: 1..%d items", PSSR_COPY_BATCH_MAX);
    int longest = 0;
    for (int i = 0; i < n_items; ++i) {
        PSSR_CHECK(items->dst[i] && items->src[i] && items->n[i] > 0, PSSR_ERR_ARG, "f64_to_f32_batch: item %d is empty", i);
        if (items->n[i] > longest) longest = items->n[i];
    }
    hipLaunchKernelGGL(f64_to_f32_batch_kernel, dim3(cdiv(longest, STRIPE_CH), n_items), dim3(STRIPE_CH, STRIPE_SG), 0, (hipStream_t)s, *items, stripes);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

}  // extern "C"
