// Atrous / PSP-pooling variants of the models (pssr/models/_blocks.py:43-92: ResBlockA, PSP_Pooling): the HBM-bound pieces.
//
// A dilated 3x3 convolution is run as im2col (9 shifted, zero-padded copies of the activated input side by side along the
// channel axis: pssr_im2col_dil, with the pre-activation BatchNorm + ReLU of ResBlockA fused into the gather) followed by the
// ordinary 1x1 implicit-GEMM kernels over K = 9 * C (forward, input gradient, weight gradient: pssr_conv2d / pssr_conv2d_wgrad
// with weights packed in mode 4 / 5), and pssr_col2im_dil folds the gradient of the 9 copies back (with the ReLU mask and the
// BatchNorm-backward statistics of the pre-activation fused).  Dilations reach 31 (a 63 x 63 footprint): a halo tile as in the
// fused 3x3 loop would not fit in LDS, and these variants are in no benchmark configuration -- the im2col buffer costs 9x the
// activation's traffic and buys the tuned GEMM loops for every dilation.
// PSP_Pooling: k x k max pooling (k = 1, 2, 4, 8), bilinear resize back (align_corners = False) and their gradients.
#include "common.h"

namespace {

constexpr int TPB = 256;

template <typename T> __device__ __forceinline__ float ld(const void* p, long i) { return (float)((const T*)p)[i]; }
template <typename T> __device__ __forceinline__ void st(void* p, long i, float v) { ((T*)p)[i] = (T)v; }
template <typename T> __device__ __forceinline__ float rnd(float v) { return (float)(T)v; }

struct Slice { const void* p; int cs, co; };
struct MSlice { void* p; int cs, co; };

template <typename T>
__global__ void input_plain_kernel(const float* __restrict__ x, void* out, int n, int c, long hw, int out_cs, float a, float b) {
    const long total = (long)n * hw * out_cs;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % out_cs);
        const long pix = i / out_cs;
        const long img = pix / hw, q = pix % hw;
        st<T>(out, i, ci < c ? fmaf(x[(img * c + ci) * hw + q], a, b) : 0.f);
    }
}

// col[pix][t * cp + ci] = act(in[pix + off_t][ci]), zero outside the image and for ci >= c
template <typename T>
__global__ void im2col_dil_kernel(Slice in, int c, const float* __restrict__ scale, const float* __restrict__ shift, void* col, int cp,
                                  int n, int h, int w, int dil) {
    const long total = (long)n * h * w * 9 * cp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cp);
        long r = i / cp;
        const int t = (int)(r % 9);
        const long pix = r / 9;
        const int x = (int)(pix % w), y = (int)((pix / w) % h);
        const long img = pix / ((long)w * h);
        const int sy = y + (t / 3 - 1) * dil, sx = x + (t % 3 - 1) * dil;
        float v = 0.f;
        if (ci < c && sy >= 0 && sy < h && sx >= 0 && sx < w) {
            v = ld<T>(in.p, ((img * h + sy) * w + sx) * in.cs + in.co + ci);
            if (scale) v = rnd<T>(fmaxf(fmaf(v, scale[ci], shift[ci]), 0.f));
        }
        st<T>(col, i, v);
    }
}

// channel-fixed-per-thread walk: thread (tid % CW) owns channel cbase + tid % CW, rows tid / CW of the block's pixel group
struct Walk {
    int cw, rows;       // channels per pass (power of two <= 256), pixel rows per block
};
__host__ __device__ inline Walk make_walk(int c) {
    int cw = 1;
    while (cw < c && cw < TPB) cw <<= 1;
    return Walk{cw, TPB / cw};
}

// per-block partial sums (NS sums per channel) -> f64 atomics into the striped destination [stripe][NS * c]
template <int NS>
__device__ __forceinline__ void flush(const Walk& wk, int cbase, int c, float (&acc)[NS], double* stats, float* lds) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int s = 0; s < NS; ++s) lds[s * TPB + tid] = acc[s];
    __syncthreads();
    if (tid < wk.cw && cbase + tid < c) {
        double* dst = stats + (long)(blockIdx.x % PSSR_STAT_STRIPES) * NS * c;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float t = 0.f;
            for (int k = tid; k < TPB; k += wk.cw) t += lds[s * TPB + k];
            stat_add(dst + (long)s * c + cbase + tid, (long)PSSR_STAT_STRIPES * NS * c, t);
        }
    }
    __syncthreads();
}

// out[pix][ci] = sum_t dcol[pix - off_t][t * cp + ci], then (optional) the ReLU mask of the pre-activation y*scale+shift and the
// BatchNorm-backward statistics [sum g, sum g * xhat(y)]
template <typename T>
__global__ void col2im_dil_kernel(const void* dcol, int cp, MSlice out, int c, int n, int h, int w, int dil, Slice y,
                                  const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
                                  const float* __restrict__ invstd, double* stats) {
    __shared__ float lds[2 * TPB];
    const Walk wk = make_walk(c);
    const int tid = threadIdx.x, lc = tid % wk.cw, lr = tid / wk.cw;
    const long npix = (long)n * h * w;
    for (int cbase = 0; cbase < c; cbase += wk.cw) {
        const int ci = cbase + lc;
        float acc[2] = {0.f, 0.f};
        if (ci < c) {
            const float sc = y.p ? scale[ci] : 0.f, sh = y.p ? shift[ci] : 0.f;
            const float mu = stats ? mean[ci] : 0.f, is = stats ? invstd[ci] : 0.f;
            for (long pix = (long)blockIdx.x * wk.rows + lr; pix < npix; pix += (long)gridDim.x * wk.rows) {
                const int x = (int)(pix % w), yy = (int)((pix / w) % h);
                const long img = pix / ((long)w * h);
                float g = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int sy = yy - (t / 3 - 1) * dil, sx = x - (t % 3 - 1) * dil;
                    if (sy >= 0 && sy < h && sx >= 0 && sx < w) g += ld<T>(dcol, (((img * h + sy) * w + sx) * 9 + t) * cp + ci);
                }
                if (y.p) {
                    const float yv = ld<T>(y.p, pix * y.cs + y.co + ci);
                    g = fmaf(yv, sc, sh) > 0.f ? g : 0.f;
                    g = rnd<T>(g);
                    acc[0] += g;
                    acc[1] += g * (yv - mu) * is;
                }
                st<T>(out.p, pix * out.cs + out.co + ci, g);
            }
        }
        if (stats) flush<2>(wk, cbase, c, acc, stats, lds);
    }
}

// stats[0:c] += sum x, stats[c:2c] += sum x^2 over the pixels of an NHWC slice
template <typename T>
__global__ void channel_stats_kernel(Slice x, int c, long npix, double* stats) {
    __shared__ float lds[2 * TPB];
    const Walk wk = make_walk(c);
    const int tid = threadIdx.x, lc = tid % wk.cw, lr = tid / wk.cw;
    for (int cbase = 0; cbase < c; cbase += wk.cw) {
        const int ci = cbase + lc;
        float acc[2] = {0.f, 0.f};
        if (ci < c)
            for (long pix = (long)blockIdx.x * wk.rows + lr; pix < npix; pix += (long)gridDim.x * wk.rows) {
                const float v = ld<T>(x.p, pix * x.cs + x.co + ci);
                acc[0] += v; acc[1] += v * v;
            }
        flush<2>(wk, cbase, c, acc, stats, lds);
    }
}

struct SumArgs { const void* p[8]; int cs[8], co[8]; int n; };

template <typename T>
__global__ void sum_relu_kernel(SumArgs a, MSlice out, long npix, int c, int relu) {
    const long total = npix * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        const long pix = i / c;
        float v = 0.f;
        for (int k = 0; k < a.n; ++k) v += ld<T>(a.p[k], pix * a.cs[k] + a.co[k] + ci);
        if (relu) v = fmaxf(v, 0.f);
        st<T>(out.p, pix * out.cs + out.co + ci, v);
    }
}

// dz = dout where out > 0 (mode 0), or out = relu(x * scale + shift) (mode 1: `a` is x, `b` unused)
template <typename T>
__global__ void mask_or_affine_kernel(Slice a, Slice b, const float* __restrict__ scale, const float* __restrict__ shift, MSlice out, long npix, int c,
                                      int mode) {
    const long total = npix * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        const long pix = i / c;
        const float av = ld<T>(a.p, pix * a.cs + a.co + ci);
        float v;
        if (mode == 0) v = ld<T>(b.p, pix * b.cs + b.co + ci) > 0.f ? av : 0.f;
        else v = fmaxf(fmaf(av, scale[ci], shift[ci]), 0.f);
        st<T>(out.p, pix * out.cs + out.co + ci, v);
    }
}

// F.max_pool2d(x, k): floor mode, windows [k*oy, k*oy + k)
template <typename T>
__global__ void maxpool_k_kernel(Slice in, MSlice out, int n, int h, int w, int c, int k) {
    const int ho = h / k, wo = w / k;
    const long total = (long)n * ho * wo * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int ox = (int)(r % wo); r /= wo;
        const int oy = (int)(r % ho);
        const long img = r / ho;
        float m = -__builtin_inff();
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) m = fmaxf(m, ld<T>(in.p, ((img * h + oy * k + dy) * w + ox * k + dx) * in.cs + in.co + ci));
        st<T>(out.p, ((img * ho + oy) * wo + ox) * out.cs + out.co + ci, m);
    }
}

// gradient of the above: each input pixel belongs to at most one window; the gradient goes to the window's first maximum in
// row-major scan order (torch's choice); pixels outside every window (h % k rows / w % k columns) get zero
template <typename T>
__global__ void maxpool_k_bwd_kernel(Slice act, Slice dpool, MSlice dx, int n, int h, int w, int c, int k) {
    const int ho = h / k, wo = w / k;
    const long total = (long)n * h * w * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const long img = r / h;
        const int oy = y / k, ox = x / k;
        float g = 0.f;
        if (oy < ho && ox < wo) {
            const float mine = ld<T>(act.p, ((img * h + y) * w + x) * act.cs + act.co + ci);
            bool first = true;
            for (int dy = 0; dy < k && first; ++dy)
                for (int dx_ = 0; dx_ < k; ++dx_) {
                    const int yy = oy * k + dy, xx = ox * k + dx_;
                    const float v = ld<T>(act.p, ((img * h + yy) * w + xx) * act.cs + act.co + ci);
                    if (v > mine || (v == mine && (yy < y || (yy == y && xx < x)))) { first = false; break; }
                }
            if (first) g = ld<T>(dpool.p, ((img * ho + oy) * wo + ox) * dpool.cs + dpool.co + ci);
        }
        st<T>(dx.p, ((img * h + y) * w + x) * dx.cs + dx.co + ci, g);
    }
}

// torch's bilinear source index for align_corners = False: src = max((dst + 0.5) * in / out - 0.5, 0)
__device__ __forceinline__ void bil_src(int dst, int in_size, int out_size, int& i0, int& i1, float& l1) {
    const float scale = (float)in_size / (float)out_size;
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
    l1 = src - (float)i0;
}

template <typename T>
__global__ void bilinear_up_kernel(Slice in, MSlice out, int n, int hs, int ws, int h, int w, int c) {
    const long total = (long)n * h * w * c;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int x = (int)(r % w); r /= w;
        const int y = (int)(r % h);
        const long img = r / h;
        int y0, y1, x0, x1; float ly, lx;
        bil_src(y, hs, h, y0, y1, ly); bil_src(x, ws, w, x0, x1, lx);
        auto at = [&](int yy, int xx) { return ld<T>(in.p, ((img * hs + yy) * ws + xx) * in.cs + in.co + ci); };
        const float v = (1.f - ly) * ((1.f - lx) * at(y0, x0) + lx * at(y0, x1)) + ly * ((1.f - lx) * at(y1, x0) + lx * at(y1, x1));
        st<T>(out.p, ((img * h + y) * w + x) * out.cs + out.co + ci, v);
    }
}

// gradient in gather form: a source pixel collects from the destination pixels whose 2 x 2 footprint contains it
template <typename T>
__global__ void bilinear_up_bwd_kernel(Slice dout, MSlice din, int n, int hs, int ws, int h, int w, int c) {
    const long total = (long)n * hs * ws * c;
    const float ry = (float)h / (float)hs, rx = (float)w / (float)ws;      // destination pixels per source pixel (not always an integer)
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        long r = i / c;
        const int qx = (int)(r % ws); r /= ws;
        const int qy = (int)(r % hs);
        const long img = r / hs;
        float g = 0.f;
        // source index of destination y is (y + 0.5) / ry - 0.5: it lies in [qy - 1, qy + 1) for y in [(qy - 0.5) ry - 0.5, (qy + 1.5) ry - 0.5)
        int ylo = (int)floorf(((float)qy - 1.f) * ry) - 1, yhi = (int)ceilf(((float)qy + 2.f) * ry) + 1;
        int xlo = (int)floorf(((float)qx - 1.f) * rx) - 1, xhi = (int)ceilf(((float)qx + 2.f) * rx) + 1;
        ylo = ylo < 0 ? 0 : ylo; yhi = yhi > h ? h : yhi; xlo = xlo < 0 ? 0 : xlo; xhi = xhi > w ? w : xhi;
        for (int y = ylo; y < yhi; ++y) {
            int y0, y1; float ly;
            bil_src(y, hs, h, y0, y1, ly);
            const float wy = (y0 == qy ? 1.f - ly : 0.f) + (y1 == qy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int x = xlo; x < xhi; ++x) {
                int x0, x1; float lx;
                bil_src(x, ws, w, x0, x1, lx);
                const float wx = (x0 == qx ? 1.f - lx : 0.f) + (x1 == qx ? lx : 0.f);
                if (wx != 0.f) g += wy * wx * ld<T>(dout.p, ((img * h + y) * w + x) * dout.cs + dout.co + ci);
            }
        }
        st<T>(din.p, ((img * hs + qy) * ws + qx) * din.cs + din.co + ci, g);
    }
}

inline unsigned grid_for(long total) {
    long b = (total + TPB - 1) / TPB;
    return (unsigned)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

#define PSSR_DT_SWITCH(dtype, CALL)                                                           \
    switch (dtype) {                                                                          \
    case PSSR_F32: { using T = float; CALL; } break;                                          \
    case PSSR_BF16: { using T = bf16_t; CALL; } break;                                        \
    case PSSR_F16: { using T = f16_t; CALL; } break;                                          \
    default: pssr_set_error("bad dtype %d", dtype); return PSSR_ERR_ARG;                      \
    }

extern "C" int pssr_input_plain(const float* x_nchw, void* out, int n, int c, int h, int w, int out_cs, float pre_scale, float pre_shift,
                                int dtype, pssr_stream_t s) {
    PSSR_CHECK(x_nchw && out && n > 0 && c > 0 && h > 0 && w > 0 && out_cs >= c, PSSR_ERR_ARG, "input_plain: bad args");
    const long hw = (long)h * w;
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(input_plain_kernel<T>, dim3(grid_for((long)n * hw * out_cs)), dim3(TPB), 0, (hipStream_t)s, x_nchw, out, n, c, hw,
                                             out_cs, pre_scale, pre_shift))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_im2col_dil(const void* in, int in_cs, int in_co, int c, const float* scale, const float* shift, void* col, int cp,
                               int n, int h, int w, int dil, int dtype, pssr_stream_t s) {
    PSSR_CHECK(in && col && c > 0 && cp >= c && n > 0 && h > 0 && w > 0 && dil >= 1 && in_co + c <= in_cs, PSSR_ERR_ARG, "im2col_dil: bad args");
    PSSR_CHECK((scale == nullptr) == (shift == nullptr), PSSR_ERR_ARG, "im2col_dil: scale and shift go together");
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(im2col_dil_kernel<T>, dim3(grid_for((long)n * h * w * 9 * cp)), dim3(TPB), 0, (hipStream_t)s,
                                             Slice{in, in_cs, in_co}, c, scale, shift, col, cp, n, h, w, dil))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_col2im_dil(const void* dcol, int cp, void* out, int out_cs, int out_co, int c, int n, int h, int w, int dil,
                               const void* y, int y_cs, int y_co, const float* scale, const float* shift, const float* mean,
                               const float* invstd, double* stats, int dtype, pssr_stream_t s) {
    PSSR_CHECK(dcol && out && c > 0 && cp >= c && n > 0 && h > 0 && w > 0 && dil >= 1 && out_co + c <= out_cs, PSSR_ERR_ARG, "col2im_dil: bad args");
    PSSR_CHECK(!y || (scale && shift), PSSR_ERR_ARG, "col2im_dil: the ReLU mask needs scale / shift");
    PSSR_CHECK(!stats || (y && mean && invstd), PSSR_ERR_ARG, "col2im_dil: statistics need the pre-activation, mean and invstd");
    const Walk wk = make_walk(c);
    long blocks = ((long)n * h * w + wk.rows - 1) / wk.rows;
    if (blocks > 2048) blocks = 2048;
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(col2im_dil_kernel<T>, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)s, dcol, cp,
                                             MSlice{out, out_cs, out_co}, c, n, h, w, dil, Slice{y, y_cs, y_co}, scale, shift, mean, invstd, stats))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_channel_stats_nhwc(const void* x, int cs, int co, int c, int64_t npix, double* stats, int dtype, pssr_stream_t s) {
    PSSR_CHECK(x && stats && c > 0 && npix > 0 && co + c <= cs, PSSR_ERR_ARG, "channel_stats_nhwc: bad args");
    const Walk wk = make_walk(c);
    long blocks = (npix + wk.rows - 1) / wk.rows;
    if (blocks > 2048) blocks = 2048;
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(channel_stats_kernel<T>, dim3((unsigned)blocks), dim3(TPB), 0, (hipStream_t)s, Slice{x, cs, co}, c, (long)npix, stats))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_sum_relu(const void* const* ins, const int* in_cs, const int* in_co, int n_in, void* out, int out_cs, int out_co,
                             int64_t npix, int c, int relu, int dtype, pssr_stream_t s) {
    PSSR_CHECK(ins && in_cs && in_co && out && n_in >= 1 && n_in <= 8 && npix > 0 && c > 0, PSSR_ERR_ARG, "sum_relu: 1..8 inputs");
    SumArgs a;
    a.n = n_in;
    for (int k = 0; k < n_in; ++k) { a.p[k] = ins[k]; a.cs[k] = in_cs[k]; a.co[k] = in_co[k]; }
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(sum_relu_kernel<T>, dim3(grid_for((long)npix * c)), dim3(TPB), 0, (hipStream_t)s, a, MSlice{out, out_cs, out_co},
                                             (long)npix, c, relu))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_relu_mask(const void* dout, int do_cs, int do_co, const void* out, int o_cs, int o_co, void* dz, int dz_cs, int dz_co,
                              int64_t npix, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(dout && out && dz && npix > 0 && c > 0, PSSR_ERR_ARG, "relu_mask: bad args");
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(mask_or_affine_kernel<T>, dim3(grid_for((long)npix * c)), dim3(TPB), 0, (hipStream_t)s, Slice{dout, do_cs, do_co},
                                             Slice{out, o_cs, o_co}, (const float*)nullptr, (const float*)nullptr, MSlice{dz, dz_cs, dz_co}, (long)npix, c, 0))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_affine_relu(const void* x, int cs, int co, const float* scale, const float* shift, void* out, int out_cs, int out_co,
                                int64_t npix, int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(x && scale && shift && out && npix > 0 && c > 0, PSSR_ERR_ARG, "affine_relu: bad args");
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(mask_or_affine_kernel<T>, dim3(grid_for((long)npix * c)), dim3(TPB), 0, (hipStream_t)s, Slice{x, cs, co},
                                             Slice{nullptr, 0, 0}, scale, shift, MSlice{out, out_cs, out_co}, (long)npix, c, 1))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_maxpool_k(const void* in, int in_cs, int in_co, void* out, int out_cs, int out_co, int n, int h, int w, int c, int k,
                              int dtype, pssr_stream_t s) {
    PSSR_CHECK(in && out && n > 0 && c > 0 && k >= 1 && h >= k && w >= k, PSSR_ERR_ARG, "maxpool_k: bad args (k=%d on %dx%d)", k, h, w);
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(maxpool_k_kernel<T>, dim3(grid_for((long)n * (h / k) * (w / k) * c)), dim3(TPB), 0, (hipStream_t)s,
                                             Slice{in, in_cs, in_co}, MSlice{out, out_cs, out_co}, n, h, w, c, k))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_maxpool_k_bwd(const void* act, int act_cs, int act_co, const void* dpool, int dp_cs, int dp_co, void* dx, int dx_cs, int dx_co,
                                  int n, int h, int w, int c, int k, int dtype, pssr_stream_t s) {
    PSSR_CHECK(act && dpool && dx && n > 0 && c > 0 && k >= 1 && h >= k && w >= k, PSSR_ERR_ARG, "maxpool_k_bwd: bad args");
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(maxpool_k_bwd_kernel<T>, dim3(grid_for((long)n * h * w * c)), dim3(TPB), 0, (hipStream_t)s, Slice{act, act_cs, act_co},
                                             Slice{dpool, dp_cs, dp_co}, MSlice{dx, dx_cs, dx_co}, n, h, w, c, k))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_bilinear_up(const void* in, int in_cs, int in_co, void* out, int out_cs, int out_co, int n, int hs, int ws, int h, int w,
                                int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(in && out && n > 0 && c > 0 && hs > 0 && ws > 0 && h >= hs && w >= ws, PSSR_ERR_ARG, "bilinear_up: bad args");
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(bilinear_up_kernel<T>, dim3(grid_for((long)n * h * w * c)), dim3(TPB), 0, (hipStream_t)s, Slice{in, in_cs, in_co},
                                             MSlice{out, out_cs, out_co}, n, hs, ws, h, w, c))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_bilinear_up_bwd(const void* dout, int do_cs, int do_co, void* din, int di_cs, int di_co, int n, int hs, int ws, int h, int w,
                                    int c, int dtype, pssr_stream_t s) {
    PSSR_CHECK(dout && din && n > 0 && c > 0 && hs > 0 && ws > 0 && h >= hs && w >= ws, PSSR_ERR_ARG, "bilinear_up_bwd: bad args");
    PSSR_DT_SWITCH(dtype, hipLaunchKernelGGL(bilinear_up_bwd_kernel<T>, dim3(grid_for((long)n * hs * ws * c)), dim3(TPB), 0, (hipStream_t)s, Slice{dout, do_cs, do_co},
                                             MSlice{din, di_cs, di_co}, n, hs, ws, h, w, c))
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}
