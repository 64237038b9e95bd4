// normalize_preds (pssr/util.py:139-191) for uint8 image pairs of equal size, bit-exact with the reference's numpy float32 /
// float64 arithmetic.  Both inputs are bytes, so every per-pixel quantity is a 256-entry table (normalised value, mean-removed
// value, rescaled value, output byte); what has to be reproduced exactly is
//   * np.percentile's float32 virtual index + lerp on the order statistics (from a histogram),
//   * numpy's float32 summation for np.mean / np.var of the float32 images, in pixel order: buffer-sized pieces of 8192
//     elements accumulated in order, each summed PAIRWISE (blocks of <= 128 elements with 8 accumulators, halves rounded down
//     to a multiple of 8),
// while the float64 parts (np.cov, the prediction's rescaling) are insensitive to summation order at uint8 resolution.
// One workgroup per image.
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int NPY_BUF = 8192;      // numpy's default ufunc buffer size in elements: the granularity of its reductions

struct Leaf { int start, len; };

// numpy pairwise_sum: split until n <= 128 (explicit stacks: device recursion depth is not something to rely on)
__device__ void build_leaves(int lo0, int n0, Leaf* leaves, int& count) {
    int st_lo[48], st_n[48], sp = 0;
    st_lo[sp] = lo0; st_n[sp] = n0; ++sp;
    while (sp) {
        --sp;
        const int lo = st_lo[sp], n = st_n[sp];
        if (n <= 128) { leaves[count].start = lo; leaves[count].len = n; ++count; continue; }
        int n2 = n / 2;
        n2 -= n2 % 8;
        st_lo[sp] = lo + n2; st_n[sp] = n - n2; ++sp;       // right half (popped after the left one)
        st_lo[sp] = lo; st_n[sp] = n2; ++sp;
    }
}
__device__ float combine_leaves(int n_total, const float* sums, int& idx) {
    int fn[48], stage[48], sp = 0;
    float left[48], ret = 0.f;
    fn[0] = n_total; stage[0] = 0; sp = 1;
    while (sp) {
        const int t = sp - 1;
        if (fn[t] <= 128) { ret = sums[idx++]; --sp; continue; }
        int n2 = fn[t] / 2;
        n2 -= n2 % 8;
        if (stage[t] == 0) { stage[t] = 1; fn[sp] = n2; stage[sp] = 0; ++sp; }
        else if (stage[t] == 1) { left[t] = ret; stage[t] = 2; fn[sp] = fn[t] - n2; stage[sp] = 0; ++sp; }
        else { ret = __fadd_rn(left[t], ret); --sp; }
    }
    return ret;
}
template <class F>
__device__ float leaf_sum(int start, int n, F val) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, val(start + i));
        return res;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = val(start + j);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], val(start + i + j));
    float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __fadd_rn(res, val(start + i));
    return res;
}
// whole-image float32 pairwise sum of val(i), i in [0, n): result broadcast through *shared_out
template <class F>
__device__ float pairwise_f32(int n, const Leaf* leaves, int n_leaves, float* sums, F val, float* shared_out) {
    for (int l = threadIdx.x; l < n_leaves; l += NT) sums[l] = leaf_sum(leaves[l].start, leaves[l].len, val);
    __threadfence_block();
    __syncthreads();
    if (threadIdx.x == 0) {
        // numpy's reduction walks the array in buffer-sized pieces (np.getbufsize() = 8192 elements): result = 0, then
        // result += pairwise(piece) for each piece in order
        int idx = 0;
        float acc = 0.f;
        for (int c0 = 0; c0 < n; c0 += NPY_BUF) acc = __fadd_rn(acc, combine_leaves(n - c0 < NPY_BUF ? n - c0 : NPY_BUF, sums, idx));
        *shared_out = acc;
    }
    __syncthreads();
    const float r = *shared_out;
    __syncthreads();
    return r;
}
__device__ double block_sum_f64(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    const double r = red[0];
    __syncthreads();
    return r;
}
// np.percentile(float32 array, p) with method "linear": float32 quantile, virtual index, gamma and lerp
__device__ float percentile_f32(const int* cum /* inclusive cumulative histogram */, int n, float p) {
    const float q = __fdiv_rn(p, 100.f);
    const float vi = __fmul_rn((float)(n - 1), q);          // numpy's "linear" method: (n - 1) * q in the array's dtype
    long lo, hi;
    if (vi >= (float)(n - 1)) { lo = hi = n - 1; }
    else if (vi < 0.f) { lo = hi = 0; }
    else { lo = (long)floorf(vi); hi = lo + 1; }
    auto order_stat = [&](long k) { int v = 0; while (cum[v] <= k) ++v; return (float)v; };
    const float a = order_stat(lo), b = order_stat(hi);
    const float t = __fsub_rn(vi, floorf(vi));
    const float d = __fsub_rn(b, a);
    if (t >= 0.5f) return __fsub_rn(b, __fmul_rn(d, __fsub_rn(1.f, t)));
    return __fadd_rn(a, __fmul_rn(d, t));
}

// `n_hat` != n (a prediction of another size, pssr/util.py:176-179): the covariance of the RESIZED prediction with the ground truth
// arrives as per-workgroup partial sums [sum r, sum r * hr] of resize_cov_sums_kernel (`cov_part`, `n_part` pairs per image); every
// other quantity of the prediction (mean, variance, the rescaling mean) runs over its own n_hat pixels.
__global__ __launch_bounds__(NT) void normalize_preds_kernel(const uint8_t* __restrict__ hr_all, const uint8_t* __restrict__ hat_all,
                                                             uint8_t* __restrict__ out_hr_all, uint8_t* __restrict__ out_hat_all, int n, int n_hat,
                                                             float pmin, float pmax, char* __restrict__ ws_all, long ws_per_image,
                                                             const double* __restrict__ cov_part, int n_part) {
    __shared__ int hist[256], hist2[256], cum[256];
    __shared__ float hn[256], hn2[256], a_tab[256], hat2[256];
    __shared__ double b_tab[256];
    __shared__ unsigned char out_a[256], out_b[256];
    __shared__ double red[NT];
    __shared__ float bc;
    __shared__ int n_leaves_s;
    const uint8_t* hr = hr_all + (long)blockIdx.x * n;
    const uint8_t* hat = hat_all + (long)blockIdx.x * n_hat;
    const int n_big = n > n_hat ? n : n_hat;
    Leaf* leaves = (Leaf*)(ws_all + (long)blockIdx.x * ws_per_image);
    Leaf* leaves_h = leaves + (n_big / 32 + 16);                       // leaves of an n_hat-element reduction (== leaves when n_hat == n)
    float* sums = (float*)(leaves_h + (n_big / 32 + 16));
    __shared__ int n_leaves_h_s;
    const int tid = threadIdx.x;
    hist[tid] = 0; hist2[tid] = 0;
    if (tid == 0) {
        int c = 0;
        for (int c0 = 0; c0 < n; c0 += NPY_BUF) build_leaves(c0, n - c0 < NPY_BUF ? n - c0 : NPY_BUF, leaves, c);
        n_leaves_s = c;
        c = 0;
        for (int c0 = 0; c0 < n_hat; c0 += NPY_BUF) build_leaves(c0, n_hat - c0 < NPY_BUF ? n_hat - c0 : NPY_BUF, leaves_h, c);
        n_leaves_h_s = c;
    }
    __syncthreads();
    const int n_leaves = n_leaves_s, n_leaves_h = n_leaves_h_s;
    for (int i = tid; i < n; i += NT) atomicAdd(&hist[hr[i]], 1);
    for (int i = tid; i < n_hat; i += NT) atomicAdd(&hist2[hat[i]], 1);
    __syncthreads();
    if (tid == 0) { int c = 0; for (int v = 0; v < 256; ++v) { c += hist[v]; cum[v] = c; } }
    __syncthreads();
    // ---- hr side, all float32 (pssr/util.py:165-183)
    const float base_max = percentile_f32(cum, n, pmax);
    const float base_mean = __fdiv_rn(pairwise_f32(n, leaves, n_leaves, sums, [&](int i) { return (float)hr[i]; }, &bc), (float)n);
    const float x_min = percentile_f32(cum, n, pmin), x_max = base_max;
    const float denom = __fadd_rn(__fsub_rn(x_max, x_min), 1e-20f);
    hn[tid] = __fdiv_rn(__fsub_rn((float)tid, x_min), denom);
    __syncthreads();
    const float mean_hn = __fdiv_rn(pairwise_f32(n, leaves, n_leaves, sums, [&](int i) { return hn[hr[i]]; }, &bc), (float)n);
    hn2[tid] = __fsub_rn(hn[tid], mean_hn);
    const float mean_hat = __fdiv_rn(pairwise_f32(n_hat, leaves_h, n_leaves_h, sums, [&](int i) { return (float)hat[i]; }, &bc), (float)n_hat);
    hat2[tid] = __fsub_rn((float)tid, mean_hat);
    __syncthreads();
    int vmin = 0;
    while (hist[vmin] == 0) ++vmin;
    const float mn = hn2[vmin];                                        // hr_norm.min(): the map x -> hn2 is monotone
    // np.var(hat2) in float32: mean (pairwise), deviations, squares, pairwise sum / n
    const float m2 = __fdiv_rn(pairwise_f32(n_hat, leaves_h, n_leaves_h, sums, [&](int i) { return hat2[hat[i]]; }, &bc), (float)n_hat);
    const float var_hat = __fdiv_rn(pairwise_f32(n_hat, leaves_h, n_leaves_h, sums, [&](int i) { const float d = __fsub_rn(hat2[hat[i]], m2); return __fmul_rn(d, d); }, &bc), (float)n_hat);
    double cov;
    if (cov_part == nullptr) {
        // np.cov(hat2, hn2)[0, 1] in float64 (order-insensitive at this resolution)
        double s_h = 0.0, s_n = 0.0;
        for (int i = tid; i < n; i += NT) { s_h += (double)hat2[hat[i]]; s_n += (double)hn2[hr[i]]; }
        const double avg_h = block_sum_f64(s_h, red) / n, avg_n = block_sum_f64(s_n, red) / n;
        double s_c = 0.0;
        for (int i = tid; i < n; i += NT) s_c += ((double)hat2[hat[i]] - avg_h) * ((double)hn2[hr[i]] - avg_n);
        cov = block_sum_f64(s_c, red) / (double)(n - 1);
    } else {
        // np.cov(resize(hat2), hn2)[0, 1]: the resize is linear with weights summing to one and hn2 is affine in the ground-truth byte
        // (hn2 = (hr - x_min) / denom - mean), so cov = [sum r*hr - sum r * sum hr / n] / (n - 1) / denom with r = resize(hat)
        const double* part = cov_part + (long)blockIdx.x * n_part * 2;
        double s_r = 0.0, s_rh = 0.0;
        for (int i = tid; i < n_part; i += NT) { s_r += part[2 * i]; s_rh += part[2 * i + 1]; }
        s_r = block_sum_f64(s_r, red);
        s_rh = block_sum_f64(s_rh, red);
        double s_hr = 0.0;
        s_hr = block_sum_f64((double)tid * (double)hist[tid], red);
        cov = (s_rh - s_r * s_hr / (double)n) / (double)(n - 1) / (double)denom;
    }
    const double amp = cov / (double)var_hat;
    // ---- rescale to the initial intensity (pssr/util.py:181-184)
    a_tab[tid] = __fmul_rn(__fsub_rn(hn2[tid], mn), base_max);
    b_tab[tid] = (amp * (double)hat2[tid] - (double)mn) * (double)base_max;
    __syncthreads();
    const float a_mean = __fdiv_rn(pairwise_f32(n, leaves, n_leaves, sums, [&](int i) { return a_tab[hr[i]]; }, &bc), (float)n);
    double s_b = 0.0;
    for (int i = tid; i < n_hat; i += NT) s_b += b_tab[hat[i]];
    const double b_mean = block_sum_f64(s_b, red) / n_hat;
    const float a_div = __fdiv_rn(a_mean, base_mean);
    const double b_div = b_mean / (double)base_mean;
    {
        float a = __fdiv_rn(a_tab[tid], a_div);
        a = a < 0.f ? 0.f : (a > 255.f ? 255.f : a);
        out_a[tid] = (unsigned char)a;
        double b = b_tab[tid] / b_div;
        b = b < 0.0 ? 0.0 : (b > 255.0 ? 255.0 : b);
        out_b[tid] = (unsigned char)b;
    }
    __syncthreads();
    uint8_t* out_hr = out_hr_all + (long)blockIdx.x * n;
    uint8_t* out_hat = out_hat_all + (long)blockIdx.x * n_hat;
    for (int i = tid; i < n; i += NT) out_hr[i] = out_a[hr[i]];
    for (int i = tid; i < n_hat; i += NT) out_hat[i] = out_b[hat[i]];
}

// ---- skimage.transform.resize(prediction, ground-truth shape) of pssr/util.py:179, restated in oracle/metrics_ref.py:resize_restated
// (scikit-image >= 0.19 = scipy.ndimage: Gaussian pre-filter along shrinking axes, mirror boundary, then linear interpolation at
// (i + 0.5) * in / out - 0.5 with mirrored indices; float64 arithmetic rounded to float32 per stage, as scipy does for float32 images)
__device__ __forceinline__ int mirror_idx(int i, int n) {
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    i = (i < 0 ? -i : i) % period;
    return i >= n ? period - i : i;
}
template <typename S>
__global__ void resize_gauss_axis_kernel(const S* __restrict__ in_all, float* __restrict__ out_all, int h, int w, double sigma, int axis) {
    const S* in = in_all + (long)blockIdx.y * h * w;
    float* out = out_all + (long)blockIdx.y * h * w;
    const int r = (int)(4.0 * sigma + 0.5);
    double wsum = 0.0;
    for (int t = -r; t <= r; ++t) wsum += exp(-0.5 / (sigma * sigma) * (double)t * t);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)h * w; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)(i / w);
        double acc = 0.0;
        for (int t = -r; t <= r; ++t) {
            const double wt = exp(-0.5 / (sigma * sigma) * (double)t * t) / wsum;
            const long j = axis == 0 ? (long)mirror_idx(y + t, h) * w + x : (long)y * w + mirror_idx(x + t, w);
            acc += wt * (double)in[j];
        }
        out[i] = (float)acc;
    }
}
// per-workgroup partial sums [sum r, sum r * hr] over the ground-truth grid, r = float32(interpolated prediction)
template <typename S>
__global__ __launch_bounds__(NT) void resize_cov_sums_kernel(const S* __restrict__ src_all, int h, int w, const uint8_t* __restrict__ hr_all, int H,
                                                             int W, double* __restrict__ part_all) {
    __shared__ double red[NT];
    const S* src = src_all + (long)blockIdx.y * h * w;
    const uint8_t* hr = hr_all + (long)blockIdx.y * H * W;
    const double fy = (double)h / (double)H, fx = (double)w / (double)W;
    double s_r = 0.0, s_rh = 0.0;
    for (long i = blockIdx.x * (long)NT + threadIdx.x; i < (long)H * W; i += (long)gridDim.x * NT) {
        const int X = (int)(i % W), Y = (int)(i / W);
        const double cy = ((double)Y + 0.5) * fy - 0.5, cx = ((double)X + 0.5) * fx - 0.5;
        const double fly = floor(cy), flx = floor(cx);
        const double ty = cy - fly, tx = cx - flx;
        const int y0 = mirror_idx((int)fly, h), y1 = mirror_idx((int)fly + 1, h), x0 = mirror_idx((int)flx, w), x1 = mirror_idx((int)flx + 1, w);
        const double v = (1.0 - ty) * (1.0 - tx) * (double)src[(long)y0 * w + x0] + (1.0 - ty) * tx * (double)src[(long)y0 * w + x1]
                       + ty * (1.0 - tx) * (double)src[(long)y1 * w + x0] + ty * tx * (double)src[(long)y1 * w + x1];
        const double r = (double)(float)v;
        s_r += r;
        s_rh += r * (double)hr[i];
    }
    const double a = block_sum_f64(s_r, red), b = block_sum_f64(s_rh, red);
    if (threadIdx.x == 0) {
        double* part = part_all + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        part[0] = a; part[1] = b;
    }
}


// ---- PSNR / SSIM of uint8 image pairs (pssr/predict.py:199-203 -> skimage.metrics.peak_signal_noise_ratio /
// structural_similarity with data_range 255: uniform 7x7 window, K1 = .01, K2 = .03, sample covariance, mean over the interior
// that a full window covers).  Window sums of a, b, a^2, b^2, ab over bytes are exact in 32-bit integers, so the local
// statistics are computed from exact sums (the reference's float64 running sums round in the 16th digit) and only the SSIM
// expression itself is float64; the squared-difference sum is an exact 64-bit integer.  Per-tile float64 partials are written
// to the workspace and summed in a fixed order, so the result is reproducible bit for bit.
constexpr int MT = 32, MW = 7, MH = MT + MW - 1;      // 32x32 output pixels per workgroup, 38x38 input halo

__global__ void __launch_bounds__(NT) ssim_tiles_kernel(const uint8_t* __restrict__ hr_all, const uint8_t* __restrict__ hat_all,
                                                        double* __restrict__ partial, int h, int w, int tiles_x, int tiles) {
    __shared__ uint8_t A[MH][MH + 2], B[MH][MH + 2];
    __shared__ unsigned HS[5][MH][MT];
    __shared__ double red[NT / 64];
    const int tid = threadIdx.x, tile = blockIdx.x, img = blockIdx.y;
    const int ty0 = tile / tiles_x * MT, tx0 = tile % tiles_x * MT;      // top-left of the tile in interior (output) coordinates
    const uint8_t* hr = hr_all + (long)img * h * w;
    const uint8_t* hat = hat_all + (long)img * h * w;
    for (int i = tid; i < MH * MH; i += NT) {
        const int r = i / MH, c = i % MH, y = ty0 + r, x = tx0 + c;
        const bool in = y < h && x < w;
        A[r][c] = in ? hr[(long)y * w + x] : 0;
        B[r][c] = in ? hat[(long)y * w + x] : 0;
    }
    __syncthreads();
    for (int i = tid; i < MH * MT; i += NT) {
        const int r = i / MT, c = i % MT;
        unsigned sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
#pragma unroll
        for (int k = 0; k < MW; ++k) {
            const unsigned a = A[r][c + k], b = B[r][c + k];
            sa += a; sb += b; saa += a * a; sbb += b * b; sab += a * b;
        }
        HS[0][r][c] = sa; HS[1][r][c] = sb; HS[2][r][c] = saa; HS[3][r][c] = sbb; HS[4][r][c] = sab;
    }
    __syncthreads();
    const double inv = 1.0 / (MW * MW), cov_norm = (double)(MW * MW) / (MW * MW - 1);
    const double c1 = (0.01 * 255) * (0.01 * 255), c2 = (0.03 * 255) * (0.03 * 255);
    double acc = 0.0;
    for (int i = tid; i < MT * MT; i += NT) {
        const int r = i / MT, c = i % MT;
        if (ty0 + r >= h - (MW - 1) || tx0 + c >= w - (MW - 1)) continue;
        unsigned s[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < MW; ++k)
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q] += HS[q][r + k][c];
        const double ux = s[0] * inv, uy = s[1] * inv, uxx = s[2] * inv, uyy = s[3] * inv, uxy = s[4] * inv;
        const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
        const double a1 = 2 * ux * uy + c1, a2 = 2 * vxy + c2, b1 = ux * ux + uy * uy + c1, b2 = vx + vy + c2;
        acc += (a1 * a2) / (b1 * b2);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partial[(long)img * tiles + tile] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[img] = {sum of squared differences (exact), mean SSIM}; one workgroup per image, fixed summation order
__global__ void __launch_bounds__(NT) image_metrics_finish_kernel(const uint8_t* __restrict__ hr_all, const uint8_t* __restrict__ hat_all,
                                                                  const double* __restrict__ partial, double* __restrict__ out, long px,
                                                                  int tiles, long interior) {
    __shared__ unsigned long long sred[NT];
    __shared__ double dred[NT];
    const int tid = threadIdx.x, img = blockIdx.x;
    const uint8_t* hr = hr_all + img * px;
    const uint8_t* hat = hat_all + img * px;
    unsigned long long ssd = 0;
    for (long i = tid; i < px; i += NT) { const int d = (int)hr[i] - (int)hat[i]; ssd += (unsigned)(d * d); }
    double s = 0.0;
    for (int i = tid; i < tiles; i += NT) s += partial[(long)img * tiles + i];
    sred[tid] = ssd; dred[tid] = s;
    __syncthreads();
    for (int o = NT / 2; o; o >>= 1) {
        if (tid < o) { sred[tid] += sred[tid + o]; dred[tid] += dred[tid + o]; }
        __syncthreads();
    }
    if (tid == 0) { out[2 * img] = (double)sred[0]; out[2 * img + 1] = dred[0] / (double)interior; }
}

}  // namespace

extern "C" int64_t pssr_normalize_preds_workspace_bytes(int64_t pixels_per_image) {
    return ((pixels_per_image / 32 + 16) * (int64_t)(2 * sizeof(Leaf) + sizeof(float)) + 15) / 16 * 16;
}

extern "C" int pssr_normalize_preds_u8(const uint8_t* hr, const uint8_t* hr_hat, uint8_t* hr_norm, uint8_t* hr_hat_norm, int n_images,
                                       int64_t pixels_per_image, float pmin, float pmax, void* workspace, pssr_stream_t s) {
    PSSR_CHECK(hr && hr_hat && hr_norm && hr_hat_norm && workspace && n_images > 0, PSSR_ERR_ARG, "normalize_preds: null pointer / no images");
    PSSR_CHECK(pixels_per_image >= 2 && pixels_per_image < (1L << 24), PSSR_ERR_ARG, "normalize_preds: %ld pixels per image (2 .. 2^24 - 1)", (long)pixels_per_image);
    PSSR_CHECK(pmin >= 0.f && pmax <= 100.f && pmin <= pmax, PSSR_ERR_ARG, "normalize_preds: percentiles");
    hipLaunchKernelGGL(normalize_preds_kernel, dim3(n_images), dim3(NT), 0, (hipStream_t)s, hr, hr_hat, hr_norm, hr_hat_norm, (int)pixels_per_image,
                       (int)pixels_per_image, pmin, pmax, (char*)workspace, (long)pssr_normalize_preds_workspace_bytes(pixels_per_image), nullptr, 0);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

// ---- prediction and ground truth of different sizes (pssr/util.py:176-179)
constexpr int COV_PARTS = 256;
static inline int64_t nr_ws_norm(int64_t px_big) { return ((px_big / 32 + 16) * (int64_t)(2 * sizeof(Leaf) + sizeof(float)) + 15) / 16 * 16; }

extern "C" int64_t pssr_normalize_preds_resized_workspace_bytes(int n_images, int H, int W, int h, int w) {
    if (n_images <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return 0;
    const int64_t big = (int64_t)H * W > (int64_t)h * w ? (int64_t)H * W : (int64_t)h * w;
    return n_images * (nr_ws_norm(big) + (int64_t)COV_PARTS * 2 * sizeof(double) + 2 * (int64_t)h * w * sizeof(float));
}

extern "C" int pssr_normalize_preds_resized_u8(const uint8_t* hr, int H, int W, const uint8_t* hr_hat, int h, int w, uint8_t* hr_norm,
                                               uint8_t* hr_hat_norm, int n_images, float pmin, float pmax, void* workspace, pssr_stream_t s) {
    PSSR_CHECK(hr && hr_hat && hr_norm && hr_hat_norm && workspace && n_images > 0, PSSR_ERR_ARG, "normalize_preds_resized: null pointer / no images");
    PSSR_CHECK(H > 0 && W > 0 && h > 0 && w > 0 && (long)H * W >= 2 && (long)H * W < (1L << 24) && (long)h * w >= 2 && (long)h * w < (1L << 24), PSSR_ERR_ARG,
               "normalize_preds_resized: image sizes %dx%d / %dx%d (2 .. 2^24 - 1 pixels)", H, W, h, w);
    PSSR_CHECK(pmin >= 0.f && pmax <= 100.f && pmin <= pmax, PSSR_ERR_ARG, "normalize_preds_resized: percentiles");
    const int64_t big = (int64_t)H * W > (int64_t)h * w ? (int64_t)H * W : (int64_t)h * w;
    char* ws = (char*)workspace;
    char* ws_norm = ws;
    double* parts = (double*)(ws + n_images * nr_ws_norm(big));
    float* fa = (float*)(parts + (int64_t)n_images * COV_PARTS * 2);
    float* fb = fa + (int64_t)n_images * h * w;
    hipStream_t st = (hipStream_t)s;
    const double sy = ((double)h / H - 1.0) / 2.0, sx = ((double)w / W - 1.0) / 2.0;
    const bool shrink = H < h || W < w;
    const float* filtered = nullptr;
    const dim3 fgrid((unsigned)(((long)h * w + 255) / 256 < 1024 ? ((long)h * w + 255) / 256 : 1024), n_images);
    if (shrink) {
        // axis 0 first, then axis 1 (scipy.ndimage.gaussian_filter); an axis that does not shrink has sigma 0 and is skipped
        if (sy > 0.0) {
            hipLaunchKernelGGL(resize_gauss_axis_kernel<uint8_t>, fgrid, dim3(256), 0, st, hr_hat, fa, h, w, sy, 0);
            filtered = fa;
            if (sx > 0.0) { hipLaunchKernelGGL(resize_gauss_axis_kernel<float>, fgrid, dim3(256), 0, st, (const float*)fa, fb, h, w, sx, 1); filtered = fb; }
        } else {
            hipLaunchKernelGGL(resize_gauss_axis_kernel<uint8_t>, fgrid, dim3(256), 0, st, hr_hat, fa, h, w, sx, 1);
            filtered = fa;
        }
    }
    if (filtered) hipLaunchKernelGGL(resize_cov_sums_kernel<float>, dim3(COV_PARTS, n_images), dim3(NT), 0, st, filtered, h, w, hr, H, W, parts);
    else hipLaunchKernelGGL(resize_cov_sums_kernel<uint8_t>, dim3(COV_PARTS, n_images), dim3(NT), 0, st, hr_hat, h, w, hr, H, W, parts);
    hipLaunchKernelGGL(normalize_preds_kernel, dim3(n_images), dim3(NT), 0, st, hr, hr_hat, hr_norm, hr_hat_norm, H * W, h * w, pmin, pmax, ws_norm,
                       (long)nr_ws_norm(big), (const double*)parts, COV_PARTS);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

static inline int metric_tiles(int h, int w, int* tiles_x) {
    const int tx = (w - (MW - 1) + MT - 1) / MT, ty = (h - (MW - 1) + MT - 1) / MT;
    if (tiles_x) *tiles_x = tx;
    return tx * ty;
}

extern "C" int64_t pssr_image_metrics_workspace_bytes(int n_images, int h, int w) {
    if (n_images <= 0 || h < MW || w < MW) return 0;
    return (int64_t)n_images * metric_tiles(h, w, nullptr) * (int64_t)sizeof(double);
}

extern "C" int pssr_image_metrics_u8(const uint8_t* hr, const uint8_t* hr_hat, double* out, int n_images, int h, int w, void* workspace,
                                     pssr_stream_t s) {
    PSSR_CHECK(hr && hr_hat && out && workspace && n_images > 0 && n_images <= 65535, PSSR_ERR_ARG, "image_metrics: null pointer / image count");
    PSSR_CHECK(h >= MW && w >= MW && h <= 32768 && w <= 32768, PSSR_ERR_ARG,
               "image_metrics: %dx%d image (the 7x7 SSIM window must fit, as in skimage; at most 32768 a side)", h, w);
    int tiles_x;
    const int tiles = metric_tiles(h, w, &tiles_x);
    hipLaunchKernelGGL(ssim_tiles_kernel, dim3(tiles, n_images), dim3(NT), 0, (hipStream_t)s, hr, hr_hat, (double*)workspace, h, w, tiles_x, tiles);
    PSSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(image_metrics_finish_kernel, dim3(n_images), dim3(NT), 0, (hipStream_t)s, hr, hr_hat, (const double*)workspace, out,
                       (long)h * w, tiles, (long)(h - (MW - 1)) * (w - (MW - 1)));
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}
