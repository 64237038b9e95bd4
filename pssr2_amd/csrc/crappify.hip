// Crappifier / pair-generation kernels (pssr/data.py:471-495, pssr/crappifiers.py): integer-exact
// Pillow BILINEAR reduction of uint8 tiles, additive-Gaussian / Poisson noise from a counter-based
// Philox4x32-10 stream keyed by (seed, tile), Gaussian blur, and the round-half-even + clip that ends
// _gen_pair.  All HBM-bound byte/float work: one thread per output element, coalesced rows.
#include "common.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow ImagingResample 8bpc fixed point

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for one output index (bilinear = triangle filter).
// Evaluated in double in Pillow's operation order, so the integer taps are the ones Pillow uses.
struct Taps { int xmin, n; int k[40]; };
__device__ __forceinline__ void pil_taps(int xx, int in_size, int out_size, Taps& t) {
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    const int n = xmax - xmin;
    double w[40];
    double ww = 0.0;
    for (int x = 0; x < n && x < 40; ++x) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        const double v = a < 1.0 ? 1.0 - a : 0.0;
        w[x] = v; ww += v;
    }
    t.xmin = xmin; t.n = n < 40 ? n : 40;
    for (int x = 0; x < t.n; ++x) {
        const double v = ww != 0.0 ? w[x] / ww : w[x];
        t.k[x] = v < 0 ? (int)(-0.5 + v * (1 << PRECISION_BITS)) : (int)(0.5 + v * (1 << PRECISION_BITS));
    }
}
__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: [planes][H][W] -> [planes][H][w]
__global__ void resample_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int planes, int H, int W, int w) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    if (ox >= w) return;
    Taps t; pil_taps(ox, W, w, t);
    for (long row = blockIdx.y; row < (long)planes * H; row += gridDim.y) {
        const uint8_t* src = in + row * W + t.xmin;
        int acc = 1 << (PRECISION_BITS - 1);
        for (int x = 0; x < t.n; ++x) acc += (int)src[x] * t.k[x];
        out[row * w + ox] = clip8(acc);
    }
}
// vertical pass: [planes][H][w] -> [planes][h][w]
__global__ void resample_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int planes, int H, int h, int w) {
    const int oy = blockIdx.y;
    Taps t; pil_taps(oy, H, h, t);
    for (int pl = blockIdx.z; pl < planes; pl += gridDim.z)
        for (int ox = blockIdx.x * blockDim.x + threadIdx.x; ox < w; ox += gridDim.x * blockDim.x) {
            const uint8_t* src = in + ((long)pl * H + t.xmin) * w + ox;
            int acc = 1 << (PRECISION_BITS - 1);
            for (int y = 0; y < t.n; ++y) acc += (int)src[(long)y * w] * t.k[y];
            out[((long)pl * h + oy) * w + ox] = clip8(acc);
        }
}

__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

// ------------------------------------------------------------------ Philox4x32-10
struct Philox {
    uint32_t key[2];
    __device__ __forceinline__ Philox(uint64_t seed, uint64_t stream) {
        // the stream (tile index) is folded into the key so that results do not depend on how tiles are batched
        const uint64_t k = seed ^ (stream * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
        key[0] = (uint32_t)k; key[1] = (uint32_t)(k >> 32);
    }
    __device__ __forceinline__ uint4 operator()(uint64_t counter, uint32_t sub) const {
        uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = sub, c3 = 0x1BD11BDAu;
        uint32_t k0 = key[0], k1 = key[1];
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        return make_uint4(c0, c1, c2, c3);
    }
};
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {   // (0,1) with 53 bits
    const uint64_t v = (((uint64_t)hi << 32) | lo) >> 11;
    return ((double)v + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double normal_from(const uint4& r) {
    const double u1 = u01(r.x, r.y), u2 = u01(r.z, r.w);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}
__device__ __forceinline__ double finish(double v, int flags) {
    if (flags & 2) v = rint(v);                 // np.round: round-half-to-even (pssr/data.py:487)
    if (flags & 3) v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
    return v;
}
// per-tile intensity: max(N(intensity, spread), 0) if spread > 0 else intensity
__device__ __forceinline__ double tile_intensity(const Philox& ph, float intensity, float spread) {
    if (!(spread > 0.f)) return intensity;
    const double v = (double)intensity + (double)spread * normal_from(ph(~0ull, 7u));
    return v > 0.0 ? v : 0.0;
}

__global__ void gaussian_noise_kernel(const float* __restrict__ in, float* __restrict__ out, int tiles, long per_tile, float intensity,
                                      float gain, float spread, uint64_t seed, uint64_t tile_offset, const double* __restrict__ noise, int flags,
                                      const uint64_t* __restrict__ tile_counter) {
    if (tile_counter) tile_offset += tile_counter[0];
    const long total = (long)tiles * per_tile;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long tile = i / per_tile, pix = i % per_tile;
        double nz;
        if (noise) nz = noise[i];
        else {
            const Philox ph(seed, tile_offset + tile);
            nz = (double)gain + tile_intensity(ph, intensity, spread) * normal_from(ph((uint64_t)pix, 0u));
        }
        out[i] = (float)finish((double)in[i] + nz, flags);
    }
}

// Poisson(lambda): Knuth product for lambda < 10, Hoermann's PTRS transformed rejection otherwise.
__device__ double poisson_draw(const Philox& ph, uint64_t pix, double lam) {
    if (lam <= 0.0) return 0.0;
    uint32_t sub = 1;
    if (lam < 10.0) {
        const double enlam = exp(-lam);
        double prod = 1.0; long x = 0;
        for (int it = 0; it < 64; ++it) {
            const uint4 r = ph(pix, sub++);
            const double u[2] = {u01(r.x, r.y), u01(r.z, r.w)};
            for (int j = 0; j < 2; ++j) { prod *= u[j]; if (prod > enlam) ++x; else return (double)x; }
        }
        return (double)x;
    }
    const double slam = sqrt(lam), loglam = log(lam);
    const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
    const double invalpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (int it = 0; it < 256; ++it) {
        const uint4 r = ph(pix, sub++);
        const double U = u01(r.x, r.y) - 0.5, V = u01(r.z, r.w);
        const double us = 0.5 - fabs(U);
        const double k = floor((2.0 * a / us + b) * U + lam + 0.43);
        if (us >= 0.07 && V <= vr) return k;
        if (k < 0.0 || (us < 0.013 && V > us)) continue;
        if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + k * loglam - lgamma(k + 1.0)) return k;
    }
    return floor(lam + 0.5);
}

__global__ void poisson_noise_kernel(const float* __restrict__ in, float* __restrict__ out, int tiles, long per_tile, float intensity,
                                     float gain, float spread, uint64_t seed, uint64_t tile_offset, int flags,
                                     const uint64_t* __restrict__ tile_counter) {
    if (tile_counter) tile_offset += tile_counter[0];
    const long total = (long)tiles * per_tile;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long tile = i / per_tile, pix = i % per_tile;
        const Philox ph(seed, tile_offset + tile);
        const double x = (double)in[i];
        const double y = poisson_draw(ph, (uint64_t)pix, x > 0.0 ? x : 0.0);
        const double mixv = tile_intensity(ph, intensity, spread);
        out[i] = (float)finish(x * (1.0 - mixv) + y * mixv + (double)gain, flags);
    }
}

// Poisson with the samples handed in (exact-parity tests: the reference draws them from numpy's frozen legacy stream).  The arithmetic
// is numpy's for pssr/crappifiers.py:81-86: `x * (1 - intensity)` stays float32 (a Python float is a weak scalar), `y * intensity`
// is int64 * float -> float64, their sum and `+ gain` are float64; then np.round / clip as flags say (pssr/data.py:487).
__global__ void poisson_samples_kernel(const float* __restrict__ in, const double* __restrict__ samples, float* __restrict__ out, long n,
                                       double intensity, double gain, int flags) {
    const float keep = (float)(1.0 - intensity);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float t1 = __fmul_rn(in[i], keep);
        const double v = __dadd_rn(__dadd_rn((double)t1, __dmul_rn(samples[i], intensity)), gain);
        out[i] = (float)finish(v, flags);
    }
}

// separable Gaussian, edge replicate, truncate 4 sigma; f64 accumulate, f32 between the passes (scipy.ndimage)
__global__ void blur_pass_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int h, int w, float sigma, int axis,
                                 float gain, int flags) {
    const int r = (int)(4.0f * sigma + 0.5f);
    const long total = (long)planes * h * w;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = i % w, y = (i / w) % h;
        const long base = i - (long)y * w - x;
        double wsum = 0.0, acc = 0.0;
        for (int t = -r; t <= r; ++t) {
            const double wt = exp(-0.5 / ((double)sigma * sigma) * (double)t * t);
            int yy = y, xx = x;
            if (axis == 0) { yy = y + t; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy); }
            else { xx = x + t; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx); }
            wsum += wt; acc += wt * (double)in[base + (long)yy * w + xx];
        }
        double v = (double)(float)(acc / wsum);
        if (axis == 1) v = finish((double)(float)v + (double)gain, flags);
        out[i] = (float)v;
    }
}

// SaltPepper (pssr/crappifiers.py:88-105 -> skimage.util.random_noise(mode="s&p")): clip(x + gain, 0, 255), then a fraction
// `amount` of the pixels becomes 255 (salt, probability 1/2) or 0 (pepper).  Counter-based like the other noises; the
// reference draws from an unseeded default_rng(), so there is no stream to reproduce: parity is statistical.
__global__ void saltpepper_kernel(const float* __restrict__ in, float* __restrict__ out, int tiles, long per_tile, float amount, float gain,
                                  float spread, uint64_t seed, uint64_t tile_offset, int flags, const uint64_t* __restrict__ tile_counter) {
    if (tile_counter) tile_offset += tile_counter[0];
    const long total = (long)tiles * per_tile;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long tile = i / per_tile, pix = i % per_tile;
        const Philox ph(seed, tile_offset + tile);
        const double a = tile_intensity(ph, amount, spread);
        const uint4 r = ph((uint64_t)pix, 0u);
        double v = (double)in[i] + (double)gain;
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        if (u01(r.x, r.y) <= a) v = u01(r.z, r.w) <= 0.5 ? 255.0 : 0.0;
        out[i] = (float)finish(v, flags);
    }
}

// Blur with a per-tile sigma = max(N(intensity, spread), 0) (Blur(spread > 0): pssr/crappifiers.py:107-124 draws it per call,
// i.e. per tile): same separable pass as blur_pass_kernel with the tile's own radius; sigma <= 0 leaves the tile unchanged.
__global__ void blur_pass_tiles_kernel(const float* __restrict__ in, float* __restrict__ out, int tiles, int planes_per_tile, int h, int w,
                                       float intensity, float spread, uint64_t seed, uint64_t tile_offset, int axis, float gain, int flags,
                                       const uint64_t* __restrict__ tile_counter) {
    if (tile_counter) tile_offset += tile_counter[0];
    const long per_tile = (long)planes_per_tile * h * w, total = (long)tiles * per_tile;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long tile = i / per_tile;
        const Philox ph(seed, tile_offset + tile);
        const double sigma = tile_intensity(ph, intensity, spread);
        const int x = i % w, y = (i / w) % h;
        const long base = i - (long)y * w - x;
        double v = (double)in[i];
        if (sigma > 0.0) {
            const int r = (int)(4.0 * (double)(float)sigma + 0.5);
            double wsum = 0.0, acc = 0.0;
            for (int t = -r; t <= r; ++t) {
                const double wt = exp(-0.5 / (sigma * sigma) * (double)t * t);
                int yy = y, xx = x;
                if (axis == 0) { yy = y + t; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy); }
                else { xx = x + t; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx); }
                wsum += wt; acc += wt * (double)in[base + (long)yy * w + xx];
            }
            v = (double)(float)(acc / wsum);
        }
        if (axis == 1) v = finish((double)(float)v + (double)gain, flags);
        out[i] = (float)v;
    }
}

// Geometry of _gen_pair (pssr/data.py:471-482): centred square crop to at most `res`, reflect padding up to `res` at the
// bottom / right (np.pad(mode="reflect")), then np.rot90 in the (H, W) plane when rot, then np.flip along axis flip_axis
// (0 = frames, 1 = rows, 2 = columns, 3 = rows and columns = axis (1, 2); -1 = none).  One source stack [c][sh][sw] per tile, addressed through a pointer table
// so that stacks of different sizes can be batched; the random draws stay on the host (reference order).
struct GatherItem { const uint8_t* src; int sh, sw, rot, flip_axis; };

__global__ void gen_pair_geometry_kernel(const GatherItem* __restrict__ items, uint8_t* __restrict__ out, int c, int res) {
    const GatherItem it = items[blockIdx.y];
    const long per_tile = (long)c * res * res;
    const int size = it.sh < it.sw ? (it.sh < res ? it.sh : res) : (it.sw < res ? it.sw : res);   // min(h, w, res)
    const bool exact = it.sh == res && it.sw == res;
    const int sx = exact ? 0 : (it.sh - size) / 2, sy = exact ? 0 : (it.sw - size) / 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per_tile; i += (long)gridDim.x * blockDim.x) {
        int ox = (int)(i % res), oy = (int)((i / res) % res), oc = (int)(i / ((long)res * res));
        // undo flip, then rot90, to find the coordinate in the padded crop
        if (it.flip_axis == 0) oc = c - 1 - oc;
        if (it.flip_axis == 1 || it.flip_axis == 3) oy = res - 1 - oy;
        if (it.flip_axis == 2 || it.flip_axis == 3) ox = res - 1 - ox;
        int py = oy, px = ox;
        if (it.rot) { py = ox; px = res - 1 - oy; }        // np.rot90(m)[i][j] = m[j][n - 1 - i]
        // reflect padding (no edge repeat) beyond the crop: index p >= size maps to 2 * (size - 1) - p, periodically
        if (size > 1) {
            const int period = 2 * (size - 1);
            py %= period; px %= period;
            if (py >= size) py = period - py;
            if (px >= size) px = period - px;
        } else { py = 0; px = 0; }
        out[blockIdx.y * per_tile + i] = it.src[((long)oc * it.sh + sx + py) * it.sw + sy + px];
    }
}

static inline int grid1d(long total) { long b = (total + 255) / 256; return (int)(b < 8192 ? (b > 0 ? b : 1) : 8192); }

}  // namespace

extern "C" {

int pssr_bilinear_down_u8(const uint8_t* hr, uint8_t* tmp, uint8_t* lr, int planes, int H, int W, int h, int w, pssr_stream_t s) {
    PSSR_CHECK(hr && tmp && lr && planes > 0 && H > 0 && W > 0 && h > 0 && w > 0, PSSR_ERR_ARG, "bilinear_down_u8: bad args");
    PSSR_CHECK(H >= h && W >= w && (double)W / w <= 19.0 && (double)H / h <= 19.0, PSSR_ERR_UNSUPPORTED, "bilinear_down_u8: reduction ratio must be in [1, 19]");
    long rows = (long)planes * H;
    // a thread computes its output column's taps once (float arithmetic) and walks ~16 rows with them: with one row per thread the launch
    // spent its time on the taps (66 -> 20 us for a c2 batch)
    const long row_groups = (rows + 15) / 16;
    hipLaunchKernelGGL(resample_h_kernel, dim3(cdiv(w, 64), (unsigned)(row_groups < 65535 ? row_groups : 65535)), dim3(64), 0, (hipStream_t)s, hr, tmp, planes, H, W, w);
    PSSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(resample_v_kernel, dim3(cdiv(w, 128), h, planes < 1024 ? planes : 1024), dim3(128), 0, (hipStream_t)s, tmp, lr, planes, H, h, w);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_u8_to_f32(const uint8_t* in, float* out, int64_t n, pssr_stream_t s) {
    PSSR_CHECK(in && out && n > 0, PSSR_ERR_ARG, "u8_to_f32: bad args");
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)s, in, out, (long)n);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_crappify_gaussian(const float* in, float* out, int tiles, int64_t per_tile, float intensity, float gain, float spread,
                           uint64_t seed, uint64_t tile_offset, const double* noise, int flags, const uint64_t* tile_counter,
                           pssr_stream_t s) {
    PSSR_CHECK(in && out && tiles > 0 && per_tile > 0 && flags >= 0 && flags <= 3, PSSR_ERR_ARG, "crappify_gaussian: bad args");
    hipLaunchKernelGGL(gaussian_noise_kernel, dim3(grid1d((long)tiles * per_tile)), dim3(256), 0, (hipStream_t)s, in, out, tiles, (long)per_tile,
                       intensity, gain, spread, seed, tile_offset, noise, flags, tile_counter);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_crappify_poisson(const float* in, float* out, int tiles, int64_t per_tile, float intensity, float gain, float spread,
                          uint64_t seed, uint64_t tile_offset, int flags, const uint64_t* tile_counter, pssr_stream_t s) {
    PSSR_CHECK(in && out && tiles > 0 && per_tile > 0 && flags >= 0 && flags <= 3, PSSR_ERR_ARG, "crappify_poisson: bad args");
    hipLaunchKernelGGL(poisson_noise_kernel, dim3(grid1d((long)tiles * per_tile)), dim3(256), 0, (hipStream_t)s, in, out, tiles, (long)per_tile,
                       intensity, gain, spread, seed, tile_offset, flags, tile_counter);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_crappify_poisson_samples(const float* in, const double* samples, float* out, int64_t n, double intensity, double gain, int flags,
                                  pssr_stream_t s) {
    PSSR_CHECK(in && samples && out && n > 0 && flags >= 0 && flags <= 3, PSSR_ERR_ARG, "crappify_poisson_samples: bad args");
    hipLaunchKernelGGL(poisson_samples_kernel, dim3(grid1d((long)n)), dim3(256), 0, (hipStream_t)s, in, samples, out, (long)n, intensity, gain, flags);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

__global__ void counter_add_kernel(uint64_t* c, uint64_t inc) { c[0] += inc; }

int pssr_counter_add(uint64_t* counter, uint64_t inc, pssr_stream_t s) {
    PSSR_CHECK(counter != nullptr, PSSR_ERR_ARG, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, counter, inc);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_gaussian_blur(const float* in, float* tmp, float* out, int planes, int h, int w, float sigma, float gain, int flags, pssr_stream_t s) {
    PSSR_CHECK(in && tmp && out && planes > 0 && h > 0 && w > 0 && sigma > 0.f && flags >= 0 && flags <= 3, PSSR_ERR_ARG, "gaussian_blur: bad args");
    const long total = (long)planes * h * w;
    hipLaunchKernelGGL(blur_pass_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)s, in, tmp, planes, h, w, sigma, 0, 0.f, 0);
    PSSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(blur_pass_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)s, tmp, out, planes, h, w, sigma, 1, gain, flags);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_crappify_saltpepper(const float* in, float* out, int tiles, int64_t per_tile, float amount, float gain, float spread, uint64_t seed,
                             uint64_t tile_offset, int flags, const uint64_t* tile_counter, pssr_stream_t s) {
    PSSR_CHECK(in && out && tiles > 0 && per_tile > 0 && flags >= 0 && flags <= 3 && amount >= 0.f, PSSR_ERR_ARG, "crappify_saltpepper: bad args");
    hipLaunchKernelGGL(saltpepper_kernel, dim3(grid1d((long)tiles * per_tile)), dim3(256), 0, (hipStream_t)s, in, out, tiles, (long)per_tile,
                       amount, gain, spread, seed, tile_offset, flags, tile_counter);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_gaussian_blur_tiles(const float* in, float* tmp, float* out, int tiles, int planes_per_tile, int h, int w, float sigma, float spread,
                             float gain, uint64_t seed, uint64_t tile_offset, int flags, const uint64_t* tile_counter, pssr_stream_t s) {
    PSSR_CHECK(in && tmp && out && tiles > 0 && planes_per_tile > 0 && h > 0 && w > 0 && sigma >= 0.f && spread >= 0.f && flags >= 0 && flags <= 3,
               PSSR_ERR_ARG, "gaussian_blur_tiles: bad args");
    const long total = (long)tiles * planes_per_tile * h * w;
    hipLaunchKernelGGL(blur_pass_tiles_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)s, in, tmp, tiles, planes_per_tile, h, w, sigma,
                       spread, seed, tile_offset, 0, 0.f, 0, tile_counter);
    PSSR_LAUNCH_CHECK();
    hipLaunchKernelGGL(blur_pass_tiles_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)s, tmp, out, tiles, planes_per_tile, h, w, sigma,
                       spread, seed, tile_offset, 1, gain, flags, tile_counter);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_gen_pair_geometry_u8(const pssr_gather_item* items_dev, int n_items, uint8_t* out, int c, int res, pssr_stream_t s) {
    PSSR_CHECK(items_dev && out && n_items > 0 && n_items <= 65535 && c > 0 && res > 0, PSSR_ERR_ARG, "gen_pair_geometry: bad args");
    static_assert(sizeof(pssr_gather_item) == sizeof(GatherItem), "pssr_gather_item layout");
    const long per_tile = (long)c * res * res;
    long gx = (per_tile + 255) / 256; if (gx > 256) gx = 256;
    hipLaunchKernelGGL(gen_pair_geometry_kernel, dim3((unsigned)gx, n_items), dim3(256), 0, (hipStream_t)s, (const GatherItem*)items_dev, out, c, res);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

}  // extern "C"
