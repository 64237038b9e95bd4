// SSIM / MS-SSIM + Gaussian-weighted L1 loss (pssr/util.py:10-52 -> pytorch_msssim), forward and
// backward, on f32 NCHW planes.  HBM-bound: every level reads its two maps once per pass; the five
// Gaussian-filtered maps (mu_x, mu_y, E[x^2], E[y^2], E[xy]) exist only in LDS.
//
// forward  (per level): tile of TS x TS *valid* positions; separable 11-tap filter through LDS;
//                       per-plane f64 sums of the cs map and the ssim map (+ weighted |x-y| at level 0).
// weights  (tiny)     : loss value and d loss / d(map element) per plane and level, on device.
// backward (per level): recompute the filtered maps on a (TS+10)^2 halo, form the three adjoint maps
//                       (wrt mu_x, E[x^2], E[xy]), apply the transposed filter, add the avg-pool
//                       gradient arriving from the coarser level and the L1 term.
#include "common.h"

namespace {

constexpr int TS = 32;       // tile side
constexpr int MAXW = 33;     // largest supported window

struct Win { float g[MAXW]; int k; };

// ------------------------------------------------------------------------------------------------
template <int K_DUMMY>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                       Win win, float C1, float C2, double* sums /*[planes][2]*/,
                                                       double* l1_sum /*nullable*/) {
    // LDS: xs, ys [(TS+k-1)^2]; hp [5][(TS+k-1)][TS]
    extern __shared__ float lds[];
    const int k = win.k, halo = k - 1, IN = TS + halo;
    float* xs = lds;
    float* ys = xs + IN * IN;
    float* hp = ys + IN * IN;
    const int plane = blockIdx.z;
    const int oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;       // valid-map tile origin == input origin
    const int VH = H - halo, VW = W - halo;
    const float* xp = X + (long)plane * H * W;
    const float* yp = Y + (long)plane * H * W;
    const int tid = threadIdx.x;
    float l1 = 0.f;
    for (int i = tid; i < IN * IN; i += 256) {
        const int r = i / IN, c = i % IN;
        const int gy = oy0 + r, gx = ox0 + c;
        float xv = 0.f, yv = 0.f;
        if (gy < H && gx < W) { xv = xp[(long)gy * W + gx]; yv = yp[(long)gy * W + gx]; }
        xs[i] = xv; ys[i] = yv;
    }
    if (l1_sum) {
        // L1 term: each pixel once (tiles partition the image when extended to cover it; see launcher)
        const int ly0 = blockIdx.y * TS, lx0 = blockIdx.x * TS;
        const int r5 = k / 2;
        for (int i = tid; i < TS * TS; i += 256) {
            const int gy = ly0 + i / TS, gx = lx0 + i % TS;
            if (gy < H && gx < W) {
                float sy = 0.f, sx = 0.f;
                for (int t = 0; t < k; ++t) {
                    const int yy = gy + t - r5, xx = gx + t - r5;
                    if (yy >= 0 && yy < H) sy += win.g[t];
                    if (xx >= 0 && xx < W) sx += win.g[t];
                }
                l1 += fabsf(xp[(long)gy * W + gx] - yp[(long)gy * W + gx]) * sy * sx;
            }
        }
    }
    __syncthreads();
    // horizontal pass: rows 0..IN-1, cols 0..TS-1
    for (int i = tid; i < IN * TS; i += 256) {
        const int r = i / TS, c = i % TS;
        float mx = 0, my = 0, xx = 0, yy = 0, xy = 0;
        for (int t = 0; t < k; ++t) {
            const float a = xs[r * IN + c + t], b = ys[r * IN + c + t], gw = win.g[t];
            mx = fmaf(gw, a, mx); my = fmaf(gw, b, my);
            xx = fmaf(gw, a * a, xx); yy = fmaf(gw, b * b, yy); xy = fmaf(gw, a * b, xy);
        }
        hp[0 * IN * TS + i] = mx; hp[1 * IN * TS + i] = my; hp[2 * IN * TS + i] = xx; hp[3 * IN * TS + i] = yy; hp[4 * IN * TS + i] = xy;
    }
    __syncthreads();
    float cs_acc = 0.f, ss_acc = 0.f;
    for (int i = tid; i < TS * TS; i += 256) {
        const int r = i / TS, c = i % TS;
        if (oy0 + r >= VH || ox0 + c >= VW) continue;
        float m[5] = {0, 0, 0, 0, 0};
        for (int t = 0; t < k; ++t) {
            const float gw = win.g[t];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] = fmaf(gw, hp[q * IN * TS + (r + t) * TS + c], m[q]);
        }
        const float mx = m[0], my = m[1];
        const float sxx = m[2] - mx * mx, syy = m[3] - my * my, sxy = m[4] - mx * my;
        const float cs = (2.f * sxy + C2) / (sxx + syy + C2);
        const float lum = (2.f * mx * my + C1) / (mx * mx + my * my + C1);
        cs_acc += cs; ss_acc += lum * cs;
    }
    __shared__ float red[3][256];
    red[0][tid] = cs_acc; red[1][tid] = ss_acc; red[2][tid] = l1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        atomicAdd(sums + plane * 2, (double)red[0][0]);
        atomicAdd(sums + plane * 2 + 1, (double)red[1][0]);
        if (l1_sum) atomicAdd(l1_sum, (double)red[2][0]);
    }
}

// avg_pool2d(kernel 2, stride 2, padding = size % 2, count_include_pad) on planes
__global__ void avgpool_kernel(const float* __restrict__ in0, float* __restrict__ out0, const float* __restrict__ in1, float* __restrict__ out1,
                               int planes, int H, int W, int HO, int WO, float in_mul) {
#pragma clang fp contract(off)          // the scaled inputs are rounded products (what the quotient tensor would hold), not FMA operands
    const float* in = blockIdx.y ? in1 : in0;          // (two tensors per launch: the x and y pyramids of the loss)
    float* out = blockIdx.y ? out1 : out0;
    const int py = H & 1, px = W & 1;
    const long total = (long)planes * HO * WO;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ox = i % WO, oy = (i / WO) % HO;
        const long pl = i / ((long)WO * HO);
        const float* src = in + pl * H * W;
        float s = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int y = 2 * oy - py + dy, x = 2 * ox - px + dx;
                if (y >= 0 && y < H && x >= 0 && x < W) s += __fmul_rn(src[(long)y * W + x], in_mul);      // (a rounded product, as the quotient tensor would hold: no FMA contraction)
            }
        out[i] = 0.25f * s;
    }
}

// loss and upstream weights.  sums: [levels][planes][2] (cs sum, ssim sum); nvalid[l] = valid positions per plane.
__global__ void weights_kernel(const double* sums_in, int levels, int planes, const double* nvalid, const float* lvl_w, int ms,
                               float mix, const double* l1_sum_in, double l1_numel, const float* grad_out,
                               float* loss_out, float* wts /*[levels][planes]*/, float* l1_coef, int stripes, long stripe_stride,
                               double* folded /* [levels * planes * 2 + 1] scratch when stripes > 1 */) {
    // striped sums (ssim_fwd_adj_k): fold the copies first, in a fixed order
    const double* sums = sums_in;
    const double* l1_sum = l1_sum_in;
    if (stripes > 1) {
        // (fixed order; the loads of a value are independent and issued together: the serial loop cost 24 us on the step's chain)
        const int nvals = levels * planes * 2;
        for (int i = threadIdx.x; i <= nvals; i += 256) {
            const double* src = i < nvals ? sums_in + i : l1_sum_in;
            double t = 0.0;
            if (src) {
                int k = 0;
                for (; k + 8 <= 2 * stripes; k += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = src[(long)(k + u) * stripe_stride];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t += v[u];
                }
                for (; k < 2 * stripes; ++k) t += src[(long)k * stripe_stride];
            }
            folded[i] = t;
        }
        __syncthreads();
        sums = folded;
        if (l1_sum_in) l1_sum = folded + nvals;
    }
    __shared__ double acc[256];
    const int tid = threadIdx.x;
    const float go = grad_out ? grad_out[0] : 1.f;
    double local = 0.0;
    for (int p = tid; p < planes; p += 256) {
        if (ms) {
            double prod = 1.0;
            double v[8];
            for (int l = 0; l < levels; ++l) {
                const double mean = sums[((long)l * planes + p) * 2 + (l == levels - 1 ? 1 : 0)] / nvalid[l];
                v[l] = mean > 0 ? mean : 0.0;
                prod *= pow(v[l], (double)lvl_w[l]);
            }
            local += prod;
            for (int l = 0; l < levels; ++l) {
                // d loss / d v_l = -mix/planes * w_l * prod / v_l ; spread over nvalid[l] map elements
                const double d = v[l] > 0 ? -(double)mix / planes * lvl_w[l] * prod / v[l] : 0.0;
                wts[(long)l * planes + p] = (float)(d / nvalid[l] * go);
            }
        } else {
            const double mean = sums[(long)p * 2 + 1] / nvalid[0];
            local += mean;      // nonnegative_ssim=False: no relu on plain SSIM
            wts[p] = (float)(-(double)mix / planes / nvalid[0] * go);
        }
    }
    acc[tid] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) acc[tid] += acc[tid + o]; __syncthreads(); }
    if (tid == 0) {
        double loss = mix * (1.0 - acc[0] / planes);
        double lc = 0.0;
        if (l1_sum) { loss = loss + (1.0 - mix) * l1_sum[0] / l1_numel; lc = (1.0 - mix) / l1_numel * go; }
        else loss = (1.0 - acc[0] / planes);          // mix == 1: the reference skips the L1 branch entirely
        loss_out[0] = (float)loss;
        l1_coef[0] = (float)lc;
    }
}

__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, Win win,
                                                       float C1, float C2, const float* __restrict__ wts /*[planes]*/, int use_ssim,
                                                       const float* __restrict__ dcoarse, int HC, int WC,
                                                       const float* l1_coef_p, float* __restrict__ dX) {
    // output tile: TS x TS pixels at (qy0, qx0); adjoint maps needed at valid positions p in [q0-halo, q0+TS-1]
    extern __shared__ float lds[];
    const int k = win.k, halo = k - 1, AD = TS + halo, IN = TS + 2 * halo;
    float* xs = lds;                       // [IN][IN]
    float* ys = xs + IN * IN;              // [IN][IN]
    float* hp = ys + IN * IN;              // [5][IN][AD]  (horizontal pass)  -> later reused for adjoint h-pass [3][AD][TS]
    float* ad = hp + 5 * IN * AD;          // [3][AD][AD]  adjoint maps a, b, c
    const int plane = blockIdx.z;
    const int qy0 = blockIdx.y * TS, qx0 = blockIdx.x * TS;
    const int VH = H - halo, VW = W - halo;
    const float* xp = X + (long)plane * H * W;
    const float* yp = Y + (long)plane * H * W;
    const int tid = threadIdx.x;
    const float wt = wts[plane];
    // inputs rows [qy0-halo, qy0+TS+halo)
    for (int i = tid; i < IN * IN; i += 256) {
        const int r = i / IN, c = i % IN;
        const int gy = qy0 - halo + r, gx = qx0 - halo + c;
        float xv = 0.f, yv = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) { xv = xp[(long)gy * W + gx]; yv = yp[(long)gy * W + gx]; }
        xs[i] = xv; ys[i] = yv;
    }
    __syncthreads();
    for (int i = tid; i < IN * AD; i += 256) {
        const int r = i / AD, c = i % AD;
        float mx = 0, my = 0, xx = 0, yy = 0, xy = 0;
        for (int t = 0; t < k; ++t) {
            const float a = xs[r * IN + c + t], b = ys[r * IN + c + t], gw = win.g[t];
            mx = fmaf(gw, a, mx); my = fmaf(gw, b, my);
            xx = fmaf(gw, a * a, xx); yy = fmaf(gw, b * b, yy); xy = fmaf(gw, a * b, xy);
        }
        hp[0 * IN * AD + i] = mx; hp[1 * IN * AD + i] = my; hp[2 * IN * AD + i] = xx; hp[3 * IN * AD + i] = yy; hp[4 * IN * AD + i] = xy;
    }
    __syncthreads();
    for (int i = tid; i < AD * AD; i += 256) {
        const int r = i / AD, c = i % AD;
        const int py = qy0 - halo + r, px = qx0 - halo + c;     // valid-map position
        float a = 0.f, b = 0.f, cc = 0.f;
        if (py >= 0 && py < VH && px >= 0 && px < VW) {
            float m[5] = {0, 0, 0, 0, 0};
            for (int t = 0; t < k; ++t) {
                const float gw = win.g[t];
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] = fmaf(gw, hp[q * IN * AD + (r + t) * AD + c], m[q]);
            }
            const float mx = m[0], my = m[1];
            const float sxx = m[2] - mx * mx, syy = m[3] - my * my, sxy = m[4] - mx * my;
            const float Dcs = sxx + syy + C2, cs = (2.f * sxy + C2) / Dcs;
            const float dcs_dmx = 2.f * (cs * mx - my) / Dcs, dcs_dexx = -cs / Dcs, dcs_dexy = 2.f / Dcs;
            if (use_ssim) {
                const float Dl = mx * mx + my * my + C1, lum = (2.f * mx * my + C1) / Dl;
                const float dl_dmx = 2.f * (my - lum * mx) / Dl;
                a = wt * (lum * dcs_dmx + cs * dl_dmx); b = wt * lum * dcs_dexx; cc = wt * lum * dcs_dexy;
            } else {
                a = wt * dcs_dmx; b = wt * dcs_dexx; cc = wt * dcs_dexy;
            }
        }
        ad[0 * AD * AD + i] = a; ad[1 * AD * AD + i] = b; ad[2 * AD * AD + i] = cc;
    }
    __syncthreads();
    // transposed filter, horizontal: out col q gets sum_t g[t] * ad[.., q + halo - t]  (ad col index = p - (q0-halo))
    float* th = hp;                        // [3][AD][TS]
    for (int i = tid; i < AD * TS; i += 256) {
        const int r = i / TS, c = i % TS;
        float s0 = 0, s1 = 0, s2 = 0;
        for (int t = 0; t < k; ++t) {
            const float gw = win.g[t];
            const int src = r * AD + c + halo - t;
            s0 = fmaf(gw, ad[src], s0); s1 = fmaf(gw, ad[AD * AD + src], s1); s2 = fmaf(gw, ad[2 * AD * AD + src], s2);
        }
        th[i] = s0; th[AD * TS + i] = s1; th[2 * AD * TS + i] = s2;
    }
    __syncthreads();
    const float l1c = l1_coef_p ? l1_coef_p[0] : 0.f;
    const int r5 = k / 2;
    float* dxp = dX + (long)plane * H * W;
    for (int i = tid; i < TS * TS; i += 256) {
        const int r = i / TS, c = i % TS;
        const int gy = qy0 + r, gx = qx0 + c;
        if (gy >= H || gx >= W) continue;
        float s0 = 0, s1 = 0, s2 = 0;
        for (int t = 0; t < k; ++t) {
            const float gw = win.g[t];
            const int src = (r + halo - t) * TS + c;
            s0 = fmaf(gw, th[src], s0); s1 = fmaf(gw, th[AD * TS + src], s1); s2 = fmaf(gw, th[2 * AD * TS + src], s2);
        }
        const float xv = xs[(r + halo) * IN + c + halo], yv = ys[(r + halo) * IN + c + halo];
        float g = s0 + 2.f * xv * s1 + yv * s2;
        if (dcoarse) {
            const int cy = (gy + (H & 1)) >> 1, cx = (gx + (W & 1)) >> 1;
            g += 0.25f * dcoarse[((long)plane * HC + cy) * WC + cx];
        }
        if (l1c != 0.f) {
            float sy = 0.f, sx = 0.f;
            for (int t = 0; t < k; ++t) {
                const int yy = gy + t - r5, xx = gx + t - r5;
                if (yy >= 0 && yy < H) sy += win.g[t];
                if (xx >= 0 && xx < W) sx += win.g[t];
            }
            const float d = xv - yv;
            g += l1c * sy * sx * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        dxp[(long)gy * W + gx] = g;
    }
}

// ------------------------------------------------------------------------------------------------
// Register-blocked variants for a compile-time window K: a thread produces SEG consecutive outputs of a 1-D filter pass
// from a sliding window of SEG+K-1 inputs held in registers (6x fewer LDS reads than one output per thread), all tap
// loops unrolled (window weights come from scalar registers).  Same LDS images and arithmetic order per output as above.
template <int K, int SEG, int NQ>
__device__ __forceinline__ void fir_seg(const float (*v)[SEG + K - 1], const Win& win, float (*out)[SEG]) {
#pragma unroll
    for (int o = 0; o < SEG; ++o)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < K; ++t) s = fmaf(win.g[t], v[q][o + t], s);
            out[q][o] = s;
        }
}

template <int K>
__global__ __launch_bounds__(256) void ssim_fwd_k(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                  Win win, float C1, float C2, double* sums, double* l1_sum) {
    constexpr int halo = K - 1, IN = TS + halo, SEG = 8, NS = TS / SEG;
    constexpr int HPW = TS + 1;                // row pitch of the h-pass images: the 8 rows a lane group writes fall on different banks
    extern __shared__ float lds[];
    float* xs = lds;
    float* ys = xs + IN * IN;
    float* hp = ys + IN * IN;                  // [5][IN][HPW]
    const int plane = blockIdx.z;
    const int oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
    const int VH = H - halo, VW = W - halo;
    const float* xp = X + (long)plane * H * W;
    const float* yp = Y + (long)plane * H * W;
    const int tid = threadIdx.x;
    float l1 = 0.f;
    for (int i = tid; i < IN * IN; i += 256) {
        const int r = i / IN, c = i % IN;
        const int gy = oy0 + r, gx = ox0 + c;
        float xv = 0.f, yv = 0.f;
        if (gy < H && gx < W) { xv = xp[(long)gy * W + gx]; yv = yp[(long)gy * W + gx]; }
        xs[i] = xv; ys[i] = yv;
    }
    __syncthreads();
    if (l1_sum) {
        // L1 term from the staged tile (its origin is the tile origin); the border weight of an interior pixel is the whole
        // window sum, added in the same order as the clipped sums so the value is the same float
        constexpr int r5 = K / 2;
        float wsum = 0.f;
#pragma unroll
        for (int t = 0; t < K; ++t) wsum += win.g[t];
        for (int i = tid; i < TS * TS; i += 256) {
            const int r = i / TS, c = i % TS;
            const int gy = oy0 + r, gx = ox0 + c;
            if (gy < H && gx < W) {
                float sy = wsum, sx = wsum;
                if (gy < r5 || gy >= H - r5) {
                    sy = 0.f;
#pragma unroll
                    for (int t = 0; t < K; ++t) { const int yy = gy + t - r5; if (yy >= 0 && yy < H) sy += win.g[t]; }
                }
                if (gx < r5 || gx >= W - r5) {
                    sx = 0.f;
#pragma unroll
                    for (int t = 0; t < K; ++t) { const int xx = gx + t - r5; if (xx >= 0 && xx < W) sx += win.g[t]; }
                }
                l1 += fabsf(xs[r * IN + c] - ys[r * IN + c]) * sy * sx;
            }
        }
    }
    // horizontal pass: item = (row, segment of SEG output columns)
    for (int it = tid; it < IN * NS; it += 256) {
        const int r = it / NS, c0 = (it % NS) * SEG;
        float v[5][SEG + K - 1];
#pragma unroll
        for (int j = 0; j < SEG + K - 1; ++j) {
            const float a = xs[r * IN + c0 + j], b = ys[r * IN + c0 + j];
            v[0][j] = a; v[1][j] = b; v[2][j] = a * a; v[3][j] = b * b; v[4][j] = a * b;
        }
        float o[5][SEG];
        fir_seg<K, SEG, 5>(v, win, o);
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < SEG; ++j) hp[q * IN * HPW + r * HPW + c0 + j] = o[q][j];
    }
    __syncthreads();
    float cs_acc = 0.f, ss_acc = 0.f;
    // vertical pass: item = (column, segment of SEG output rows)
    for (int it = tid; it < TS * NS; it += 256) {
        const int c = it % TS, r0 = (it / TS) * SEG;
        float v[5][SEG + K - 1];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < SEG + K - 1; ++j) v[q][j] = hp[q * IN * HPW + (r0 + j) * HPW + c];
        float m[5][SEG];
        fir_seg<K, SEG, 5>(v, win, m);
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
            if (oy0 + r0 + j >= VH || ox0 + c >= VW) continue;
            const float mx = m[0][j], my = m[1][j];
            const float sxx = m[2][j] - mx * mx, syy = m[3][j] - my * my, sxy = m[4][j] - mx * my;
            const float Dcs = sxx + syy + C2;
            const float cs = (2.f * sxy + C2) / Dcs;
            const float Dl = mx * mx + my * my + C1;
            const float lum = (2.f * mx * my + C1) / Dl;
            cs_acc += cs; ss_acc += lum * cs;
        }
    }
    __shared__ float red[3][256];
    red[0][tid] = cs_acc; red[1][tid] = ss_acc; red[2][tid] = l1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        atomicAdd(sums + plane * 2, (double)red[0][0]);
        atomicAdd(sums + plane * 2 + 1, (double)red[1][0]);
        if (l1_sum) atomicAdd(l1_sum, (double)red[2][0]);
    }
}

// Training forward (11-tap window): as ssim_fwd_k, plus the adjoint maps for ssim_bwd_adj_k.  The five filtered maps go through ONE
// horizontal-pass image one after the other (xs / ys + one map: 19.7 KB per workgroup instead of 41.6 KB, so 5-8 workgroups per CU
// instead of 3: this kernel, too, spent most of its wave cycles in waits), the vertical results stay in registers; the per-plane sums
// are spread over `stripes` copies (a level-0 launch ended with 256 f64 atomics per address and 8192 on the L1 sum: 59 of 217 us).
template <int K>
__global__ __launch_bounds__(256) void ssim_fwd_adj_k(const float* __restrict__ X, const float* __restrict__ Y, int H, int W,
                                                      Win win, float C1, float C2, double* sums, double* l1_sum, int stripes, long stripe_stride,
                                                      float* __restrict__ adj, int use_ssim, float in_mul) {
    constexpr int halo = K - 1, IN = TS + halo, SEG = 8, NS = TS / SEG;
    constexpr int HPW = TS + 1;
    extern __shared__ float lds[];
    float* xs = lds;
    float* ys = xs + IN * IN;
    float* hp = ys + IN * IN;                  // [IN][HPW]: one map at a time
    const int plane = blockIdx.z;
    const int oy0 = blockIdx.y * TS, ox0 = blockIdx.x * TS;
    const int VH = H - halo, VW = W - halo;
    const float* xp = X + (long)plane * H * W;
    const float* yp = Y + (long)plane * H * W;
    const int tid = threadIdx.x;
    float l1 = 0.f;
    for (int i = tid; i < IN * IN; i += 256) {
        const int r = i / IN, c = i % IN;
        const int gy = oy0 + r, gx = ox0 + c;
        float xv = 0.f, yv = 0.f;
        // (in_mul = 1 / divisor in f32: the loss of x / divisor and y / divisor without materialising the quotients -- train_paired
        // passes images / 255, two full passes over 33 MB tensors per step and a third one for the gradient; torch divides a tensor
        // by a scalar as a multiplication by the f32 reciprocal, so do these kernels)
        if (gy < H && gx < W) { xv = __fmul_rn(xp[(long)gy * W + gx], in_mul); yv = __fmul_rn(yp[(long)gy * W + gx], in_mul); }
        xs[i] = xv; ys[i] = yv;
    }
    __syncthreads();
    if (l1_sum) {
        constexpr int r5 = K / 2;
        float wsum = 0.f;
#pragma unroll
        for (int t = 0; t < K; ++t) wsum += win.g[t];
        for (int i = tid; i < TS * TS; i += 256) {
            const int r = i / TS, c = i % TS;
            const int gy = oy0 + r, gx = ox0 + c;
            if (gy < H && gx < W) {
                float sy = wsum, sx = wsum;
                if (gy < r5 || gy >= H - r5) {
                    sy = 0.f;
#pragma unroll
                    for (int t = 0; t < K; ++t) { const int yy = gy + t - r5; if (yy >= 0 && yy < H) sy += win.g[t]; }
                }
                if (gx < r5 || gx >= W - r5) {
                    sx = 0.f;
#pragma unroll
                    for (int t = 0; t < K; ++t) { const int xx = gx + t - r5; if (xx >= 0 && xx < W) sx += win.g[t]; }
                }
                l1 += fabsf(xs[r * IN + c] - ys[r * IN + c]) * sy * sx;
            }
        }
    }
    // vertical item of this thread (the first TS * NS threads): column vc, rows vr0 .. vr0 + SEG - 1; its five filtered values
    const bool vitem = tid < TS * NS;
    const int vc = tid % TS, vr0 = (tid / TS) * SEG;
    float m[5][SEG];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        // horizontal pass of map q (x, y, x^2, y^2, xy): item = (row, segment of SEG output columns)
        for (int it = tid; it < IN * NS; it += 256) {
            const int r = it / NS, c0 = (it % NS) * SEG;
            float v[1][SEG + K - 1];
#pragma unroll
            for (int j = 0; j < SEG + K - 1; ++j) {
                const float a = (q == 1 || q == 3) ? 0.f : xs[r * IN + c0 + j];
                const float b = (q == 0 || q == 2) ? 0.f : ys[r * IN + c0 + j];
                v[0][j] = q == 0 ? a : q == 1 ? b : q == 2 ? a * a : q == 3 ? b * b : a * b;
            }
            float o[1][SEG];
            fir_seg<K, SEG, 1>(v, win, o);
#pragma unroll
            for (int j = 0; j < SEG; ++j) hp[r * HPW + c0 + j] = o[0][j];
        }
        __syncthreads();
        if (vitem) {
            float v[1][SEG + K - 1];
#pragma unroll
            for (int j = 0; j < SEG + K - 1; ++j) v[0][j] = hp[(vr0 + j) * HPW + vc];
            float o[1][SEG];
            fir_seg<K, SEG, 1>(v, win, o);
#pragma unroll
            for (int j = 0; j < SEG; ++j) m[q][j] = o[0][j];
        }
        if (q < 4) __syncthreads();
    }
    float cs_acc = 0.f, ss_acc = 0.f;
    if (vitem) {
#pragma unroll
        for (int j = 0; j < SEG; ++j) {
            if (oy0 + vr0 + j >= VH || ox0 + vc >= VW) continue;
            const float mx = m[0][j], my = m[1][j];
            const float sxx = m[2][j] - mx * mx, syy = m[3][j] - my * my, sxy = m[4][j] - mx * my;
            const float Dcs = sxx + syy + C2;
            const float cs = (2.f * sxy + C2) / Dcs;
            const float Dl = mx * mx + my * my + C1;
            const float lum = (2.f * mx * my + C1) / Dl;
            cs_acc += cs; ss_acc += lum * cs;
            // the map's derivatives wrt (mu_x, E[x^2], E[xy]) at this position, per unit of upstream weight: what the backward
            // pass used to recompute from x and y on a (TS + 20) x (TS + 10) halo (5 filtered maps) before its own 3 filters
            const float dcs_dmx = 2.f * (cs * mx - my) / Dcs, dcs_dexx = -cs / Dcs, dcs_dexy = 2.f / Dcs;
            float a = dcs_dmx, b = dcs_dexx, cc = dcs_dexy;
            if (use_ssim) {
                const float dl_dmx = 2.f * (my - lum * mx) / Dl;
                a = lum * dcs_dmx + cs * dl_dmx; b = lum * dcs_dexx; cc = lum * dcs_dexy;
            }
            float* ap = adj + ((long)plane * 3 * H + (oy0 + vr0 + j)) * W + ox0 + vc;
            ap[0] = a; ap[(long)H * W] = b; ap[2L * H * W] = cc;
        }
    }
    __shared__ float red[3][256];
    red[0][tid] = cs_acc; red[1][tid] = ss_acc; red[2][tid] = l1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; red[2][tid] += red[2][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        // (two exact pieces per sum, as stat_add: rows [0, stripes) and [stripes, 2 stripes) -- order-independent)
        const long so = (long)((blockIdx.x + blockIdx.y * 5) % stripes) * stripe_stride, lo = (long)stripes * stripe_stride;
        stat_add(sums + so + plane * 2, lo, red[0][0]);
        stat_add(sums + so + plane * 2 + 1, lo, red[1][0]);
        if (l1_sum) stat_add(l1_sum + so, lo, red[2][0]);
    }
}

// Backward from the adjoint maps the forward pass stored ([plane][3][H][W], valid positions only): scale by the plane's upstream
// weight, apply the transposed separable filter, add the chain-rule factors (1, 2x, y), the avg-pool gradient of the coarser level
// and the L1 term.  3 filtered maps on a (TS + 10)^2 halo instead of 5 + 3: 78 k instead of 295 k FMAs per 32 x 32 tile.
// The three maps go through ONE pair of LDS images one after the other (13.8 KB per workgroup: 8 workgroups per CU instead of 3 --
// the counters of the all-at-once version: 71 % of the wave cycles in waits, 15 % issuing vector instructions), the next map's tile
// is requested before the current one is filtered, and the per-tile index arithmetic is done once.
template <int K>
__global__ __launch_bounds__(256) void ssim_bwd_adj_k(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ ADJ,
                                                      int H, int W, Win win, const float* __restrict__ wts,
                                                      const float* __restrict__ dcoarse, int HC, int WC, const float* l1_coef_p,
                                                      float* __restrict__ dX, float in_mul) {
    constexpr int halo = K - 1, AD = TS + halo;
    constexpr int SEG = 8, NS = TS / SEG;
    constexpr int ADC = AD + 1;                            // row pitch of the adjoint tile (odd: the rows of a lane group on different banks)
    constexpr int THR = AD + SEG, THW = TS + 1;
    constexpr int NL = (AD * AD + 255) / 256;              // tile elements per thread
    extern __shared__ float lds[];
    float* ad = lds;                       // [AD][ADC]
    float* th = lds + AD * ADC;            // [THR][THW]
    const int plane = blockIdx.z;
    const int qy0 = blockIdx.y * TS, qx0 = blockIdx.x * TS;
    const int VH = H - halo, VW = W - halo;
    const float* xp = X + (long)plane * H * W;
    const float* yp = Y + (long)plane * H * W;
    const float* ap = ADJ + (long)plane * 3 * H * W;
    const int tid = threadIdx.x;
    const float wt = wts[plane];
    // tile elements of this thread: offset in a map (or -1 outside the valid region) and LDS slot
    int goff[NL], lslot[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int i = tid + u * 256;
        const int r = i / AD, c = i - r * AD;
        const int py = qy0 - halo + r, px = qx0 - halo + c;
        lslot[u] = i < AD * AD ? r * ADC + c : -1;
        goff[u] = (i < AD * AD && py >= 0 && py < VH && px >= 0 && px < VW) ? py * W + px : -1;
    }
    float pre[NL];
#define SB_LOAD(Q)                                                                                                \
    _Pragma("unroll") for (int u = 0; u < NL; ++u) pre[u] = goff[u] >= 0 ? wt * ap[(long)(Q) * H * W + goff[u]] : 0.f;
    SB_LOAD(0)
    // vertical item of this thread (the first TS * NS threads): column c, rows r0 .. r0 + SEG - 1; its x / y values
    const bool vitem = tid < TS * NS;
    const int vc = tid % TS, vr0 = (tid / TS) * SEG;
    float xv[SEG], yv[SEG], gsum[SEG];
#pragma unroll
    for (int o = 0; o < SEG; ++o) {
        const int gy = qy0 + vr0 + o, gx = qx0 + vc;
        const bool ok = vitem && gy < H && gx < W;
        xv[o] = ok ? __fmul_rn(xp[(long)gy * W + gx], in_mul) : 0.f;
        yv[o] = ok ? __fmul_rn(yp[(long)gy * W + gx], in_mul) : 0.f;
        gsum[o] = 0.f;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int u = 0; u < NL; ++u) if (lslot[u] >= 0) ad[lslot[u]] = pre[u];
        __syncthreads();
        if (q < 2) { SB_LOAD(q + 1) }
        // transposed filter, horizontal: th[r][x] = sum_t g[t] * ad[r][x + halo - t]
        for (int it = tid; it < AD * NS; it += 256) {
            const int r = it / NS, c0 = (it % NS) * SEG;
            float v[SEG + K - 1];
#pragma unroll
            for (int j = 0; j < SEG + K - 1; ++j) v[j] = ad[r * ADC + c0 + j];
#pragma unroll
            for (int o = 0; o < SEG; ++o) {
                float s_ = 0.f;
#pragma unroll
                for (int t = 0; t < K; ++t) s_ = fmaf(win.g[t], v[o + halo - t], s_);
                th[r * THW + c0 + o] = s_;
            }
        }
        __syncthreads();
        if (vitem) {
            float v[SEG + K - 1];
#pragma unroll
            for (int j = 0; j < SEG + K - 1; ++j) v[j] = th[(vr0 + j) * THW + vc];
#pragma unroll
            for (int o = 0; o < SEG; ++o) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < K; ++t) a = fmaf(win.g[t], v[o + halo - t], a);
                // g = s0 + 2 x s1 + y s2, summed in that order (as the all-at-once kernel did)
                if (q == 0) gsum[o] = a;
                else if (q == 1) gsum[o] = gsum[o] + 2.f * xv[o] * a;
                else gsum[o] = gsum[o] + yv[o] * a;
            }
        }
    }
#undef SB_LOAD
    if (!vitem) return;
    const float l1c = l1_coef_p ? l1_coef_p[0] : 0.f;
    constexpr int r5 = K / 2;
    float wsum = 0.f;
#pragma unroll
    for (int t = 0; t < K; ++t) wsum += win.g[t];
    float* dxp = dX + (long)plane * H * W;
#pragma unroll
    for (int o = 0; o < SEG; ++o) {
        const int gy = qy0 + vr0 + o, gx = qx0 + vc;
        if (gy >= H || gx >= W) continue;
        float g = gsum[o];
        if (dcoarse) {
            const int cy = (gy + (H & 1)) >> 1, cx = (gx + (W & 1)) >> 1;
            g += 0.25f * dcoarse[((long)plane * HC + cy) * WC + cx];
        }
        if (l1c != 0.f) {
            float sy = wsum, sx = wsum;
            if (gy < r5 || gy >= H - r5) {
                sy = 0.f;
#pragma unroll
                for (int t = 0; t < K; ++t) { const int yy = gy + t - r5; if (yy >= 0 && yy < H) sy += win.g[t]; }
            }
            if (gx < r5 || gx >= W - r5) {
                sx = 0.f;
#pragma unroll
                for (int t = 0; t < K; ++t) { const int xx = gx + t - r5; if (xx >= 0 && xx < W) sx += win.g[t]; }
            }
            const float d = xv[o] - yv[o];
            g += l1c * sy * sx * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
        }
        dxp[(long)gy * W + gx] = __fmul_rn(g, in_mul);      // chain rule of x * in_mul (what autograd does for x / divisor)
    }
}

template <int K>
__global__ __launch_bounds__(256, 2) void ssim_bwd_k(const float* __restrict__ X, const float* __restrict__ Y, int H, int W, Win win,
                                                  float C1, float C2, const float* __restrict__ wts, int use_ssim,
                                                  const float* __restrict__ dcoarse, int HC, int WC, const float* l1_coef_p,
                                                  float* __restrict__ dX) {
    constexpr int halo = K - 1, AD = TS + halo, IN = TS + 2 * halo;
    constexpr int SA = 7, NSA = (AD + SA - 1) / SA;       // segments over the AD-wide adjoint region (42 = 6 x 7 for K = 11)
    constexpr int SEG = 8, NS = TS / SEG;                 // segments over the TS-wide output region
    constexpr int ADP = NSA * SA;                         // padded row length of the h-pass image
    extern __shared__ float lds[];
    float* xs = lds;                       // [IN][IN + pad]
    constexpr int INP = IN + SA;           // row pitch with room for the last (partial) segment's window
    float* ys = xs + IN * INP;
    float* hp = ys + IN * INP;             // [5][IN][ADP]  -> later th [3][AD][TS]
    float* ad = lds;                       // [3][AD][AD] adjoint maps: alias xs / ys, which are dead after the horizontal pass
    constexpr int ADR = AD, ADC = AD;
    static_assert(3 * AD * AD <= 2 * IN * INP, "adjoint maps must fit into the input images");
    const int plane = blockIdx.z;
    const int qy0 = blockIdx.y * TS, qx0 = blockIdx.x * TS;
    const int VH = H - halo, VW = W - halo;
    const float* xp = X + (long)plane * H * W;
    const float* yp = Y + (long)plane * H * W;
    const int tid = threadIdx.x;
    const float wt = wts[plane];
    for (int i = tid; i < IN * INP; i += 256) {
        const int r = i / INP, c = i % INP;
        const int gy = qy0 - halo + r, gx = qx0 - halo + c;
        float xv = 0.f, yv = 0.f;
        if (c < IN && gy >= 0 && gy < H && gx >= 0 && gx < W) { xv = xp[(long)gy * W + gx]; yv = yp[(long)gy * W + gx]; }
        xs[i] = xv; ys[i] = yv;
    }
    __syncthreads();
    // horizontal pass over [IN rows][AD cols]
    for (int it = tid; it < IN * NSA; it += 256) {
        const int r = it / NSA, c0 = (it % NSA) * SA;
        float v[5][SA + K - 1];
#pragma unroll
        for (int j = 0; j < SA + K - 1; ++j) {
            const float a = xs[r * INP + c0 + j], b = ys[r * INP + c0 + j];
            v[0][j] = a; v[1][j] = b; v[2][j] = a * a; v[3][j] = b * b; v[4][j] = a * b;
        }
        float o[5][SA];
        fir_seg<K, SA, 5>(v, win, o);
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < SA; ++j) hp[q * IN * ADP + r * ADP + c0 + j] = o[q][j];
    }
    __syncthreads();
    // vertical pass + adjoint maps at [AD rows][AD cols]; stored at column offset 0, rows 0..AD-1 of the padded image
    for (int it = tid; it < AD * NSA; it += 256) {
        const int c = it % AD, r0 = (it / AD) * SA;
        float v[5][SA + K - 1];
#pragma unroll
        for (int q = 0; q < 5; ++q)
#pragma unroll
            for (int j = 0; j < SA + K - 1; ++j) {
                const int rr = r0 + j;
                v[q][j] = rr < IN ? hp[q * IN * ADP + rr * ADP + c] : 0.f;
            }
        float m[5][SA];
        fir_seg<K, SA, 5>(v, win, m);
#pragma unroll
        for (int j = 0; j < SA; ++j) {
            const int r = r0 + j;
            if (r >= AD) continue;
            const int py = qy0 - halo + r, px = qx0 - halo + c;
            float a = 0.f, b = 0.f, cc = 0.f;
            if (py >= 0 && py < VH && px >= 0 && px < VW) {
                const float mx = m[0][j], my = m[1][j];
                const float sxx = m[2][j] - mx * mx, syy = m[3][j] - my * my, sxy = m[4][j] - mx * my;
                const float Dcs = sxx + syy + C2, cs = (2.f * sxy + C2) / Dcs;
                const float dcs_dmx = 2.f * (cs * mx - my) / Dcs, dcs_dexx = -cs / Dcs, dcs_dexy = 2.f / Dcs;
                if (use_ssim) {
                    const float Dl = mx * mx + my * my + C1, lum = (2.f * mx * my + C1) / Dl;
                    const float dl_dmx = 2.f * (my - lum * mx) / Dl;
                    a = wt * (lum * dcs_dmx + cs * dl_dmx); b = wt * lum * dcs_dexx; cc = wt * lum * dcs_dexy;
                } else {
                    a = wt * dcs_dmx; b = wt * dcs_dexx; cc = wt * dcs_dexy;
                }
            }
            ad[0 * ADR * ADC + r * ADC + c] = a; ad[1 * ADR * ADC + r * ADC + c] = b; ad[2 * ADR * ADC + r * ADC + c] = cc;
        }
    }
    __syncthreads();
    // transposed filter, horizontal: th[r][q] = sum_t g[t] * ad[r][q + halo - t] = sum_u g[K-1-u] * ad[r][q + u]
    float* th = hp;                        // [3][AD + SEG][TS]
    constexpr int THR = AD + SEG;
    for (int it = tid; it < AD * NS; it += 256) {
        const int r = it / NS, c0 = (it % NS) * SEG;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v[SEG + K - 1];
#pragma unroll
            for (int j = 0; j < SEG + K - 1; ++j) v[j] = ad[q * ADR * ADC + r * ADC + c0 + j];
#pragma unroll
            for (int o = 0; o < SEG; ++o) {
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < K; ++t) s = fmaf(win.g[t], v[o + halo - t], s);
                th[q * THR * TS + r * TS + c0 + o] = s;
            }
        }
    }
    __syncthreads();
    const float l1c = l1_coef_p ? l1_coef_p[0] : 0.f;
    constexpr int r5 = K / 2;
    float wsum = 0.f;
#pragma unroll
    for (int t = 0; t < K; ++t) wsum += win.g[t];
    float* dxp = dX + (long)plane * H * W;
    for (int it = tid; it < TS * NS; it += 256) {
        const int c = it % TS, r0 = (it / TS) * SEG;
        float s[3][SEG];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float v[SEG + K - 1];
#pragma unroll
            for (int j = 0; j < SEG + K - 1; ++j) v[j] = th[q * THR * TS + (r0 + j) * TS + c];
#pragma unroll
            for (int o = 0; o < SEG; ++o) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < K; ++t) a = fmaf(win.g[t], v[o + halo - t], a);
                s[q][o] = a;
            }
        }
#pragma unroll
        for (int o = 0; o < SEG; ++o) {
            const int r = r0 + o;
            const int gy = qy0 + r, gx = qx0 + c;
            if (gy >= H || gx >= W) continue;
            const float xv = xp[(long)gy * W + gx], yv = yp[(long)gy * W + gx];      // (the LDS copies were overwritten by `ad`)
            float g = s[0][o] + 2.f * xv * s[1][o] + yv * s[2][o];
            if (dcoarse) {
                const int cy = (gy + (H & 1)) >> 1, cx = (gx + (W & 1)) >> 1;
                g += 0.25f * dcoarse[((long)plane * HC + cy) * WC + cx];
            }
            if (l1c != 0.f) {
                // border weights: the whole window sum (added in the same order) for interior pixels, clipped sums on the border
                float sy = wsum, sx = wsum;
                if (gy < r5 || gy >= H - r5) {
                    sy = 0.f;
#pragma unroll
                    for (int t = 0; t < K; ++t) { const int yy = gy + t - r5; if (yy >= 0 && yy < H) sy += win.g[t]; }
                }
                if (gx < r5 || gx >= W - r5) {
                    sx = 0.f;
#pragma unroll
                    for (int t = 0; t < K; ++t) { const int xx = gx + t - r5; if (xx >= 0 && xx < W) sx += win.g[t]; }
                }
                const float d = xv - yv;
                g += l1c * sy * sx * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            }
            dxp[(long)gy * W + gx] = g;
        }
    }
}

template <int K> constexpr size_t bwd_lds_k() {
    constexpr int halo = K - 1, AD = TS + halo, IN = TS + 2 * halo, SA = 7, NSA = (AD + SA - 1) / SA, ADP = NSA * SA, INP = IN + SA;
    return (size_t)(2 * IN * INP + 5 * IN * ADP) * sizeof(float);
}

Win make_win(const float* g, int k) {
    Win w; w.k = k;
    for (int i = 0; i < MAXW; ++i) w.g[i] = i < k ? g[i] : 0.f;
    return w;
}

}  // namespace

extern "C" {

int pssr_ssim_level_fwd(const float* x, const float* y, int planes, int h, int w, const float* win_host, int k,
                        float c1, float c2, double* sums, double* l1_sum, pssr_stream_t s) {
    PSSR_CHECK(x && y && sums && win_host && planes > 0 && k > 0 && k <= MAXW && (k & 1), PSSR_ERR_ARG, "ssim_level_fwd: bad args");
    PSSR_CHECK(h >= k && w >= k, PSSR_ERR_ARG, "ssim_level_fwd: image %dx%d smaller than window %d", h, w, k);
    const int IN = TS + k - 1;
    const size_t lds = (size_t)(2 * IN * IN + 5 * IN * (TS + 1)) * sizeof(float);
    // when the L1 term is requested the tiles must cover the whole image, not only the valid region
    const int eh = l1_sum ? h : h - k + 1, ew = l1_sum ? w : w - k + 1;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)ssim_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        (void)hipFuncSetAttribute((const void*)ssim_fwd_k<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        attr = true;
    }
    if (k == 11)
        hipLaunchKernelGGL(ssim_fwd_k<11>, dim3(cdiv(ew, TS), cdiv(eh, TS), planes), dim3(256), lds, (hipStream_t)s, x, y, h, w,
                           make_win(win_host, k), c1, c2, sums, l1_sum);
    else
        hipLaunchKernelGGL(ssim_fwd_kernel<0>, dim3(cdiv(ew, TS), cdiv(eh, TS), planes), dim3(256), lds, (hipStream_t)s, x, y, h, w,
                           make_win(win_host, k), c1, c2, sums, l1_sum);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_ssim_level_fwd_adj(const float* x, const float* y, float in_div, int planes, int h, int w, const float* win_host, int k, float c1,
                            float c2, int use_ssim, double* sums, double* l1_sum, int stripes, int64_t stripe_stride, float* adj,
                            pssr_stream_t s) {
    PSSR_CHECK(x && y && sums && win_host && planes > 0 && adj && k == 11, PSSR_ERR_ARG,
               "ssim_level_fwd_adj: needs the adjoint buffer and the 11-tap window (k=%d)", k);
    PSSR_CHECK(h >= k && w >= k && stripes >= 1 && (stripes == 1 || stripe_stride > 0) && in_div > 0.f, PSSR_ERR_ARG, "ssim_level_fwd_adj: bad size / stripes / in_div");
    const int IN = TS + k - 1;
    const size_t lds = (size_t)(2 * IN * IN + IN * (TS + 1)) * sizeof(float);
    const int eh = l1_sum ? h : h - k + 1, ew = l1_sum ? w : w - k + 1;
    hipLaunchKernelGGL(ssim_fwd_adj_k<11>, dim3(cdiv(ew, TS), cdiv(eh, TS), planes), dim3(256), lds, (hipStream_t)s, x, y, h, w,
                       make_win(win_host, k), c1, c2, sums, l1_sum, stripes, (long)stripe_stride, adj, use_ssim, in_div == 1.f ? 1.f : 1.f / in_div);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_ssim_level_bwd_adj(const float* x, const float* y, float in_div, const float* adj, int planes, int h, int w, const float* win_host,
                            int k, const float* wts, const float* dcoarse, int hc, int wc, const float* l1_coef, float* dx, pssr_stream_t s) {
    PSSR_CHECK(x && y && adj && wts && dx && win_host && planes > 0 && k == 11, PSSR_ERR_ARG, "ssim_level_bwd_adj: bad args (k=%d)", k);
    constexpr int AD = TS + 10;
    const size_t lds = (size_t)(AD * (AD + 1) + (AD + 8) * (TS + 1)) * sizeof(float);
    hipLaunchKernelGGL(ssim_bwd_adj_k<11>, dim3(cdiv(w, TS), cdiv(h, TS), planes), dim3(256), lds, (hipStream_t)s, x, y, adj, h, w,
                       make_win(win_host, k), wts, dcoarse, hc, wc, l1_coef, dx, in_div == 1.f ? 1.f : 1.f / in_div);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_avgpool2_planes(const float* in, float* out, int planes, int h, int w, pssr_stream_t s) {
    return pssr_avgpool2_planes_div(in, 1.f, out, planes, h, w, s);
}

int pssr_avgpool2_planes_div(const float* in, float in_div, float* out, int planes, int h, int w, pssr_stream_t s) {
    PSSR_CHECK(in && out && planes > 0 && h > 0 && w > 0 && in_div > 0.f, PSSR_ERR_ARG, "avgpool2_planes: bad args");
    const int ho = (h + 2 * (h & 1) - 2) / 2 + 1, wo = (w + 2 * (w & 1) - 2) / 2 + 1;
    const long total = (long)planes * ho * wo;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(avgpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, in, out, in, out, planes, h, w, ho, wo, in_div == 1.f ? 1.f : 1.f / in_div);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_avgpool2_pair_div(const float* x, const float* y, float in_div, float* xo, float* yo, int planes, int h, int w, pssr_stream_t s) {
    PSSR_CHECK(x && y && xo && yo && planes > 0 && h > 0 && w > 0 && in_div > 0.f, PSSR_ERR_ARG, "avgpool2_pair: bad args");
    const int ho = (h + 2 * (h & 1) - 2) / 2 + 1, wo = (w + 2 * (w & 1) - 2) / 2 + 1;
    const long total = (long)planes * ho * wo;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(avgpool_kernel, dim3(blocks, 2), dim3(256), 0, (hipStream_t)s, x, xo, y, yo, planes, h, w, ho, wo, in_div == 1.f ? 1.f : 1.f / in_div);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_msssim_weights(const double* sums, int levels, int planes, const double* nvalid, const float* level_weights, int ms,
                        float mix, const double* l1_sum, double l1_numel, const float* grad_out, float* loss_out, float* wts,
                        float* l1_coef, pssr_stream_t s) {
    PSSR_CHECK(sums && nvalid && loss_out && wts && l1_coef && levels > 0 && levels <= 8 && planes > 0, PSSR_ERR_ARG, "msssim_weights: bad args");
    hipLaunchKernelGGL(weights_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, sums, levels, planes, nvalid, level_weights, ms, mix,
                       l1_sum, l1_numel, grad_out, loss_out, wts, l1_coef, 1, 0L, (double*)nullptr);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_msssim_weights_striped(const double* sums, int stripes, int64_t stripe_stride, double* folded, int levels, int planes,
                                const double* nvalid, const float* level_weights, int ms, float mix, const double* l1_sum,
                                double l1_numel, const float* grad_out, float* loss_out, float* wts, float* l1_coef, pssr_stream_t s) {
    PSSR_CHECK(sums && folded && stripes >= 1 && stripe_stride > 0 && nvalid && loss_out && wts && l1_coef && levels > 0 && levels <= 8 && planes > 0,
               PSSR_ERR_ARG, "msssim_weights_striped: bad args");
    hipLaunchKernelGGL(weights_kernel, dim3(1), dim3(256), 0, (hipStream_t)s, sums, levels, planes, nvalid, level_weights, ms, mix,
                       l1_sum, l1_numel, grad_out, loss_out, wts, l1_coef, stripes, (long)stripe_stride, folded);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

int pssr_ssim_level_bwd(const float* x, const float* y, int planes, int h, int w, const float* win_host, int k, float c1, float c2,
                        const float* wts, int use_ssim, const float* dcoarse, int hc, int wc, const float* l1_coef, float* dx,
                        pssr_stream_t s) {
    PSSR_CHECK(x && y && wts && dx && win_host && planes > 0 && k > 0 && k <= MAXW && (k & 1), PSSR_ERR_ARG, "ssim_level_bwd: bad args");
    const int halo = k - 1, AD = TS + halo, IN = TS + 2 * halo;
    const size_t lds = (size_t)(2 * IN * IN + 5 * IN * AD + 3 * AD * AD) * sizeof(float);
    PSSR_CHECK(lds <= 156 * 1024, PSSR_ERR_UNSUPPORTED, "ssim_level_bwd: window %d needs %zu bytes of LDS", k, lds);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)ssim_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        (void)hipFuncSetAttribute((const void*)ssim_bwd_k<11>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        attr = true;
    }
    if (k == 11)
        hipLaunchKernelGGL(ssim_bwd_k<11>, dim3(cdiv(w, TS), cdiv(h, TS), planes), dim3(256), bwd_lds_k<11>(), (hipStream_t)s, x, y, h, w,
                           make_win(win_host, k), c1, c2, wts, use_ssim, dcoarse, hc, wc, l1_coef, dx);
    else
        hipLaunchKernelGGL(ssim_bwd_kernel, dim3(cdiv(w, TS), cdiv(h, TS), planes), dim3(256), lds, (hipStream_t)s, x, y, h, w,
                           make_win(win_host, k), c1, c2, wts, use_ssim, dcoarse, hc, wc, l1_coef, dx);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

}  // extern "C"
