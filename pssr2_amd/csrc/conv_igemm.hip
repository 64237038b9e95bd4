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
#include <type_traits>

#include "common.h"

namespace {

struct ConvArgs {
    int N, H, W;
    int tiles_x, tiles_y, tiles_n;
    const void* in[2]; int in_cs[2]; int in_co[2]; int nchunks[2]; int taps[2]; const void* w[2];
    int prologue; const float* pro_scale; const float* pro_shift;
    void* out; int out_cs, out_co, cout, n_pad;
    const float* bias;
    int epi, flags;
    const void* aux; int aux_cs, aux_co;
    const float* aux_scale; const float* aux_shift; const float* aux_mean; const float* aux_invstd;
    double* stats;
    int in0_blk, out_blk, aux_blk;
    float out_scale, out_shift;
};

template <int GEO> struct Geo;
template <> struct Geo<0> { static constexpr int TWL = 4, THL = 3; };   // 8 x 16, 1 image
template <> struct Geo<1> { static constexpr int TWL = 3, THL = 3; };   // 8 x 8,  2 images
template <> struct Geo<2> { static constexpr int TWL = 2, THL = 2; };   // 4 x 4,  8 images
template <> struct Geo<3> { static constexpr int TWL = 1, THL = 1; };   // 2 x 2, 32 images
template <> struct Geo<4> { static constexpr int TWL = 0, THL = 0; };   // 1 x 1, 128 images

template <int BN, int GEO> struct Cfg {
    static constexpr int TWL = Geo<GEO>::TWL, THL = Geo<GEO>::THL;
    static constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    static constexpr int HW2 = TW + 2, HPI = (TH + 2) * (TW + 2), HP = NI * HPI;
    static constexpr int A_ITEMS = (2 * HP + 255) / 256;
    static constexpr int A_BYTES = HP * 32;
    static constexpr int B_BYTES = 9 * BN * 32;
    static constexpr int B_ITEMS = (9 * BN * 2 + 255) / 256;
    static constexpr int WM = (BN == 32) ? 4 : 2, WN = 4 / WM;
    static constexpr int MI = 4 / WM, NJ = BN / (32 * WN);
    static constexpr int E_BYTES = WM * 32 * BN * 4;
    static constexpr int RED_BYTES = 4 * BN * 2 * 4;
    static constexpr int MAIN_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = (MAIN_BYTES > E_BYTES + RED_BYTES) ? MAIN_BYTES : (E_BYTES + RED_BYTES);
};

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

    const int bid = blockIdx.x;
    const int tn = bid % p.tiles_n;
    int tmi = bid / p.tiles_n;
    const int tile_x = tmi % p.tiles_x; tmi /= p.tiles_x;
    const int tile_y = tmi % p.tiles_y;
    const int tile_i = tmi / p.tiles_y;
    const int x0 = tile_x << C::TWL, y0 = tile_y << C::THL, img0 = tile_i * C::NI, n0 = tn * BN;

    // ---- per-thread staging descriptors for the input halo (same pixels for every chunk)
    long a_pix[C::A_ITEMS], a_pix1[C::A_ITEMS];   // source 0 (possibly blocked order); source 1 is always plain NHWC
    int a_lds[C::A_ITEMS];
    bool a_ok[C::A_ITEMS];
    int a_half[C::A_ITEMS];
#pragma unroll
    for (int it = 0; it < C::A_ITEMS; ++it) {
        const int idx = tid + it * 256;
        const int pp = idx >> 1, half = idx & 1;
        const int img = pp / C::HPI, rem = pp % C::HPI;
        const int hy = rem / C::HW2, hx = rem % C::HW2;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1, gi = img0 + img;
        const bool inb = (idx < 2 * C::HP) && gi < p.N && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        a_ok[it] = inb;
        a_pix[it] = inb ? pix_index(gi, gy, gx, p.H, p.W, p.in0_blk) : 0;
        a_pix1[it] = inb ? ((long)gi * p.H + gy) * p.W + gx : 0;
        a_lds[it] = (idx < 2 * C::HP) ? (pp * 32 + ((half ^ ((pp >> 3) & 1)) << 4)) : -1;
        a_half[it] = half;
    }
    // ---- fragment read bases
    int a_p0[C::MI];
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi) {
        const int m = wm * C::MI * 32 + mi * 32 + r;
        const int tx = m & (C::TW - 1), ty = (m >> C::TWL) & (C::TH - 1), img = m >> (C::TWL + C::THL);
        a_p0[mi] = img * C::HPI + ty * C::HW2 + tx;
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

    // NB: the three phases are macros, not lambdas: with lambdas hipcc keeps a_reg/b_reg in scratch.
#define PSSR_ISSUE(I, TAPS)                                                                                       \
    {                                                                                                             \
        const int s_ = ((I) >= p.nchunks[0]) ? 1 : 0;                                                             \
        const int chunk_ = s_ ? (I) - p.nchunks[0] : (I);                                                         \
        const T* in_ = (const T*)p.in[s_];                                                                        \
        const int cs_ = p.in_cs[s_], co_ = p.in_co[s_] + chunk_ * KCH;                                            \
        /* unconditional loads (out-of-image items read pixel 0 and are zeroed at commit): a guarded load   \
           makes hipcc park the staging registers in scratch behind a vmcnt(0) each */                      \
        _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it)                                                 \
            a_reg[it] = *(const u32x4*)(in_ + (s_ ? a_pix1[it] : a_pix[it]) * cs_ + co_ + a_half[it] * EPS);      \
        const char* wsrc_ = (const char*)p.w[s_] + ((long)chunk_ * (TAPS) * p.n_pad + n0) * 32;                   \
        _Pragma("unroll") for (int it = 0; it < C::B_ITEMS; ++it) {                                               \
            int idx = tid + it * 256;                                                                             \
            idx = idx < (TAPS) * BN * 2 ? idx : (TAPS) * BN * 2 - 1;                                              \
            const int tap = idx / (BN * 2), rem = idx % (BN * 2);                                                 \
            b_reg[it] = *(const u32x4*)(wsrc_ + (long)tap * p.n_pad * 32 + rem * 16);                             \
        }                                                                                                         \
    }
#define PSSR_COMMIT(I, TAPS)                                                                                      \
    {                                                                                                             \
        const int s_ = ((I) >= p.nchunks[0]) ? 1 : 0;                                                             \
        const int chunk_ = s_ ? (I) - p.nchunks[0] : (I);                                                         \
        const bool pro_ = (s_ == 0) && (p.prologue == PSSR_PRO_BN_RELU);                                          \
        const bool gelu_ = (s_ == 0) && (p.prologue == PSSR_PRO_GELU);                                            \
        _Pragma("unroll") for (int it = 0; it < C::A_ITEMS; ++it) {                                               \
            if (a_lds[it] >= 0) {                                                                                 \
                u32x4 v = a_reg[it];                                                                              \
                if (!a_ok[it]) v = u32x4{0u, 0u, 0u, 0u};                                          \
                if (pro_ && a_ok[it]) {                                                                           \
                    const int c0 = chunk_ * KCH + a_half[it] * EPS;                                               \
                    float f[EPS];                                                                                 \
                    X::unpack(v, f);                                                                              \
                    _Pragma("unroll") for (int e = 0; e < EPS; e += 4) {                                          \
                        const float4 sc = *(const float4*)(p.pro_scale + c0 + e);                                 \
                        const float4 sh = *(const float4*)(p.pro_shift + c0 + e);                                 \
                        f[e + 0] = fmaxf(fmaf(f[e + 0], sc.x, sh.x), 0.f);                                        \
                        f[e + 1] = fmaxf(fmaf(f[e + 1], sc.y, sh.y), 0.f);                                        \
                        f[e + 2] = fmaxf(fmaf(f[e + 2], sc.z, sh.z), 0.f);                                        \
                        f[e + 3] = fmaxf(fmaf(f[e + 3], sc.w, sh.w), 0.f);                                        \
                    }                                                                                             \
                    v = X::pack(f);                                                                               \
                }                                                                                                 \
                if (gelu_) {      /* gelu(0) == 0: padding stays zero */                                          \
                    float f[EPS];                                                                                 \
                    X::unpack(v, f);                                                                              \
                    _Pragma("unroll") for (int e = 0; e < EPS; ++e) f[e] = gelu_f(f[e]);                          \
                    v = X::pack(f);                                                                               \
                }                                                                                                 \
                *(u32x4*)(As + a_lds[it]) = v;                                                                    \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < C::B_ITEMS; ++it) {                                               \
            const int idx = tid + it * 256;                                                                       \
            if (idx < (TAPS) * BN * 2) *(u32x4*)(Bs + idx * 16) = b_reg[it];                                      \
        }                                                                                                         \
    }
#define PSSR_COMPUTE(TAPS)                                                                                        \
    {                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < (TAPS); ++t) {                                                      \
            const int ky = ((TAPS) == 9) ? t / 3 : 1, kx = ((TAPS) == 9) ? t % 3 : 1;                             \
            u32x4 af[C::MI], bf[C::NJ];                                                                           \
            _Pragma("unroll") for (int mi = 0; mi < C::MI; ++mi) {                                                \
                const int pa = a_p0[mi] + ky * C::HW2 + kx;                                                       \
                af[mi] = *(const u32x4*)(As + pa * 32 + ((h ^ ((pa >> 3) & 1)) << 4));                            \
            }                                                                                                     \
            _Pragma("unroll") for (int nj = 0; nj < C::NJ; ++nj) bf[nj] = *(const u32x4*)(Bs + t * BN * 32 + b_off[nj]); \
            _Pragma("unroll") for (int mi = 0; mi < C::MI; ++mi)                                                  \
                _Pragma("unroll") for (int nj = 0; nj < C::NJ; ++nj) X::mma(acc[mi][nj], af[mi], bf[nj]);         \
        }                                                                                                         \
    }

    // source 0 (TAPS0 taps) then the optional 1x1 source 1; the loads of the next chunk fly during the MFMAs
    const int n0c = p.nchunks[0];
    PSSR_ISSUE(0, TAPS0)
    for (int i = 0; i < n0c; ++i) {
        PSSR_COMMIT(i, TAPS0)
        __syncthreads();
        if (i + 1 < n0c) PSSR_ISSUE(i + 1, TAPS0)
        else if (i + 1 < total) PSSR_ISSUE(i + 1, 1)
        PSSR_COMPUTE(TAPS0)
        __syncthreads();
    }
    for (int i = n0c; i < total; ++i) {
        PSSR_COMMIT(i, 1)
        __syncthreads();
        if (i + 1 < total) PSSR_ISSUE(i + 1, 1)
        PSSR_COMPUTE(1)
        __syncthreads();
    }
#undef PSSR_ISSUE
#undef PSSR_COMMIT
#undef PSSR_COMPUTE

    // ------------------------------------------------------------------ epilogue
    float* Es = (float*)smem;                       // [WM*32][BN] f32
    float* Red = (float*)(smem + C::E_BYTES);       // [4 waves][BN][2]
    constexpr int CG = BN / 4;                      // 4-channel groups per row
    constexpr int PASSES = C::WM * 32 * CG / 256;
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
        if (mi) __syncthreads();
#pragma unroll
        for (int nj = 0; nj < C::NJ; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int col = wn * C::NJ * 32 + nj * 32 + r;
                Es[row * BN + col] = acc[mi][nj][e];
            }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int piece = tid + ps * 256;
            const int row = piece / CG;
            const int m = (row >> 5) * C::MI * 32 + mi * 32 + (row & 31);
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
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= gelu_grad_f(a[e]);
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
            atomicAdd(st + n0 + tid, (double)t1);
            atomicAdd(st + p.cout + n0 + tid, (double)t2);
        }
    }
}

template <typename T, int BN, int GEO, int TAPS0>
int launch_t(const ConvArgs& a, hipStream_t stream) {
    using C = Cfg<BN, GEO>;
    ConvArgs p = a;
    p.tiles_x = cdiv(a.W, C::TW);
    p.tiles_y = cdiv(a.H, C::TH);
    p.tiles_n = cdiv(a.cout, BN);
    const long blocks = (long)p.tiles_x * p.tiles_y * cdiv(a.N, C::NI) * p.tiles_n;
    PSSR_CHECK(blocks > 0 && blocks < (1L << 31), PSSR_ERR_ARG, "conv2d: bad grid %ld", blocks);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BN, GEO, TAPS0>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_igemm_kernel<T, BN, GEO, TAPS0>), dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, stream, p);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

template <typename T, int BN, int GEO>
int launch(const ConvArgs& a, hipStream_t s) {
    return a.taps[0] == 9 ? launch_t<T, BN, GEO, 9>(a, s) : launch_t<T, BN, GEO, 1>(a, s);
}

template <typename T, int BN>
int launch_geo(const ConvArgs& a, hipStream_t s) {
    const int w = a.W;
    if (w > 8) return launch<T, BN, 0>(a, s);
    if (w > 4) return launch<T, BN, 1>(a, s);
    if (w > 2) return launch<T, BN, 2>(a, s);
    if (w > 1) return launch<T, BN, 3>(a, s);
    return launch<T, BN, 4>(a, s);
}

template <typename T>
int launch_bn(const ConvArgs& a, hipStream_t s) {
    if (a.cout > 64) return launch_geo<T, 128>(a, s);
    if (a.cout > 32) return launch_geo<T, 64>(a, s);
    return launch_geo<T, 32>(a, s);
}

}  // namespace

extern "C" int pssr_conv2d(const pssr_conv_desc* d, pssr_stream_t stream) {
    PSSR_CHECK(d != nullptr, PSSR_ERR_ARG, "conv2d: null desc");
    PSSR_CHECK(d->dtype == PSSR_F32 || d->dtype == PSSR_BF16, PSSR_ERR_ARG, "conv2d: bad dtype %d", d->dtype);
    const int kch = d->dtype == PSSR_BF16 ? 16 : 8;
    const int esz = d->dtype == PSSR_BF16 ? 2 : 4;
    PSSR_CHECK(d->n > 0 && d->h > 0 && d->w > 0, PSSR_ERR_ARG, "conv2d: bad shape %dx%dx%d", d->n, d->h, d->w);
    PSSR_CHECK(d->in0 && d->w0 && d->out, PSSR_ERR_ARG, "conv2d: null pointer");
    PSSR_CHECK(d->cin0 > 0 && d->cin0 % kch == 0, PSSR_ERR_ARG, "conv2d: cin0=%d must be a positive multiple of %d", d->cin0, kch);
    PSSR_CHECK(d->taps0 == 9 || d->taps0 == 1, PSSR_ERR_ARG, "conv2d: taps0=%d", d->taps0);
    PSSR_CHECK((d->in0_cstride * esz) % 16 == 0 && (d->in0_coff * esz) % 16 == 0, PSSR_ERR_ARG, "conv2d: in0 stride/offset not 16-byte aligned");
    PSSR_CHECK(d->in0_coff + d->cin0 <= d->in0_cstride, PSSR_ERR_ARG, "conv2d: in0 slice exceeds stride");
    if (d->cin1) {
        PSSR_CHECK(d->in1 && d->w1 && d->cin1 % kch == 0 && d->taps1 == 1, PSSR_ERR_ARG, "conv2d: source 1 must be a 1x1 (or flat-K) source");
        PSSR_CHECK((d->in1_cstride * esz) % 16 == 0 && (d->in1_coff * esz) % 16 == 0 && d->in1_coff + d->cin1 <= d->in1_cstride, PSSR_ERR_ARG, "conv2d: in1 stride/offset");
    }
    if (d->epilogue == PSSR_EPI_FINAL) {
        PSSR_CHECK(d->cout > 0 && d->cout <= 32 && d->flags == 0, PSSR_ERR_ARG, "conv2d: EPI_FINAL needs 0 < cout <= 32 and no flags");
    } else {
        PSSR_CHECK(d->cout > 0 && d->cout % 4 == 0, PSSR_ERR_ARG, "conv2d: cout=%d must be a positive multiple of 4", d->cout);
        PSSR_CHECK(d->out_coff % 4 == 0 && d->out_cstride % 4 == 0 && d->out_coff + d->cout <= d->out_cstride, PSSR_ERR_ARG, "conv2d: out stride/offset");
    }
    PSSR_CHECK(d->n_pad % 128 == 0 && d->n_pad >= d->cout, PSSR_ERR_ARG, "conv2d: n_pad=%d", d->n_pad);
    PSSR_CHECK(d->prologue >= 0 && d->prologue <= PSSR_PRO_GELU, PSSR_ERR_ARG, "conv2d: prologue=%d", d->prologue);
    PSSR_CHECK(d->prologue != PSSR_PRO_BN_RELU || (d->pro_scale && d->pro_shift), PSSR_ERR_ARG, "conv2d: prologue needs scale/shift");
    PSSR_CHECK(d->epilogue >= 0 && d->epilogue <= 4, PSSR_ERR_ARG, "conv2d: epilogue=%d", d->epilogue);
    PSSR_CHECK(d->in0_blk >= 0 && d->out_blk >= 0 && d->aux_blk >= 0 && d->in0_blk <= 3 && d->out_blk <= 3 && d->aux_blk <= 3, PSSR_ERR_ARG, "conv2d: blocked order");
    {
        const int mb = d->in0_blk > d->out_blk ? (d->in0_blk > d->aux_blk ? d->in0_blk : d->aux_blk) : (d->out_blk > d->aux_blk ? d->out_blk : d->aux_blk);
        PSSR_CHECK(d->h % (1 << mb) == 0 && d->w % (1 << mb) == 0, PSSR_ERR_ARG, "conv2d: blocked order needs H,W multiples of r");
    }
    if (d->epilogue == PSSR_EPI_TAIL || d->epilogue == PSSR_EPI_DGRAD_MASK) {
        PSSR_CHECK(d->aux && d->aux_scale && d->aux_shift && d->aux_coff % 4 == 0 && d->aux_cstride % 4 == 0, PSSR_ERR_ARG, "conv2d: epilogue needs aux tensor");
    }
    if (d->epilogue == PSSR_EPI_DGRAD_GELU)
        PSSR_CHECK(d->aux && d->aux_coff % 4 == 0 && d->aux_cstride % 4 == 0, PSSR_ERR_ARG, "conv2d: GELU backward needs the pre-activation as aux");
    if (d->flags & PSSR_FLAG_STATS) {
        PSSR_CHECK(d->stats != nullptr, PSSR_ERR_ARG, "conv2d: stats buffer missing");
        PSSR_CHECK(d->epilogue != PSSR_EPI_DGRAD_MASK || (d->aux_mean && d->aux_invstd), PSSR_ERR_ARG, "conv2d: mask stats need mean/invstd");
        PSSR_CHECK(d->epilogue != PSSR_EPI_TAIL, PSSR_ERR_ARG, "conv2d: no stats on tail");
    }
    ConvArgs a;
    a.N = d->n; a.H = d->h; a.W = d->w;
    a.in[0] = d->in0; a.in_cs[0] = d->in0_cstride; a.in_co[0] = d->in0_coff; a.nchunks[0] = d->cin0 / kch; a.taps[0] = d->taps0; a.w[0] = d->w0;
    a.in[1] = d->in1; a.in_cs[1] = d->in1_cstride; a.in_co[1] = d->in1_coff; a.nchunks[1] = d->cin1 ? d->cin1 / kch : 0; a.taps[1] = d->cin1 ? d->taps1 : 1; a.w[1] = d->w1;
    a.prologue = d->prologue; a.pro_scale = d->pro_scale; a.pro_shift = d->pro_shift;
    a.out = d->out; a.out_cs = d->out_cstride; a.out_co = d->out_coff; a.cout = d->cout; a.n_pad = d->n_pad;
    a.bias = d->bias; a.epi = d->epilogue; a.flags = d->flags;
    a.aux = d->aux; a.aux_cs = d->aux_cstride; a.aux_co = d->aux_coff;
    a.aux_scale = d->aux_scale; a.aux_shift = d->aux_shift; a.aux_mean = d->aux_mean; a.aux_invstd = d->aux_invstd;
    a.stats = d->stats;
    a.in0_blk = d->in0_blk; a.out_blk = d->out_blk; a.aux_blk = d->aux_blk;
    a.out_scale = d->out_scale; a.out_shift = d->out_shift;
    a.tiles_x = a.tiles_y = a.tiles_n = 0;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == PSSR_BF16 ? launch_bn<bf16_t>(a, s) : launch_bn<float>(a, s);
}
