// Shared device helpers for the pssr2_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pssr_mi355.h"

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // 16-byte staging register (native vector: stays in VGPRs)

// A "slot" is 16 bytes of consecutive channels of one pixel: 8 bf16 or 4 f32.  Every tile in
// LDS is made of 32-byte rows (2 slots) so that the bf16 and f32 builds share one byte geometry.
template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int EPS = 4;      // elements per slot
    static constexpr int KCH = 8;      // channels per K-chunk (2 slots)
    typedef f32x4 frag_t;
    static __device__ __forceinline__ void unpack(const u32x4& raw, float* f) {
        f[0] = __uint_as_float(raw[0]); f[1] = __uint_as_float(raw[1]);
        f[2] = __uint_as_float(raw[2]); f[3] = __uint_as_float(raw[3]);
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {
        u32x4 r = {__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
        return r;
    }
    static __device__ __forceinline__ float round(float v) { return v; }
    // relu(x * scale + shift) of one slot (the BatchNorm + ReLU prologue of the loaders)
    static __device__ __forceinline__ u32x4 bn_relu(const u32x4& raw, const float* sc, const float* sh) {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = __float_as_uint(fmaxf(fmaf(__uint_as_float(raw[i]), sc[i], sh[i]), 0.f));
        return o;
    }
    // 32x32 tile, K = one slot per lane half: 4 x v_mfma_f32_32x32x2_f32 (exact f32 fma chain)
    static __device__ __forceinline__ void mma(f32x16& c, const u32x4& a, const u32x4& b) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[0]), __uint_as_float(b[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[1]), __uint_as_float(b[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[2]), __uint_as_float(b[2]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[3]), __uint_as_float(b[3]), c, 0, 0, 0);
    }
};
template <> struct TT<bf16_t> {
    static constexpr int EPS = 8;
    static constexpr int KCH = 16;
    typedef bf16x8 frag_t;
    static __device__ __forceinline__ void unpack(const u32x4& raw, float* f) {
        const uint32_t w[4] = {raw[0], raw[1], raw[2], raw[3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {      // pairwise: one v_cvt_pk_bf16_f32 per dword
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 y = {f[2 * i], f[2 * i + 1]};
            o[i] = __builtin_bit_cast(unsigned int, __builtin_convertvector(y, bf16x2));
        }
        return o;
    }
    static __device__ __forceinline__ float round(float v) { return (float)(bf16_t)v; }
    // relu(x * scale + shift) of one slot, rounded to bf16: packed f32 fma, packed convert, ReLU as a signed 16-bit max
    // with 0 on the rounded values (same result as rounding relu(.) since rounding keeps the sign)
    static __device__ __forceinline__ u32x4 bn_relu(const u32x4& raw, const float* sc, const float* sh) {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 x = {__uint_as_float(raw[i] << 16), __uint_as_float(raw[i] & 0xffff0000u)};
            const f32x2 s = {sc[2 * i], sc[2 * i + 1]}, t = {sh[2 * i], sh[2 * i + 1]};
            const f32x2 y = __builtin_elementwise_fma(x, s, t);
            const s16x2 z = {0, 0};
            o[i] = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(y, bf16x2)), z));
        }
        return o;
    }
    // one v_mfma_f32_32x32x16_bf16
    static __device__ __forceinline__ void mma(f32x16& c, const u32x4& a, const u32x4& b) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

template <> struct TT<f16_t> {      // fp16 storage, f32 accumulate (config "fp16": needs loss scaling on the host side)
    static constexpr int EPS = 8;
    static constexpr int KCH = 16;
    typedef f16x8 frag_t;
    static __device__ __forceinline__ void unpack(const u32x4& raw, float* f) {
        const f16x8 v = __builtin_bit_cast(f16x8, raw);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
    static __device__ __forceinline__ u32x4 pack(const float* f) {
        f16x8 v;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (f16_t)f[i];
        return __builtin_bit_cast(u32x4, v);
    }
    static __device__ __forceinline__ float round(float v) { return (float)(f16_t)v; }
    static __device__ __forceinline__ u32x4 bn_relu(const u32x4& raw, const float* sc, const float* sh) {
        const f16x8 v = __builtin_bit_cast(f16x8, raw);
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 x = {(float)v[2 * i], (float)v[2 * i + 1]};
            const f32x2 s = {sc[2 * i], sc[2 * i + 1]}, t = {sh[2 * i], sh[2 * i + 1]};
            const f32x2 y = __builtin_elementwise_fma(x, s, t);
            const s16x2 z = {0, 0};
            o[i] = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(y, f16x2)), z));
        }
        return o;
    }
    // one v_mfma_f32_32x32x16_f16
    static __device__ __forceinline__ void mma(f32x16& c, const u32x4& a, const u32x4& b) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// store 4 consecutive channels
__device__ __forceinline__ void store4(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ void store4(bf16_t* p, const float* v) {
    union { bf16x4 v; uint2 u; } cv;
#pragma unroll
    for (int i = 0; i < 4; ++i) cv.v[i] = (bf16_t)v[i];
    *(uint2*)p = cv.u;
}
__device__ __forceinline__ void store4(f16_t* p, const float* v) {
    union { f16x4 v; uint2 u; } cv;
#pragma unroll
    for (int i = 0; i < 4; ++i) cv.v[i] = (f16_t)v[i];
    *(uint2*)p = cv.u;
}
__device__ __forceinline__ void load4(const f16_t* p, float* v) {
    union { f16x4 v; uint2 u; } cv;
    cv.u = *(const uint2*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)cv.v[i];
}
__device__ __forceinline__ void load4(const float* p, float* v) {
    float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float* v) {
    uint2 t = *(const uint2*)p;
    v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
    v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

// exact (erf) GELU of nn.GELU() (pssr/models/_rdnet.py:185,200) and its derivative
__device__ __forceinline__ float gelu_f(float z) { return 0.5f * z * (1.f + erff(z * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float z) {
    return 0.5f * (1.f + erff(z * 0.70710678118654752f)) + z * 0.39894228040143268f * expf(-0.5f * z * z);
}

// 16-bit storage builds: GELU and its derivative from polynomials in s = 2 min(|z|, 5) / 5 - 1, no transcendental instructions,
// evaluated for TWO values per instruction with packed f32 FMAs (v_pk_fma_f32) and for all pairs of a call in lockstep (independent
// Horner chains: no wait states between the dependent FMAs):
//   gelu(z)  = max(z, 0) - u T(s),          T = 1 - Phi on [0, 5]
//   gelu'(z) = 1/2 + sign(z) D(s),          D = Phi - 1/2 + u phi on [0, 5]
// Least-squares Chebyshev fits on 400 k points, Horner in f32; beyond |z| = 5 both are their limits to 2e-6.  Degree 12
// (|error| of u T <= 7e-6, of D <= 5e-5) for both storage types (degree 10 -- 1.3e-4 / 3.3e-4 -- is 8 % cheaper and moved the bf16
// gradients of tests/test_gpu_atrous.py's untrained RD net measurably).  The
// Abramowitz-Stegun 7.1.26 form used before (v_rcp + v_exp + 15 more instructions per value) cost twice as much, and RDNet's 1x1
// layers apply GELU or its derivative to every element of the 4C-wide tensor three times per step: conv2's forward and
// weight-gradient kernels ran 45 % faster with the GELU left out, its input gradient 15 %.  The exact-f32 build keeps erff.
// (The same chains with scalar v_fma_f32 measured 3-8 % slower on those kernels, 0.8 % on the c3 step: the packed form stays.)
typedef float pssr_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pssr_v2f pk_fma(pssr_v2f a, pssr_v2f b, pssr_v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pssr_v2f pk_bcast(float v) { return pssr_v2f{v, v}; }
template <typename T> struct GeluPoly;
template <> struct GeluPoly<f16_t> {
    static constexpr int DEG = 12;
    static constexpr float CT[13] = {6.210066012e-03f, -4.382169402e-02f, 1.368972628e-01f, -2.396008408e-01f, 2.325344506e-01f, -6.565579615e-02f,
                                     -1.301242450e-01f, 1.636429658e-01f, -2.667531663e-02f, -7.848973485e-02f, 4.258114333e-02f, 1.392573103e-02f,
                                     -1.142495623e-02f};
    static constexpr float CD[13] = {5.376079593e-01f, -1.861071133e-01f, 3.084409169e-01f, 2.716495812e-02f, -8.390708424e-01f, 1.182164307e+00f,
                                     -2.109314755e-01f, -1.116054970e+00f, 8.931616973e-01f, 3.827852021e-01f, -5.670671519e-01f, -3.991985896e-02f,
                                     1.278773880e-01f};
};
template <> struct GeluPoly<bf16_t> : GeluPoly<f16_t> {};
// t[p] = poly(s[p]) for NP pairs, the chains interleaved
template <typename T, int WHICH, int NP> __device__ __forceinline__ void gelu_poly_pairs(const pssr_v2f (&sv)[NP], pssr_v2f (&t)[NP]) {
    using G = GeluPoly<T>;
#pragma unroll
    for (int p = 0; p < NP; ++p) t[p] = pk_bcast(WHICH ? G::CD[G::DEG] : G::CT[G::DEG]);
#pragma unroll
    for (int k = G::DEG - 1; k >= 0; --k)
#pragma unroll
        for (int p = 0; p < NP; ++p) t[p] = pk_fma(t[p], sv[p], pk_bcast(WHICH ? G::CD[k] : G::CT[k]));
}
// f[i] = gelu(f[i]) for N (even) values
template <typename T, int N> __device__ __forceinline__ void gelu_vec(float* f) {
    static_assert(N % 2 == 0, "pairs");
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < N; ++e) f[e] = gelu_f(f[e]);
    } else {
        pssr_v2f u[N / 2], sv[N / 2], t[N / 2];
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            u[p] = pssr_v2f{fminf(fabsf(f[2 * p]), 5.f), fminf(fabsf(f[2 * p + 1]), 5.f)};
            sv[p] = pk_fma(u[p], pk_bcast(0.4f), pk_bcast(-1.f));
        }
        gelu_poly_pairs<T, 0, N / 2>(sv, t);
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            pssr_v2f g = pk_fma(-u[p], t[p], pssr_v2f{fmaxf(f[2 * p], 0.f), fmaxf(f[2 * p + 1], 0.f)});
            // min / max return their non-NaN operand: without this a NaN (or an fp16 overflow to inf) in z would come out as a finite
            // value and never reach the loss or the LossScaler's check.  z * 0 is +-0 for finite z, NaN otherwise (one packed FMA per pair)
            g = pk_fma(pssr_v2f{f[2 * p], f[2 * p + 1]}, pk_bcast(0.f), g);
            f[2 * p] = g.x; f[2 * p + 1] = g.y;
        }
    }
}
// v[i] *= gelu'(z[i]) for N (even) values
template <typename T, int N> __device__ __forceinline__ void gelu_grad_mul_vec(const float* z, float* v) {
    static_assert(N % 2 == 0, "pairs");
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] *= gelu_grad_f(z[e]);
    } else {
        pssr_v2f sv[N / 2], t[N / 2];
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            const pssr_v2f u = {fminf(fabsf(z[2 * p]), 5.f), fminf(fabsf(z[2 * p + 1]), 5.f)};
            sv[p] = pk_fma(u, pk_bcast(0.4f), pk_bcast(-1.f));
        }
        gelu_poly_pairs<T, 1, N / 2>(sv, t);
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            pssr_v2f d = {0.5f + __builtin_copysignf(t[p].x, z[2 * p]), 0.5f + __builtin_copysignf(t[p].y, z[2 * p + 1])};
            d = pk_fma(pssr_v2f{z[2 * p], z[2 * p + 1]}, pk_bcast(0.f), d);        // NaN / inf in z propagate (see gelu_vec)
            v[2 * p] *= d.x;
            v[2 * p + 1] *= d.y;
        }
    }
}
template <typename T> __device__ __forceinline__ float gelu_t(float z) { float f[2] = {z, z}; gelu_vec<T, 2>(f); return f[0]; }
template <typename T> __device__ __forceinline__ float gelu_grad_t(float z) { float zz[2] = {z, z}, v[2] = {1.f, 1.f}; gelu_grad_mul_vec<T, 2>(zz, v); return v[0]; }

// Pixel linearisation: plain NHWC, or "blocked" order of an r-times (r = 1<<blk) upsampled image
// (see pssr_conv_desc in include/pssr_mi355.h).
// One f32 partial sum into a statistic buffer [PSSR_STAT_ROWS][row_len] (include/pssr_mi355.h): `slot` addresses the element in the
// workgroup's stripe row, `lo_off` = PSSR_STAT_STRIPES * row_len doubles further lies the same element of the remainder row.
// Both pieces are multiples of a fixed power of two, so the f64 atomics add them without rounding: the result does not depend on
// the order of arrival.
__device__ __forceinline__ void stat_add(double* slot, long lo_off, float v) {
    const double d = (double)v;
    const double hi = __builtin_rint(d * 0x1p20) * 0x1p-20;
    const double lo = __builtin_rint((d - hi) * 0x1p64) * 0x1p-64;
    atomicAdd(slot, hi);
    atomicAdd(slot + lo_off, lo);
}

__device__ __forceinline__ long pix_index(int gi, int gy, int gx, int H, int W, int blk) {
    if (blk == 0) return ((long)gi * H + gy) * W + gx;
    const int R = 1 << blk;
    return ((((long)gi * (H >> blk) + (gy >> blk)) * (W >> blk) + (gx >> blk)) << (2 * blk)) + ((gy & (R - 1)) << blk) + (gx & (R - 1));
}

// status helpers for the C ABI
void pssr_set_error(const char* fmt, ...);
#define PSSR_CHECK(cond, code, ...)                         \
    do {                                                    \
        if (!(cond)) { pssr_set_error(__VA_ARGS__); return (code); } \
    } while (0)
#define PSSR_LAUNCH_CHECK()                                                          \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) { pssr_set_error("launch failed: %s", hipGetErrorString(e__)); return PSSR_ERR_LAUNCH; } \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
