// Weight gradient of the 3x3 / 1x1 convolution for gfx950:  dW[n][tap][k] = sum_pixels dy[p][n] * a[p+tap][k]
//
// GEMM view per tap: rows = output channels n, columns = input channels k, reduction = pixels.  Both
// operands are pixel-major (NHWC) in memory, i.e. the reduction index is the *strided* one, so the
// MFMA fragments (8 consecutive reduction elements per lane) are fetched from LDS with the CDNA4
// transposing read ds_read_b64_tr_b16 (bf16 build) or with plain 4-byte reads feeding
// v_mfma_f32_32x32x2_f32 (exact-f32 build).  A workgroup owns a (CO_T x CI_T x taps) slab of dW,
// keeps all of it in accumulators (9 tiles per wave), and walks a strided subset of the pixel
// tiles; per tile it stages dy (128 px) and the input halo tile once and reuses the halo for the 9
// taps.  Partial slabs are combined with f32 atomics (one add per element per workgroup).

#include "common.h"
#include "tunables.h"

namespace {

struct WgradArgs {
    int N, H, W;
    int tiles_x, tiles_y, tiles_i, n_tiles, split;
    const void* dy; int dy_cs, dy_co, dy_blk, cout;        // rows of dW
    const void* in; int in_cs, in_co, in_blk, cin_pad;     // columns of dW (k), multiple of 16
    int taps;
    int prologue; const float* pro_scale; const float* pro_shift;
    float* dw;                                             // [parts][cout][taps][cin_pad] f32 (parts == 0: one slab, atomics)
    int co_tiles, ci_tiles;
    int parts; long part_stride;
    unsigned* stamps;                                      // diagnostic build (-DPSSR_WG_STAMPS) only: per-(workgroup, wave) phase cycle sums
    int dbg;                                               // IGEMM_DBG bits (timing experiments): 16 skip the multiply, 32 skip the tile staging
};

template <int GEO> struct WGeo;
template <> struct WGeo<0> { static constexpr int TWL = 4, THL = 3; };
template <> struct WGeo<1> { static constexpr int TWL = 3, THL = 3; };
template <> struct WGeo<2> { static constexpr int TWL = 2, THL = 2; };

// fragment fetch: 8 (bf16) / 4 (f32) reduction rows for one 32-wide channel sub-tile
struct Frag16 {      // any 16-bit element type
    static constexpr int KP = 16;   // pixels per k-step
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    // `rowaddr0/1`: LDS byte address of (this lane's row q, its 4 columns) for reduction rows q and q+4
    static __device__ __forceinline__ u32x4 load(const char* a0, const char* a1) {
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0));
        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1));
        u32x4 r = {lo[0], lo[1], hi[0], hi[1]};
        return r;
    }
};

template <typename T, int CO_T, int CI_T, int GEO, int TAPS>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && GEO < 2) ? 2 : 1) void conv_wgrad_kernel(const WgradArgs p) {
    using X = TT<T>;
    constexpr int EPS = X::EPS;
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    constexpr int HW2 = TW + 2, HPI = (TH + 2) * (TW + 2), HP = NI * HPI;
    constexpr int ROWB = 32 * (int)sizeof(T);            // bytes per LDS row (32 channels)
    constexpr int PPS = ROWB / 16;                        // 16-byte pieces per sub-tile row
    constexpr int CO_S = CO_T / 32, CI_S = CI_T / 32;     // 32-channel sub-tiles
    static_assert(CO_S * CI_S == 4, "one (cout sub, cin sub) pair per wave");
    constexpr int DY_BYTES = CO_S * 128 * ROWB;
    constexpr int DY_PIECES = 128 * CO_S * PPS, AH_PIECES = HP * CI_S * PPS;
    constexpr int DY_ITEMS = (DY_PIECES + 255) / 256, AH_ITEMS = (AH_PIECES + 255) / 256;
    constexpr int KP = (sizeof(T) == 2) ? 16 : 8;         // pixels per k-step
    // bf16: the loads of pixel tile i+1 are in flight (in registers) while tile i is multiplied; the exact-f32 build
    // (twice the staging registers) loads synchronously
    constexpr bool PREFETCH = sizeof(T) == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Dy = smem;
    char* Ah = smem + DY_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cosub = wave % CO_S, cisub = wave / CO_S;
    const int ct = blockIdx.y;
    const int n0 = (ct % p.co_tiles) * CO_T, k0 = (ct / p.co_tiles) * CI_T;
    const T* dyp = (const T*)p.dy;
    const T* inp = (const T*)p.in;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const char* dy_img = Dy + cosub * 128 * ROWB;
    const char* ah_img = Ah + cisub * HP * ROWB;

    u32x4 dy_reg[DY_ITEMS], ah_reg[AH_ITEMS];
    unsigned long long dy_ok = 0, ah_ok = 0;      // validity bits of the staged items (out-of-range items load pixel 0 and are zeroed at commit)

    // NB: macros, not lambdas (with lambdas hipcc keeps the staging registers in scratch); loads are unconditional
#define WG_ISSUE(TILE)                                                                                            \
    {                                                                                                             \
        int tmi_ = (TILE);                                                                                        \
        const int tile_x_ = tmi_ % p.tiles_x; tmi_ /= p.tiles_x;                                                  \
        const int tile_y_ = tmi_ % p.tiles_y;                                                                     \
        const int tile_i_ = tmi_ / p.tiles_y;                                                                     \
        const int x0_ = tile_x_ << TWL, y0_ = tile_y_ << THL, img0_ = tile_i_ * NI;                               \
        dy_ok = 0; ah_ok = 0;                                                                                     \
        _Pragma("unroll") for (int it = 0; it < DY_ITEMS; ++it) {                                                 \
            const int idx = tid + it * 256;                                                                       \
            const int m = idx / (CO_S * PPS), pc = idx % (CO_S * PPS);                                            \
            const int tx = m & (TW - 1), ty = (m >> TWL) & (TH - 1), img = m >> (TWL + THL);                      \
            const int gy = y0_ + ty, gx = x0_ + tx, gi = img0_ + img;                                             \
            const int ch = n0 + pc * EPS;                                                                         \
            const bool ok = idx < DY_PIECES && gi < p.N && gy < p.H && gx < p.W && ch < p.cout;                   \
            const long pix = ok ? pix_index(gi, gy, gx, p.H, p.W, p.dy_blk) : 0;                                  \
            dy_reg[it] = *(const u32x4*)(dyp + pix * p.dy_cs + p.dy_co + (ok ? ch : 0));                          \
            dy_ok |= (ok ? 1ull : 0ull) << it;                                                                        \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < AH_ITEMS; ++it) {                                                 \
            const int idx = tid + it * 256;                                                                       \
            const int pp = idx / (CI_S * PPS), pc = idx % (CI_S * PPS);                                           \
            const int img = pp / HPI, rem = pp % HPI;                                                             \
            const int hy = rem / HW2, hx = rem % HW2;                                                             \
            const int gy = y0_ + hy - 1, gx = x0_ + hx - 1, gi = img0_ + img;                                     \
            const int ch = k0 + pc * EPS;                                                                         \
            const bool ok = idx < AH_PIECES && gi < p.N && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W && ch < p.cin_pad; \
            const long pix = ok ? pix_index(gi, gy, gx, p.H, p.W, p.in_blk) : 0;                                  \
            ah_reg[it] = *(const u32x4*)(inp + pix * p.in_cs + p.in_co + (ok ? ch : 0));                          \
            ah_ok |= (ok ? 1ull : 0ull) << it;                                                                        \
        }                                                                                                         \
    }
#define WG_COMMIT()                                                                                               \
    {                                                                                                             \
        _Pragma("unroll") for (int it = 0; it < DY_ITEMS; ++it) {                                                 \
            const int idx = tid + it * 256;                                                                       \
            if (idx < DY_PIECES) {                                                                                \
                const int m = idx / (CO_S * PPS), pc = idx % (CO_S * PPS);                                        \
                const int sub = pc / PPS, pin = pc % PPS;                                                         \
                u32x4 v = dy_reg[it];                                                                             \
                if (!((dy_ok >> it) & 1ull)) v = u32x4{0u, 0u, 0u, 0u};                                             \
                *(u32x4*)(Dy + sub * 128 * ROWB + m * ROWB + pin * 16) = v;                                       \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int it = 0; it < AH_ITEMS; ++it) {                                                 \
            const int idx = tid + it * 256;                                                                       \
            if (idx < AH_PIECES) {                                                                                \
                const int pp = idx / (CI_S * PPS), pc = idx % (CI_S * PPS);                                       \
                const int sub = pc / PPS, pin = pc % PPS;                                                         \
                const int ch = k0 + pc * EPS;                                                                     \
                u32x4 v = ah_reg[it];                                                                             \
                const bool ok = (ah_ok >> it) & 1ull;                                                               \
                if (!ok) v = u32x4{0u, 0u, 0u, 0u};                                                               \
                if (ok && p.prologue == PSSR_PRO_BN_RELU) {                                                       \
                    float f[EPS];                                                                                 \
                    X::unpack(v, f);                                                                              \
                    _Pragma("unroll") for (int e = 0; e < EPS; e += 4) {                                          \
                        const float4 sc = *(const float4*)(p.pro_scale + ch + e);                                 \
                        const float4 sh = *(const float4*)(p.pro_shift + ch + e);                                 \
                        f[e + 0] = fmaxf(fmaf(f[e + 0], sc.x, sh.x), 0.f);                                        \
                        f[e + 1] = fmaxf(fmaf(f[e + 1], sc.y, sh.y), 0.f);                                        \
                        f[e + 2] = fmaxf(fmaf(f[e + 2], sc.z, sh.z), 0.f);                                        \
                        f[e + 3] = fmaxf(fmaf(f[e + 3], sc.w, sh.w), 0.f);                                        \
                    }                                                                                             \
                    v = X::pack(f);                                                                               \
                } else if (p.prologue == PSSR_PRO_GELU) {      /* gelu(0) == 0 */                                 \
                    float f[EPS];                                                                                 \
                    X::unpack(v, f);                                                                              \
                    gelu_vec<T, EPS>(f);                          \
                    v = X::pack(f);                                                                               \
                }                                                                                                 \
                *(u32x4*)(Ah + sub * HP * ROWB + pp * ROWB + pin * 16) = v;                                       \
            }                                                                                                     \
        }                                                                                                         \
    }

    int tile = blockIdx.x;
    if (tile < p.n_tiles) WG_ISSUE(tile)
    for (; tile < p.n_tiles; tile += p.split) {
        WG_COMMIT()
        __syncthreads();
        const int nt = tile + p.split;
        if (PREFETCH && nt < p.n_tiles) WG_ISSUE(nt)

        // ---- multiply: 128/KP k-steps x taps
        for (int s = 0; s < 128 / KP; ++s) {
            u32x4 af;
            int hb[2];   // halo byte offsets of this lane's reduction rows (tap (0,0))
            if constexpr (sizeof(T) == 2) {
                // ds_read_b64_tr_b16: lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3
                const int g = lane >> 4, i = lane & 15, q = i >> 2, pcol = i & 3;
                const int colb = ((g & 1) * 16 + pcol * 4) * 2;
                const int m0 = s * 16 + (g >> 1) * 8 + q;
                af = Frag16::load(dy_img + m0 * ROWB + colb, dy_img + (m0 + 4) * ROWB + colb);
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) {
                    const int m = m0 + 4 * rd;
                    const int tx = m & (TW - 1), ty = (m >> TWL) & (TH - 1), img = m >> (TWL + THL);
                    hb[rd] = (img * HPI + ty * HW2 + tx) * ROWB + colb;
                }
            } else {
                // 4 x 32x32x2: MFMA t consumes reduction rows s*8 + 4h + t; lane r reads channel r of that row
                const int r = lane & 31, h = lane >> 5;
                const int m0 = s * 8 + 4 * h;
                float a4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[t] = *(const float*)(dy_img + (m0 + t) * ROWB + r * 4);
                af = u32x4{__float_as_uint(a4[0]), __float_as_uint(a4[1]), __float_as_uint(a4[2]), __float_as_uint(a4[3])};
                const int tx = m0 & (TW - 1), ty = (m0 >> TWL) & (TH - 1), img = m0 >> (TWL + THL);
                hb[0] = (img * HPI + ty * HW2 + tx) * ROWB + r * 4;   // rows m0..m0+3 are consecutive in x when TW >= 4
                hb[1] = 0;
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int ky = (TAPS == 9) ? t / 3 : 1, kx = (TAPS == 9) ? t % 3 : 1;
                const int toff = (ky * HW2 + kx) * ROWB;
                u32x4 bf;
                if constexpr (sizeof(T) == 2) {
                    bf = Frag16::load(ah_img + hb[0] + toff, ah_img + hb[1] + toff);
                } else {
                    float b4[4];
                    if constexpr (TW >= 4) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) b4[u] = *(const float*)(ah_img + hb[0] + toff + u * ROWB);
                    } else {
                        const int r = lane & 31, h = lane >> 5;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int m = s * 8 + 4 * h + u;
                            const int tx = m & (TW - 1), ty = (m >> TWL) & (TH - 1), img = m >> (TWL + THL);
                            b4[u] = *(const float*)(ah_img + (img * HPI + ty * HW2 + tx) * ROWB + r * 4 + toff);
                        }
                    }
                    bf = u32x4{__float_as_uint(b4[0]), __float_as_uint(b4[1]), __float_as_uint(b4[2]), __float_as_uint(b4[3])};
                }
                X::mma(acc[t], af, bf);
            }
        }
        __syncthreads();
        if (!PREFETCH && nt < p.n_tiles) WG_ISSUE(nt)
    }
#undef WG_ISSUE
#undef WG_COMMIT

    // ---- combine.  parts > 0: this workgroup's partial slab goes to part blockIdx.x with plain stores (every element of
    // the part is written by exactly one workgroup; pssr_unpack_conv_wgrad_parts sums the parts).  parts == 0: f32
    // atomics into one caller-zeroed slab.
    const int kcol = k0 + cisub * 32 + (lane & 31);
    if (kcol < p.cin_pad) {
        float* dst = p.dw + (p.parts > 0 ? (long)blockIdx.x * p.part_stride : 0L);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + cosub * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (n < p.cout) {
                    float* q = dst + ((long)n * TAPS + t) * p.cin_pad + kcol;
                    if (p.parts > 0) *q = acc[t][e];
                    else atomicAdd(q, acc[t][e]);
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// Lean-loader build of the same kernel for 16-bit types when every tile of the launch is a full tile (W % TW == 0,
// H % TH == 0, N % NI == 0: every layer of the models).  Measured on the general kernel above: 10.4 vector-ALU
// instructions per MFMA, nearly all of them per-tile index arithmetic, validity tests and 64-bit addresses; here
//   * every per-item quantity that does not depend on the tile (pixel offset relative to the tile origin, LDS slot,
//     which border a halo pixel lies on, the BatchNorm coefficients of the thread's 8 channels) is computed once;
//   * per tile only a buffer resource is rebuilt (scalar) and out-of-image halo items are redirected to an
//     out-of-range offset (the buffer load returns zeros, no memory access): 3 vector instructions per halo item;
//   * fragment addresses are a per-lane base plus compile-time offsets (no vector arithmetic in the MFMA loop).
constexpr unsigned WG_BIAS = 1u << 30;      // relative offsets can be negative (halo row -1): base - BIAS, offset + BIAS

// halo pixel slot of output pixel m of the tile (tap (0,0))
template <int GEO> __host__ __device__ constexpr int wg_halo_pix(int m) {
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL, TW = 1 << TWL, TH = 1 << THL;
    return (m >> (TWL + THL)) * ((TH + 2) * (TW + 2)) + ((m >> TWL) & (TH - 1)) * (TW + 2) + (m & (TW - 1));
}

// pixel index of (img, y0 + u, x0 + v) minus that of (0, y0, x0), for tile origins that are multiples of 1 << blk
__device__ __forceinline__ long wg_rel_pix(int img, int u, int v, int H, int W, int blk) {
    const long base = (long)img * H * W;
    if (blk == 0) return base + (long)u * W + v;
    const int R = 1 << blk;
    return base + ((((long)(u >> blk) * (W >> blk) + (v >> blk))) << (2 * blk)) + ((u & (R - 1)) << blk) + (v & (R - 1));
}

template <typename T, int CO_T, int CI_T, int GEO, int TAPS>
__global__ __launch_bounds__(256, 2) void conv_wgrad16_kernel(const WgradArgs p) {
    using X = TT<T>;
    static_assert(sizeof(T) == 2 && GEO < 2, "16-bit types, tiles of at least 8x8 pixels");
    constexpr int EPS = 8, ESZ = 2;
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    constexpr int HW2 = TW + 2, HPI = (TH + 2) * (TW + 2), HP = NI * HPI;
    constexpr int ROWB = 64, PPS = 4;
    constexpr int CO_S = CO_T / 32, CI_S = CI_T / 32;
    static_assert(CO_S * CI_S == 4, "one (cout sub, cin sub) pair per wave");
    constexpr int DY_BYTES = CO_S * 128 * ROWB;
    constexpr int DY_PPP = CO_S * PPS, AH_PPP = CI_S * PPS;           // 16-byte pieces per pixel
    constexpr int DY_PIECES = 128 * DY_PPP, AH_PIECES = HP * AH_PPP;
    constexpr int DY_ITEMS = DY_PIECES / 256, AH_ITEMS = (AH_PIECES + 255) / 256;
    static_assert(DY_PIECES % 256 == 0 && 256 % DY_PPP == 0 && 256 % AH_PPP == 0 && AH_ITEMS <= 16, "item geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Dy = smem;
    char* Ah = smem + DY_BYTES;
#ifdef PSSR_WG_STAMPS
    unsigned long long st_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0) :: "memory");
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cosub = wave % CO_S, cisub = wave / CO_S;
    const int ct = blockIdx.y;
    const int n0 = (ct % p.co_tiles) * CO_T, k0 = (ct / p.co_tiles) * CI_T;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- tile-independent item descriptors
    unsigned dy_off[DY_ITEMS], ah_off[AH_ITEMS];
    unsigned long long cls = 0;                      // 4 bits per halo item: lies on the left / right / top / bottom halo ring
    int dy_lds, ah_lds;
    {
        const int pc = tid % DY_PPP, sub = pc / PPS, pin = pc % PPS;
        const int ch = n0 + pc * EPS;
        dy_lds = sub * 128 * ROWB + (tid / DY_PPP) * ROWB + pin * 16;
#pragma unroll
        for (int it = 0; it < DY_ITEMS; ++it) {
            const int m = tid / DY_PPP + it * (256 / DY_PPP);
            const int tx = m & (TW - 1), ty = (m >> TWL) & (TH - 1), img = m >> (TWL + THL);
            dy_off[it] = ch < p.cout ? (unsigned)(wg_rel_pix(img, ty, tx, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ + ch * ESZ) + WG_BIAS : 0xffffffffu;
        }
    }
    float sc[EPS], sh[EPS];
    {
        const int pc = tid % AH_PPP, sub = pc / PPS, pin = pc % PPS;
        const int ch = k0 + pc * EPS;
        const bool ch_ok = ch < p.cin_pad;
        ah_lds = sub * HP * ROWB + (tid / AH_PPP) * ROWB + pin * 16;
#pragma unroll
        for (int it = 0; it < AH_ITEMS; ++it) {
            const int idx = tid + it * 256;
            const int pp = tid / AH_PPP + it * (256 / AH_PPP);
            const int img = pp / HPI, rem = pp % HPI;
            const int hy = rem / HW2, hx = rem % HW2;
            const bool live = idx < AH_PIECES && ch_ok;
            ah_off[it] = live ? (unsigned)(wg_rel_pix(img, hy - 1, hx - 1, p.H, p.W, p.in_blk) * p.in_cs * ESZ + ch * ESZ) + WG_BIAS : 0xffffffffu;
            cls |= (unsigned long long)((hx == 0 ? 1u : 0u) | (hx == TW + 1 ? 2u : 0u) | (hy == 0 ? 4u : 0u) | (hy == TH + 1 ? 8u : 0u)) << (4 * it);
        }
        // a thread stages the same 8 channels of every item and tile: its prologue coefficients live in registers
        // (channels past cin_pad get 0 / 0, so their zero-filled loads stay zero)
#pragma unroll
        for (int e = 0; e < EPS; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
        if (p.prologue == PSSR_PRO_BN_RELU && ch_ok) {
            load4(p.pro_scale + ch, sc); load4(p.pro_scale + ch + 4, sc + 4);
            load4(p.pro_shift + ch, sh); load4(p.pro_shift + ch + 4, sh + 4);
        }
    }
    // ---- fragment read bases: ds_read_b64_tr_b16, lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pcol = li & 3;
    const int colb = ((g & 1) * 16 + pcol * 4) * 2;
    const int m00 = (g >> 1) * 8 + q;
    const char* const dy_rd = Dy + cosub * 128 * ROWB + m00 * ROWB + colb;
    const char* const ah_rd0 = Ah + cisub * HP * ROWB + wg_halo_pix<GEO>(m00) * ROWB + colb;
    const char* const ah_rd1 = Ah + cisub * HP * ROWB + wg_halo_pix<GEO>(m00 + 4) * ROWB + colb;

    const char* const dy_base = (const char*)p.dy + (long)p.dy_co * ESZ - WG_BIAS;
    const char* const in_base = (const char*)p.in + (long)p.in_co * ESZ - WG_BIAS;

    u32x4 dy_reg[DY_ITEMS], ah_reg[AH_ITEMS];
    unsigned long long ah_bad = 0;          // cls bits of the staged tile that fall outside the image

#define WG_ISSUE(TILE)                                                                                            \
    {                                                                                                             \
        int tmi_ = (TILE);                                                                                        \
        const int tile_x_ = tmi_ % p.tiles_x; tmi_ /= p.tiles_x;                                                  \
        const int tile_y_ = tmi_ % p.tiles_y;                                                                     \
        const int tile_i_ = tmi_ / p.tiles_y;                                                                     \
        const int x0_ = tile_x_ << TWL, y0_ = tile_y_ << THL, img0_ = tile_i_ * NI;                               \
        const unsigned tb_ = (x0_ == 0 ? 1u : 0u) | (x0_ + TW == p.W ? 2u : 0u) | (y0_ == 0 ? 4u : 0u) | (y0_ + TH == p.H ? 8u : 0u); \
        ah_bad = cls & (tb_ * 0x1111111111111111ull);                                                                      \
        const __amdgpu_buffer_rsrc_t rdy_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(dy_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        const __amdgpu_buffer_rsrc_t rin_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(in_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.in_blk) * p.in_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        _Pragma("unroll") for (int it = 0; it < DY_ITEMS; ++it)                                                   \
            dy_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy_, (int)dy_off[it], 0, 0)); \
        _Pragma("unroll") for (int it = 0; it < AH_ITEMS; ++it) {                                                 \
            const unsigned off_ = ((ah_bad >> (4 * it)) & 0xfu) ? 0xffffffffu : ah_off[it];                       \
            ah_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin_, (int)off_, 0, 0)); \
        }                                                                                                         \
    }
#define WG_COMMIT()                                                                                               \
    {                                                                                                             \
        _Pragma("unroll") for (int it = 0; it < DY_ITEMS; ++it)                                                   \
            *(u32x4*)(Dy + dy_lds + it * (256 / DY_PPP) * ROWB) = dy_reg[it];                                     \
        _Pragma("unroll") for (int it = 0; it < AH_ITEMS; ++it) {                                                 \
            if ((it + 1) * 256 <= AH_PIECES || tid < AH_PIECES - it * 256) {                                      \
                u32x4 v = ah_reg[it];                                                                             \
                if (p.prologue == PSSR_PRO_BN_RELU) {                                                             \
                    v = X::bn_relu(v, sc, sh);                                                                    \
                    if ((ah_bad >> (4 * it)) & 0xfu) v = u32x4{0u, 0u, 0u, 0u};                                   \
                } else if (p.prologue == PSSR_PRO_GELU) {      /* gelu(0) == 0 */                                 \
                    float f[EPS];                                                                                 \
                    X::unpack(v, f);                                                                              \
                    gelu_vec<T, EPS>(f);                          \
                    v = X::pack(f);                                                                               \
                }                                                                                                 \
                *(u32x4*)(Ah + ah_lds + it * (256 / AH_PPP) * ROWB) = v;                                          \
            }                                                                                                     \
        }                                                                                                         \
    }

#ifdef PSSR_WG_STAMPS
    unsigned long long st_prev = 0;
    unsigned st_sum[6] = {0, 0, 0, 0, 0, 0};
#define WG_STAMP(I)                                                                                               \
    {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        unsigned long long t_;                                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((I) >= 0) st_sum[(I) < 0 ? 0 : (I)] += (unsigned)(t_ - st_prev);                                      \
        st_prev = t_;                                                                                             \
    }
#else
#define WG_STAMP(I)
#endif
    int tile = blockIdx.x;
    WG_STAMP(-1)
#ifdef PSSR_WG_STAMPS
    const unsigned st_pro = (unsigned)(st_prev - st_t0);
#endif
    if (tile < p.n_tiles) WG_ISSUE(tile)
    const bool first_only = p.dbg & 32;
    for (; tile < p.n_tiles; tile += p.split) {
        WG_STAMP(5)
#ifdef PSSR_WG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // segment 0 = exposed load latency
#endif
        WG_STAMP(0)
        if (!first_only || tile == (int)blockIdx.x) WG_COMMIT()
        WG_STAMP(1)
        __syncthreads();
        WG_STAMP(2)
        const int nt = tile + p.split;
        if (nt < p.n_tiles && !first_only) WG_ISSUE(nt)
        WG_STAMP(3)

        // ---- multiply: 8 k-steps of 16 pixels x taps; addresses = lane base + compile-time offsets
        if (p.dbg & 16) {
        } else if constexpr (GEO == 0 && TAPS == 9) {
            // a k-step is one 16-pixel tile row, so the input fragment of tap (ky, kx) at k-step s is the fragment of halo row s + ky
            // at column offset kx: each halo row is fetched once (3 fragments) into a 3-row register window and multiplied by
            // the dy fragments of the three k-steps that see it -- 4 fragment fetches per 9 MFMAs instead of 10 (the LDS read
            // port, not the matrix pipe, bounded the 10-fetch loop: 20 x 512 B per 9 x 8 clocks per CU > 128 B/clk)
            u32x4 win[3][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) win[r][kx] = Frag16::load(ah_rd0 + (r * HW2 + kx) * ROWB, ah_rd1 + (r * HW2 + kx) * ROWB);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 af = Frag16::load(dy_rd + s * 16 * ROWB, dy_rd + (s * 16 + 4) * ROWB);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    win[(s + 2) % 3][kx] = Frag16::load(ah_rd0 + ((s + 2) * HW2 + kx) * ROWB, ah_rd1 + ((s + 2) * HW2 + kx) * ROWB);
#pragma unroll
                for (int t = 0; t < 9; ++t) X::mma(acc[t], af, win[(s + t / 3) % 3][t % 3]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 af = Frag16::load(dy_rd + s * 16 * ROWB, dy_rd + (s * 16 + 4) * ROWB);
                const int hs = wg_halo_pix<GEO>(s * 16) * ROWB;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int ky = (TAPS == 9) ? t / 3 : 1, kx = (TAPS == 9) ? t % 3 : 1;
                    const int toff = (ky * HW2 + kx) * ROWB + hs;
                    const u32x4 bf = Frag16::load(ah_rd0 + toff, ah_rd1 + toff);
                    X::mma(acc[t], af, bf);
                }
            }
        }
        WG_STAMP(4)
        __syncthreads();
    }
    WG_STAMP(5)
#undef WG_ISSUE
#undef WG_COMMIT
#undef WG_STAMP

    const int kcol = k0 + cisub * 32 + (lane & 31);
    if (kcol < p.cin_pad) {
        float* dst = p.dw + (p.parts > 0 ? (long)blockIdx.x * p.part_stride : 0L);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + cosub * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (n < p.cout) {
                    float* qd = dst + ((long)n * TAPS + t) * p.cin_pad + kcol;
                    if (p.parts > 0) *qd = acc[t][e];
                    else atomicAdd(qd, acc[t][e]);
                }
            }
        }
    }
#ifdef PSSR_WG_STAMPS
    if (p.stamps) {
        unsigned long long t1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        if (lane == 0) {
            unsigned* q = p.stamps + ((long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
            for (int i = 0; i < 6; ++i) q[i] = st_sum[i];
            q[6] = ((p.n_tiles - (int)blockIdx.x + p.split - 1) / p.split) | (st_pro << 8);
            q[7] = (unsigned)(t1 - st_prev);
        }
    }
#endif
}

// -----------------------------------------------------------------------------------------------------------------
// All-DMA build of the lean 3x3 kernel for a prologue-free input (round 3).  Phase stamps of conv_wgrad16_kernel per 128-pixel tile:
// multiply 2580 clocks (72 MFMAs = 2304), commit 1830 (BatchNorm/ReLU prologue + ten ds_write_b128), issue 1140, barriers 400 -- the
// matrix pipe works 38 % of a workgroup's life because every staged byte crosses the register file and the vector ALU.  When the
// input needs no prologue (the engine materialises relu(bn(y)) once per layer, on the second stream under the forward pass:
// Engine._materialise) BOTH operands go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: out-of-range offsets = zero fill = the
// conv's zero padding and the channel tails), double-buffered: the 40 KiB of pixel tile i + 1 land while tile i is multiplied.  No
// staging registers, no commit, ~10 vector instructions per tile per wave besides the MFMAs and their fragment reads.
// LDS image per buffer: dy [CO_S][128 px][64 B] | input halo [CI_S][HPP px][64 B] (HPP = halo pixels rounded up to whole 16-pixel
// pieces, an even number of them per sub-tile); a piece = one wave instruction = 16 pixels x 64 B of one 32-channel sub-tile = 1 KiB, lane l -> (pixel l / 4, 16 B l % 4).
template <typename T, int CO_T, int CI_T, int GEO, int TAPS>
__global__ __launch_bounds__(256, 2) void conv_wgrad16d_kernel(const WgradArgs p) {
    using X = TT<T>;
    static_assert(sizeof(T) == 2 && GEO < 2, "16-bit types, tiles of at least 8x8 pixels");
    constexpr int EPS = 8, ESZ = 2;
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    constexpr int HW2 = TW + 2, HPI = (TH + 2) * (TW + 2), HP = NI * HPI, HPP = (HP + 31) / 32 * 32;
    constexpr int ROWB = 64;
    constexpr int CO_S = CO_T / 32, CI_S = CI_T / 32;
    static_assert(CO_S * CI_S == 4, "one (cout sub, cin sub) pair per wave");
    constexpr int DY_BYTES = CO_S * 128 * ROWB, AH_BYTES = CI_S * HPP * ROWB, BUF = DY_BYTES + AH_BYTES;
    constexpr int DY_PCS = DY_BYTES / 1024, AH_PCS = AH_BYTES / 1024, PCS = DY_PCS + AH_PCS;
    static_assert(PCS % 4 == 0, "pieces per wave");
    constexpr int PW = PCS / 4;                                           // pieces a wave issues per pixel tile
    static_assert(PW <= 16, "class bits");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cosub = wave % CO_S, cisub = wave / CO_S;
    const int ct = blockIdx.y;
    const int n0 = (ct % p.co_tiles) * CO_T, k0 = (ct / p.co_tiles) * CI_T;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- tile-independent piece descriptors: piece q = wave + 4 j
    unsigned voff[PW];                               // byte offset relative to the tile origin (+ WG_BIAS), or out of range
    unsigned long long cls = 0;                      // 4 bits per piece: this lane's halo pixel lies on the left / right / top / bottom ring
    const int lp = lane >> 2, pin = lane & 3;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        const int q = wave + 4 * j;
        if (q < DY_PCS) {
            const int sub = q / (128 / 16), m = (q % (128 / 16)) * 16 + lp;
            const int tx = m & (TW - 1), ty = (m >> TWL) & (TH - 1), img = m >> (TWL + THL);
            const int ch = n0 + sub * 32 + pin * EPS;
            voff[j] = ch < p.cout ? (unsigned)(wg_rel_pix(img, ty, tx, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ + ch * ESZ) + WG_BIAS : 0xffffffffu;
        } else {
            const int qa = q - DY_PCS;
            const int sub = qa / (HPP / 16), pp = (qa % (HPP / 16)) * 16 + lp;
            const int img = pp / HPI, rem = pp % HPI;
            const int hy = rem / HW2, hx = rem % HW2;
            const int ch = k0 + sub * 32 + pin * EPS;
            const bool live = pp < HP && ch < p.cin_pad;
            voff[j] = live ? (unsigned)(wg_rel_pix(img, hy - 1, hx - 1, p.H, p.W, p.in_blk) * p.in_cs * ESZ + ch * ESZ) + WG_BIAS : 0xffffffffu;
            cls |= (unsigned long long)((hx == 0 ? 1u : 0u) | (hx == TW + 1 ? 2u : 0u) | (hy == 0 ? 4u : 0u) | (hy == TH + 1 ? 8u : 0u)) << (4 * j);
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // ---- fragment read bases (buffer 0): ds_read_b64_tr_b16, lane 4q+p of each 16-lane group addresses row q, columns 4p..4p+3
    const int g = lane >> 4, li = lane & 15, qr = li >> 2, pcol = li & 3;
    const int colb = ((g & 1) * 16 + pcol * 4) * 2;
    const int m00 = (g >> 1) * 8 + qr;
    const char* const dy_rd0 = smem + cosub * 128 * ROWB + m00 * ROWB + colb;
    const char* const ah_rd00 = smem + DY_BYTES + cisub * HPP * ROWB + wg_halo_pix<GEO>(m00) * ROWB + colb;
    const char* const ah_rd10 = smem + DY_BYTES + cisub * HPP * ROWB + wg_halo_pix<GEO>(m00 + 4) * ROWB + colb;

    const char* const dy_base = (const char*)p.dy + (long)p.dy_co * ESZ - WG_BIAS;
    const char* const in_base = (const char*)p.in + (long)p.in_co * ESZ - WG_BIAS;

    // requests pixel tile TILE into LDS buffer B: PW buffer-load-to-LDS instructions per wave (M0 = LDS address of the piece)
#define WD_ISSUE(TILE, B)                                                                                         \
    {                                                                                                             \
        int tmi_ = (TILE);                                                                                        \
        const int tile_x_ = tmi_ % p.tiles_x; tmi_ /= p.tiles_x;                                                  \
        const int tile_y_ = tmi_ % p.tiles_y;                                                                     \
        const int tile_i_ = tmi_ / p.tiles_y;                                                                     \
        const int x0_ = tile_x_ << TWL, y0_ = tile_y_ << THL, img0_ = tile_i_ * NI;                               \
        const unsigned tb_ = (x0_ == 0 ? 1u : 0u) | (x0_ + TW == p.W ? 2u : 0u) | (y0_ == 0 ? 4u : 0u) | (y0_ + TH == p.H ? 8u : 0u); \
        const unsigned long long bad_ = cls & (tb_ * 0x1111111111111111ull);                                      \
        const __amdgpu_buffer_rsrc_t rdy_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(dy_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        const __amdgpu_buffer_rsrc_t rin_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(in_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.in_blk) * p.in_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        const unsigned lb_ = lds0 + (unsigned)(B) * BUF + (unsigned)wave * 1024u;                                 \
        _Pragma("unroll") for (int j = 0; j < PW; ++j) {                                                          \
            const unsigned off_ = ((bad_ >> (4 * j)) & 0xfu) ? 0xffffffffu : voff[j];                             \
            const unsigned la_ = __builtin_amdgcn_readfirstlane(lb_ + (unsigned)j * 4096u);                       \
            if (wave + 4 * j < DY_PCS)                                                                            \
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la_), "v"(off_), "s"(rdy_) : "memory"); \
            else                                                                                                  \
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(la_), "v"(off_), "s"(rin_) : "memory"); \
        }                                                                                                         \
    }

    int tile = blockIdx.x;
    int buf = 0;
    if (tile < p.n_tiles) WD_ISSUE(tile, 0)
    for (; tile < p.n_tiles; tile += p.split) {
        const int nt = tile + p.split;
        if (nt < p.n_tiles) {
            WD_ISSUE(nt, buf ^ 1)
            // everything but the PW pieces just requested has landed
            if constexpr (PW == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if constexpr (PW == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else if constexpr (PW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");                          // every wave's pieces of this buffer have landed
        const char* const dy_rd = dy_rd0 + buf * BUF;
        const char* const ah_rd0 = ah_rd00 + buf * BUF;
        const char* const ah_rd1 = ah_rd10 + buf * BUF;
        if constexpr (GEO == 0 && TAPS == 9) {
            // a k-step is one 16-pixel tile row: each halo row is fetched once into a 3-row register window (conv_wgrad16_kernel)
            u32x4 win[3][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) win[r][kx] = Frag16::load(ah_rd0 + (r * HW2 + kx) * ROWB, ah_rd1 + (r * HW2 + kx) * ROWB);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 af = Frag16::load(dy_rd + s * 16 * ROWB, dy_rd + (s * 16 + 4) * ROWB);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    win[(s + 2) % 3][kx] = Frag16::load(ah_rd0 + ((s + 2) * HW2 + kx) * ROWB, ah_rd1 + ((s + 2) * HW2 + kx) * ROWB);
#pragma unroll
                for (int t = 0; t < 9; ++t) X::mma(acc[t], af, win[(s + t / 3) % 3][t % 3]);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const u32x4 af = Frag16::load(dy_rd + s * 16 * ROWB, dy_rd + (s * 16 + 4) * ROWB);
                const int hs = wg_halo_pix<GEO>(s * 16) * ROWB;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const int ky = (TAPS == 9) ? t / 3 : 1, kx = (TAPS == 9) ? t % 3 : 1;
                    const int toff = (ky * HW2 + kx) * ROWB + hs;
                    const u32x4 bf = Frag16::load(ah_rd0 + toff, ah_rd1 + toff);
                    X::mma(acc[t], af, bf);
                }
            }
        }
        // this buffer is requested again two tiles on: every wave must have read its fragments first
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        buf ^= 1;
    }
#undef WD_ISSUE

    const int kcol = k0 + cisub * 32 + (lane & 31);
    if (kcol < p.cin_pad) {
        float* dst = p.dw + (p.parts > 0 ? (long)blockIdx.x * p.part_stride : 0L);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + cosub * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (n < p.cout) {
                    float* qd = dst + ((long)n * TAPS + t) * p.cin_pad + kcol;
                    if (p.parts > 0) *qd = acc[t][e];
                    else atomicAdd(qd, acc[t][e]);
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// Two-group build of the lean 3x3 kernel (16x8-pixel tiles).  Phase stamps of the kernel above at one workgroup per CU (one
// wave per SIMD): of ~6000 clocks per pixel tile only ~2580 are the 72 MFMAs; ~1830 are the commit (BatchNorm/ReLU prologue +
// LDS writes) and ~1140 the issue of the next tile (tile coordinates by division, descriptors), and nothing overlaps them.
// Here a workgroup is 8 waves = two groups of 4; both groups accumulate the SAME slab over alternate pixel tiles of the
// workgroup's subset, one phase apart:   A: commit+issue | multiply | commit+issue | ...
//                                        B:      -       | commit+issue | multiply | ...
// so each SIMD holds one wave in its MFMA block and one in its vector/LDS-write block.  Each group has its own LDS tile;
// the phases are separated by workgroup barriers.  At the end group B hands its accumulators to group A through LDS (3 taps
// per round), so the partial-slab traffic stays that of 256 workgroups.  Tile coordinates advance incrementally (no division).
template <typename T, int CO_T, int CI_T>
__global__ __launch_bounds__(512, 1) void conv_wgrad16x2_kernel(const WgradArgs p) {
    using X = TT<T>;
    static_assert(sizeof(T) == 2, "16-bit types");
    constexpr int GEO = 0, TAPS = 9, EPS = 8, ESZ = 2;
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 1;
    constexpr int HW2 = TW + 2, HPI = (TH + 2) * (TW + 2), HP = NI * HPI;
    constexpr int ROWB = 64, PPS = 4;
    constexpr int CO_S = CO_T / 32, CI_S = CI_T / 32;
    static_assert(CO_S * CI_S == 4, "one (cout sub, cin sub) pair per wave");
    constexpr int DY_BYTES = CO_S * 128 * ROWB, AH_BYTES = CI_S * HP * ROWB, GRP_BYTES = DY_BYTES + AH_BYTES;
    constexpr int DY_PPP = CO_S * PPS, AH_PPP = CI_S * PPS;
    constexpr int DY_PIECES = 128 * DY_PPP, AH_PIECES = HP * AH_PPP;
    constexpr int DY_ITEMS = DY_PIECES / 256, AH_ITEMS = (AH_PIECES + 255) / 256;
    static_assert(DY_PIECES % 256 == 0 && 256 % DY_PPP == 0 && 256 % AH_PPP == 0 && AH_ITEMS <= 16, "item geometry");
    static_assert(2 * GRP_BYTES >= 12 * 256 * 16, "hand-over buffer");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef X2_GRP0
    const int grp = 0;
#else
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
#endif
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
#ifdef X2_SAMELDS
    char* const Dy = smem;
#else
    char* const Dy = smem + grp * GRP_BYTES;
#endif
    char* const Ah = Dy + DY_BYTES;
    const int cosub = wave % CO_S, cisub = wave / CO_S;
    const int ct = blockIdx.y;
    const int n0 = (ct % p.co_tiles) * CO_T, k0 = (ct / p.co_tiles) * CI_T;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- tile-independent item descriptors (as in conv_wgrad16_kernel)
    unsigned dy_off[DY_ITEMS], ah_off[AH_ITEMS];
    unsigned long long cls = 0;
    int dy_lds, ah_lds;
    {
        const int pc = tid % DY_PPP, sub = pc / PPS, pin = pc % PPS;
        const int ch = n0 + pc * EPS;
        dy_lds = sub * 128 * ROWB + (tid / DY_PPP) * ROWB + pin * 16;
#pragma unroll
        for (int it = 0; it < DY_ITEMS; ++it) {
            const int m = tid / DY_PPP + it * (256 / DY_PPP);
            const int tx = m & (TW - 1), ty = (m >> TWL) & (TH - 1);
            dy_off[it] = ch < p.cout ? (unsigned)(wg_rel_pix(0, ty, tx, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ + ch * ESZ) + WG_BIAS : 0xffffffffu;
        }
    }
    float sc[EPS], sh[EPS];
    {
        const int pc = tid % AH_PPP, sub = pc / PPS, pin = pc % PPS;
        const int ch = k0 + pc * EPS;
        const bool ch_ok = ch < p.cin_pad;
        ah_lds = sub * HP * ROWB + (tid / AH_PPP) * ROWB + pin * 16;
#pragma unroll
        for (int it = 0; it < AH_ITEMS; ++it) {
            const int idx = tid + it * 256;
            const int pp = tid / AH_PPP + it * (256 / AH_PPP);
            const int hy = pp / HW2, hx = pp % HW2;
            const bool live = idx < AH_PIECES && ch_ok;
            ah_off[it] = live ? (unsigned)(wg_rel_pix(0, hy - 1, hx - 1, p.H, p.W, p.in_blk) * p.in_cs * ESZ + ch * ESZ) + WG_BIAS : 0xffffffffu;
            cls |= (unsigned long long)((hx == 0 ? 1u : 0u) | (hx == TW + 1 ? 2u : 0u) | (hy == 0 ? 4u : 0u) | (hy == TH + 1 ? 8u : 0u)) << (4 * it);
        }
#pragma unroll
        for (int e = 0; e < EPS; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
        if (p.prologue == PSSR_PRO_BN_RELU && ch_ok) {
            load4(p.pro_scale + ch, sc); load4(p.pro_scale + ch + 4, sc + 4);
            load4(p.pro_shift + ch, sh); load4(p.pro_shift + ch + 4, sh + 4);
        }
    }
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pcol = li & 3;
    const int colb = ((g & 1) * 16 + pcol * 4) * 2;
    const int m00 = (g >> 1) * 8 + q;
    const char* const dy_rd = Dy + cosub * 128 * ROWB + m00 * ROWB + colb;
    const char* const ah_rd0 = Ah + cisub * HP * ROWB + wg_halo_pix<GEO>(m00) * ROWB + colb;
    const char* const ah_rd1 = Ah + cisub * HP * ROWB + wg_halo_pix<GEO>(m00 + 4) * ROWB + colb;

    const char* const dy_base = (const char*)p.dy + (long)p.dy_co * ESZ - WG_BIAS;
    const char* const in_base = (const char*)p.in + (long)p.in_co * ESZ - WG_BIAS;

    u32x4 dy_reg[DY_ITEMS], ah_reg[AH_ITEMS];
    unsigned long long ah_bad = 0;

    // ---- this group's tiles: ordinals grp, grp + 2, ... of the workgroup's subset blockIdx.x + o * split
    const int cnt = (int)blockIdx.x < p.n_tiles ? (p.n_tiles - (int)blockIdx.x + p.split - 1) / p.split : 0;
    const int n_mine = (cnt - grp + 1) >> 1, n_a = (cnt + 1) >> 1;
    int tc_x, tc_y, tc_i;                         // coordinates of the next tile to issue (scalar registers)
    int st_x, st_y, st_i;                         // 2 * split in the same mixed radix
    {
        int t0 = (int)blockIdx.x + grp * p.split;
        tc_x = t0 % p.tiles_x; t0 /= p.tiles_x; tc_y = t0 % p.tiles_y; tc_i = t0 / p.tiles_y;
        int s2 = 2 * p.split;
        st_x = s2 % p.tiles_x; s2 /= p.tiles_x; st_y = s2 % p.tiles_y; st_i = s2 / p.tiles_y;
    }

#define WX_ISSUE()                                                                                                \
    {                                                                                                             \
        const int x0_ = tc_x << TWL, y0_ = tc_y << THL, img0_ = tc_i;                                             \
        const unsigned tb_ = (x0_ == 0 ? 1u : 0u) | (x0_ + TW == p.W ? 2u : 0u) | (y0_ == 0 ? 4u : 0u) | (y0_ + TH == p.H ? 8u : 0u); \
        ah_bad = cls & (tb_ * 0x1111111111111111ull);                                                             \
        const __amdgpu_buffer_rsrc_t rdy_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(dy_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        const __amdgpu_buffer_rsrc_t rin_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(in_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.in_blk) * p.in_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        _Pragma("unroll") for (int it = 0; it < DY_ITEMS; ++it)                                                   \
            dy_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy_, (int)dy_off[it], 0, 0)); \
        _Pragma("unroll") for (int it = 0; it < AH_ITEMS; ++it) {                                                 \
            const unsigned off_ = ((ah_bad >> (4 * it)) & 0xfu) ? 0xffffffffu : ah_off[it];                       \
            ah_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin_, (int)off_, 0, 0)); \
        }                                                                                                         \
        tc_x += st_x;                                                                                             \
        const int cx_ = tc_x >= p.tiles_x ? 1 : 0;                                                                \
        tc_x -= cx_ ? p.tiles_x : 0;                                                                              \
        tc_y += st_y + cx_;                                                                                       \
        const int cy_ = tc_y >= p.tiles_y ? 1 : 0;                                                                \
        tc_y -= cy_ ? p.tiles_y : 0;                                                                              \
        tc_i += st_i + cy_;                                                                                       \
    }
#define WX_COMMIT()                                                                                               \
    {                                                                                                             \
        _Pragma("unroll") for (int it = 0; it < DY_ITEMS; ++it)                                                   \
            *(u32x4*)(Dy + dy_lds + it * (256 / DY_PPP) * ROWB) = dy_reg[it];                                     \
        _Pragma("unroll") for (int it = 0; it < AH_ITEMS; ++it) {                                                 \
            if ((it + 1) * 256 <= AH_PIECES || tid < AH_PIECES - it * 256) {                                      \
                u32x4 v = ah_reg[it];                                                                             \
                if (p.prologue == PSSR_PRO_BN_RELU) {                                                             \
                    v = X::bn_relu(v, sc, sh);                                                                    \
                    if ((ah_bad >> (4 * it)) & 0xfu) v = u32x4{0u, 0u, 0u, 0u};                                   \
                } else if (p.prologue == PSSR_PRO_GELU) {                                                         \
                    float f[EPS];                                                                                 \
                    X::unpack(v, f);                                                                              \
                    gelu_vec<T, EPS>(f);                          \
                    v = X::pack(f);                                                                               \
                }                                                                                                 \
                *(u32x4*)(Ah + ah_lds + it * (256 / AH_PPP) * ROWB) = v;                                          \
            }                                                                                                     \
        }                                                                                                         \
    }
#define WX_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define WX_MULTIPLY()                                                                                              \
    {                                                                                                             \
        u32x4 win[3][3];                                                                                          \
        _Pragma("unroll") for (int r = 0; r < 2; ++r)                                                             \
            _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                      \
                win[r][kx] = Frag16::load(ah_rd0 + (r * HW2 + kx) * ROWB, ah_rd1 + (r * HW2 + kx) * ROWB);        \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) {                                                           \
            const u32x4 af = Frag16::load(dy_rd + s * 16 * ROWB, dy_rd + (s * 16 + 4) * ROWB);                    \
            _Pragma("unroll") for (int kx = 0; kx < 3; ++kx)                                                      \
                win[(s + 2) % 3][kx] = Frag16::load(ah_rd0 + ((s + 2) * HW2 + kx) * ROWB, ah_rd1 + ((s + 2) * HW2 + kx) * ROWB); \
            _Pragma("unroll") for (int t = 0; t < 9; ++t) X::mma(acc[t], af, win[(s + t / 3) % 3][t % 3]);        \
        }                                                                                                         \
    }
    // Both groups run the same loop  { commit | barrier | issue next, multiply | barrier };  group B enters it one barrier
    // later, so A's multiply epoch is B's commit epoch and vice versa (the hardware barrier only counts arrivals).  The
    // barrier counts are evened out after the loop: A has n_a tiles, B n_a or n_a - 1.
#ifdef PSSR_WG_STAMPS
    unsigned long long st_prev = 0;
    unsigned st_sum[6] = {0, 0, 0, 0, 0, 0};
#define WX_STAMP(I)                                                                                               \
    {                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        unsigned long long t_;                                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                              \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        if ((I) >= 0) st_sum[(I) < 0 ? 0 : (I)] += (unsigned)(t_ - st_prev);                                      \
        st_prev = t_;                                                                                             \
    }
#else
#define WX_STAMP(I)
#endif
    if (n_mine > 0) WX_ISSUE()
    if (grp == 1) WX_BARRIER();
    WX_STAMP(-1)
    for (int j = 0; j < n_mine; ++j) {
#ifdef PSSR_WG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // segment 0 = exposed load latency
#endif
        WX_STAMP(0)
        WX_COMMIT()
        WX_STAMP(1)
        WX_BARRIER();
        WX_STAMP(2)
        if (j + 1 < n_mine) WX_ISSUE()
        WX_STAMP(3)
        WX_MULTIPLY()
        WX_STAMP(4)
        WX_BARRIER();
        WX_STAMP(5)
    }
#ifdef PSSR_WG_STAMPS
    if (p.stamps && lane == 0) {
        unsigned* q = p.stamps + (((long)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + grp) * 4 + wave) * 8;
        for (int i = 0; i < 6; ++i) q[i] = st_sum[i];
        q[6] = n_mine;
    }
#endif
#undef WX_STAMP
    if (grp == 0) WX_BARRIER();
    else if (n_mine < n_a) { WX_BARRIER(); WX_BARRIER(); }
#undef WX_MULTIPLY
#undef WX_ISSUE
#undef WX_COMMIT

    // ---- group B hands its accumulators over, 3 taps per round ([slot][thread] float4: conflict-free both ways)
    float4* const xb = (float4*)smem;
#ifndef X2_NO_HANDOVER
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (grp == 1) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    xb[(tt * 4 + qq) * 256 + tid] = make_float4(acc[3 * r + tt][4 * qq], acc[3 * r + tt][4 * qq + 1], acc[3 * r + tt][4 * qq + 2], acc[3 * r + tt][4 * qq + 3]);
        }
        WX_BARRIER();
        if (grp == 0) {
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 v = xb[(tt * 4 + qq) * 256 + tid];
                    acc[3 * r + tt][4 * qq] += v.x; acc[3 * r + tt][4 * qq + 1] += v.y; acc[3 * r + tt][4 * qq + 2] += v.z; acc[3 * r + tt][4 * qq + 3] += v.w;
                }
        }
        WX_BARRIER();
    }
#endif
#undef WX_BARRIER
    if (grp != 0) return;

    const int kcol = k0 + cisub * 32 + (lane & 31);
    if (kcol < p.cin_pad) {
        float* dst = p.dw + (p.parts > 0 ? (long)blockIdx.x * p.part_stride : 0L);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + cosub * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (n < p.cout) {
                    float* qd = dst + ((long)n * TAPS + t) * p.cin_pad + kcol;
                    if (p.parts > 0) *qd = acc[t][e];
                    else atomicAdd(qd, acc[t][e]);
                }
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// 1x1 weight gradients (16-bit, full tiles): with one tap the slab kernel above multiplies 8 MFMAs per wave and pixel
// tile against a full staging round.  Here a workgroup owns a 128 x 128 slab of dW, a wave 2 x 2 of its 32 x 32 tiles
// (64 accumulator registers, 32 MFMAs per wave and pixel tile from 4 transposed fragments per k-step), no halo.
template <typename T, int GEO>
__global__ __launch_bounds__(256, 2) void conv_wgrad16_1x1_kernel(const WgradArgs p) {
    using X = TT<T>;
    static_assert(sizeof(T) == 2 && GEO < 2, "16-bit types, tiles of at least 8x8 pixels");
    constexpr int EPS = 8, ESZ = 2;
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    constexpr int ROWB = 64, PPP = 16, ITEMS = 8;                  // 16 pieces of 16 bytes per pixel and operand, 8 items per thread
    constexpr int OP_BYTES = 4 * 128 * ROWB;                       // [4 sub-tiles][128 px][64 B]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Dy = smem;
    char* Xt = smem + OP_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cw = wave & 1, kw = wave >> 1;
    const int ct = blockIdx.y;
    const int n0 = (ct % p.co_tiles) * 128, k0 = (ct / p.co_tiles) * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // tile-independent descriptors: piece tid + it * 256 = (pixel tid / 16 + 16 it, 16-byte piece tid % 16)
    const int pc = tid % PPP, sub = pc / 4, pin = pc % 4;
    const int lds0 = sub * 128 * ROWB + (tid / PPP) * ROWB + pin * 16;
    unsigned dy_off[ITEMS], x_off[ITEMS];
    const int chd = n0 + pc * EPS, chx = k0 + pc * EPS;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int m = tid / PPP + it * 16;
        const int tx = m & (TW - 1), ty = (m >> TWL) & (TH - 1), img = m >> (TWL + THL);
        dy_off[it] = chd < p.cout ? (unsigned)(wg_rel_pix(img, ty, tx, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ + chd * ESZ) + WG_BIAS : 0xffffffffu;
        x_off[it] = chx < p.cin_pad ? (unsigned)(wg_rel_pix(img, ty, tx, p.H, p.W, p.in_blk) * p.in_cs * ESZ + chx * ESZ) + WG_BIAS : 0xffffffffu;
    }
    float sc[EPS], sh[EPS];
#pragma unroll
    for (int e = 0; e < EPS; ++e) { sc[e] = 0.f; sh[e] = 0.f; }
    if (p.prologue == PSSR_PRO_BN_RELU && chx < p.cin_pad) {
        load4(p.pro_scale + chx, sc); load4(p.pro_scale + chx + 4, sc + 4);
        load4(p.pro_shift + chx, sh); load4(p.pro_shift + chx + 4, sh + 4);
    }
    const int g = lane >> 4, li = lane & 15, q = li >> 2, pcol = li & 3;
    const int fr_off = ((g >> 1) * 8 + q) * ROWB + ((g & 1) * 16 + pcol * 4) * 2;
    const char* const dy_rd = Dy + (2 * cw) * 128 * ROWB + fr_off;
    const char* const x_rd = Xt + (2 * kw) * 128 * ROWB + fr_off;
    const char* const dy_base = (const char*)p.dy + (long)p.dy_co * ESZ - WG_BIAS;
    const char* const in_base = (const char*)p.in + (long)p.in_co * ESZ - WG_BIAS;

    u32x4 dy_reg[ITEMS], x_reg[ITEMS];
#define W1_ISSUE(TILE)                                                                                            \
    {                                                                                                             \
        int tmi_ = (TILE);                                                                                        \
        const int tile_x_ = tmi_ % p.tiles_x; tmi_ /= p.tiles_x;                                                  \
        const int tile_y_ = tmi_ % p.tiles_y;                                                                     \
        const int tile_i_ = tmi_ / p.tiles_y;                                                                     \
        const int x0_ = tile_x_ << TWL, y0_ = tile_y_ << THL, img0_ = tile_i_ * NI;                               \
        const __amdgpu_buffer_rsrc_t rdy_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(dy_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.dy_blk) * p.dy_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        const __amdgpu_buffer_rsrc_t rin_ = __builtin_amdgcn_make_buffer_rsrc(                                    \
            (void*)(in_base + pix_index(img0_, y0_, x0_, p.H, p.W, p.in_blk) * p.in_cs * ESZ), 0, (int)0xfffffff0u, 0x00020000); \
        _Pragma("unroll") for (int it = 0; it < ITEMS; ++it) {                                                    \
            dy_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rdy_, (int)dy_off[it], 0, 0)); \
            x_reg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin_, (int)x_off[it], 0, 0));   \
        }                                                                                                         \
    }
    int tile = blockIdx.x;
    if (tile < p.n_tiles) W1_ISSUE(tile)
    for (; tile < p.n_tiles; tile += p.split) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            *(u32x4*)(Dy + lds0 + it * 16 * ROWB) = dy_reg[it];
            u32x4 v = x_reg[it];
            if (p.prologue == PSSR_PRO_BN_RELU) v = X::bn_relu(v, sc, sh);
            else if (p.prologue == PSSR_PRO_GELU) {
                float f[EPS];
                X::unpack(v, f);
                gelu_vec<T, EPS>(f);
                v = X::pack(f);
            }
            *(u32x4*)(Xt + lds0 + it * 16 * ROWB) = v;
        }
        __syncthreads();
        const int nt = tile + p.split;
        if (nt < p.n_tiles) W1_ISSUE(nt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            u32x4 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = Frag16::load(dy_rd + a * 128 * ROWB + s * 16 * ROWB, dy_rd + a * 128 * ROWB + (s * 16 + 4) * ROWB);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = Frag16::load(x_rd + b * 128 * ROWB + s * 16 * ROWB, x_rd + b * 128 * ROWB + (s * 16 + 4) * ROWB);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) X::mma(acc[a][b], af[a], bf[b]);
        }
        __syncthreads();
    }
#undef W1_ISSUE
    float* dst = p.dw + (p.parts > 0 ? (long)blockIdx.x * p.part_stride : 0L);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int kcol = k0 + (2 * kw + b) * 32 + (lane & 31);
            if (kcol >= p.cin_pad) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + (2 * cw + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (n < p.cout) {
                    float* qd = dst + (long)n * p.cin_pad + kcol;
                    if (p.parts > 0) *qd = acc[a][b][e];
                    else atomicAdd(qd, acc[a][b][e]);
                }
            }
        }
}

// full tiles and offsets within the bias window: what the lean-loader kernels need (same answer at query and launch time)
template <int GEO> bool wg_lean_ok(const WgradArgs& p) {
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    const int lean = pssr_tunables().wgrad_lean;
    const bool full = p.W % TW == 0 && p.H % TH == 0 && p.N % NI == 0;
    const long span_dy = (p.dy_blk ? (long)p.H * p.W * NI : (long)(TH + 2) * p.W + (long)(NI - 1) * p.H * p.W) * p.dy_cs * 2;
    const long span_in = (p.in_blk ? (long)p.H * p.W * NI : (long)(TH + 2) * p.W + (long)(NI - 1) * p.H * p.W) * p.in_cs * 2;
    return lean && full && span_dy < (1L << 30) && span_in < (1L << 30);
}

template <typename T, int GEO>
int launch_1x1(WgradArgs p, hipStream_t stream, int* query) {
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    constexpr int LDS = 2 * 4 * 128 * 64;
    p.tiles_x = cdiv(p.W, TW); p.tiles_y = cdiv(p.H, TH); p.tiles_i = cdiv(p.N, NI);
    p.n_tiles = p.tiles_x * p.tiles_y * p.tiles_i;
    p.co_tiles = cdiv(p.cout, 128); p.ci_tiles = cdiv(p.cin_pad, 128);
    const int slabs = p.co_tiles * p.ci_tiles;
    int split;
    if (query != nullptr || p.parts > 0) {
        const int target = pssr_tunables().wgrad_blocks_1x1;
        split = cdiv(target, slabs);
        if (split > p.n_tiles) split = p.n_tiles;
        if (split < 1) split = 1;
        if (query != nullptr) { *query = split; return PSSR_OK; }
        PSSR_CHECK(p.parts == split, PSSR_ERR_ARG, "wgrad: dw_parts=%d but this shape needs %d (ask pssr_conv2d_wgrad_parts)", p.parts, split);
        p.part_stride = (long)p.cout * p.cin_pad;
    } else {
        split = p.n_tiles / 8;
        if (split > cdiv(1024, slabs)) split = cdiv(1024, slabs);
        if (split < 1) split = 1;
        if (slabs * split < 256) split = cdiv(256, slabs);
        if (split > p.n_tiles) split = p.n_tiles;
        p.part_stride = 0;
    }
    p.split = split;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad16_1x1_kernel<T, GEO>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    hipLaunchKernelGGL((conv_wgrad16_1x1_kernel<T, GEO>), dim3(split, slabs), dim3(256), LDS, stream, p);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

template <typename T, int CO_T, int CI_T, int GEO, int TAPS>
int launch_t(WgradArgs p, hipStream_t stream, int* query) {
    constexpr int TWL = WGeo<GEO>::TWL, THL = WGeo<GEO>::THL;
    constexpr int TW = 1 << TWL, TH = 1 << THL, NI = 128 >> (TWL + THL);
    constexpr int HP = NI * (TH + 2) * (TW + 2);
    constexpr int ROWB = 32 * (int)sizeof(T);
    constexpr int LDS = (CO_T / 32) * 128 * ROWB + (CI_T / 32) * HP * ROWB;
    p.tiles_x = cdiv(p.W, TW); p.tiles_y = cdiv(p.H, TH); p.tiles_i = cdiv(p.N, NI);
    p.n_tiles = p.tiles_x * p.tiles_y * p.tiles_i;
    p.co_tiles = cdiv(p.cout, CO_T); p.ci_tiles = cdiv(p.cin_pad, CI_T);
    const int slabs = p.co_tiles * p.ci_tiles;
    int split;
    if (query != nullptr || p.parts > 0) {
        // partial-slab mode: ~1 workgroup per CU; every workgroup stores its slab once (plain stores), so the
        // extra traffic is split * |dW| written + read back by the unpack/reduce pass.  The weight gradients run on a second
        // stream under the backward chain: with 512 workgroups they finished sooner but took more of the chip and twice the
        // slab traffic from the kernels on that chain (measured per step, c2: 768 -> 14.2 ms, 512 -> 13.6, 384 -> 13.5,
        // 256 -> 13.1, 192 -> 13.5, 128 -> 14.0)
        const int target = pssr_tunables().wgrad_blocks;
        split = cdiv(target, slabs);
        if (split > p.n_tiles) split = p.n_tiles;
        if (split < 1) split = 1;
        if (query != nullptr) { *query = split; return PSSR_OK; }
        PSSR_CHECK(p.parts == split, PSSR_ERR_ARG, "wgrad: dw_parts=%d but this shape needs %d (ask pssr_conv2d_wgrad_parts)", p.parts, split);
        p.part_stride = (long)p.cout * TAPS * p.cin_pad;
    } else {
        // atomic mode: every workgroup ends with one f32 atomic add per slab element; at the chip-wide float-atomic
        // rate (~1.3 TB/s) that hides behind the MFMAs only if a workgroup reduces over >= 8 pixel tiles first
        split = p.n_tiles / 8;
        if (split > cdiv(1024, slabs)) split = cdiv(1024, slabs);
        if (split < 1) split = 1;
        if (slabs * split < 256) split = cdiv(256, slabs);
        if (split > p.n_tiles) split = p.n_tiles;
        p.part_stride = 0;
    }
    p.split = split;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<T, CO_T, CI_T, GEO, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    if constexpr (sizeof(T) == 2 && GEO < 2) {
        // lean-loader kernel: full tiles only, relative offsets within +-2^30 bytes of the tile origin
        const bool full = p.W % TW == 0 && p.H % TH == 0 && p.N % NI == 0;
        const long span_dy = (p.dy_blk ? (long)p.H * p.W * NI : (long)(TH + 2) * p.W + (long)(NI - 1) * p.H * p.W) * p.dy_cs * 2;
        const long span_in = (p.in_blk ? (long)p.H * p.W * NI : (long)(TH + 2) * p.W + (long)(NI - 1) * p.H * p.W) * p.in_cs * 2;
        const int lean = pssr_tunables().wgrad_lean;
        if (lean && full && span_dy < (1L << 30) && span_in < (1L << 30)) {
            if constexpr (GEO == 0 && TAPS == 9) {
                if (pssr_tunables().wgrad_x2) {
                    static bool attrx2_done = false;
                    if (!attrx2_done) {
                        (void)hipFuncSetAttribute((const void*)conv_wgrad16x2_kernel<T, CO_T, CI_T>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * LDS);
                        attrx2_done = true;
                    }
                    hipLaunchKernelGGL((conv_wgrad16x2_kernel<T, CO_T, CI_T>), dim3(split, slabs), dim3(512), 2 * LDS, stream, p);
                    PSSR_LAUNCH_CHECK();
                    return PSSR_OK;
                }
            }
            if constexpr (TAPS == 9 && CO_T == 64 && CI_T == 64) {
                if (p.prologue == PSSR_PRO_NONE && pssr_tunables().wgrad_dma) {
                    constexpr int HPP = (HP + 31) / 32 * 32;
                    constexpr int LDS_D = 2 * ((CO_T / 32) * 128 * 64 + (CI_T / 32) * HPP * 64);
                    static bool attrd_done = false;
                    if (!attrd_done) {
                        (void)hipFuncSetAttribute((const void*)conv_wgrad16d_kernel<T, CO_T, CI_T, GEO, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_D);
                        attrd_done = true;
                    }
                    hipLaunchKernelGGL((conv_wgrad16d_kernel<T, CO_T, CI_T, GEO, TAPS>), dim3(split, slabs), dim3(256), LDS_D, stream, p);
                    PSSR_LAUNCH_CHECK();
                    return PSSR_OK;
                }
            }
            static bool attr16_done = false;
            if (!attr16_done) {
                (void)hipFuncSetAttribute((const void*)conv_wgrad16_kernel<T, CO_T, CI_T, GEO, TAPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                attr16_done = true;
            }
            hipLaunchKernelGGL((conv_wgrad16_kernel<T, CO_T, CI_T, GEO, TAPS>), dim3(split, slabs), dim3(256), LDS, stream, p);
            PSSR_LAUNCH_CHECK();
            return PSSR_OK;
        }
    }
    hipLaunchKernelGGL((conv_wgrad_kernel<T, CO_T, CI_T, GEO, TAPS>), dim3(split, slabs), dim3(256), LDS, stream, p);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

template <typename T, int CO_T, int CI_T, int GEO>
int launch(const WgradArgs& p, hipStream_t stream, int* query) {
    return p.taps == 9 ? launch_t<T, CO_T, CI_T, GEO, 9>(p, stream, query) : launch_t<T, CO_T, CI_T, GEO, 1>(p, stream, query);
}

template <typename T, int CO_T, int CI_T>
int launch_geo(const WgradArgs& a, hipStream_t s, int* query) {
    const int w = a.W;
    if (w > 8) return launch<T, CO_T, CI_T, 0>(a, s, query);
    if (w > 4) return launch<T, CO_T, CI_T, 1>(a, s, query);
    if (w > 2) return launch<T, CO_T, CI_T, 2>(a, s, query);
    // 2x2 / 1x1 images: the 9x halo blow-up does not fit LDS; training at such sizes is out of scope
    pssr_set_error("wgrad: spatial width %d < 3 is not supported", w);
    return PSSR_ERR_UNSUPPORTED;
}

template <typename T>
int launch_shape(const WgradArgs& a, hipStream_t s, int* query) {
    if constexpr (sizeof(T) == 2) {
        // 1x1 with both channel counts beyond one 32-wide sub-tile: the 128 x 128 slab kernel (lean-loader launches only)
        if (a.taps == 1 && a.cout > 32 && a.cin_pad > 32) {
            if (a.W > 8 && wg_lean_ok<0>(a)) return launch_1x1<T, 0>(a, s, query);
            if (a.W > 4 && a.W <= 8 && wg_lean_ok<1>(a)) return launch_1x1<T, 1>(a, s, query);
        }
    }
    if (a.cout > 64 && a.cin_pad <= 32) return launch_geo<T, 128, 32>(a, s, query);
    if (a.cout <= 32) return launch_geo<T, 32, 128>(a, s, query);
    return launch_geo<T, 64, 64>(a, s, query);
}

}  // namespace

static int wgrad_entry(const pssr_wgrad_desc* d, pssr_stream_t stream, int* query);

#ifdef PSSR_WG_STAMPS
static unsigned* g_wg_stamp_buf = nullptr;
extern "C" void pssr_debug_wgrad_stamp_buffer(void* p) { g_wg_stamp_buf = (unsigned*)p; }
#endif

extern "C" int pssr_conv2d_wgrad(const pssr_wgrad_desc* d, pssr_stream_t stream) { return wgrad_entry(d, stream, nullptr); }

extern "C" int pssr_conv2d_wgrad_parts(const pssr_wgrad_desc* d) {
    int parts = 0;
    const int rc = wgrad_entry(d, nullptr, &parts);
    return rc != PSSR_OK ? rc : parts;
}

static int wgrad_entry(const pssr_wgrad_desc* d, pssr_stream_t stream, int* query) {
    PSSR_CHECK(d != nullptr, PSSR_ERR_ARG, "wgrad: null desc");
    PSSR_CHECK(d->dtype == PSSR_F32 || d->dtype == PSSR_BF16 || d->dtype == PSSR_F16, PSSR_ERR_ARG, "wgrad: bad dtype %d", d->dtype);
    const int esz = d->dtype == PSSR_F32 ? 4 : 2;
    PSSR_CHECK(d->n > 0 && d->h > 0 && d->w > 0, PSSR_ERR_ARG, "wgrad: bad shape");
    PSSR_CHECK(query != nullptr || (d->dy && d->in && d->dw), PSSR_ERR_ARG, "wgrad: null pointer");
    PSSR_CHECK(d->dw_parts >= 0, PSSR_ERR_ARG, "wgrad: dw_parts=%d", d->dw_parts);
    PSSR_CHECK(d->taps == 9 || d->taps == 1, PSSR_ERR_ARG, "wgrad: taps=%d", d->taps);
    PSSR_CHECK(d->cin_pad > 0 && d->cin_pad % (d->dtype == PSSR_F32 ? 8 : 16) == 0, PSSR_ERR_ARG, "wgrad: cin_pad=%d must be a multiple of the K-chunk", d->cin_pad);
    PSSR_CHECK(d->cout > 0 && (d->cout * esz) % 16 == 0, PSSR_ERR_ARG, "wgrad: cout=%d must fill whole 16-byte slots", d->cout);
    PSSR_CHECK((d->dy_cstride * esz) % 16 == 0 && (d->dy_coff * esz) % 16 == 0 && d->dy_coff + d->cout <= d->dy_cstride, PSSR_ERR_ARG, "wgrad: dy stride/offset");
    PSSR_CHECK((d->in_cstride * esz) % 16 == 0 && (d->in_coff * esz) % 16 == 0 && d->in_coff + d->cin_pad <= d->in_cstride, PSSR_ERR_ARG, "wgrad: in stride/offset");
    PSSR_CHECK(d->prologue >= 0 && d->prologue <= PSSR_PRO_GELU, PSSR_ERR_ARG, "wgrad: prologue=%d", d->prologue);
    PSSR_CHECK(d->prologue != PSSR_PRO_BN_RELU || (d->pro_scale && d->pro_shift), PSSR_ERR_ARG, "wgrad: prologue needs scale/shift");
    PSSR_CHECK(d->dy_blk >= 0 && d->in_blk >= 0 && (d->h % (1 << d->dy_blk)) == 0 && (d->w % (1 << d->in_blk)) == 0, PSSR_ERR_ARG, "wgrad: blocked layout");
    WgradArgs a;
    a.N = d->n; a.H = d->h; a.W = d->w;
    a.dy = d->dy; a.dy_cs = d->dy_cstride; a.dy_co = d->dy_coff; a.dy_blk = d->dy_blk; a.cout = d->cout;
    a.in = d->in; a.in_cs = d->in_cstride; a.in_co = d->in_coff; a.in_blk = d->in_blk; a.cin_pad = d->cin_pad;
    a.taps = d->taps; a.prologue = d->prologue; a.pro_scale = d->pro_scale; a.pro_shift = d->pro_shift;
    a.dw = d->dw; a.parts = d->dw_parts; a.part_stride = 0;
    a.dbg = pssr_tunables().igemm_dbg;
    a.stamps = nullptr;
#ifdef PSSR_WG_STAMPS
    a.stamps = g_wg_stamp_buf;
#endif
    hipStream_t s = (hipStream_t)stream;
#ifdef PSSR_WGRAD_DEV       // development builds: instantiate only the kernel being worked on (resource usage / ISA in seconds)
    hipLaunchKernelGGL((conv_wgrad16x2_kernel<bf16_t, 64, 64>), dim3(1, 1), dim3(512), 0, s, a);
    return PSSR_OK;
#else
    return d->dtype == PSSR_BF16 ? launch_shape<bf16_t>(a, s, query)
         : d->dtype == PSSR_F16 ? launch_shape<f16_t>(a, s, query) : launch_shape<float>(a, s, query);
#endif
}
