// Weight packing (OIHW f32 -> K-chunked LDS-image order in the compute dtype) and the inverse
// scatter for weight gradients.  Pure data movement: one thread per packed element.
#include "common.h"

namespace {

// Maps a packed position (chunk, tap, n, slotpos, e) to the source OIHW element, or -1 for padding.
struct PackMap {
    int cout, cin, ks, ci_begin, ci_count, mode, taps;
    const int32_t* n_perm;
    int center;   // modes 2/3 only: the tensor is a 1x1 weight seen as the centre tap of a 3x3 kernel (ks must be 3)
    __device__ __forceinline__ long src_index(int k, int tap, int n) const {
        // k: GEMM-K index (un-padded range checked by caller), n: GEMM-N index
        const int kk = ks * ks;
        if (center) {
            const int flat = mode == 2 ? k : n, other = mode == 2 ? n : k;
            if (flat >= ci_count * kk || other >= cout || flat % kk != kk / 2) return -1;
            const int co = n_perm ? n_perm[other] : other;
            return (long)co * cin + ci_begin + flat / kk;
        }
        if (mode == 0) {          // forward: K = ci, N = co
            if (k >= ci_count || n >= cout) return -1;
            const int co = n_perm ? n_perm[n] : n;
            return ((long)co * cin + ci_begin + k) * kk + tap;
        } else if (mode == 1) {   // dgrad: K = co, N = ci, taps flipped
            if (k >= cout || n >= ci_count) return -1;
            const int co = n_perm ? n_perm[k] : k;
            return ((long)co * cin + ci_begin + n) * kk + (kk - 1 - tap);
        } else if (mode == 2) {   // flat-K: K = (ci - ci_begin)*kk + tap', single tap
            if (k >= ci_count * kk || n >= cout) return -1;
            const int co = n_perm ? n_perm[n] : n;
            return ((long)co * cin + ci_begin) * kk + k;
        } else if (mode == 3) {   // flat-K dgrad: K = co, N = (ci - ci_begin)*kk + tap', single tap
            if (k >= cout || n >= ci_count * kk) return -1;
            const int co = n_perm ? n_perm[k] : k;
            return ((long)co * cin + ci_begin) * kk + n;
        } else if (mode == 4) {   // space-to-depth: K = tap'*cpad + ci (cpad = cin rounded up to 16), N = co, single tap
            const int cpad = (cin + 15) / 16 * 16;
            const int tp = k / cpad, ci = k % cpad;
            if (tp >= kk || ci >= cin || n >= cout) return -1;
            return ((long)n * cin + ci) * kk + tp;
        } else {                  // space-to-depth dgrad: K = co, N = tap'*cpad + ci
            const int cpad = (cin + 15) / 16 * 16;
            const int tp = n / cpad, ci = n % cpad;
            if (tp >= kk || ci >= cin || k >= cout) return -1;
            return ((long)k * cin + ci) * kk + tp;
        }
    }
};

// generic path: one packed element per thread
template <typename T>
__device__ __forceinline__ void pack_elems(const float* __restrict__ w, T* __restrict__ packed, const PackMap& m, int k_pad, int n_pad) {
    constexpr int EPS = TT<T>::EPS, KCH = TT<T>::KCH;
    const long total = (long)m.taps * k_pad * n_pad;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = i % EPS;
        long t = i / EPS;
        const int slotpos = t % 2; t /= 2;
        const int n = t % n_pad; t /= n_pad;
        const int tap = t % m.taps;
        const int chunk = t / m.taps;
        const int slot = slotpos ^ ((n >> 3) & 1);
        const int k = chunk * KCH + slot * EPS + e;
        const long s = m.src_index(k, tap, n);
        packed[i] = s >= 0 ? (T)w[s] : (T)0.f;
    }
}

// 3x3 weights, forward (mode 0) or input-gradient (mode 1) layout: a workgroup-iteration builds the 9 tap slices of one
// (K chunk, 128-wide N tile) in LDS from source runs that are contiguous in OIHW order (mode 0: 9*KCH floats per output
// channel, mode 1: 9*128 floats per output channel) and stores each slice as 4 KB of contiguous 16-byte pieces.
template <typename T>
__device__ __forceinline__ void pack_tiles33(const float* __restrict__ w, T* __restrict__ packed, const PackMap& m, int k_pad, int n_pad, char* lds) {
    constexpr int EPS = TT<T>::EPS, KCH = TT<T>::KCH;
    const int tid = threadIdx.x;
    const int n_tiles = n_pad / 128, tiles = (k_pad / KCH) * n_tiles;
    const int gk = m.mode == 0 ? m.ci_count : m.cout, gn = m.mode == 0 ? m.cout : m.ci_count;
    // item = (output-channel row nl of the tile, pair of adjacent k): its 2 x 9 source floats (mode 0: 18 consecutive floats -- two
    // input channels of one output channel; mode 1: two runs of 9) become nine 2-element LDS stores, one per tap slice.  Lanes are
    // (8 k-pairs) x (8 rows): a wave's stores to a slice are 256 contiguous bytes, its loads 8 segments of 288-576 bytes.  (The
    // per-element walk this replaces spent ~25 integer instructions per element on index arithmetic and stored 2 bytes at a time:
    // 228 us per c2 step for 60 M parameters in two layouts, 2.1 TB/s.)
    constexpr int KP = KCH / 2;
    typedef T pair_t __attribute__((ext_vector_type(2)));
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int c = tile / n_tiles, n0 = (tile % n_tiles) * 128;
        for (int item = tid; item < 128 * KP; item += 256) {
            const int kp = item % KP, nl = item / KP;
            const int kl = 2 * kp, k = c * KCH + kl, n = n0 + nl;
            float v0[9], v1[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) { v0[t] = 0.f; v1[t] = 0.f; }
            if (n < gn) {
                if (m.mode == 0) {       // o = n, ci = k, k + 1
                    const int co = m.n_perm ? m.n_perm[n] : n;
                    const float* src = w + ((long)co * m.cin + m.ci_begin + k) * 9;
                    if (k < gk) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) v0[t] = src[t];
                    }
                    if (k + 1 < gk) {
#pragma unroll
                        for (int t = 0; t < 9; ++t) v1[t] = src[9 + t];
                    }
                } else {                 // o = k, k + 1, ci = n
                    if (k < gk) {
                        const int co = m.n_perm ? m.n_perm[k] : k;
                        const float* src = w + ((long)co * m.cin + m.ci_begin + n) * 9;
#pragma unroll
                        for (int t = 0; t < 9; ++t) v0[t] = src[t];
                    }
                    if (k + 1 < gk) {
                        const int co = m.n_perm ? m.n_perm[k + 1] : k + 1;
                        const float* src = w + ((long)co * m.cin + m.ci_begin + n) * 9;
#pragma unroll
                        for (int t = 0; t < 9; ++t) v1[t] = src[t];
                    }
                }
            }
            char* dst = lds + nl * 32 + (((kl / EPS) ^ ((n >> 3) & 1)) << 4) + (kl % EPS) * (int)sizeof(T);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int tap = m.mode == 0 ? t : 8 - t;
                pair_t pv = {(T)v0[t], (T)v1[t]};
                *(pair_t*)(dst + tap * 4096) = pv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            *(u32x4*)((char*)packed + (((long)c * 9 + tap) * n_pad + n0) * 32 + tid * 16) = *(const u32x4*)(lds + tap * 4096 + tid * 16);
        __syncthreads();
    }
}

template <typename T>
__device__ __forceinline__ void pack_any(const float* __restrict__ w, T* __restrict__ packed, const PackMap& m, int k_pad, int n_pad, char* lds) {
    if (m.ks == 3 && m.mode <= 1 && !m.center) pack_tiles33<T>(w, packed, m, k_pad, n_pad, lds);
    else pack_elems<T>(w, packed, m, k_pad, n_pad);
}

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, T* __restrict__ packed, PackMap m, int k_pad, int n_pad, long total) {
    __shared__ __attribute__((aligned(16))) char lds[9 * 128 * 32];
    pack_any<T>(w, packed, m, k_pad, n_pad, lds);
}

// every weight of a model in ONE launch: blockIdx.y selects the item (a training step re-packs ~100 weights)
__global__ __launch_bounds__(256) void pack_batch_kernel(const pssr_pack_item* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) char lds[9 * 128 * 32];
    const pssr_pack_item it = items[blockIdx.y];
    const PackMap m{it.cout, it.cin, it.ks, it.ci_begin, it.ci_count, it.mode, it.mode >= 2 ? 1 : it.ks * it.ks, it.n_perm, it.center};
    if (it.dtype == PSSR_BF16) pack_any<bf16_t>(it.w, (bf16_t*)it.packed, m, it.k_pad, it.n_pad, lds);
    else if (it.dtype == PSSR_F16) pack_any<f16_t>(it.w, (f16_t*)it.packed, m, it.k_pad, it.n_pad, lds);
    else pack_any<float>(it.w, (float*)it.packed, m, it.k_pad, it.n_pad, lds);
}

// dW packed layout produced by the wgrad kernel: f32 [parts][rows][tap][k_pad].  blockIdx.y owns a strided subset of the
// parts; with one subset (gridDim.y == 1) the sum is stored directly, otherwise the subsets are combined with f32 atomics
// (the destination is zeroed by `unpack_zero_kernel` first unless the caller accumulates into it).
__global__ void unpack_kernel(const float* __restrict__ dwp, int parts, long part_stride, float* __restrict__ dw, PackMap m, int k_pad,
                              int accumulate, long total) {
    // 4 consecutive k per thread (k_pad is a multiple of 16): 16-byte loads, 4 parts in flight
    const int kq = k_pad >> 2;
    for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < (total >> 2); i4 += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i4 % kq) * 4;
        long t = i4 / kq;
        const int tap = t % m.taps;
        const int n = t / m.taps;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = dwp + i4 * 4;
        int q = blockIdx.y;
        for (; q + 3 * (int)gridDim.y < parts; q += 4 * gridDim.y) {
            const float4 a = *(const float4*)(src + (long)q * part_stride);
            const float4 b = *(const float4*)(src + (long)(q + gridDim.y) * part_stride);
            const float4 c = *(const float4*)(src + (long)(q + 2 * gridDim.y) * part_stride);
            const float4 d = *(const float4*)(src + (long)(q + 3 * gridDim.y) * part_stride);
            v.x += (a.x + b.x) + (c.x + d.x); v.y += (a.y + b.y) + (c.y + d.y);
            v.z += (a.z + b.z) + (c.z + d.z); v.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; q < parts; q += gridDim.y) {
            const float4 a = *(const float4*)(src + (long)q * part_stride);
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long s = m.src_index(k + e, tap, n);
            if (s < 0) continue;
            if (gridDim.y > 1) atomicAdd(dw + s, ve[e]);
            else dw[s] = accumulate ? dw[s] + ve[e] : ve[e];
        }
    }
}

// 3x3 forward-layout gradients (mode 0, 9 taps): the destination OIHW order has the 9 taps of one (co, ci) adjacent, the
// packed order has them k_pad apart.  A workgroup-iteration covers 1024 >> pl (row, k) pairs: its 256 threads are
// (256 >> pl) positions of 4 consecutive k  x  (1 << pl) part lanes; every thread sums its share of the parts with 16-byte
// loads, the 9216 partial values meet in LDS, are summed over the part lanes there, and leave as runs that are
// contiguous in the destination (when the weight has no padding / permutation: `contig`) -- no atomics, no zero pass.
template <int TAPS>
__global__ __launch_bounds__(256) void unpack9_kernel(const float* __restrict__ dwp, int parts, long part_stride, float* __restrict__ dw, PackMap m,
                                                      int k_pad, int accumulate, long total_nk, int contig, int pl) {
    __shared__ float buf[1024 * TAPS];
    const int tid = threadIdx.x;
    const int PL = 1 << pl, ppb = 1024 >> pl;            // part lanes, pairs per workgroup-iteration
    const int pos = tid & ((256 >> pl) - 1), plane = tid >> (8 - pl);
    // workgroups are dealt round-robin to the 8 XCDs, and with many part lanes a workgroup reads only 16 * (256 >> pl) bytes of every
    // 128-byte line of the partial slabs: neighbours in `base` share lines, so they are numbered onto ONE XCD (its L2 then serves the
    // second reader; with the plain numbering every line came from HBM twice: 65 MB fetched per 37.7 MB of slabs, profiles/traffic.json)
    const unsigned G = gridDim.x;
    const unsigned bid = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    for (long base = (long)bid * ppb; base < total_nk; base += (long)gridDim.x * ppb) {
        const long i = base + pos * 4;
        float4 v[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < total_nk) {
            const long n = i / k_pad;
            const int k = (int)(i - n * k_pad);
            const float* src = dwp + n * TAPS * k_pad + k;
            for (int q = plane; q < parts; q += PL) {
                const float* sq = src + (long)q * part_stride;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    const float4 a = *(const float4*)(sq + (long)t * k_pad);
                    v[t].x += a.x; v[t].y += a.y; v[t].z += a.z; v[t].w += a.w;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            buf[(tid * 4 + 0) * TAPS + t] = v[t].x; buf[(tid * 4 + 1) * TAPS + t] = v[t].y;
            buf[(tid * 4 + 2) * TAPS + t] = v[t].z; buf[(tid * 4 + 3) * TAPS + t] = v[t].w;
        }
        __syncthreads();
        const long left = total_nk - base;
        const int np = (int)(left < ppb ? left : ppb) * TAPS;
        for (int pp = tid; pp < np; pp += 256) {
            float sum = 0.f;
            for (int l = 0; l < PL; ++l) sum += buf[l * (ppb * TAPS) + pp];
            long sidx;
            if (contig) sidx = base * TAPS + pp;
            else {
                const int j = pp / TAPS, t = pp - j * TAPS;
                const long ii = base + j;
                const long n = ii / k_pad;
                sidx = m.src_index((int)(ii - n * k_pad), t, (int)n);
                if (sidx < 0) continue;
            }
            dw[sidx] = accumulate ? dw[sidx] + sum : sum;
        }
        __syncthreads();
    }
}

__global__ void unpack_zero_kernel(float* __restrict__ dw, PackMap m, int k_pad, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i % k_pad;
        long t = i / k_pad;
        const long s = m.src_index(k, (int)(t % m.taps), (int)(t / m.taps));
        if (s >= 0) dw[s] = 0.f;
    }
}

}  // namespace

extern "C" int64_t pssr_packed_weight_bytes(int taps, int k_pad, int n_pad, int dtype) {
    return (int64_t)taps * k_pad * n_pad * (dtype == PSSR_F32 ? 4 : 2);
}

extern "C" int pssr_pack_conv_weight(const float* w, void* packed, int cout, int cin, int ks, int ci_begin, int ci_count,
                                     int mode, const int32_t* n_perm, int k_pad, int n_pad, int dtype, pssr_stream_t stream) {
    PSSR_CHECK(w && packed, PSSR_ERR_ARG, "pack: null pointer");
    PSSR_CHECK(ks >= 1 && ks <= 3, PSSR_ERR_ARG, "pack: ks=%d", ks);
    PSSR_CHECK(mode >= 0 && mode <= 5, PSSR_ERR_ARG, "pack: mode=%d", mode);
    PSSR_CHECK(ks != 2 || mode >= 2, PSSR_ERR_ARG, "pack: a 2x2 (stride-2) kernel is only consumed in a flat-K / space-to-depth mode");
    PSSR_CHECK(ci_begin >= 0 && ci_count > 0 && ci_begin + ci_count <= cin, PSSR_ERR_ARG, "pack: channel range");
    PSSR_CHECK(k_pad % 16 == 0 && n_pad % 128 == 0, PSSR_ERR_ARG, "pack: k_pad=%d n_pad=%d", k_pad, n_pad);
    const int s2dk = ks * ks * ((cin + 15) / 16 * 16);
    const int gk = mode == 0 ? ci_count : (mode == 1 || mode == 3 || mode == 5) ? cout : mode == 4 ? s2dk : ci_count * ks * ks;
    const int gn = mode == 1 ? ci_count : mode == 3 ? ci_count * ks * ks : mode == 5 ? s2dk : cout;
    PSSR_CHECK(k_pad >= gk && n_pad >= gn, PSSR_ERR_ARG, "pack: padding smaller than GEMM dims (%d<%d or %d<%d)", k_pad, gk, n_pad, gn);
    PackMap m{cout, cin, ks, ci_begin, ci_count, mode, mode >= 2 ? 1 : ks * ks, n_perm, 0};
    const long total = (long)m.taps * k_pad * n_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (dtype == PSSR_BF16)
        hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)packed, m, k_pad, n_pad, total);
    else if (dtype == PSSR_F16)
        hipLaunchKernelGGL(pack_kernel<f16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (f16_t*)packed, m, k_pad, n_pad, total);
    else if (dtype == PSSR_F32)
        hipLaunchKernelGGL(pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (float*)packed, m, k_pad, n_pad, total);
    else
        PSSR_CHECK(false, PSSR_ERR_ARG, "pack: dtype=%d", dtype);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_pack_conv_weight_batch(const pssr_pack_item* items_dev, int n_items, pssr_stream_t stream) {
    PSSR_CHECK(items_dev && n_items > 0 && n_items <= 65535, PSSR_ERR_ARG, "pack_batch: bad args");
    hipLaunchKernelGGL(pack_batch_kernel, dim3(384, n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_unpack_conv_wgrad_parts(const float* dwp, int parts, int rows, float* dw, int cout, int cin, int ks, int ci_begin,
                                            int ci_count, int mode, const int32_t* n_perm, int k_pad, int accumulate, pssr_stream_t stream) {
    PSSR_CHECK(dwp && dw && parts > 0 && rows >= cout, PSSR_ERR_ARG, "unpack: bad args");
    PSSR_CHECK(mode == 0 || mode == 2 || mode == 4, PSSR_ERR_ARG, "unpack: mode=%d", mode);
    PackMap m{cout, cin, ks, ci_begin, ci_count, mode, mode >= 2 ? 1 : ks * ks, n_perm, 0};
    const long total = (long)cout * m.taps * k_pad;
    const int blocks = (int)((total / 4 + 255) / 256 < 4096 ? (total / 4 + 255) / 256 : 4096);
    // enough workgroups to pull the partial slabs at HBM rate: split the parts over blockIdx.y when the slab is small
    if (mode == 0 && (m.taps == 9 || m.taps == 1) && !m.center) {
        const long total_nk = (long)cout * k_pad;
        int pl = 0;      // part lanes: as many as keep the grid within ~2048 workgroups (and no more than there are parts)
        while (pl < 6 && (2 << pl) <= parts && (total_nk << (pl + 1)) / 1024 <= 2048) ++pl;
        const long b9 = ((total_nk << pl) + 1023) / 1024;
        const int contig = n_perm == nullptr && ci_begin == 0 && ci_count == cin && k_pad == cin;
        const dim3 grid((unsigned)(b9 < 4096 ? b9 : 4096));
        if (m.taps == 9)
            hipLaunchKernelGGL(unpack9_kernel<9>, grid, dim3(256), 0, (hipStream_t)stream, dwp, parts, (long)rows * m.taps * k_pad, dw, m, k_pad,
                               accumulate, total_nk, contig, pl);
        else
            hipLaunchKernelGGL(unpack9_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, dwp, parts, (long)rows * m.taps * k_pad, dw, m, k_pad,
                               accumulate, total_nk, contig, pl);
        PSSR_LAUNCH_CHECK();
        return PSSR_OK;
    }
    // one subset of the parts per element: the sum order is fixed (splitting them over blockIdx.y and combining with f32 atomics was
    // up to 2x faster on the few small weights that come here, and made the training step irreproducible)
    int psplit = 1;
    if (psplit > 1 && !accumulate)
        hipLaunchKernelGGL(unpack_zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dw, m, k_pad, total);
    hipLaunchKernelGGL(unpack_kernel, dim3(blocks, psplit), dim3(256), 0, (hipStream_t)stream, dwp, parts, (long)rows * m.taps * k_pad, dw, m,
                       k_pad, accumulate, total);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_unpack_conv_wgrad(const float* dwp, float* dw, int cout, int cin, int ks, int ci_begin, int ci_count,
                                      int mode, const int32_t* n_perm, int k_pad, int accumulate, pssr_stream_t stream) {
    return pssr_unpack_conv_wgrad_parts(dwp, 1, cout, dw, cout, cin, ks, ci_begin, ci_count, mode, n_perm, k_pad, accumulate, stream);
}
