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

template <typename T>
__global__ void pack_kernel(const float* __restrict__ w, T* __restrict__ packed, PackMap m, int k_pad, int n_pad, long total) {
    constexpr int EPS = TT<T>::EPS, KCH = TT<T>::KCH;
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

// every weight of a model in ONE launch: blockIdx.y selects the item (a training step re-packs ~100 weights)
template <typename T>
__device__ __forceinline__ void pack_item(const pssr_pack_item& it, const PackMap& m) {
    constexpr int EPS = TT<T>::EPS, KCH = TT<T>::KCH;
    const long total = (long)m.taps * it.k_pad * it.n_pad;
    T* packed = (T*)it.packed;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int e = i % EPS;
        long t = i / EPS;
        const int slotpos = t % 2; t /= 2;
        const int n = t % it.n_pad; t /= it.n_pad;
        const int tap = t % m.taps;
        const int chunk = t / m.taps;
        const int slot = slotpos ^ ((n >> 3) & 1);
        const int k = chunk * KCH + slot * EPS + e;
        const long s = m.src_index(k, tap, n);
        packed[i] = s >= 0 ? (T)it.w[s] : (T)0.f;
    }
}

__global__ void pack_batch_kernel(const pssr_pack_item* __restrict__ items) {
    const pssr_pack_item it = items[blockIdx.y];
    const PackMap m{it.cout, it.cin, it.ks, it.ci_begin, it.ci_count, it.mode, it.mode >= 2 ? 1 : it.ks * it.ks, it.n_perm, it.center};
    if (it.dtype == PSSR_BF16) pack_item<bf16_t>(it, m);
    else if (it.dtype == PSSR_F16) pack_item<f16_t>(it, m);
    else pack_item<float>(it, m);
}

// dW packed layout produced by the wgrad kernel: f32 [parts][rows][tap][k_pad].  blockIdx.y owns a strided subset of the
// parts; with one subset (gridDim.y == 1) the sum is stored directly, otherwise the subsets are combined with f32 atomics
// (the destination is zeroed by `unpack_zero_kernel` first unless the caller accumulates into it).
__global__ void unpack_kernel(const float* __restrict__ dwp, int parts, long part_stride, float* __restrict__ dw, PackMap m, int k_pad,
                              int accumulate, long total) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i % k_pad;
        long t = i / k_pad;
        const int tap = t % m.taps;
        const int n = t / m.taps;
        const long s = m.src_index(k, tap, n);
        if (s < 0) continue;
        float v = 0.f;
        for (int q = blockIdx.y; q < parts; q += gridDim.y) v += dwp[q * part_stride + i];
        if (gridDim.y > 1) atomicAdd(dw + s, v);
        else dw[s] = accumulate ? dw[s] + v : v;
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
    hipLaunchKernelGGL(pack_batch_kernel, dim3(96, n_items), dim3(256), 0, (hipStream_t)stream, items_dev);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_unpack_conv_wgrad_parts(const float* dwp, int parts, int rows, float* dw, int cout, int cin, int ks, int ci_begin,
                                            int ci_count, int mode, const int32_t* n_perm, int k_pad, int accumulate, pssr_stream_t stream) {
    PSSR_CHECK(dwp && dw && parts > 0 && rows >= cout, PSSR_ERR_ARG, "unpack: bad args");
    PSSR_CHECK(mode == 0 || mode == 2 || mode == 4, PSSR_ERR_ARG, "unpack: mode=%d", mode);
    PackMap m{cout, cin, ks, ci_begin, ci_count, mode, mode >= 2 ? 1 : ks * ks, n_perm, 0};
    const long total = (long)cout * m.taps * k_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    // enough workgroups to pull the partial slabs at HBM rate: split the parts over blockIdx.y when the slab is small
    int psplit = 2048 / blocks;
    if (psplit > parts / 4) psplit = parts / 4;
    if (psplit < 1) psplit = 1;
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
