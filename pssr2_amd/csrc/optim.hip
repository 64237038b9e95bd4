// Fused AdamW over a flat f32 parameter/gradient/state buffer (torch.optim.AdamW semantics,
// decoupled weight decay), one pass over 4 arrays: 16 B read + 12 B written per parameter.
#include "common.h"

namespace {
__global__ void advance_kernel(long long* step) { step[0] += 1; }

// Dynamic loss scaling on the device (fp16 storage; no upstream counterpart: the reference trains in fp32).  amp = int32[4]:
// [0] loss scale (f32 bits), [1] good steps in a row, [2] steps skipped so far, [3] a gradient of the current step is not finite.
// A step is: amp_check (sets [3]) -> amp_advance (the AdamW step count moves only when [3] is clear) -> adamw_kernel (returns at once
// when [3] is set; divides the gradients by the scale) -> amp_update (the torch.amp.GradScaler policy; clears [3]).  Nothing goes
// through the host, so the whole fp16 step replays as one graph.
__global__ __launch_bounds__(256) void amp_check_kernel(const float* __restrict__ g, long n, int* __restrict__ amp) {
    const long n4 = n / 4;
    bool bad = false;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const uint4 v = ((const uint4*)g)[i];
        bad |= (v.x & 0x7f800000u) == 0x7f800000u || (v.y & 0x7f800000u) == 0x7f800000u || (v.z & 0x7f800000u) == 0x7f800000u ||
               (v.w & 0x7f800000u) == 0x7f800000u;
    }
    for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        bad |= (__float_as_uint(g[i]) & 0x7f800000u) == 0x7f800000u;
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(amp + 3, 1);
}
__global__ void amp_advance_kernel(long long* step, const int* amp) {
    if (!amp[3]) step[0] += 1;
}
__global__ void amp_update_kernel(int* amp, float growth, float backoff, int interval) {
    float s = __int_as_float(amp[0]);
    if (amp[3]) {
        s *= backoff; amp[1] = 0; amp[2] += 1;
    } else {
        amp[1] += 1;
        if (amp[1] % interval == 0) s *= growth;
    }
    amp[0] = __float_as_int(s);
    amp[3] = 0;
}

// `dev` (optional): device-resident [step (int64), lr (f32 bits in the next 4 bytes)] so that a captured
// hipGraph replays with the right bias correction and learning rate
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                             float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                             const long long* dev, const int* amp) {
    if (amp) {
        if (amp[3]) return;                         // a gradient of this step is not finite: the step is skipped
        gscale /= __int_as_float(amp[0]);
    }
    if (dev) {
        const float st = (float)dev[0];
        lr = __uint_as_float((unsigned)((const unsigned long long*)dev)[1]);
        bc1 = 1.f - powf(b1, st);
        bc2_sqrt = sqrtf(1.f - powf(b2, st));
    }
    const long n4 = n / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pv = ((float4*)p)[i], gv = ((const float4*)g)[i], mv = ((float4*)m)[i], vv = ((float4*)v)[i];
        float* pp = (float*)&pv; float* gg = (float*)&gv; float* mm = (float*)&mv; float* vw = (float*)&vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * gscale;
            pp[e] *= (1.f - lr * wd);
            mm[e] = b1 * mm[e] + (1.f - b1) * gr;
            vw[e] = b2 * vw[e] + (1.f - b2) * gr * gr;
            const float denom = sqrtf(vw[e]) / bc2_sqrt + eps;
            pp[e] -= (lr / bc1) * (mm[e] / denom);
        }
        ((float4*)p)[i] = pv; ((float4*)m)[i] = mv; ((float4*)v)[i] = vv;
    }
    // tail
    for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gr = g[i] * gscale;
        float pw = p[i] * (1.f - lr * wd);
        const float mn = b1 * m[i] + (1.f - b1) * gr, vn = b2 * v[i] + (1.f - b2) * gr * gr;
        pw -= (lr / bc1) * (mn / (sqrtf(vn) / bc2_sqrt + eps));
        p[i] = pw; m[i] = mn; v[i] = vn;
    }
}
}  // namespace

extern "C" int pssr_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, float grad_scale, pssr_stream_t s) {
    PSSR_CHECK(p && g && m && v && n > 0 && step > 0, PSSR_ERR_ARG, "adamw_step: bad args");
    PSSR_CHECK(((uintptr_t)p % 16 == 0 && (uintptr_t)g % 16 == 0 && (uintptr_t)m % 16 == 0 && (uintptr_t)v % 16 == 0) || n < 4, PSSR_ERR_ARG,
               "adamw_step: buffers must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, lr, beta1, beta2, eps,
                       weight_decay, bc1, bc2s, grad_scale, (const long long*)nullptr, (const int*)nullptr);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, int64_t* state, float beta1, float beta2,
                                   float eps, float weight_decay, float grad_scale, pssr_stream_t s) {
    PSSR_CHECK(p && g && m && v && state && n > 0, PSSR_ERR_ARG, "adamw_step_dev: bad args");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (long long*)state);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, 0.f, beta1, beta2, eps,
                       weight_decay, 1.f, 1.f, grad_scale, (const long long*)state, (const int*)nullptr);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_amp_check(const float* g, int64_t n, int32_t* amp, pssr_stream_t s) {
    PSSR_CHECK(g && amp && n > 0 && (uintptr_t)g % 16 == 0, PSSR_ERR_ARG, "amp_check: bad args");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(amp_check_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, g, (long)n, (int*)amp);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}

extern "C" int pssr_adamw_step_amp(float* p, const float* g, float* m, float* v, int64_t n, int64_t* state, float beta1, float beta2,
                                   float eps, float weight_decay, float grad_scale, int32_t* amp, float growth, float backoff,
                                   int interval, pssr_stream_t s) {
    PSSR_CHECK(p && g && m && v && state && amp && n > 0 && interval > 0 && growth > 0.f && backoff > 0.f, PSSR_ERR_ARG, "adamw_step_amp: bad args");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(amp_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (long long*)state, (const int*)amp);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, 0.f, beta1, beta2, eps,
                       weight_decay, 1.f, 1.f, grad_scale, (const long long*)state, (const int*)amp);
    hipLaunchKernelGGL(amp_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, (int*)amp, growth, backoff, interval);
    PSSR_LAUNCH_CHECK();
    return PSSR_OK;
}
