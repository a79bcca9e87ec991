// Fused AdamW over one flat fp32 buffer (all 81.65 M parameters of SimCLRSkinV32 in one launch),
// with the GradScaler-style unscale and skip-on-overflow folded in.  HBM-bound: reads p,g,m,v and
// writes p,m,v once, float4 per lane.
//
// Reference call sites replaced: torch.optim.AdamW(lr, weight_decay=args.wd, eps=1e-5)
// (tools/backbone_train.py:525-527) stepped through GradScaler (:125-127).
#include "common.h"

namespace {

// dyn (nullable): {loss scale, number of optimizer steps taken so far} on the device -- the fp16 mode's GradScaler state.
// With it the gradient is also divided by the loss scale and the bias corrections use steps_taken + 1: a step skipped
// for an overflow does not advance Adam's step count, as with torch.optim.AdamW under GradScaler.step().
struct AdamDyn {
    const float* loss_scale;
    const int32_t* steps_taken;
};
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float inv_bc1,
                                                    float inv_sqrt_bc2, float gscale, const int32_t* found_inf,
                                                    AdamDyn dyn) {
    if (found_inf && *found_inf) return;
    if (dyn.loss_scale) {
        gscale /= dyn.loss_scale[0];
        const float step = (float)(dyn.steps_taken[0] + 1);
        inv_bc1 = 1.f / (1.f - powf(b1, step));
        inv_sqrt_bc2 = 1.f / sqrtf(1.f - powf(b2, step));
    }
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= gscale;
        pp *= (1.f - lr * wd);
        mm = b1 * mm + (1.f - b1) * gg;
        vv = b2 * vv + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pp -= lr * inv_bc1 * (mm / denom);
    };
    for (int64_t i = t0; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        upd(pp.x, gg.x, mm.x, vv.x);
        upd(pp.y, gg.y, mm.y, vv.y);
        upd(pp.z, gg.z, mm.z, vv.z);
        upd(pp.w, gg.w, mm.w, vv.w);
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (int64_t i = (n4 << 2) + t0; i < n; i += stride) upd(p[i], g[i], m[i], v[i]);
}

__global__ __launch_bounds__(256) void check_finite_kernel(const float* __restrict__ g, int64_t n, int32_t* found) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        bad |= !isfinite(g[i]);
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(found, 1);
}

}  // namespace

extern "C" int sm3_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int step, float grad_scale, const int32_t* found_inf,
                         void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return SM3_EINVAL;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return SM3_EALIGN;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                       beta1, beta2, eps, weight_decay, (float)(1.0 / bc1), (float)(1.0 / sqrt(bc2)), grad_scale,
                       found_inf, AdamDyn{nullptr, nullptr});
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_adamw_dynamic(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, float grad_scale, const float* loss_scale,
                                 const int32_t* steps_taken, const int32_t* found_inf, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || !loss_scale || !steps_taken || !found_inf) return SM3_EINVAL;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return SM3_EALIGN;
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                       beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale, found_inf, AdamDyn{loss_scale, steps_taken});
    SM3_CHECK_LAUNCH();
    return 0;
}

namespace {
// torch.cuda.amp.GradScaler.update() (torch/amp/grad_scaler.py _amp_update_scale_): on overflow the scale backs off and
// the growth tracker restarts; otherwise the tracker advances and every `interval` clean steps the scale grows.  Also
// advances the count of optimizer steps actually taken and clears found_inf for the next step.
__global__ void loss_scale_update_kernel(float* scale, int32_t* found_inf, int32_t* tracker, int32_t* steps_taken,
                                         float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (*found_inf) {
        *scale *= backoff;
        *tracker = 0;
    } else {
        *steps_taken += 1;
        const int t = *tracker + 1;
        if (t >= interval) {
            *scale *= growth;
            *tracker = 0;
        } else {
            *tracker = t;
        }
    }
    *found_inf = 0;
}
}  // namespace

extern "C" int sm3_loss_scale_update(float* loss_scale, int32_t* found_inf, int32_t* growth_tracker, int32_t* steps_taken,
                                     float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    if (!loss_scale || !found_inf || !growth_tracker || !steps_taken || growth_interval < 1) return SM3_EINVAL;
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_scale, found_inf,
                       growth_tracker, steps_taken, growth_factor, backoff_factor, growth_interval);
    SM3_CHECK_LAUNCH();
    return 0;
}

namespace {
__global__ __launch_bounds__(256) void ema_update_kernel(float* __restrict__ t, const float* __restrict__ p, int64_t n, float m) {
    const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < n4; i += stride) {
        float4 a = reinterpret_cast<float4*>(t)[i];
        const float4 b = reinterpret_cast<const float4*>(p)[i];
        a.x = m * a.x + (1.f - m) * b.x; a.y = m * a.y + (1.f - m) * b.y;
        a.z = m * a.z + (1.f - m) * b.z; a.w = m * a.w + (1.f - m) * b.w;
        reinterpret_cast<float4*>(t)[i] = a;
    }
    for (int64_t i = (n4 << 2) + t0; i < n; i += stride) t[i] = m * t[i] + (1.f - m) * p[i];
}
}  // namespace

extern "C" int sm3_ema_update(float* target, const float* online, int64_t n, float momentum, void* stream) {
    if (!target || !online || n <= 0 || momentum < 0.f || momentum > 1.f) return SM3_EINVAL;
    if ((((uintptr_t)target | (uintptr_t)online) & 15) != 0) return SM3_EALIGN;
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, target, online, n, momentum);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_check_finite(const float* g, int64_t n, int32_t* found_inf, void* stream) {
    if (!g || !found_inf || n <= 0) return SM3_EINVAL;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(check_finite_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n,
                       found_inf);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_abi_version(void) { return 8; }
