// Multi-label heads of the inference model (reference inference.py:53-96, tools/mlc_eval.py): everything between
// the 8 label-token projections and the 8 prototype logits that is not a GEMM -- the GEMMs themselves (label
// projectors, attention in/out projections, feed-forward) go through sm3_conv_gather_gemm as bias-free Linear
// followed by sm3_bn_act(scale = 1, shift = bias).  Eval mode: dropout is the identity.
//
//   sm3_token_attention  softmax(Q K^T / sqrt(hd)) V over the S <= 8 label tokens of every sample, per head
//   sm3_add_layernorm    LayerNorm(a + b) * gamma + beta   (post-norm residual joins of nn.TransformerEncoderLayer)
//   sm3_token_heads      out[b, t] = <x[b, token_of[t], :] (optionally L2-normalised), W[t, :]> + bias[t]
//
// All three are tiny (a few MFLOP per sample) and latency-bound: one workgroup per sample / 4 rows, f32 math.
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ float ld1(const T* p);
template <>
__device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(p->v); }
template <>
__device__ __forceinline__ float ld1<f16_t>(const f16_t* p) { return f16_to_f32(p->v); }
template <typename T>
__device__ __forceinline__ void st1(T* p, float v);
template <>
__device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
template <>
__device__ __forceinline__ void st1<f16_t>(f16_t* p, float v) { p->v = f32_to_f16(v); }

constexpr int kMaxS = 8;

// qkv: [B*S, 3*D] rows (b*S + s), columns [q | k | v]; out: [B*S, D].  grid = B, block = 256.
template <typename T>
__global__ __launch_bounds__(256) void token_attention_kernel(const T* __restrict__ qkv, T* __restrict__ out, int S,
                                                              int D, int nhead) {
    __shared__ float sc[kMaxS * kMaxS * 8];  // [nhead <= 8][S][S]
    const int b = blockIdx.x, tid = threadIdx.x;
    const int hd = D / nhead;
    const T* base = qkv + (int64_t)b * S * 3 * D;
    const float inv_sqrt = rsqrtf((float)hd);
    // scores: 4 lanes per (head, i, j) triple
    const int ntrip = nhead * S * S;
    for (int t0 = 0; t0 < ntrip; t0 += 64) {
        const int trip = t0 + (tid >> 2), part = tid & 3;
        float a = 0.f;
        if (trip < ntrip) {
            const int h = trip / (S * S), ij = trip - h * S * S, i = ij / S, j = ij - i * S;
            const T* q = base + (int64_t)i * 3 * D + h * hd;
            const T* k = base + (int64_t)j * 3 * D + D + h * hd;
            for (int d = part; d < hd; d += 4) a += ld1<T>(q + d) * ld1<T>(k + d);
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (trip < ntrip && part == 0) sc[trip] = a * inv_sqrt;
    }
    __syncthreads();
    if (tid < nhead * S) {  // softmax of one row
        float* row = sc + tid * S;
        float m = row[0];
        for (int j = 1; j < S; ++j) m = fmaxf(m, row[j]);
        float sum = 0.f;
        for (int j = 0; j < S; ++j) {
            row[j] = __expf(row[j] - m);
            sum += row[j];
        }
        const float inv = 1.f / sum;
        for (int j = 0; j < S; ++j) row[j] *= inv;
    }
    __syncthreads();
    for (int o = tid; o < S * D; o += 256) {
        const int i = o / D, d = o - i * D, h = d / hd;
        const float* p = sc + (h * S + i) * S;
        float a = 0.f;
        for (int j = 0; j < S; ++j) a += p[j] * ld1<T>(base + (int64_t)j * 3 * D + 2 * D + d);
        st1<T>(out + ((int64_t)b * S + i) * D + d, a);
    }
}

// one wave per row; D <= 64 * 16
template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            T* __restrict__ out, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[16];
    float s = 0.f;
    int n = 0;
    for (int d = lane; d < D; d += 64, ++n) {
        v[n] = ld1<T>(a + row * D + d) + (b ? ld1<T>(b + row * D + d) : 0.f);
        s += v[n];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int i = 0; i < n; ++i) q += (v[i] - mean) * (v[i] - mean);
    const float inv = rsqrtf(wave_sum(q) / (float)D + eps);
    n = 0;
    for (int d = lane; d < D; d += 64, ++n) st1<T>(out + row * D + d, (v[n] - mean) * inv * gamma[d] + beta[d]);
}

// grid = B, block = 256 = 32 outputs x 8 lanes per pass
template <typename T>
__global__ __launch_bounds__(256) void token_heads_kernel(const T* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          const int* __restrict__ token_of, int l2_norm,
                                                          float* __restrict__ out, int S, int D, int Tn) {
    __shared__ float inv_norm[kMaxS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const T* xb = x + (int64_t)b * S * D;
    if (l2_norm) {  // F.normalize(p=2, eps=1e-12) of every token
        const int s = tid >> 5, part = tid & 31;
        float q = 0.f;
        if (s < S)
            for (int d = part; d < D; d += 32) {
                const float v = ld1<T>(xb + s * D + d);
                q += v * v;
            }
        for (int o = 1; o < 32; o <<= 1) q += __shfl_xor(q, o, 64);
        if (s < S && part == 0) inv_norm[s] = 1.f / fmaxf(sqrtf(q), 1e-12f);
        __syncthreads();
    }
    for (int t0 = 0; t0 < Tn; t0 += 32) {
        const int t = t0 + (tid >> 3), part = tid & 7;
        float a = 0.f;
        int tok = 0;
        if (t < Tn) {
            tok = token_of[t];
            const T* xr = xb + tok * D;
            const float* w = W + (int64_t)t * D;
            for (int d = part; d < D; d += 8) a += ld1<T>(xr + d) * w[d];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        if (t < Tn && part == 0) out[(int64_t)b * Tn + t] = a * (l2_norm ? inv_norm[tok] : 1.f) + bias[t];
    }
}

}  // namespace

extern "C" int sm3_token_attention(int dtype, const void* qkv, void* out, int B, int S, int D, int nhead, void* stream) {
    if (!qkv || !out || B <= 0 || S <= 0 || S > kMaxS || D <= 0 || nhead <= 0 || nhead > 8 || D % nhead) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    hipStream_t st = (hipStream_t)stream;
#define SM3_TA(T) hipLaunchKernelGGL(token_attention_kernel<T>, dim3(B), dim3(256), 0, st, (const T*)qkv, (T*)out, S, D, nhead)
    SM3_DISPATCH_DTYPE(dtype, SM3_TA);
#undef SM3_TA
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_add_layernorm(int dtype, const void* a, const void* b, const float* gamma, const float* beta,
                                 float eps, void* out, int64_t rows, int D, void* stream) {
    if (!a || !gamma || !beta || !out || rows <= 0 || D <= 0 || D > 1024) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    hipStream_t st = (hipStream_t)stream;
    const unsigned g = (unsigned)((rows + 3) / 4);
#define SM3_LN(T) \
    hipLaunchKernelGGL(add_layernorm_kernel<T>, dim3(g), dim3(256), 0, st, (const T*)a, (const T*)b, gamma, beta, eps, (T*)out, rows, D)
    SM3_DISPATCH_DTYPE(dtype, SM3_LN);
#undef SM3_LN
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_token_heads(int dtype, const void* x, const float* W, const float* bias, const int* token_of,
                               int l2_norm, float* out, int B, int S, int D, int T, void* stream) {
    if (!x || !W || !bias || !token_of || !out || B <= 0 || S <= 0 || S > kMaxS || D <= 0 || T <= 0) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    hipStream_t st = (hipStream_t)stream;
#define SM3_TH(ET) \
    hipLaunchKernelGGL(token_heads_kernel<ET>, dim3(B), dim3(256), 0, st, (const ET*)x, W, bias, token_of, l2_norm, out, S, D, T)
    SM3_DISPATCH_DTYPE(dtype, SM3_TH);
#undef SM3_TH
    SM3_CHECK_LAUNCH();
    return 0;
}
