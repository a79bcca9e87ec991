// BatchNorm backward BY LINEARITY for an expanding 1x1 convolution followed by train-mode BatchNorm
// (Bottleneck conv3 -> bn3, reference src/models/resnet.py:162-163 and their autograd backward):
//
//   forward   x = y W^T          y: [M, p] (conv input),  W: [C, p],  C = 4p;   z = gamma * (x - mu) * invstd + beta
//   backward  dx = a (dz - m1) - b (x - mu),   a = gamma invstd,  b = a invstd m2,  m1 = mean(dz),  m2 = mean(dz xhat)
//
// Every consumer of dx is linear in it, and x is linear in y, so neither dx nor x has to be read (or written):
//   P  = dz^T y  [C, p]  (the weight-gradient GEMM on dz itself),  G = y^T y [p, p],  s = sum_m y [p]
//   sum_m dz x        = rowdot(W, P)                               -> m2 without a pass over x
//   dy = dx W         = dz (diag(a) W) - y H + const,   H = W^T diag(b) W [p, p],  const = (b mu - a m1) W
//   dW = dx^T y       = diag(a) (P - m1 s^T) - diag(b) (W G - mu s^T)
// The data gradient is then one gather-GEMM over two K segments [dz | y] (sm3_conv_dgrad_seg_bnfuse), the weight
// gradient the same kernel as ever on dz (sm3_conv_wgrad_cat; the [p, p] Gram block is one more launch of it on y alone,
// made in the forward pass), and the BatchNorm-backward apply
// pass over the widest tensors of the network (read dz, read x, write dx: 12 of the 80 activation-sized transfers a
// Bottleneck costs per step) disappears together with every backward read of x.  Per view: all per-channel vectors are
// [views][C].  The kernels here are the small pieces between those GEMMs; 16-bit activation types only (the exact-f32
// parity mode keeps the two-pass BatchNorm backward).
#include "common.h"

namespace {

// ---- sm3_linbn_stats ------------------------------------------------------------------------------------------------
// blocks [0, views * C / 4): one wave per output channel: S1 = sum over the `groups` rows sm3_bn_stats_reduce's stage A left
// in ws (that kernel's stage B, folded in here), lsums[v][co] = S1, lsums[v][C + co] = invstd * (rowdot(W[co], P_v[co]) - mu S1)
// blocks beyond: s[v][ci] = sum over the colsum partial rows
template <typename T>
__global__ __launch_bounds__(256) void linbn_stats_kernel(const float* __restrict__ P, const T* __restrict__ w,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const double* __restrict__ ws, int groups,
                                                          double* __restrict__ lsums, const float* __restrict__ colsum,
                                                          int crow, float* __restrict__ s_out, int C, int p, int views,
                                                          int row_blocks) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < row_blocks) {
        const int row = blockIdx.x * 4 + wave;  // (view, co)
        if (row >= views * C) return;
        const int v = row / C, co = row - v * C;
        const float* pr = P + ((long)v * C + co) * p;
        const T* wr = w + (long)co * p;
        double acc = 0.0, s1 = 0.0;
        for (int k = lane; k < p; k += 64) acc += (double)pr[k] * (double)ElemTraits<T>::load(wr + k);
        for (int g = lane; g < groups; g += 64) s1 += ws[((long)v * groups + g) * 2 * C + co];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc += __shfl_xor(acc, o, 64);
            s1 += __shfl_xor(s1, o, 64);
        }
        if (lane == 0) {
            double* ls = lsums + (long)v * 2 * C;
            ls[co] = s1;
            ls[C + co] = (double)invstd[(long)v * C + co] * (acc - (double)mean[(long)v * C + co] * s1);
        }
        return;
    }
    // column sums: a block takes 32 columns of one view, 8 row lanes x 4 independent chains each (the partial rows are a few
    // hundred: one thread per column would walk them as one chain of dependent L2 loads)
    __shared__ double red[8][32];
    const int cb = blockIdx.x - row_blocks, pb = (p + 31) / 32;
    const int v = cb / pb, ci = (cb - v * pb) * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (ci < p) {
        const float* cp = colsum + (long)v * crow * p + ci;
        int r = rl;
        for (; r + 24 < crow; r += 32) {
            a0 += (double)cp[(long)r * p];
            a1 += (double)cp[(long)(r + 8) * p];
            a2 += (double)cp[(long)(r + 16) * p];
            a3 += (double)cp[(long)(r + 24) * p];
        }
        for (; r < crow; r += 8) a0 += (double)cp[(long)r * p];
    }
    red[rl][threadIdx.x & 31] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && ci < p) {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) t += red[j][threadIdx.x];
        s_out[(long)v * p + ci] = (float)t;
    }
}

// ---- sm3_linbn_coeffs -----------------------------------------------------------------------------------------------
// grid (p, views): block (ci, v) walks row ci of the data-gradient bank wd [p][C]:
//   wa[v][ci][co] = T(a wd),  wbn[v][ci][co] = T(-b wd),  col_const[v][ci] = sum_co -(mu f(wbn) + m1 f(wa))
// (the constant uses the ROUNDED products, so that what the GEMM adds up is centred exactly).  Block ci = 0 also leaves
// coef[v][{a, b, m1, mu}][C] for the weight-gradient finish and adds the BatchNorm's own parameter gradients.
template <typename T>
__global__ __launch_bounds__(256) void linbn_coeffs_kernel(const T* __restrict__ wd, const float* __restrict__ gamma,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const double* __restrict__ gsums, double inv_count,
                                                           const double* __restrict__ lsums, float* dgamma, float* dbeta,
                                                           T* __restrict__ wa, T* __restrict__ wbn,
                                                           float* __restrict__ col_const, float* __restrict__ coef, int C,
                                                           int p) {
    constexpr int E = 8;
    __shared__ float red[4];
    const int ci = blockIdx.x, v = blockIdx.y;
    const T* src = wd + (long)ci * C;
    T* oa = wa + ((long)v * p + ci) * C;
    T* ob = wbn + ((long)v * p + ci) * C;
    const float* mu_v = mean + (long)v * C;
    const float* is_v = invstd + (long)v * C;
    const double* gs = gsums + (long)v * 2 * C;
    float part = 0.f;
    for (int c0 = threadIdx.x * E; c0 < C; c0 += 256 * E) {
        float a[E], b[E], m1[E], mu[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = c0 + e;
            const float is = is_v[c];
            a[e] = (gamma ? gamma[c] : 1.f) * is;
            m1[e] = (float)(gs[c] * inv_count);
            b[e] = a[e] * is * (float)(gs[C + c] * inv_count);
            mu[e] = mu_v[c];
        }
        float w[E], fa[E], fb[E];
        unpack16<T>(*reinterpret_cast<const uint4*>(src + c0), w);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            fa[e] = a[e] * w[e];
            fb[e] = -b[e] * w[e];
        }
        const uint4 pa = pack16<T>(fa), pb = pack16<T>(fb);
        *reinterpret_cast<uint4*>(oa + c0) = pa;
        *reinterpret_cast<uint4*>(ob + c0) = pb;
        unpack16<T>(pa, fa);
        unpack16<T>(pb, fb);
#pragma unroll
        for (int e = 0; e < E; ++e) part -= mu[e] * fb[e] + m1[e] * fa[e];
        if (ci == 0) {
            float* cf = coef + (long)v * 4 * C;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int c = c0 + e;
                cf[c] = a[e];
                cf[C + c] = b[e];
                cf[2 * C + c] = m1[e];
                cf[3 * C + c] = mu[e];
                if (lsums) {  // parameter gradients from the LOCAL sums (atomics: views / lanes may add concurrently)
                    const double* ls = lsums + (long)v * 2 * C;
                    if (dbeta) atomicAdd(&dbeta[c], (float)ls[c]);
                    if (dgamma) atomicAdd(&dgamma[c], (float)ls[C + c]);
                }
            }
        }
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) col_const[(long)v * p + ci] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- sm3_linbn_post -------------------------------------------------------------------------------------------------
// The two small matrix products of the scheme, as 32 x 32 output tiles, one wave each, operands straight from L2 (they are
// a few hundred KB): no LDS, no barriers; loads of step t + 1 are issued before the MFMAs of step t.
//   tiles [0, views * (p/32)^2):  -H_v[k][ci] = sum_co wbn_v[k][co] * wd[ci][co]           (16-bit MFMA, K = C)
//   tiles beyond:  dw[co][ci] += sum_v a (P_v - m1 s_v^T) - b (W G_v - mu s_v^T)           (exact-f32 MFMA for W G, K = p)
// f32 tile: lane (i = l & 31, kk = l >> 5) walks k = kk * p/2 + j, j = 0 .. p/2: both operands are read along their rows.
template <typename T>
__global__ __launch_bounds__(256) void linbn_post_kernel(const T* __restrict__ wbn, const T* __restrict__ wd,
                                                         T* __restrict__ hn, const float* __restrict__ P,
                                                         const float* __restrict__ G, const T* __restrict__ w,
                                                         const float* __restrict__ s, const float* __restrict__ coef,
                                                         float* __restrict__ dw, int C, int p, int views, int h_tiles) {
    const int lane = threadIdx.x & 63, i = lane & 31, kk = lane >> 5;
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int pt = p / 32;
    if (tile < h_tiles) {
        const int v = tile / (pt * pt), rem = tile - v * pt * pt;
        const int k0 = (rem / pt) * 32, c0 = (rem % pt) * 32;
        const T* ap = wbn + ((long)v * p + k0 + i) * C + 8 * kk;
        const T* bp = wd + (long)(c0 + i) * C + 8 * kk;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        constexpr int U = 4;  // K-steps of 16 channels per batch of loads (C is a multiple of 128)
        uint4 fa[U], fb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fa[u] = *reinterpret_cast<const uint4*>(ap + 16 * u);
            fb[u] = *reinterpret_cast<const uint4*>(bp + 16 * u);
        }
        for (int c = 0; c < C; c += 16 * U) {
            uint4 na[U], nb[U];
            const int cn = c + 16 * U < C ? c + 16 * U : c;  // last batch: harmless re-load
#pragma unroll
            for (int u = 0; u < U; ++u) {
                na[u] = *reinterpret_cast<const uint4*>(ap + cn + 16 * u);
                nb[u] = *reinterpret_cast<const uint4*>(bp + cn + 16 * u);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t))
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[u]),
                                                                  __builtin_bit_cast(bf16x8, fb[u]), acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[u]),
                                                                 __builtin_bit_cast(f16x8, fb[u]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                fa[u] = na[u];
                fb[u] = nb[u];
            }
        }
        T* out = hn + ((long)v * p + k0) * p + c0 + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float q = acc[r];
            if constexpr (__is_same(T, bf16_t)) out[(long)row * p].v = f32_to_bf16(q);
            else out[(long)row * p].v = f32_to_f16(q);
        }
        return;
    }
    const int ft = tile - h_tiles;
    if (ft >= (C / 32) * pt) return;
    const int co0 = (ft / pt) * 32, ci0 = (ft % pt) * 32;
    const int half = p / 2;
    float tot[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[r] = 0.f;
    const T* ap = w + (long)(co0 + i) * p + kk * half;
    for (int v = 0; v < views; ++v) {
        const float* bp = G + ((long)v * p + ci0 + i) * p + kk * half;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float a4[4];
        float4 b4 = *reinterpret_cast<const float4*>(bp);
#pragma unroll
        for (int e = 0; e < 4; ++e) a4[e] = ElemTraits<T>::load(ap + e);
        for (int j = 0; j < half; j += 4) {
            const int jn = j + 4 < half ? j + 4 : j;
            float n4[4];
            const float4 nb = *reinterpret_cast<const float4*>(bp + jn);
#pragma unroll
            for (int e = 0; e < 4; ++e) n4[e] = ElemTraits<T>::load(ap + jn + e);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b4.w, acc, 0, 0, 0);
            b4 = nb;
#pragma unroll
            for (int e = 0; e < 4; ++e) a4[e] = n4[e];
        }
        const float* cf = coef + (long)v * 4 * C;
        const float sv = s[(long)v * p + ci0 + i];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float a = cf[co], b = cf[C + co], m1 = cf[2 * C + co], mu = cf[3 * C + co];
            const float pv = P[((long)v * C + co) * p + ci0 + i];
            tot[r] += a * (pv - m1 * sv) - b * (acc[r] - mu * sv);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        dw[(long)co * p + ci0 + i] += tot[r];
    }
}

}  // namespace

extern "C" int sm3_linbn_stats(int dtype, const float* P, const void* w_fwd, const float* mean, const float* invstd,
                               const double* reduce_ws, int groups, double* lsums, const float* colsum_partials,
                               int colsum_rows, float* s_out, int C, int p, int views, void* stream) {
    if (!P || !w_fwd || !mean || !invstd || !reduce_ws || !lsums || !colsum_partials || !s_out) return SM3_EINVAL;
    if (C <= 0 || p <= 0 || views < 1 || colsum_rows < 1 || groups < 1) return SM3_EINVAL;
    if (dtype != SM3_BF16 && dtype != SM3_F16) return SM3_EDTYPE;
    const int row_blocks = (views * C + 3) / 4, col_blocks = views * ((p + 31) / 32);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_stats_kernel<bf16_t>, dim3(row_blocks + col_blocks), dim3(256), 0, st, P, (const bf16_t*)w_fwd,
                           mean, invstd, reduce_ws, groups, lsums, colsum_partials, colsum_rows, s_out, C, p, views,
                           row_blocks);
    else
        hipLaunchKernelGGL(linbn_stats_kernel<f16_t>, dim3(row_blocks + col_blocks), dim3(256), 0, st, P, (const f16_t*)w_fwd,
                           mean, invstd, reduce_ws, groups, lsums, colsum_partials, colsum_rows, s_out, C, p, views,
                           row_blocks);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_coeffs(int dtype, const void* w_dgrad, const float* gamma, const float* mean, const float* invstd,
                                const double* global_sums, double count, const double* local_sums, float* dgamma,
                                float* dbeta, void* wa, void* wbn, float* col_const, float* coef, int C, int p, int views,
                                void* stream) {
    if (!w_dgrad || !mean || !invstd || !global_sums || !wa || !wbn || !col_const || !coef) return SM3_EINVAL;
    if (C <= 0 || p <= 0 || views < 1 || !(count > 0)) return SM3_EINVAL;
    if (dtype != SM3_BF16 && dtype != SM3_F16) return SM3_EDTYPE;
    if (C % 8) return SM3_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_coeffs_kernel<bf16_t>, dim3(p, views), dim3(256), 0, st, (const bf16_t*)w_dgrad, gamma, mean,
                           invstd, global_sums, 1.0 / count, local_sums, dgamma, dbeta, (bf16_t*)wa, (bf16_t*)wbn, col_const,
                           coef, C, p);
    else
        hipLaunchKernelGGL(linbn_coeffs_kernel<f16_t>, dim3(p, views), dim3(256), 0, st, (const f16_t*)w_dgrad, gamma, mean,
                           invstd, global_sums, 1.0 / count, local_sums, dgamma, dbeta, (f16_t*)wa, (f16_t*)wbn, col_const,
                           coef, C, p);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_post(int dtype, const void* wbn, const void* w_dgrad, void* hn, const float* P, const float* G,
                              const void* w_fwd, const float* s, const float* coef, float* dw, int C, int p, int views,
                              void* stream) {
    if (!wbn || !w_dgrad || !hn || !P || !G || !w_fwd || !s || !coef || !dw) return SM3_EINVAL;
    if (C <= 0 || p <= 0 || views < 1) return SM3_EINVAL;
    if (dtype != SM3_BF16 && dtype != SM3_F16) return SM3_EDTYPE;
    if (C % 128 || p % 32) return SM3_EALIGN;
    const int pt = p / 32, h_tiles = views * pt * pt, f_tiles = (C / 32) * pt;
    const unsigned blocks = (unsigned)((h_tiles + f_tiles + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_post_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)wbn, (const bf16_t*)w_dgrad,
                           (bf16_t*)hn, P, G, (const bf16_t*)w_fwd, s, coef, dw, C, p, views, h_tiles);
    else
        hipLaunchKernelGGL(linbn_post_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)wbn, (const f16_t*)w_dgrad,
                           (f16_t*)hn, P, G, (const f16_t*)w_fwd, s, coef, dw, C, p, views, h_tiles);
    SM3_CHECK_LAUNCH();
    return 0;
}
