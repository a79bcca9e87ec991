// BatchNorm BY LINEARITY for an expanding 1x1 convolution followed by train-mode BatchNorm
// (Bottleneck conv3 -> bn3, reference src/models/resnet.py:162-163, and their autograd backward):
//
//   forward   x = y W^T          y: [M, p] (conv input),  W: [C, p],  C = 4p;   z = gamma * (x - mu) * invstd + beta
//   backward  dx = a (dz - m1) - b (x - mu),   a = gamma invstd,  b = a invstd m2,  m1 = mean(dz),  m2 = mean(dz xhat)
//
// x is linear in y and every consumer of dx is linear in it, so the widest tensor of the block -- x, 4x the size of y --
// is needed by nobody as long as two small moments of y are at hand:  s = sum_m y [p]  and  G = y^T y [p, p]
// (sm3_bn_act_colsum; one launch of the weight-gradient kernel on y alone):
//   batch statistics   sum_m x = W s,   sum_m x^2 = rowdot(W G, W)          -> mu, invstd WITHOUT x (sm3_linbn_fwd_stats),
//                      so conv3 can apply bn3 + residual + ReLU in its own epilogue (sm3_conv_bn_act_fused): x never
//                      reaches HBM in the forward pass either
//   P = dz^T y [C, p]  the weight-gradient GEMM on dz itself (sm3_conv_wgrad_cat)
//   sum_m dz x         = rowdot(W, P)                                       -> m2 (sm3_linbn_stats)
//   dy = dx W          = dz (diag(a) W) - y H + const,  H = W^T diag(b) W [p, p],  const = (b mu - a m1) W
//                      (sm3_linbn_banks, sm3_linbn_post, then ONE gather-GEMM over the K segments [dz | y]:
//                      sm3_conv_dgrad_seg_bnfuse)
//   dW = dx^T y        = diag(a) (P - m1 s^T) - diag(b) (W G - mu s^T)      (sm3_linbn_post)
// Per Bottleneck and step this removes the write and three reads of x and the BatchNorm-backward apply pass (read dz,
// read x, write dx) -- 22 of the 80 activation-sized transfers.  All per-channel vectors are [views][C] (the two views of
// a branch share a launch but not their statistics).  16-bit activation types only: the exact-f32 parity mode keeps the
// two-pass BatchNorm.  The kernels here are the small pieces between the GEMMs.
#include "common.h"

namespace {

// 32 x 32 tile of W G on the exact-f32 MFMA, operands straight from L2 (a few hundred KB): lane (i = l & 31, kk = l >> 5)
// feeds A[i][k] = wd[k][co0 + i] (the data-gradient bank [p][C]: contiguous over i) and B[k][j] = G[k][ci0 + j] for
// k = 2 t + kk; the loads of 8 steps are issued before the first MFMA of the batch; K is split over the block's 4 waves.  acc[r] = (W G)[co0 + row(r)][ci0 + i],
// row(r) = (r & 3) + 8 (r >> 2) + 4 kk.
template <typename T>
__device__ __forceinline__ void tile_wg(const T* __restrict__ wd, int C, const float* __restrict__ G, int p, int co0, int ci0,
                                        int lane, int wave, f32x16& acc) {
    const int i = lane & 31, kk = lane >> 5;
    const T* ap = wd + (long)kk * C + co0 + i;
    const float* bp = G + (long)kk * p + ci0 + i;
    constexpr int U = 8;
    for (int t = wave * U; t < p / 2; t += 4 * U) {  // the block's 4 waves take every fourth batch of K-steps
        float a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = ElemTraits<T>::load(ap + (long)(2 * (t + u)) * C);
            b[u] = bp[(long)(2 * (t + u)) * p];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
}

// The four waves' partial tiles added up in a fixed order; wave w leaves with accumulator registers 4 w .. 4 w + 3 of the sum.
__device__ __forceinline__ void reduce4(const f32x16& acc, float* red, int wave, int lane, float out[4]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 4 + q;
        out[q] = (red[r * 64 + lane] + red[(16 + r) * 64 + lane]) + (red[(32 + r) * 64 + lane] + red[(48 + r) * 64 + lane]);
    }
}

__device__ __forceinline__ double half_sum(double v) {  // over the 32 lanes of a wave half
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- sm3_linbn_moments ----------------------------------------------------------------------------------------------
// The two moments of y in their final form, summed in a FIXED order (no atomics anywhere on the forward path: a step's
// forward is bit-reproducible, and a view's moments do not depend on whether its sibling view shared the launches):
//   blocks [0, gblocks):  G[v][e] = sum_j slabs[v][j][e]            (the slabs sm3_conv_wgrad_slabs left for y^T y; the
//                         same sum gives P = dz^T y from its slabs: n elements per view, no column sums)
//   blocks beyond:        s[v][ci] = sum_r colsum_partials[v][r][ci] (sm3_bn_act_colsum), 32 columns x 8 row lanes per block
constexpr int kMomentsLaneSlabs = 48;  // from here on the slabs of an element are walked by 8 lanes (linbn_moments_kernel)
__global__ __launch_bounds__(256) void linbn_moments_kernel(const float* __restrict__ slabs, int nslabs, long pp,
                                                            const float* __restrict__ colsum, int crow,
                                                            float* __restrict__ G, double* __restrict__ s_out, int p,
                                                            int views, int gblocks) {
    if ((int)blockIdx.x < gblocks && nslabs >= kMomentsLaneSlabs) {
        // many slabs (the rule is a function of the slab count, i.e. of the geometry): a workgroup owns 32 float4 columns x 8 slab lanes (the walk of slab_reduce_kernel): lane l adds slabs l, l + 8, ... on
        // four interleaved accumulators, the eight lane sums are added in lane order through LDS.  (One thread per float4
        // walking ALL slabs left the 64 x 64 Gram matrices of layer 1 -- 256 slabs, 8 workgroups -- at 24 us per launch.)
        __shared__ float4 gred[8][32];
        const int col = threadIdx.x & 31, jl = threadIdx.x >> 5;
        const long e4 = ((long)blockIdx.x * 32 + col) * 4;
        const bool in = e4 < (long)views * pp;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
        if (in) {
            const int v = (int)(e4 / pp);
            const float* src = slabs + (long)v * nslabs * pp + (e4 - (long)v * pp);
            int j = jl;
            for (; j + 24 < nslabs; j += 32) {
                const float4 v0 = *reinterpret_cast<const float4*>(src + (long)j * pp);
                const float4 v1 = *reinterpret_cast<const float4*>(src + (long)(j + 8) * pp);
                const float4 v2 = *reinterpret_cast<const float4*>(src + (long)(j + 16) * pp);
                const float4 v3 = *reinterpret_cast<const float4*>(src + (long)(j + 24) * pp);
                add(a0, v0); add(a1, v1); add(a2, v2); add(a3, v3);
            }
            for (; j < nslabs; j += 8) add(a0, *reinterpret_cast<const float4*>(src + (long)j * pp));
            add(a0, a1);
            add(a2, a3);
            add(a0, a2);
        }
        gred[jl][col] = a0;
        __syncthreads();
        if (jl == 0 && in) {
            float4 t = gred[0][col];
#pragma unroll
            for (int l = 1; l < 8; ++l) add(t, gred[l][col]);
            *reinterpret_cast<float4*>(G + e4) = t;
        }
        return;
    }
    if ((int)blockIdx.x < gblocks) {  // few slabs: one thread per float4 walks them all (the lanes above would idle)
        const long e4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;  // four consecutive elements of one view's matrix
        if (e4 >= (long)views * pp) return;
        const int v = (int)(e4 / pp);
        const float* src = slabs + (long)v * nslabs * pp + (e4 - (long)v * pp);
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
        int j = 0;
        for (; j + 3 < nslabs; j += 4) {
            add(a0, *reinterpret_cast<const float4*>(src + (long)j * pp));
            add(a1, *reinterpret_cast<const float4*>(src + (long)(j + 1) * pp));
            add(a2, *reinterpret_cast<const float4*>(src + (long)(j + 2) * pp));
            add(a3, *reinterpret_cast<const float4*>(src + (long)(j + 3) * pp));
        }
        for (; j < nslabs; ++j) add(a0, *reinterpret_cast<const float4*>(src + (long)j * pp));
        add(a0, a1);
        add(a2, a3);
        add(a0, a2);
        *reinterpret_cast<float4*>(G + e4) = a0;
        return;
    }
    __shared__ double red[8][32];
    const int cb = blockIdx.x - gblocks, pb = (p + 31) / 32;
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
        s_out[(long)v * p + ci] = t;
    }
}

// ---- sm3_linbn_fwd_stats --------------------------------------------------------------------------------------------
// one block per (view, 32 channels co, 32 columns ci), K split over its 4 waves: Tm[v][co][ci] = (W G_v)[co][ci], and the
// tile's share of the batch sums of x = y W^T:  ws[v][cit][co] = sum_{ci in tile} W[co][ci] s_v[ci],  ws[v][cit][C + co] =
// sum_{ci} Tm W[co][ci] -- the [views][groups = p/32][2C] fp64 layout sm3_bn_finalize sums over.
template <typename T>
__global__ __launch_bounds__(256) void linbn_fwd_stats_kernel(const float* __restrict__ G, const T* __restrict__ wd,
                                                              const T* __restrict__ w, const double* __restrict__ s,
                                                              float* __restrict__ Tm, double* __restrict__ ws, int C,
                                                              int p, int views) {
    __shared__ float red[4 * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
    const int pt = p / 32, ct = C / 32;
    const int tile = blockIdx.x;
    const int v = tile / (ct * pt), rem = tile - v * ct * pt;
    const int co0 = (rem / pt) * 32, cit = rem % pt, ci0 = cit * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    tile_wg<T>(wd, C, G + (long)v * p * p, p, co0, ci0, lane, wave, acc);
    float t4[4];
    reduce4(acc, red, wave, lane, t4);
    const double sv = s[(long)v * p + ci0 + i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 4 + q;
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        Tm[((long)v * C + co) * p + ci0 + i] = t4[q];
        const double wv = (double)ElemTraits<T>::load(w + (long)co * p + ci0 + i);
        const double s1 = half_sum(wv * sv), s2 = half_sum(wv * (double)t4[q]);
        if (i == 0) {
            double* o = ws + ((long)v * pt + cit) * 2 * C;
            o[co] = s1;
            o[C + co] = s2;
        }
    }
}

// ---- sm3_linbn_fold -------------------------------------------------------------------------------------------------
// out[v][e] = sum_g ws[v][g][e] in a fixed order: the partial rows sm3_linbn_fwd_stats left, folded BEFORE a data-parallel
// exchange so that ranks trade [views][2C] sums per BatchNorm, not [views][p/32][2C].
__global__ __launch_bounds__(256) void linbn_fold_kernel(const double* __restrict__ ws, int groups, int n, int views,
                                                         double* __restrict__ out) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)views * n) return;
    const int v = (int)(e / n);
    const double* src = ws + (long)v * groups * n + (e - (long)v * n);
    double a = 0.0;
    for (int g = 0; g < groups; ++g) a += src[(long)g * n];
    out[e] = a;
}

// ---- sm3_linbn_stats ------------------------------------------------------------------------------------------------
// one wave per (view, channel): S1 = sum over the `groups` rows stage A of sm3_bn_stats_reduce left in ws (its stage B,
// folded in here), S2 = invstd (rowdot(W[co], P_v[co]) - mu S1); lsums[v] = (S1 | S2); dgamma += S2, dbeta += S1.
// inv_count > 0 (no exchange of the sums between ranks to wait for): also coef[v][co] = (a, b, m1, mu).
template <typename T>
__global__ __launch_bounds__(256) void linbn_stats_kernel(const float* __restrict__ P, const T* __restrict__ w,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const double* __restrict__ ws,
                                                          int groups, double* __restrict__ lsums, float* dgamma,
                                                          float* dbeta, double inv_count, float4* __restrict__ coef, int C,
                                                          int p, int views) {
    const int lane = threadIdx.x & 63;
    const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (co >= C) return;
    const T* wr = w + (long)co * p;
    // one wave per channel walks the views of the launch IN ORDER, so d(gamma) / d(beta) receive their addends in a fixed
    // order (a step's gradients are a function of its inputs); atomics only because both lanes of a SHARED projector may add
    // to the same parameter from two streams
    for (int v = 0; v < views; ++v) {
        const float* pr = P + ((long)v * C + co) * p;
        double acc = 0.0, s1 = 0.0;
        for (int k = lane; k < p; k += 64) acc += (double)pr[k] * (double)ElemTraits<T>::load(wr + k);
        for (int g = lane; g < groups; g += 64) s1 += ws[((long)v * groups + g) * 2 * C + co];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc += __shfl_xor(acc, o, 64);
            s1 += __shfl_xor(s1, o, 64);
        }
        if (lane == 0) {
            const float is = invstd[(long)v * C + co], mu = mean[(long)v * C + co];
            const double s2 = (double)is * (acc - (double)mu * s1);
            double* ls = lsums + (long)v * 2 * C;
            ls[co] = s1;
            ls[C + co] = s2;
            if (dbeta) atomicAdd(&dbeta[co], (float)s1);
            if (dgamma) atomicAdd(&dgamma[co], (float)s2);
            if (inv_count > 0) {
                const float a = (gamma ? gamma[co] : 1.f) * is;
                coef[(long)v * C + co] = make_float4(a, a * is * (float)(s2 * inv_count), (float)(s1 * inv_count), mu);
            }
        }
    }
}

// data parallel: the same coefficients from the all-reduced sums
__global__ void linbn_coef_kernel(const double* __restrict__ gsums, double inv_count, const float* __restrict__ gamma,
                                  const float* __restrict__ mean, const float* __restrict__ invstd,
                                  float4* __restrict__ coef, int C, int views) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= views * C) return;
    const int v = t / C, co = t - v * C;
    const float is = invstd[t], a = (gamma ? gamma[co] : 1.f) * is;
    const double* gs = gsums + (long)v * 2 * C;
    coef[t] = make_float4(a, a * is * (float)(gs[C + co] * inv_count), (float)(gs[co] * inv_count), mean[t]);
}

// ---- sm3_linbn_banks ------------------------------------------------------------------------------------------------
// grid (p, views): block (ci, v) walks row ci of the data-gradient bank wd [p][C]:
//   wa[v][ci][co] = T(a wd),  wbn[v][ci][co] = T(-b wd),  col_const[v][ci] = sum_co -(mu f(wbn) + m1 f(wa))
// (the constant uses the ROUNDED products, so that what the GEMM adds up is centred exactly).
template <typename T>
__device__ __forceinline__ void banks_row(const T* __restrict__ wd, const float4* __restrict__ coef, T* __restrict__ wa,
                                          T* __restrict__ wbn, float* __restrict__ col_const, int C, int p, int ci, int v,
                                          float* red) {
    constexpr int E = 8;
    const T* src = wd + (long)ci * C;
    T* oa = wa + ((long)v * p + ci) * C;
    T* ob = wbn ? wbn + ((long)v * p + ci) * C : nullptr;  // (sm3_linbn_banks_post keeps -b wd in registers only)
    const float4* cf = coef + (long)v * C;
    float part = 0.f;
    for (int c0 = threadIdx.x * E; c0 < C; c0 += 256 * E) {
        float4 q[E];
#pragma unroll
        for (int e = 0; e < E; ++e) q[e] = cf[c0 + e];
        float w[E], fa[E], fb[E];
        unpack16<T>(*reinterpret_cast<const uint4*>(src + c0), w);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            fa[e] = q[e].x * w[e];
            fb[e] = -q[e].y * w[e];
        }
        const uint4 pa = pack16<T>(fa), pb = pack16<T>(fb);
        *reinterpret_cast<uint4*>(oa + c0) = pa;
        if (ob) *reinterpret_cast<uint4*>(ob + c0) = pb;
        unpack16<T>(pa, fa);
        unpack16<T>(pb, fb);
#pragma unroll
        for (int e = 0; e < E; ++e)  // explicit fma: every instantiation of this row (with / without the stored -b wd bank) rounds alike
            part -= __builtin_fmaf(q[e].w, fb[e], q[e].z * fa[e]);
    }
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) col_const[(long)v * p + ci] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename T>
__global__ __launch_bounds__(256) void linbn_banks_kernel(const T* __restrict__ wd, const float4* __restrict__ coef,
                                                          T* __restrict__ wa, T* __restrict__ wbn,
                                                          float* __restrict__ col_const, int C, int p) {
    __shared__ float red[4];
    banks_row<T>(wd, coef, wa, wbn, col_const, C, p, blockIdx.x, blockIdx.y, red);
}

// ---- sm3_linbn_scale_banks ------------------------------------------------------------------------------------------
// Forward of a Bottleneck's join with a downsample branch in ONE two-segment GEMM: the two BatchNorm scales go into the
// filter banks, out3[v][c][:] = T(scale3[v][c] w3[c][:]), outd[v][c][:] = T(scaled[v][c] wd[c][:]), and the two shifts add
// up to the GEMM's column bias.  grid (C, views).
template <typename T>
__global__ __launch_bounds__(256) void linbn_scale_banks_kernel(const T* __restrict__ w3, int K3, const float* __restrict__ sc3,
                                                                T* __restrict__ out3, const T* __restrict__ wd, int Kd,
                                                                const float* __restrict__ scd, T* __restrict__ outd,
                                                                const float* __restrict__ sh3, const float* __restrict__ shd,
                                                                float* __restrict__ bias, int C) {
    const int c = blockIdx.x, v = blockIdx.y;
    const float a = sc3[(long)v * C + c], b = scd[(long)v * C + c];
    for (int k0 = threadIdx.x * 8; k0 < K3 + Kd; k0 += 256 * 8) {
        const bool first = k0 < K3;
        const int k = first ? k0 : k0 - K3;
        const T* src = first ? w3 + (long)c * K3 + k : wd + (long)c * Kd + k;
        T* dst = first ? out3 + ((long)v * C + c) * K3 + k : outd + ((long)v * C + c) * Kd + k;
        float f[8];
        unpack16<T>(*reinterpret_cast<const uint4*>(src), f);
        const float q = first ? a : b;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= q;
        *reinterpret_cast<uint4*>(dst) = pack16<T>(f);
    }
    if (threadIdx.x == 0) bias[(long)v * C + c] = sh3[(long)v * C + c] + shd[(long)v * C + c];
}

// ---- sm3_linbn_post -------------------------------------------------------------------------------------------------
// 32 x 32 output tiles, one wave each, no LDS:
//   tiles [0, views (p/32)^2):  hn[v][k][ci] = sum_co wbn_v[k][co] wd[ci][co] = -H_v          (16-bit MFMA, K = C)
//   tiles beyond:  dw[co][ci] += sum_v a (P_v - m1 s_v^T) - b (W G_v - mu s_v^T),  W G_v read from Tm (saved by
//                  sm3_linbn_fwd_stats) or, Tm == nullptr, recomputed here on the exact-f32 MFMA
// wbn == nullptr (sm3_linbn_banks_post): the A fragments of the H tiles, T(-b wd), are formed here from wd and coef -- the
// same fp32 product and the same rounding sm3_linbn_banks applies, so H has the same bits as with the stored bank.
template <typename T>
__device__ __forceinline__ void post_tile(const T* __restrict__ wbn, const T* __restrict__ wd, T* __restrict__ hn,
                                          const float* __restrict__ P, const float* __restrict__ G,
                                          const float* __restrict__ Tm, const double* __restrict__ s,
                                          const float4* __restrict__ coef, float* __restrict__ dw, int C, int p, int views,
                                          int h_tiles, int tile, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, kk = lane >> 5;
    const int pt = p / 32;
    if (tile < h_tiles) {
        const int v = tile / (pt * pt), rem = tile - v * pt * pt;
        const int k0 = (rem / pt) * 32, c0 = (rem % pt) * 32;
        const T* ap = wbn ? wbn + ((long)v * p + k0 + i) * C + 8 * kk : wd + (long)(k0 + i) * C + 8 * kk;
        const T* bp = wd + (long)(c0 + i) * C + 8 * kk;
        const float4* cf = coef + (long)v * C + 8 * kk;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        constexpr int U = 8;  // K-steps of 16 channels per batch of loads (C is a multiple of 128)
        for (int c = wave * 16 * U; c < C; c += 4 * 16 * U) {
            uint4 fa[U], fb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                fa[u] = *reinterpret_cast<const uint4*>(ap + c + 16 * u);
                fb[u] = *reinterpret_cast<const uint4*>(bp + c + 16 * u);
            }
            if (!wbn) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float w8[8];
                    unpack16<T>(fa[u], w8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) w8[e] = -cf[c + 16 * u + e].y * w8[e];
                    fa[u] = pack16<T>(w8);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if constexpr (__is_same(T, bf16_t))
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[u]),
                                                                  __builtin_bit_cast(bf16x8, fb[u]), acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[u]),
                                                                 __builtin_bit_cast(f16x8, fb[u]), acc, 0, 0, 0);
            }
        }
        float t4[4];
        reduce4(acc, red, wave, lane, t4);
        T* out = hn + ((long)v * p + k0) * p + c0 + i;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kk;
            if constexpr (__is_same(T, bf16_t)) out[(long)row * p].v = f32_to_bf16(t4[q]);
            else out[(long)row * p].v = f32_to_f16(t4[q]);
        }
        return;
    }
    const int ft = tile - h_tiles;
    const int co0 = (ft / pt) * 32, ci0 = (ft % pt) * 32;
    float tot[4] = {0.f, 0.f, 0.f, 0.f};
    for (int v = 0; v < views; ++v) {
        float t4[4];
        if (Tm) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = wave * 4 + q;
                t4[q] = Tm[((long)v * C + co0 + (r & 3) + 8 * (r >> 2) + 4 * kk) * p + ci0 + i];
            }
        } else {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            tile_wg<T>(wd, C, G + (long)v * p * p, p, co0, ci0, lane, wave, acc);
            if (v) __syncthreads();  // the previous view's partial tiles have been read
            reduce4(acc, red, wave, lane, t4);
        }
        const float sv = (float)s[(long)v * p + ci0 + i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
            const float4 c4 = coef[(long)v * C + co];  // (a, b, m1, mu)
            const float pv = P[((long)v * C + co) * p + ci0 + i];
            tot[q] += c4.x * (pv - c4.z * sv) - c4.y * (t4[q] - c4.w * sv);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 4 + q;
        const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        dw[(long)co * p + ci0 + i] += tot[q];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void linbn_post_kernel(const T* __restrict__ wbn, const T* __restrict__ wd,
                                                         T* __restrict__ hn, const float* __restrict__ P,
                                                         const float* __restrict__ G, const float* __restrict__ Tm,
                                                         const double* __restrict__ s, const float4* __restrict__ coef,
                                                         float* __restrict__ dw, int C, int p, int views, int h_tiles) {
    __shared__ float red[4 * 16 * 64];
    post_tile<T>(wbn, wd, hn, P, G, Tm, s, coef, dw, C, p, views, h_tiles, blockIdx.x, red);  // one 32 x 32 tile per block
}

// ---- sm3_linbn_banks_post -------------------------------------------------------------------------------------------
// sm3_linbn_banks and sm3_linbn_post as ONE launch (round 5: the chain between the GEMMs of a unit's backward is launch
// latency, 13 us per link): blocks [0, tiles) are the -H and weight-gradient tiles of post_tile (the -b wd operand formed on
// the fly: nothing of the bank rows below is read by them), the p * views blocks behind them write diag(a) wd and the
// constant row.  The -diag(b) wd bank is never stored.
template <typename T>
__global__ __launch_bounds__(256) void linbn_banks_post_kernel(const T* __restrict__ wd, const float4* __restrict__ coef,
                                                               T* __restrict__ wa, float* __restrict__ col_const,
                                                               T* __restrict__ hn, const float* __restrict__ P,
                                                               const float* __restrict__ G, const float* __restrict__ Tm,
                                                               const double* __restrict__ s, float* __restrict__ dw, int C,
                                                               int p, int views, int h_tiles, int tiles) {
    __shared__ float red[4 * 16 * 64];
    const int b = blockIdx.x;
    if (b < tiles) {
        post_tile<T>(nullptr, wd, hn, P, G, Tm, s, coef, dw, C, p, views, h_tiles, b, red);
        return;
    }
    const int r = b - tiles;
    banks_row<T>(wd, coef, wa, nullptr, col_const, C, p, r % p, r / p, red);
}


inline bool lin16(int dtype) { return dtype == SM3_BF16 || dtype == SM3_F16; }

}  // namespace

extern "C" int sm3_linbn_moments(const float* slabs, int nslabs, int64_t n, float* out, const float* colsum_partials,
                                 int colsum_rows, double* s_out, int p, int views, void* stream) {
    if (!slabs || !out || nslabs < 1 || n <= 0 || views < 1) return SM3_EINVAL;
    if (colsum_partials && (!s_out || colsum_rows < 1 || p <= 0)) return SM3_EINVAL;
    if (n % 4) return SM3_EALIGN;
    const long per_block = nslabs >= kMomentsLaneSlabs ? 32 : 256;  // float4 columns per workgroup
    const int gblocks = (int)(((long)views * n / 4 + per_block - 1) / per_block), cblocks = colsum_partials ? views * ((p + 31) / 32) : 0;
    hipLaunchKernelGGL(linbn_moments_kernel, dim3(gblocks + cblocks), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, (long)n,
                       colsum_partials, colsum_rows, out, s_out, p, views, gblocks);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_fwd_stats(int dtype, const float* G, const void* w_dgrad, const void* w_fwd, const double* s,
                                   float* Tm, double* sums_ws, int C, int p, int views, void* stream) {
    if (!G || !w_dgrad || !w_fwd || !s || !Tm || !sums_ws || C <= 0 || p <= 0 || views < 1) return SM3_EINVAL;
    if (!lin16(dtype)) return SM3_EDTYPE;
    if (C % 32 || p % 32) return SM3_EALIGN;
    const unsigned blocks = (unsigned)(views * (C / 32) * (p / 32));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_fwd_stats_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, G, (const bf16_t*)w_dgrad,
                           (const bf16_t*)w_fwd, s, Tm, sums_ws, C, p, views);
    else
        hipLaunchKernelGGL(linbn_fwd_stats_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, G, (const f16_t*)w_dgrad,
                           (const f16_t*)w_fwd, s, Tm, sums_ws, C, p, views);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_fold(const double* sums_ws, int groups, int n, int views, double* out, void* stream) {
    if (!sums_ws || !out || groups < 1 || n <= 0 || views < 1) return SM3_EINVAL;
    hipLaunchKernelGGL(linbn_fold_kernel, dim3((unsigned)(((long)views * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       sums_ws, groups, n, views, out);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_stats(int dtype, const float* P, const void* w_fwd, const float* mean, const float* invstd,
                               const float* gamma, const double* reduce_ws, int groups, double* lsums, float* dgamma,
                               float* dbeta, double count, float* coef, int C, int p, int views, void* stream) {
    if (!P || !w_fwd || !mean || !invstd || !reduce_ws || !lsums) return SM3_EINVAL;
    if (C <= 0 || p <= 0 || views < 1 || groups < 1 || (count > 0 && !coef)) return SM3_EINVAL;
    if (!lin16(dtype)) return SM3_EDTYPE;
    const unsigned blocks = (unsigned)((C + 3) / 4);  // one wave per channel, views walked in order
    const double inv = count > 0 ? 1.0 / count : 0.0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_stats_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, P, (const bf16_t*)w_fwd, mean, invstd,
                           gamma, reduce_ws, groups, lsums, dgamma, dbeta, inv, (float4*)coef, C, p, views);
    else
        hipLaunchKernelGGL(linbn_stats_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, P, (const f16_t*)w_fwd, mean, invstd,
                           gamma, reduce_ws, groups, lsums, dgamma, dbeta, inv, (float4*)coef, C, p, views);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_coef(const double* global_sums, double count, const float* gamma, const float* mean,
                              const float* invstd, float* coef, int C, int views, void* stream) {
    if (!global_sums || !mean || !invstd || !coef || C <= 0 || views < 1 || !(count > 0)) return SM3_EINVAL;
    hipLaunchKernelGGL(linbn_coef_kernel, dim3((views * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, global_sums,
                       1.0 / count, gamma, mean, invstd, (float4*)coef, C, views);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_banks(int dtype, const void* w_dgrad, const float* coef, void* wa, void* wbn, float* col_const,
                               int C, int p, int views, void* stream) {
    if (!w_dgrad || !coef || !wa || !wbn || !col_const || C <= 0 || p <= 0 || views < 1) return SM3_EINVAL;
    if (!lin16(dtype)) return SM3_EDTYPE;
    if (C % 8) return SM3_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_banks_kernel<bf16_t>, dim3(p, views), dim3(256), 0, st, (const bf16_t*)w_dgrad,
                           (const float4*)coef, (bf16_t*)wa, (bf16_t*)wbn, col_const, C, p);
    else
        hipLaunchKernelGGL(linbn_banks_kernel<f16_t>, dim3(p, views), dim3(256), 0, st, (const f16_t*)w_dgrad,
                           (const float4*)coef, (f16_t*)wa, (f16_t*)wbn, col_const, C, p);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_scale_banks(int dtype, const void* w3, int K3, const float* scale3, const float* shift3, void* out3,
                                     const void* wd, int Kd, const float* scaled, const float* shiftd, void* outd,
                                     float* bias, int C, int views, void* stream) {
    if (!w3 || !scale3 || !shift3 || !out3 || !wd || !scaled || !shiftd || !outd || !bias) return SM3_EINVAL;
    if (C <= 0 || K3 <= 0 || Kd <= 0 || views < 1) return SM3_EINVAL;
    if (!lin16(dtype)) return SM3_EDTYPE;
    if (K3 % 8 || Kd % 8) return SM3_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_scale_banks_kernel<bf16_t>, dim3(C, views), dim3(256), 0, st, (const bf16_t*)w3, K3, scale3,
                           (bf16_t*)out3, (const bf16_t*)wd, Kd, scaled, (bf16_t*)outd, shift3, shiftd, bias, C);
    else
        hipLaunchKernelGGL(linbn_scale_banks_kernel<f16_t>, dim3(C, views), dim3(256), 0, st, (const f16_t*)w3, K3, scale3,
                           (f16_t*)out3, (const f16_t*)wd, Kd, scaled, (f16_t*)outd, shift3, shiftd, bias, C);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_post(int dtype, const void* wbn, const void* w_dgrad, void* hn, const float* P, const float* G,
                              const float* Tm, const double* s, const float* coef, float* dw, int C, int p, int views,
                              void* stream) {
    if (!wbn || !w_dgrad || !hn || !P || (!G && !Tm) || !s || !coef || !dw) return SM3_EINVAL;
    if (C <= 0 || p <= 0 || views < 1) return SM3_EINVAL;
    if (!lin16(dtype)) return SM3_EDTYPE;
    if (C % 128 || p % 32) return SM3_EALIGN;
    const int pt = p / 32, h_tiles = views * pt * pt, f_tiles = (C / 32) * pt;
    const int h_pad = h_tiles;  // one tile per block
    const unsigned blocks = (unsigned)(h_tiles + f_tiles);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_post_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)wbn, (const bf16_t*)w_dgrad,
                           (bf16_t*)hn, P, G, Tm, s, (const float4*)coef, dw, C, p, views, h_pad);
    else
        hipLaunchKernelGGL(linbn_post_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)wbn, (const f16_t*)w_dgrad,
                           (f16_t*)hn, P, G, Tm, s, (const float4*)coef, dw, C, p, views, h_pad);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_linbn_banks_post(int dtype, const void* w_dgrad, const float* coef, void* wa, float* col_const, void* hn,
                                    const float* P, const float* G, const float* Tm, const double* s, float* dw, int C, int p,
                                    int views, void* stream) {
    if (!w_dgrad || !coef || !wa || !col_const || !hn || !P || (!G && !Tm) || !s || !dw) return SM3_EINVAL;
    if (C <= 0 || p <= 0 || views < 1) return SM3_EINVAL;
    if (!lin16(dtype)) return SM3_EDTYPE;
    if (C % 128 || p % 32) return SM3_EALIGN;
    const int pt = p / 32, h_tiles = views * pt * pt, tiles = h_tiles + (C / 32) * pt;
    const unsigned blocks = (unsigned)(tiles + p * views);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(linbn_banks_post_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)w_dgrad,
                           (const float4*)coef, (bf16_t*)wa, col_const, (bf16_t*)hn, P, G, Tm, s, dw, C, p, views, h_tiles, tiles);
    else
        hipLaunchKernelGGL(linbn_banks_post_kernel<f16_t>, dim3(blocks), dim3(256), 0, st, (const f16_t*)w_dgrad,
                           (const float4*)coef, (f16_t*)wa, col_const, (f16_t*)hn, P, G, Tm, s, dw, C, p, views, h_tiles, tiles);
    SM3_CHECK_LAUNCH();
    return 0;
}
