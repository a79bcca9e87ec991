// Training of the multi-label heads (reference tools/mlc_train.py:58-90 Model, :241-283 loop; tools/mlc_eval.py): the
// non-GEMM parts of one nn.TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout) over the S <= 8 label
// tokens of every sample in TRAIN mode (post-norm, ReLU, four dropouts), the prototype heads, the pseudo-label
// cross-entropy, and the spherical k-means that produces the pseudo-labels (mlc_train.py:116-189).  All fp32 (the
// reference runs this part in fp32, TF32 allowed: mlc_train.py:294-295); the GEMMs around these kernels go through
// sm3_conv_gather_gemm / sm3_conv_wgrad in the exact-f32 MFMA mode.  Token rows are addressed as row(b, s) =
// b*samp_stride + s*tok_stride: (S, 1) for the sample-major layout of the inference path, (1, B) for the reference's own
// [S, B, D] stacking (mlc_train.py:79), where every label projector's GEMM output is one dense block.
//
// Dropout keeps no mask: element e of stream `seed` is kept iff hash(seed, e) >= p, forward and backward recompute it.
#include "common.h"

namespace {

constexpr int kMaxS = 8;

__device__ __forceinline__ bool keep(uint32_t seed, uint32_t idx, float p) {
    uint32_t h = idx * 0x9E3779B1u ^ seed;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (float)(h >> 8) * (1.f / 16777216.f) >= p;
}

// ---- attention over the S tokens of a sample, train mode (dropout on the probabilities) -------------------------
// LDS: probabilities after softmax (and after dropout) [nhead][S][S]
#define ROWP(ptr, i, width) ((ptr) + ((int64_t)b * ss + (int64_t)(i) * ts) * (width))
__device__ __forceinline__ void attention_probs(const float* __restrict__ qkv, int b, int64_t ss, int64_t ts, int S, int D,
                                                int nhead, float* sc) {
    const int tid = threadIdx.x, hd = D / nhead;
    const float inv_sqrt = rsqrtf((float)hd);
    const int ntrip = nhead * S * S;
    for (int t0 = 0; t0 < ntrip; t0 += 64) {
        const int trip = t0 + (tid >> 2), part = tid & 3;
        float a = 0.f;
        if (trip < ntrip) {
            const int h = trip / (S * S), ij = trip - h * S * S, i = ij / S, j = ij - i * S;
            const float* q = ROWP(qkv, i, 3 * D) + h * hd;
            const float* k = ROWP(qkv, j, 3 * D) + D + h * hd;
            for (int d = part; d < hd; d += 4) a += q[d] * k[d];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (trip < ntrip && part == 0) sc[trip] = a * inv_sqrt;
    }
    __syncthreads();
    if (tid < nhead * S) {
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
}

__global__ __launch_bounds__(256) void mlc_attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                int S, int D, int nhead, float p, uint32_t seed, int64_t ss,
                                                                int64_t ts) {
    __shared__ float sc[kMaxS * kMaxS * 8];
    const int b = blockIdx.x, tid = threadIdx.x, hd = D / nhead;
    attention_probs(qkv, b, ss, ts, S, D, nhead, sc);
    const float scale = 1.f / (1.f - p);
    for (int o = tid; o < S * D; o += 256) {
        const int i = o / D, d = o - i * D, h = d / hd;
        const float* pr = sc + (h * S + i) * S;
        float a = 0.f;
        for (int j = 0; j < S; ++j) {
            const uint32_t e = (((uint32_t)b * nhead + h) * S + i) * S + j;
            if (p <= 0.f || keep(seed, e, p)) a += pr[j] * ROWP(qkv, j, 3 * D)[2 * D + d];
        }
        ROWP(out, i, D)[d] = a * (p > 0.f ? scale : 1.f);
    }
}

// dqkv from dout: P' = drop(P)/(1-p); dV = P'^T dO; dP = mask/(1-p) * (dO V^T); dS = P * (dP - sum_j dP*P); dQ = dS K / sqrt(hd); dK = dS^T Q / sqrt(hd)
__global__ __launch_bounds__(256) void mlc_attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                float* __restrict__ dqkv, int S, int D, int nhead, float p,
                                                                uint32_t seed, int64_t ss, int64_t ts) {
    __shared__ float sc[kMaxS * kMaxS * 8];   // P
    __shared__ float dsc[kMaxS * kMaxS * 8];  // dP, then dS
    const int b = blockIdx.x, tid = threadIdx.x, hd = D / nhead;
    attention_probs(qkv, b, ss, ts, S, D, nhead, sc);
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const float inv_sqrt = rsqrtf((float)hd);
    const int ntrip = nhead * S * S;
    // dP[h][i][j] = mask * scale * sum_d dO[i][h,d] V[j][h,d]   (4 lanes per triple)
    for (int t0 = 0; t0 < ntrip; t0 += 64) {
        const int trip = t0 + (tid >> 2), part = tid & 3;
        float a = 0.f;
        int h = 0, i = 0, j = 0;
        if (trip < ntrip) {
            h = trip / (S * S);
            const int ij = trip - h * S * S;
            i = ij / S;
            j = ij - i * S;
            const float* go = ROWP(dout, i, D) + h * hd;
            const float* v = ROWP(qkv, j, 3 * D) + 2 * D + h * hd;
            for (int d = part; d < hd; d += 4) a += go[d] * v[d];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        if (trip < ntrip && part == 0) {
            const uint32_t e = (((uint32_t)b * nhead + h) * S + i) * S + j;
            dsc[trip] = (p <= 0.f || keep(seed, e, p)) ? a * scale : 0.f;
        }
    }
    __syncthreads();
    // dV[j][h,d] = sum_i P'[h][i][j] dO[i][h,d]
    for (int o = tid; o < S * D; o += 256) {
        const int j = o / D, d = o - j * D, h = d / hd;
        float a = 0.f;
        for (int i = 0; i < S; ++i) {
            const uint32_t e = (((uint32_t)b * nhead + h) * S + i) * S + j;
            if (p <= 0.f || keep(seed, e, p)) a += sc[(h * S + i) * S + j] * ROWP(dout, i, D)[d];
        }
        ROWP(dqkv, j, 3 * D)[2 * D + d] = a * scale;
    }
    __syncthreads();
    if (tid < nhead * S) {  // softmax backward of one row, in place dP -> dS
        float* pr = sc + tid * S;
        float* dp = dsc + tid * S;
        float dot = 0.f;
        for (int j = 0; j < S; ++j) dot += dp[j] * pr[j];
        for (int j = 0; j < S; ++j) dp[j] = pr[j] * (dp[j] - dot);
    }
    __syncthreads();
    for (int o = tid; o < S * D; o += 256) {
        const int i = o / D, d = o - i * D, h = d / hd;
        float aq = 0.f, ak = 0.f;
        for (int j = 0; j < S; ++j) {
            aq += dsc[(h * S + i) * S + j] * ROWP(qkv, j, 3 * D)[D + d];   // dQ_i = sum_j dS_ij K_j
            ak += dsc[(h * S + j) * S + i] * ROWP(qkv, j, 3 * D)[d];       // dK_i = sum_j dS_ji Q_j
        }
        ROWP(dqkv, i, 3 * D)[d] = aq * inv_sqrt;
        ROWP(dqkv, i, 3 * D)[D + d] = ak * inv_sqrt;
    }
}

// ---- out = LayerNorm(a + dropout(b)) ; stats[row] = (mean, rstd) -------------------------------------------------
__global__ __launch_bounds__(256) void mlc_add_ln_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, float p, uint32_t seed, float* __restrict__ out,
                                                             float* __restrict__ stats, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    float v[16], s = 0.f;
    int n = 0;
    for (int d = lane; d < D; d += 64, ++n) {
        const int64_t o = row * D + d;
        const float bb = (p <= 0.f || keep(seed, (uint32_t)o, p)) ? b[o] * scale : 0.f;
        v[n] = a[o] + bb;
        s += v[n];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int i = 0; i < n; ++i) q += (v[i] - mean) * (v[i] - mean);
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    n = 0;
    for (int d = lane; d < D; d += 64, ++n) out[row * D + d] = (v[n] - mean) * rstd * gamma[d] + beta[d];
    if (lane == 0) {
        stats[2 * row] = mean;
        stats[2 * row + 1] = rstd;
    }
}

// ds = gradient w.r.t. (a + dropout(b)); da = ds; db = mask/(1-p) * ds; dgamma/dbeta += (atomics)
__global__ __launch_bounds__(256) void mlc_add_ln_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                             const float* __restrict__ b, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, float p, uint32_t seed,
                                                             float* __restrict__ da, float* __restrict__ db,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float xh[16], g[16], s1 = 0.f, s2 = 0.f;
    int n = 0;
    for (int d = lane; d < D; d += 64, ++n) {
        const int64_t o = row * D + d;
        const bool k = p <= 0.f || keep(seed, (uint32_t)o, p);
        const float x = a[o] + (k ? b[o] * scale : 0.f);
        xh[n] = (x - mean) * rstd;
        const float go = dout[o];
        atomicAdd(&dgamma[d], go * xh[n]);
        atomicAdd(&dbeta[d], go);
        g[n] = go * gamma[d];
        s1 += g[n];
        s2 += g[n] * xh[n];
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    n = 0;
    for (int d = lane; d < D; d += 64, ++n) {
        const int64_t o = row * D + d;
        const float ds = rstd * (g[n] - s1 - xh[n] * s2);
        da[o] = ds;
        db[o] = (p <= 0.f || keep(seed, (uint32_t)o, p)) ? ds * scale : 0.f;
    }
}

// ---- h = relu(y + bias); hd = dropout(h) ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlc_bias_relu_drop_fwd_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                                     float p, uint32_t seed, float* __restrict__ h,
                                                                     float* __restrict__ hd, int64_t n, int N) {
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = fmaxf(y[i] + bias[i % N], 0.f);
        h[i] = v;
        hd[i] = (p <= 0.f || keep(seed, (uint32_t)i, p)) ? v * scale : 0.f;
    }
}
// dh = dhd * mask/(1-p) * (h > 0); dbias += column sums
__global__ __launch_bounds__(256) void mlc_relu_drop_bwd_kernel(const float* __restrict__ dhd, const float* __restrict__ h,
                                                                float p, uint32_t seed, float* __restrict__ dh,
                                                                float* __restrict__ dbias, int64_t n, int N) {
    const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (h[i] > 0.f && (p <= 0.f || keep(seed, (uint32_t)i, p))) ? dhd[i] * scale : 0.f;
        dh[i] = v;
        if (v != 0.f) atomicAdd(&dbias[i % N], v);
    }
}
// db[c] += sum_r dy[r][c]   (bias gradient of a Linear): one block per 64 columns x row chunk
__global__ __launch_bounds__(256) void mlc_colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, int64_t rows, int N) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N)
        for (int64_t r = (int64_t)blockIdx.y * 4 + ty; r < rows; r += (int64_t)gridDim.y * 4) s += dy[r * N + c];
    red[ty][threadIdx.x & 63] = s;
    __syncthreads();
    if (ty == 0 && c < N) atomicAdd(&db[c], (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// ---- pseudo-label cross-entropy over the heads (mlc_train.py:252-261): loss = mean_h mean_b CE(logits_h / T, target_h) ----
// logits [B][Tn] (head h owns columns off[h] .. off[h+1]); targets [H][B] int64.  ONE workgroup: thread t owns the
// (b, h) pairs t, t + 256, ... (B * H is a few thousand five-class rows), adds their terms in fp64 in that order, and the
// workgroup folds the 256 partial sums in a fixed order -- the loss is a function of the logits bit for bit, as the
// reference's mean over per-head CrossEntropyLoss values is (no float atomics).
__global__ __launch_bounds__(256) void mlc_ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ targets,
                                                     const int* __restrict__ off, int H, int B, int Tn, float inv_t,
                                                     float* __restrict__ loss, float* __restrict__ dlogits) {
    __shared__ double sh4[4];
    const float k = 1.f / ((float)B * (float)H);
    double a = 0.0;
    for (int i = threadIdx.x; i < B * H; i += 256) {
        const int b = i / H, h = i - b * H;
        const int c0 = off[h], c1 = off[h + 1];
        const float* row = logits + (int64_t)b * Tn;
        float m = -INFINITY;
        for (int c = c0; c < c1; ++c) m = fmaxf(m, row[c] * inv_t);
        float se = 0.f;
        for (int c = c0; c < c1; ++c) se += __expf(row[c] * inv_t - m);
        const float lse = m + __logf(se);
        const int t = (int)targets[(int64_t)h * B + b];
        a += (double)(k * (lse - row[c0 + t] * inv_t));
        for (int c = c0; c < c1; ++c)
            dlogits[(int64_t)b * Tn + c] = k * inv_t * (__expf(row[c] * inv_t - lse) - (c == c0 + t ? 1.f : 0.f));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) *loss += (float)(((sh4[0] + sh4[1]) + sh4[2]) + sh4[3]);
}

// ---- prototype heads backward: out[b][t] = <xn[b][tok(t)], W[t]> (+bias), xn = x or x/|x| ------------------------------
// grid = B; dx [B][S][D] (written), dW [Tn][D] += , dbias [Tn] += (nullable)
#define XR(ptr, s_) ROWP(ptr, s_, D)
// forward: out[b][t] = <xn[b][tok(t)], W[t]> + bias[t]  (sm3_token_heads with row strides, fp32)
__global__ __launch_bounds__(256) void mlc_heads_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ bias, const int* __restrict__ token_of,
                                                            int l2_norm, float* __restrict__ out, int S, int D, int Tn,
                                                            int64_t ss, int64_t ts) {
    __shared__ float inv_norm[kMaxS];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < kMaxS) inv_norm[tid] = 1.f;
    __syncthreads();
    if (l2_norm) {
        const int s = tid >> 5, part = tid & 31;
        float q = 0.f;
        if (s < S)
            for (int d = part; d < D; d += 32) q += XR(x, s)[d] * XR(x, s)[d];
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
            const float* xr = XR(x, tok);
            for (int d = part; d < D; d += 8) a += xr[d] * W[(int64_t)t * D + d];
        }
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        if (t < Tn && part == 0) out[(int64_t)b * Tn + t] = a * inv_norm[tok] + (bias ? bias[t] : 0.f);
    }
}

__global__ __launch_bounds__(256) void mlc_heads_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ x,
                                                            const float* __restrict__ W, const int* __restrict__ token_of,
                                                            int l2_norm, float* __restrict__ dx, float* __restrict__ dW,
                                                            float* __restrict__ dbias, int S, int D, int Tn, int64_t ss,
                                                            int64_t ts) {
    __shared__ float inv_norm[kMaxS], dots[kMaxS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* gl = dlogits + (int64_t)b * Tn;
    if (tid < kMaxS) {
        inv_norm[tid] = 1.f;
        dots[tid] = 0.f;
    }
    __syncthreads();
    if (l2_norm) {
        const int s = tid >> 5, part = tid & 31;
        float q = 0.f;
        if (s < S)
            for (int d = part; d < D; d += 32) q += XR(x, s)[d] * XR(x, s)[d];
        for (int o = 1; o < 32; o <<= 1) q += __shfl_xor(q, o, 64);
        if (s < S && part == 0) inv_norm[s] = 1.f / fmaxf(sqrtf(q), 1e-12f);
        __syncthreads();
    }
    // v[s][d] = sum_{t: tok(t) = s} dlogits[t] W[t][d]  -> gradient w.r.t. the (normalised) token
    for (int o = tid; o < S * D; o += 256) {
        const int s = o / D, d = o - s * D;
        float v = 0.f;
        for (int t = 0; t < Tn; ++t)
            if (token_of[t] == s) v += gl[t] * W[(int64_t)t * D + d];
        XR(dx, s)[d] = v;
    }
    for (int o = tid; o < Tn * D; o += 256) {  // dW[t][d] += dlogits[t] * xn[tok(t)][d]
        const int t = o / D, d = o - t * D, s = token_of[t];
        const float g = gl[t];
        if (g != 0.f) atomicAdd(&dW[o], g * XR(x, s)[d] * inv_norm[s]);
    }
    if (dbias && tid < Tn) atomicAdd(&dbias[tid], gl[tid]);
    if (l2_norm) {  // dx = inv * (v - xn (xn . v))
        __syncthreads();
        const int s = tid >> 5, part = tid & 31;
        float q = 0.f;
        if (s < S)
            for (int d = part; d < D; d += 32) q += XR(dx, s)[d] * XR(x, s)[d] * inv_norm[s];
        for (int o = 1; o < 32; o <<= 1) q += __shfl_xor(q, o, 64);
        if (s < S && part == 0) dots[s] = q;
        __syncthreads();
        for (int o = tid; o < S * D; o += 256) {
            const int s2 = o / D, d = o - s2 * D;
            XR(dx, s2)[d] = inv_norm[s2] * (XR(dx, s2)[d] - XR(x, s2)[d] * inv_norm[s2] * dots[s2]);
        }
    }
}

// ---- spherical k-means on the memory bank (mlc_train.py:146-177), K <= 8 ----------------------------------------------
// E step + accumulation of the M step: assign[n] = argmax_k <emb[n], cent[k]> (first maximum), sums[k] += emb[n], counts[k]++
__global__ __launch_bounds__(256) void mlc_kmeans_assign_kernel(const float* __restrict__ emb, const float* __restrict__ cent,
                                                                int64_t* __restrict__ assign, float* __restrict__ sums,
                                                                int* __restrict__ counts, int N, int D, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per embedding
    if (n >= N) return;
    const float* e = emb + (int64_t)n * D;
    float best = -INFINITY;
    int bk = 0;
    for (int k = 0; k < K; ++k) {
        float a = 0.f;
        for (int d = lane; d < D; d += 64) a += e[d] * cent[(int64_t)k * D + d];
        a = wave_sum(a);
        if (a > best) {
            best = a;
            bk = k;
        }
    }
    if (lane == 0) {
        assign[n] = bk;
        if (counts) atomicAdd(&counts[bk], 1);
    }
    if (sums)
        for (int d = lane; d < D; d += 64) atomicAdd(&sums[(int64_t)bk * D + d], e[d]);
}
// M step: centroid k = normalise(sums[k] / counts[k]) where counts[k] > 0 (empty clusters keep their centroid, then
// every centroid is L2-normalised: mlc_train.py:171-174)
__global__ __launch_bounds__(64) void mlc_kmeans_update_kernel(float* __restrict__ cent, const float* __restrict__ sums,
                                                               const int* __restrict__ counts, int D) {
    const int k = blockIdx.x, lane = threadIdx.x;
    const int c = counts[k];
    float q = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = c > 0 ? sums[(int64_t)k * D + d] / (float)c : cent[(int64_t)k * D + d];
        cent[(int64_t)k * D + d] = v;
        q += v * v;
    }
    q = wave_sum(q);
    const float inv = 1.f / fmaxf(sqrtf(q), 1e-12f);
    for (int d = lane; d < D; d += 64) cent[(int64_t)k * D + d] *= inv;
}

inline unsigned ew_grid(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (unsigned)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

#define MLC_CHECK_ATT()                                                                                          \
    if (!qkv || B <= 0 || S <= 0 || S > kMaxS || D <= 0 || nhead <= 0 || nhead > 8 || D % nhead || p < 0.f || p >= 1.f) \
        return SM3_EINVAL;

#define MLC_STRIDES() const int64_t ss = label_major ? 1 : S, ts = label_major ? B : 1

extern "C" int sm3_mlc_attention_fwd(const float* qkv, float* out, int B, int S, int D, int nhead, float p, uint32_t seed,
                                     int label_major, void* stream) {
    MLC_CHECK_ATT();
    if (!out) return SM3_EINVAL;
    MLC_STRIDES();
    hipLaunchKernelGGL(mlc_attention_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, out, S, D, nhead, p, seed,
                       ss, ts);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_attention_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int D, int nhead,
                                     float p, uint32_t seed, int label_major, void* stream) {
    MLC_CHECK_ATT();
    if (!dout || !dqkv) return SM3_EINVAL;
    MLC_STRIDES();
    hipLaunchKernelGGL(mlc_attention_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, qkv, dout, dqkv, S, D, nhead, p,
                       seed, ss, ts);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_add_ln_fwd(const float* a, const float* b, const float* gamma, const float* beta, float eps, float p,
                                  uint32_t seed, float* out, float* stats, int64_t rows, int D, void* stream) {
    if (!a || !b || !gamma || !beta || !out || !stats || rows <= 0 || D <= 0 || D > 1024 || p < 0.f || p >= 1.f) return SM3_EINVAL;
    hipLaunchKernelGGL(mlc_add_ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, gamma,
                       beta, eps, p, seed, out, stats, rows, D);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_add_ln_bwd(const float* dout, const float* a, const float* b, const float* stats, const float* gamma,
                                  float p, uint32_t seed, float* da, float* db, float* dgamma, float* dbeta, int64_t rows,
                                  int D, void* stream) {
    if (!dout || !a || !b || !stats || !gamma || !da || !db || !dgamma || !dbeta || rows <= 0 || D <= 0 || D > 1024 || p < 0.f ||
        p >= 1.f)
        return SM3_EINVAL;
    hipLaunchKernelGGL(mlc_add_ln_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dout, a, b,
                       stats, gamma, p, seed, da, db, dgamma, dbeta, rows, D);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_bias_relu_drop_fwd(const float* y, const float* bias, float p, uint32_t seed, float* h, float* hd,
                                          int64_t rows, int N, void* stream) {
    if (!y || !bias || !h || !hd || rows <= 0 || N <= 0 || p < 0.f || p >= 1.f) return SM3_EINVAL;
    hipLaunchKernelGGL(mlc_bias_relu_drop_fwd_kernel, dim3(ew_grid(rows * N)), dim3(256), 0, (hipStream_t)stream, y, bias, p,
                       seed, h, hd, rows * N, N);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_relu_drop_bwd(const float* dhd, const float* h, float p, uint32_t seed, float* dh, float* dbias,
                                     int64_t rows, int N, void* stream) {
    if (!dhd || !h || !dh || !dbias || rows <= 0 || N <= 0 || p < 0.f || p >= 1.f) return SM3_EINVAL;
    hipLaunchKernelGGL(mlc_relu_drop_bwd_kernel, dim3(ew_grid(rows * N)), dim3(256), 0, (hipStream_t)stream, dhd, h, p, seed, dh,
                       dbias, rows * N, N);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_colsum(const float* dy, float* db, int64_t rows, int N, void* stream) {
    if (!dy || !db || rows <= 0 || N <= 0) return SM3_EINVAL;
    int64_t gy = (rows + 63) / 64;
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(mlc_colsum_kernel, dim3((N + 63) / 64, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, dy, db, rows, N);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_ce(const float* logits, const int64_t* targets, const int* head_offsets, int H, int B, int Tn,
                          float temperature, float* loss, float* dlogits, void* stream) {
    if (!logits || !targets || !head_offsets || !loss || !dlogits || H <= 0 || B <= 0 || Tn <= 0 || temperature <= 0) return SM3_EINVAL;
    hipLaunchKernelGGL(mlc_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, targets, head_offsets,
                       H, B, Tn, 1.f / temperature, loss, dlogits);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_heads_fwd(const float* x, const float* W, const float* bias, const int* token_of, int l2_norm,
                                 float* out, int B, int S, int D, int Tn, int label_major, void* stream) {
    if (!x || !W || !token_of || !out || B <= 0 || S <= 0 || S > kMaxS || D <= 0 || Tn <= 0) return SM3_EINVAL;
    MLC_STRIDES();
    hipLaunchKernelGGL(mlc_heads_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, W, bias, token_of, l2_norm, out, S,
                       D, Tn, ss, ts);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_heads_bwd(const float* dlogits, const float* x, const float* W, const int* token_of, int l2_norm,
                                 float* dx, float* dW, float* dbias, int B, int S, int D, int Tn, int label_major,
                                 void* stream) {
    if (!dlogits || !x || !W || !token_of || !dx || !dW || B <= 0 || S <= 0 || S > kMaxS || D <= 0 || Tn <= 0 || Tn > 256)
        return SM3_EINVAL;
    MLC_STRIDES();
    hipLaunchKernelGGL(mlc_heads_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dlogits, x, W, token_of, l2_norm, dx,
                       dW, dbias, S, D, Tn, ss, ts);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_kmeans_assign(const float* emb, const float* centroids, int64_t* assign, float* sums, int* counts,
                                     int N, int D, int K, void* stream) {
    if (!emb || !centroids || !assign || N <= 0 || D <= 0 || K <= 0 || (sums != nullptr) != (counts != nullptr)) return SM3_EINVAL;
    hipLaunchKernelGGL(mlc_kmeans_assign_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, emb, centroids, assign,
                       sums, counts, N, D, K);
    SM3_CHECK_LAUNCH();
    return 0;
}
extern "C" int sm3_mlc_kmeans_update(float* centroids, const float* sums, const int* counts, int K, int D, void* stream) {
    if (!centroids || !sums || !counts || K <= 0 || D <= 0) return SM3_EINVAL;
    hipLaunchKernelGGL(mlc_kmeans_update_kernel, dim3(K), dim3(64), 0, (hipStream_t)stream, centroids, sums, counts, D);
    SM3_CHECK_LAUNCH();
    return 0;
}
