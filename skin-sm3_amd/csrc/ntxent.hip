// NT-Xent contrastive loss kernels (fp32 VALU, LDS-staged rows, wave shuffle reductions).
// R = 2B rows of D-dim projections (D = 128).  The arithmetic is ~0.001 GFLOP per pair, so these
// kernels are latency/launch bound; what matters is that the reference's ~15 ATen kernels with
// three boolean-mask gathers collapse to 2-3 launches with no host synchronisation.
//
// Reference call sites replaced: F.normalize(dim=1) + torch.matmul(features, features.T) +
// eye-mask / label-mask select + cat + /temperature (src/models/simclr.py:62-88 and :294-320) and
// nn.CrossEntropyLoss (tools/backbone_train.py:531 applied at :101-102,119-120).
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ void store_out(T* p, float v);
template <>
__device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
template <>
__device__ __forceinline__ void store_out<f16_t>(f16_t* p, float v) { p->v = f32_to_f16(v); }

// *loss += the fixed-order sum of the per-row loss terms of one NT-Xent / cross-entropy call (one workgroup): the loss of a
// step is then a function of the parameters bit for bit, like the reference's single CrossEntropyLoss reduction
// (tools/backbone_train.py:101-102,119-121,531).  The calls of a step are enqueued on ONE stream, so the plain
// read-modify-write of *loss happens in program order.
__global__ __launch_bounds__(256) void ordered_loss_add_kernel(const float* __restrict__ terms, int n, float* __restrict__ loss) {
    __shared__ double sh4[4];
    const double s = block256_ordered_sum(terms, n, sh4);
    if (threadIdx.x == 0) *loss += (float)s;
}

// The NT-Xent terms of a step (derm, clinic, two cross-modal: same R, D, temperature) as ONE launch per phase, blockIdx.y =
// term (sm3_ntxent_fused_batch, round 6): 12 dependent launches of 64 workgroups on the main stream -- where neither lane has
// other work -- become 3 of 256.  Per-term operands travel by value.
struct NtxBatch {
    const float* z[4];
    float* ws[4];   // per term: zn [R][D] | inv_norm [R] | lse [R] | row_terms [R]
    void* dz[4];
    float weight[4];
    int n;
};

// one wave per row: zn = z / max(||z||, 1e-12)
__device__ __forceinline__ void normalize_rows_body(const float* __restrict__ z, int R, int D, float* __restrict__ zn,
                                                    float* __restrict__ inv_norm);
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ z, int R, int D,
                                                             float* __restrict__ zn, float* __restrict__ inv_norm) {
    normalize_rows_body(z, R, D, zn, inv_norm);
}
__global__ __launch_bounds__(256) void normalize_rows_batch_kernel(const NtxBatch b, int R, int D) {
    const int t = blockIdx.y;
    normalize_rows_body(b.z[t], R, D, b.ws[t], b.ws[t] + (size_t)R * D);
}
__device__ __forceinline__ void normalize_rows_body(const float* __restrict__ z, int R, int D, float* __restrict__ zn,
                                                    float* __restrict__ inv_norm) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = z[(int64_t)row * D + d];
        s += v * v;
    }
    s = wave_sum(s);
    const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int d = lane; d < D; d += 64) zn[(int64_t)row * D + d] = z[(int64_t)row * D + d] * inv;
    if (lane == 0) inv_norm[row] = inv;
}

__device__ __forceinline__ int logit_col(int i, int j, int p) {
    // column of S[i][j] in the reference layout: positive first, then ascending j without {i, p}
    return j == p ? 0 : 1 + j - (j > i ? 1 : 0) - (j > p ? 1 : 0);
}

// 32x32 tile of S = Zn Zn^T per workgroup (256 threads, 2x2 outputs each)
__global__ __launch_bounds__(256) void sim_logits_kernel(const float* __restrict__ zn, int R, int D, float inv_t,
                                                         float* __restrict__ logits) {
    __shared__ float sa[32][33], sb[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    for (int d0 = 0; d0 < D; d0 += 32) {
        for (int e = threadIdx.x; e < 32 * 32; e += 256) {
            const int r = e >> 5, c = e & 31;
            sa[r][c] = (i0 + r < R && d0 + c < D) ? zn[(int64_t)(i0 + r) * D + d0 + c] : 0.f;
            sb[r][c] = (j0 + r < R && d0 + c < D) ? zn[(int64_t)(j0 + r) * D + d0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < 32; ++c) {
            const float a0 = sa[ty][c], a1 = sa[ty + 16][c], b0 = sb[tx][c], b1 = sb[tx + 16][c];
            acc[0][0] += a0 * b0;
            acc[0][1] += a0 * b1;
            acc[1][0] += a1 * b0;
            acc[1][1] += a1 * b1;
        }
        __syncthreads();
    }
    const int half = R >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
            if (i < R && j < R && i != j) {
                const int p = (i + half) % R;
                logits[(int64_t)i * (R - 1) + logit_col(i, j, p)] = acc[a][b] * inv_t;
            }
        }
}

// dz_i = inv_norm_i * (v - zn_i (zn_i . v)),  v = sum_j coef_ij zn_j ; one workgroup per row i.
// coef_ij = (G[i][col(i,j)] + G[j][col(j,i)]) / T  from the reference-layout logits gradient G.
template <typename T>
__global__ __launch_bounds__(256) void logits_bwd_kernel(const float* __restrict__ G, const float* __restrict__ zn,
                                                         const float* __restrict__ inv_norm, int R, int D,
                                                         float inv_t, T* __restrict__ dz) {
    extern __shared__ float coef[];  // [R] + reduction scratch [4]
    const int i = blockIdx.x, half = R >> 1, pi = (i + half) % R;
    for (int j = threadIdx.x; j < R; j += 256) {
        float c = 0.f;
        if (j != i) {
            const int pj = (j + half) % R;
            c = (G[(int64_t)i * (R - 1) + logit_col(i, j, pi)] + G[(int64_t)j * (R - 1) + logit_col(j, i, pj)]) * inv_t;
        }
        coef[j] = c;
    }
    __syncthreads();
    float* red = coef + R;
    float dot_part = 0.f;
    // each thread owns dims d = tid, tid+256, ... (D <= 1024 supported by the loop below)
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        for (int j = 0; j < R; ++j) v += coef[j] * zn[(int64_t)j * D + d];
        dot_part += v * zn[(int64_t)i * D + d];
    }
    dot_part = wave_sum(dot_part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot_part;
    __syncthreads();
    const float dot = red[0] + red[1] + red[2] + red[3];
    const float inv = inv_norm[i];
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        for (int j = 0; j < R; ++j) v += coef[j] * zn[(int64_t)j * D + d];
        store_out<T>(dz + (int64_t)i * D + d, inv * (v - zn[(int64_t)i * D + d] * dot));
    }
}

// cross entropy against label 0, one workgroup per row
__global__ __launch_bounds__(256) void ce_label0_kernel(const float* __restrict__ logits, int R, int Cc, float weight,
                                                        float* __restrict__ row_terms, float* __restrict__ dlogits) {
    __shared__ float red[8];
    const float* row = logits + (int64_t)blockIdx.x * Cc;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < Cc; c += 256) mx = fmaxf(mx, row[c]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int c = threadIdx.x; c < Cc; c += 256) s += __expf(row[c] - mx);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = s;
    __syncthreads();
    s = red[4] + red[5] + red[6] + red[7];
    const float lse = mx + __logf(s);
    if (threadIdx.x == 0 && row_terms) row_terms[blockIdx.x] = weight * (lse - row[0]) / (float)R;
    if (dlogits) {
        const float k = weight / (float)R;
        float* drow = dlogits + (int64_t)blockIdx.x * Cc;
        for (int c = threadIdx.x; c < Cc; c += 256) drow[c] = k * (__expf(row[c] - lse) - (c == 0 ? 1.f : 0.f));
    }
}

// fused path, phase A: per row i, lse_i = log sum_{j != i} exp(s_ij), row_terms[i] = w/R * (lse_i - s_ip)
__global__ __launch_bounds__(256) void fused_lse_kernel(const float* __restrict__ zn, int R, int D, float inv_t,
                                                        float weight, float* __restrict__ lse_out,
                                                        float* __restrict__ row_terms) {
    extern __shared__ float sm[];  // zi[D] + s[R] + red[8]
    float* zi = sm;
    float* srow = sm + D;
    float* red = srow + R;
    const int i = blockIdx.x, half = R >> 1, p = (i + half) % R;
    for (int d = threadIdx.x; d < D; d += 256) zi[d] = zn[(int64_t)i * D + d];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = wv; j < R; j += 4) {
        float a = 0.f;
        for (int d = lane; d < D; d += 64) a += zi[d] * zn[(int64_t)j * D + d];
        a = wave_sum(a);
        if (lane == 0) srow[j] = a * inv_t;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < R; j += 256)
        if (j != i) mx = fmaxf(mx, srow[j]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f;
    for (int j = threadIdx.x; j < R; j += 256)
        if (j != i) s += __expf(srow[j] - mx);
    s = wave_sum(s);
    if (lane == 0) red[4 + wv] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lse = mx + __logf(red[4] + red[5] + red[6] + red[7]);
        lse_out[i] = lse;
        row_terms[i] = weight * (lse - srow[p]) / (float)R;
    }
}

// fused path, phase B: coef_ij = w/(R T) * [ (P_ij - [j==p_i]) + (P_ji - [i==p_j]) ], then as logits_bwd
template <typename T>
__global__ __launch_bounds__(256) void fused_bwd_kernel(const float* __restrict__ zn, const float* __restrict__ inv_norm,
                                                        const float* __restrict__ lse, int R, int D, float inv_t,
                                                        float weight, const float* __restrict__ dz_scale,
                                                        const float* __restrict__ row_terms, float* __restrict__ loss,
                                                        T* __restrict__ dz) {
    __shared__ double sh4[4];
    if (loss && blockIdx.x == 0) {  // phase A has finished (stream order): its row terms, summed in a fixed order
        const double s = block256_ordered_sum(row_terms, R, sh4);
        if (threadIdx.x == 0) *loss += (float)s;
    }
    extern __shared__ float sm[];  // zi[D] + coef[R] + red[4]
    float* zi = sm;
    float* coef = sm + D;
    float* red = coef + R;
    const int i = blockIdx.x, half = R >> 1, p = (i + half) % R;
    for (int d = threadIdx.x; d < D; d += 256) zi[d] = zn[(int64_t)i * D + d];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // dz_scale: the dynamic loss scale of the fp16 mode (device scalar; GradScaler.scale(loss), backbone_train.py:125)
    const float k = weight * (dz_scale ? dz_scale[0] : 1.f) * inv_t / (float)R;
    const float lse_i = lse[i];
    for (int j = wv; j < R; j += 4) {
        float a = 0.f;
        for (int d = lane; d < D; d += 64) a += zi[d] * zn[(int64_t)j * D + d];
        a = wave_sum(a) * inv_t;
        if (lane == 0) {
            float c = 0.f;
            if (j != i) {
                const float pos = (j == p) ? 1.f : 0.f;  // p_j == i  <=>  j == p_i  (R even)
                c = k * ((__expf(a - lse_i) - pos) + (__expf(a - lse[j]) - pos));
            }
            coef[j] = c;
        }
    }
    __syncthreads();
    float dot_part = 0.f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        for (int j = 0; j < R; ++j) v += coef[j] * zn[(int64_t)j * D + d];
        dot_part += v * zi[d];
    }
    dot_part = wave_sum(dot_part);
    if (lane == 0) red[wv] = dot_part;
    __syncthreads();
    const float dot = red[0] + red[1] + red[2] + red[3];
    const float inv = inv_norm[i];
    for (int d = threadIdx.x; d < D; d += 256) {
        float v = 0.f;
        for (int j = 0; j < R; ++j) v += coef[j] * zn[(int64_t)j * D + d];
        store_out<T>(dz + (int64_t)i * D + d, inv * (v - zi[d] * dot));
    }
}

// ---- global negatives (data parallel): local anchors x the all-gathered candidate set -------------------------------
// S [Rl][Rg]: cosines of this rank's Rl = 2B normalised projections (rows) against the Rg = world * Rl gathered ones
// (columns; rank r's block is columns r*Rl .., its own rows sit at columns off + i).  Row i's positive is the other view
// of the same pair, column off + (i + B) % Rl; its own column is excluded.  One workgroup per row:
//   row_terms[i] = weight/Rl * (-S_ip/T + log sum_{j != self} exp(S_ij/T));   S_ij <- dloss/dS_ij (in place; 0 at self)
__global__ __launch_bounds__(256) void ntxent_rect_kernel(float* __restrict__ S, int Rl, int Rg, int off, float inv_t,
                                                          float weight, const float* __restrict__ dz_scale,
                                                          float* __restrict__ row_terms) {
    __shared__ float red[8];
    const int i = blockIdx.x, self = off + i, pos = off + (i + (Rl >> 1)) % Rl;
    float* row = S + (int64_t)i * Rg;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < Rg; j += 256)
        if (j != self) mx = fmaxf(mx, row[j] * inv_t);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float se = 0.f;
    for (int j = threadIdx.x; j < Rg; j += 256)
        if (j != self) se += __expf(row[j] * inv_t - mx);
    se = wave_sum(se);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = se;
    __syncthreads();
    se = (red[4] + red[5]) + (red[6] + red[7]);
    const float lse = mx + __logf(se);
    const float k = weight / (float)Rl;
    if (threadIdx.x == 0) row_terms[i] = k * (lse - row[pos] * inv_t);
    __syncthreads();  // (row[pos] is read above, rewritten below)
    const float g = k * inv_t * (dz_scale ? dz_scale[0] : 1.f);
    for (int j = threadIdx.x; j < Rg; j += 256) {
        float d = 0.f;
        if (j != self) d = g * (__expf(row[j] * inv_t - lse) - (j == pos ? 1.f : 0.f));
        row[j] = d;
    }
}

// gradient through zn = z / max(||z||, 1e-12): dz_i = inv_norm_i * (v_i - zn_i (zn_i . v_i)), v = d(loss)/d(zn) = a + b
template <typename T>
__global__ __launch_bounds__(256) void normalize_rows_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                 const float* __restrict__ zn,
                                                                 const float* __restrict__ inv_norm, int R, int D,
                                                                 T* __restrict__ dz) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) {
        const int64_t o = (int64_t)row * D + d;
        dot += (a[o] + (b ? b[o] : 0.f)) * zn[o];
    }
    dot = wave_sum(dot);
    const float inv = inv_norm[row];
    for (int d = lane; d < D; d += 64) {
        const int64_t o = (int64_t)row * D + d;
        store_out<T>(dz + o, inv * ((a[o] + (b ? b[o] : 0.f)) - zn[o] * dot));
    }
}

}  // namespace

extern "C" int sm3_ntxent_logits(const float* z, int R, int D, float temperature, float* zn, float* inv_norm,
                                 float* logits, void* stream) {
    if (!z || !zn || !inv_norm || !logits || R < 2 || (R & 1) || D <= 0 || temperature <= 0) return SM3_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, z, R, D, zn, inv_norm);
    SM3_CHECK_LAUNCH();
    hipLaunchKernelGGL(sim_logits_kernel, dim3((R + 31) / 32, (R + 31) / 32), dim3(256), 0, st, zn, R, D,
                       1.f / temperature, logits);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_ntxent_logits_bwd(int dtype, const float* dlogits, const float* zn, const float* inv_norm, int R,
                                     int D, float temperature, void* dz, void* stream) {
    if (!dlogits || !zn || !inv_norm || !dz || R < 2 || (R & 1) || D <= 0 || temperature <= 0) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const size_t lds = (size_t)(R + 4) * 4;
    if (lds > 60000) return SM3_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define SM3_LB(T) \
    hipLaunchKernelGGL(logits_bwd_kernel<T>, dim3(R), dim3(256), lds, st, dlogits, zn, inv_norm, R, D, 1.f / temperature, (T*)dz)
    SM3_DISPATCH_DTYPE(dtype, SM3_LB);
#undef SM3_LB
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_ce_label0(const float* logits, int R, int Cc, float weight, float* row_terms, float* loss,
                             float* dlogits, void* stream) {
    if (!logits || R <= 0 || Cc <= 0 || (loss && !row_terms)) return SM3_EINVAL;
    hipLaunchKernelGGL(ce_label0_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, R, Cc, weight,
                       loss ? row_terms : nullptr, dlogits);
    SM3_CHECK_LAUNCH();
    if (loss) {
        hipLaunchKernelGGL(ordered_loss_add_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_terms, R, loss);
        SM3_CHECK_LAUNCH();
    }
    return 0;
}

namespace {

// ---- tiled fused path (D % 4 == 0, D <= 128): 8 anchor rows per workgroup against 64-candidate tiles -------------------
// The row-per-workgroup kernels above spend a wave reduction per similarity (~95 us per launch at R = 512, D = 128, on
// the step's forward/backward turnaround where neither execution lane has other work).  Here a workgroup keeps 8 anchor
// rows and walks the candidates in tiles of 64 rows staged in LDS (row pitch D + 4 floats: the 16-byte reads of 16
// different rows fall on different banks); thread (il = t / 32, jj = t % 32) owns S[i0 + il][j0 + jj], S[..][j0 + jj + 32]
// (float4 dot products).  LSE: online max / sum per thread, folded over the 32 lanes of a row.  BWD: the tile's
// coefficients go to LDS and v[il][:] += coef[il][tile] * Zn[tile][:] re-uses the staged tile (thread (il, ch) owns the
// 16-byte chunks ch, ch + 32 of the row).  The next tile's global loads are issued before the current tile's arithmetic.
constexpr int NTX_TI = 8, NTX_TJ = 64, NTX_CP = NTX_TJ + 4;

template <bool BWD, typename T, int NV>
__device__ __forceinline__ void ntxent_tile_body(const float* __restrict__ zn, const float* __restrict__ inv_norm,
                                                 float* __restrict__ lse, int R, int D, float inv_t, float weight,
                                                 const float* __restrict__ dz_scale, float* __restrict__ row_terms,
                                                 T* __restrict__ dz);

template <bool BWD, typename T, int NV>
__global__ __launch_bounds__(256) void ntxent_tile_kernel(const float* __restrict__ zn, const float* __restrict__ inv_norm,
                                                          float* __restrict__ lse, int R, int D, float inv_t, float weight,
                                                          const float* __restrict__ dz_scale, float* __restrict__ row_terms,
                                                          float* __restrict__ loss, T* __restrict__ dz) {
    if constexpr (BWD) {
        __shared__ double sh4[4];
        if (loss && blockIdx.x == 0) {  // the LSE launch has finished (stream order): its row terms, summed in a fixed order
            const double s = block256_ordered_sum(row_terms, R, sh4);
            if (threadIdx.x == 0) *loss += (float)s;
        }
    }
    ntxent_tile_body<BWD, T, NV>(zn, inv_norm, lse, R, D, inv_t, weight, dz_scale, row_terms, dz);
}

// blockIdx.y = term.  The loss: ONE workgroup adds the terms' sums to *loss one after the other, term 0 first -- the float
// additions the per-term calls make, in their order (bit-identical to n calls of sm3_ntxent_fused).
template <bool BWD, typename T, int NV>
__global__ __launch_bounds__(256) void ntxent_tile_batch_kernel(const NtxBatch b, int R, int D, float inv_t,
                                                                const float* __restrict__ dz_scale, float* __restrict__ loss) {
    const int t = blockIdx.y;
    float* const zn = b.ws[t];
    float* const inv_norm = zn + (size_t)R * D;
    float* const lse = inv_norm + R;
    float* const row_terms = lse + R;
    if constexpr (BWD) {
        __shared__ double sh4[4];
        if (loss && blockIdx.x == 0 && t == 0) {
            for (int u = 0; u < b.n; ++u) {
                const double s = block256_ordered_sum(b.ws[u] + (size_t)R * D + 2 * (size_t)R, R, sh4);
                if (threadIdx.x == 0) *loss += (float)s;
            }
        }
    }
    ntxent_tile_body<BWD, T, NV>(zn, inv_norm, lse, R, D, inv_t, b.weight[t], dz_scale, row_terms, (T*)b.dz[t]);
}

template <bool BWD, typename T, int NV>
__device__ __forceinline__ void ntxent_tile_body(const float* __restrict__ zn, const float* __restrict__ inv_norm,
                                                 float* __restrict__ lse, int R, int D, float inv_t, float weight,
                                                 const float* __restrict__ dz_scale, float* __restrict__ row_terms,
                                                 T* __restrict__ dz) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D4 = D >> 2, DP = D + 4;
    float* zi = sm;                      // [TI][DP]
    float* zj = zi + NTX_TI * DP;        // [TJ][DP]
    float* coef = zj + NTX_TJ * DP;      // [TI][CP]   (BWD)
    float* lse_t = coef + NTX_TI * NTX_CP;  // [TJ]    (BWD)
    const int t = threadIdx.x, il = t >> 5, jj = t & 31;
    const int i0 = blockIdx.x * NTX_TI, i = i0 + il, half = R >> 1;
    const bool row_ok = i < R;
    const int p = row_ok ? (i + half) % R : -1;
    const int ntiles = (R + NTX_TJ - 1) / NTX_TJ;

    float4 pre[NV];
    float pre_lse = 0.f;
    auto fetch = [&](int j0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = t + k * 256, r = idx / D4, c = idx - r * D4;
            pre[k] = (r < NTX_TJ && j0 + r < R) ? *reinterpret_cast<const float4*>(zn + (int64_t)(j0 + r) * D + c * 4)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (BWD && t < NTX_TJ) pre_lse = (j0 + t < R) ? lse[j0 + t] : 0.f;
    };
    auto commit = [&]() {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int idx = t + k * 256, r = idx / D4, c = idx - r * D4;
            if (r < NTX_TJ) *reinterpret_cast<float4*>(zj + r * DP + c * 4) = pre[k];
        }
        if (BWD && t < NTX_TJ) lse_t[t] = pre_lse;
    };
    fetch(0);
    for (int idx = t; idx < NTX_TI * D4; idx += 256) {
        const int r = idx / D4, c = idx - r * D4;
        *reinterpret_cast<float4*>(zi + r * DP + c * 4) =
            (i0 + r < R) ? *reinterpret_cast<const float4*>(zn + (int64_t)(i0 + r) * D + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    commit();
    __syncthreads();

    // dz_scale: the dynamic loss scale of the fp16 mode (device scalar; GradScaler.scale(loss), backbone_train.py:125)
    const float k = BWD ? weight * (dz_scale ? dz_scale[0] : 1.f) * inv_t / (float)R : 0.f;
    const float lse_i = (BWD && row_ok) ? lse[i] : 0.f;
    float run_m = -INFINITY, run_s = 0.f, s_pos = 0.f;
    constexpr int NCH = (NV * 256 / NTX_TJ + 31) / 32;  // 16-byte chunks of a row per thread (BWD accumulators)
    float4 acc[NCH];
#pragma unroll
    for (int m = 0; m < NCH; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int tile = 0; tile < ntiles; ++tile) {
        const int j0 = tile * NTX_TJ;
        if (tile + 1 < ntiles) fetch(j0 + NTX_TJ);
        float a0 = 0.f, a1 = 0.f;
        const float* zr = zi + il * DP;
        const float* zb0 = zj + jj * DP;
        const float* zb1 = zj + (jj + 32) * DP;
        for (int d = 0; d < D; d += 4) {
            const float4 x = *reinterpret_cast<const float4*>(zr + d);
            const float4 b0 = *reinterpret_cast<const float4*>(zb0 + d);
            const float4 b1 = *reinterpret_cast<const float4*>(zb1 + d);
            a0 += x.x * b0.x + x.y * b0.y + x.z * b0.z + x.w * b0.w;
            a1 += x.x * b1.x + x.y * b1.y + x.z * b1.z + x.w * b1.w;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int j = j0 + jj + 32 * h;
            const float s = (h ? a1 : a0) * inv_t;
            const bool valid = row_ok && j < R && j != i;
            if constexpr (BWD) {
                float c = 0.f;
                if (valid) {
                    const float pos = (j == p) ? 1.f : 0.f;  // p_j == i  <=>  j == p_i  (R even)
                    c = k * ((__expf(s - lse_i) - pos) + (__expf(s - lse_t[jj + 32 * h]) - pos));
                }
                coef[il * NTX_CP + jj + 32 * h] = c;
            } else if (valid) {
                if (j == p) s_pos = s;
                if (s > run_m) {
                    run_s = run_s * __expf(run_m - s) + 1.f;
                    run_m = s;
                } else {
                    run_s += __expf(s - run_m);
                }
            }
        }
        if constexpr (BWD) {
            __syncthreads();  // the tile's coefficients are in LDS
            const float* cr = coef + il * NTX_CP;
            for (int j = 0; j < NTX_TJ; j += 4) {
                const float4 c4 = *reinterpret_cast<const float4*>(cr + j);
#pragma unroll
                for (int m = 0; m < NCH; ++m) {
                    const int ch = jj + 32 * m;
                    if (ch < D4) {
                        const float* col = zj + j * DP + ch * 4;
                        const float4 r0 = *reinterpret_cast<const float4*>(col);
                        const float4 r1 = *reinterpret_cast<const float4*>(col + DP);
                        const float4 r2 = *reinterpret_cast<const float4*>(col + 2 * DP);
                        const float4 r3 = *reinterpret_cast<const float4*>(col + 3 * DP);
                        acc[m].x += c4.x * r0.x + c4.y * r1.x + c4.z * r2.x + c4.w * r3.x;
                        acc[m].y += c4.x * r0.y + c4.y * r1.y + c4.z * r2.y + c4.w * r3.y;
                        acc[m].z += c4.x * r0.z + c4.y * r1.z + c4.z * r2.z + c4.w * r3.z;
                        acc[m].w += c4.x * r0.w + c4.y * r1.w + c4.z * r2.w + c4.w * r3.w;
                    }
                }
            }
        }
        __syncthreads();  // everyone is done with this tile (and its coefficients)
        if (tile + 1 < ntiles) {
            commit();
            __syncthreads();
        }
    }

    if constexpr (BWD) {
        float dot = 0.f;
#pragma unroll
        for (int m = 0; m < NCH; ++m) {
            const int ch = jj + 32 * m;
            if (ch < D4) {
                const float4 x = *reinterpret_cast<const float4*>(zi + il * DP + ch * 4);
                dot += acc[m].x * x.x + acc[m].y * x.y + acc[m].z * x.z + acc[m].w * x.w;
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);  // over the 32 lanes of the row
        if (row_ok) {
            const float inv = inv_norm[i];
#pragma unroll
            for (int m = 0; m < NCH; ++m) {
                const int ch = jj + 32 * m;
                if (ch < D4) {
                    const float4 x = *reinterpret_cast<const float4*>(zi + il * DP + ch * 4);
                    T* o = dz + (int64_t)i * D + ch * 4;
                    store_out<T>(o + 0, inv * (acc[m].x - x.x * dot));
                    store_out<T>(o + 1, inv * (acc[m].y - x.y * dot));
                    store_out<T>(o + 2, inv * (acc[m].z - x.z * dot));
                    store_out<T>(o + 3, inv * (acc[m].w - x.w * dot));
                }
            }
        }
    } else {
        float mx = run_m;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float s = (run_m == -INFINITY) ? 0.f : run_s * __expf(run_m - mx);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);
            s_pos += __shfl_xor(s_pos, o, 64);
        }
        if (jj == 0 && row_ok) {
            const float l = mx + __logf(s);
            lse[i] = l;
            row_terms[i] = weight * (l - s_pos) / (float)R;
        }
    }
}

template <typename T, int NV>
static void launch_ntxent_tiles(const float* zn, const float* inv_norm, float* lse, float* row_terms, int R, int D, float inv_t,
                                float weight, const float* dz_scale, float* loss, T* dz, hipStream_t st) {
    const size_t lds = (size_t)((NTX_TI + NTX_TJ) * (D + 4) + NTX_TI * NTX_CP + NTX_TJ) * 4;
    const dim3 grid((R + NTX_TI - 1) / NTX_TI);
    hipLaunchKernelGGL((ntxent_tile_kernel<false, T, NV>), grid, dim3(256), lds, st, zn, inv_norm, lse, R, D, inv_t, weight,
                       nullptr, row_terms, (float*)nullptr, (T*)nullptr);
    hipLaunchKernelGGL((ntxent_tile_kernel<true, T, NV>), grid, dim3(256), lds, st, zn, inv_norm, lse, R, D, inv_t, weight,
                       dz_scale, row_terms, loss, dz);
}

}  // namespace

static int ntxent_fused_impl(int dtype, const float* z, int R, int D, float temperature, float weight,
                             const float* dz_scale, float* workspace, float* loss, void* dz, void* stream) {
    if (!z || !workspace || !dz || R < 2 || (R & 1) || D <= 0 || temperature <= 0) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const size_t lds = (size_t)(D + R + 8) * 4;
    if (lds > 60000) return SM3_EINVAL;
    float* zn = workspace;                    // [R][D]
    float* inv_norm = workspace + (size_t)R * D;  // [R]
    float* lse = inv_norm + R;                // [R]
    float* row_terms = lse + R;               // [R]  per-row loss terms, summed in a fixed order by the backward launch
    hipStream_t st = (hipStream_t)stream;
    const float inv_t = 1.f / temperature;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, z, R, D, zn, inv_norm);
    SM3_CHECK_LAUNCH();
    if (D % 4 == 0 && D <= 128 && ((uintptr_t)workspace & 15) == 0) {  // tiled kernels (8 anchors x 64-candidate tiles, 40 KB of LDS)
#define SM3_NTX_TILE(T) launch_ntxent_tiles<T, 8>(zn, inv_norm, lse, row_terms, R, D, inv_t, weight, dz_scale, loss, (T*)dz, st)
        SM3_DISPATCH_DTYPE(dtype, SM3_NTX_TILE);
#undef SM3_NTX_TILE
        SM3_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(fused_lse_kernel, dim3(R), dim3(256), lds, st, zn, R, D, inv_t, weight, lse, row_terms);
    SM3_CHECK_LAUNCH();
#define SM3_NTX(T)                                                                                                       \
    hipLaunchKernelGGL(fused_bwd_kernel<T>, dim3(R), dim3(256), lds, st, zn, inv_norm, lse, R, D, inv_t, weight, dz_scale, \
                       row_terms, loss, (T*)dz)
    SM3_DISPATCH_DTYPE(dtype, SM3_NTX);
#undef SM3_NTX
    SM3_CHECK_LAUNCH();
    return 0;
}

// n <= 4 NT-Xent terms of equal shape in three launches (normalise, LSE, backward), blockIdx.y = term; every per-term result and
// the loss are bit-identical to n calls of sm3_ntxent_fused(_scaled) in term order.  Tiled path only (D % 4 == 0, D <= 128).
extern "C" int sm3_ntxent_fused_batch(int dtype, int nterms, const float* const* z, int R, int D, float temperature,
                                      const float* weights, const float* dz_scale, float* workspace, float* loss,
                                      void* const* dz, void* stream) {
    if (!z || !weights || !workspace || !dz || nterms < 1 || nterms > 4 || R < 2 || (R & 1) || D <= 0 || temperature <= 0)
        return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    if (D % 4 || D > 128 || ((uintptr_t)workspace & 15)) return SM3_EINVAL;
    const size_t per = ((size_t)R * D + 3 * (size_t)R + 3) & ~(size_t)3;  // a term's block, padded: every zn stays 16-byte aligned
    NtxBatch b;
    b.n = nterms;
    for (int t = 0; t < 4; ++t) {
        const int u = t < nterms ? t : 0;
        if (!z[u] || !dz[u]) return SM3_EINVAL;
        b.z[t] = z[u];
        b.ws[t] = workspace + (size_t)u * per;
        b.dz[t] = dz[u];
        b.weight[t] = weights[u];
    }
    hipStream_t st = (hipStream_t)stream;
    const float inv_t = 1.f / temperature;
    hipLaunchKernelGGL(normalize_rows_batch_kernel, dim3((R + 3) / 4, nterms), dim3(256), 0, st, b, R, D);
    SM3_CHECK_LAUNCH();
    const size_t lds = (size_t)((NTX_TI + NTX_TJ) * (D + 4) + NTX_TI * NTX_CP + NTX_TJ) * 4;
    const dim3 grid((R + NTX_TI - 1) / NTX_TI, nterms);
#define SM3_NTX_BATCH(T)                                                                                                 \
    hipLaunchKernelGGL((ntxent_tile_batch_kernel<false, T, 8>), grid, dim3(256), lds, st, b, R, D, inv_t,                  \
                       (const float*)nullptr, (float*)nullptr);                                                           \
    hipLaunchKernelGGL((ntxent_tile_batch_kernel<true, T, 8>), grid, dim3(256), lds, st, b, R, D, inv_t, dz_scale, loss)
    SM3_DISPATCH_DTYPE(dtype, SM3_NTX_BATCH);
#undef SM3_NTX_BATCH
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_ntxent_fused(int dtype, const float* z, int R, int D, float temperature, float weight,
                                float* workspace, float* loss, void* dz, void* stream) {
    return ntxent_fused_impl(dtype, z, R, D, temperature, weight, nullptr, workspace, loss, dz, stream);
}

extern "C" int sm3_ntxent_fused_scaled(int dtype, const float* z, int R, int D, float temperature, float weight,
                                       const float* dz_scale, float* workspace, float* loss, void* dz, void* stream) {
    if (!dz_scale) return SM3_EINVAL;
    return ntxent_fused_impl(dtype, z, R, D, temperature, weight, dz_scale, workspace, loss, dz, stream);
}

extern "C" int sm3_normalize_rows(const float* z, int R, int D, float* zn, float* inv_norm, void* stream) {
    if (!z || !zn || !inv_norm || R <= 0 || D <= 0) return SM3_EINVAL;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, R, D, zn, inv_norm);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_ntxent_rect(float* S, int Rl, int Rg, int self_offset, float temperature, float weight,
                               const float* dz_scale, float* row_terms, float* loss, void* stream) {
    if (!S || !loss || !row_terms || Rl < 2 || (Rl & 1) || Rg < Rl || self_offset < 0 || self_offset + Rl > Rg ||
        temperature <= 0)
        return SM3_EINVAL;
    hipLaunchKernelGGL(ntxent_rect_kernel, dim3(Rl), dim3(256), 0, (hipStream_t)stream, S, Rl, Rg, self_offset,
                       1.f / temperature, weight, dz_scale, row_terms);
    SM3_CHECK_LAUNCH();
    hipLaunchKernelGGL(ordered_loss_add_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_terms, Rl, loss);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_normalize_rows_bwd(int dtype, const float* dzn_a, const float* dzn_b, const float* zn,
                                      const float* inv_norm, int R, int D, void* dz, void* stream) {
    if (!dzn_a || !zn || !inv_norm || !dz || R <= 0 || D <= 0) return SM3_EINVAL;
    hipStream_t st = (hipStream_t)stream;
#define SM3_NB(T) \
    hipLaunchKernelGGL(normalize_rows_bwd_kernel<T>, dim3((R + 3) / 4), dim3(256), 0, st, dzn_a, dzn_b, zn, inv_norm, R, D, (T*)dz)
    SM3_DISPATCH_DTYPE(dtype, SM3_NB);
#undef SM3_NB
    SM3_CHECK_LAUNCH();
    return 0;
}
