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

// one wave per row: zn = z / max(||z||, 1e-12)
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ z, int R, int D,
                                                             float* __restrict__ zn, float* __restrict__ inv_norm) {
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
                                                        float* __restrict__ loss, float* __restrict__ dlogits) {
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
    if (threadIdx.x == 0 && loss) atomicAdd(loss, weight * (lse - row[0]) / (float)R);
    if (dlogits) {
        const float k = weight / (float)R;
        float* drow = dlogits + (int64_t)blockIdx.x * Cc;
        for (int c = threadIdx.x; c < Cc; c += 256) drow[c] = k * (__expf(row[c] - lse) - (c == 0 ? 1.f : 0.f));
    }
}

// fused path, phase A: per row i, lse_i = log sum_{j != i} exp(s_ij), loss += w/R * (lse_i - s_ip)
__global__ __launch_bounds__(256) void fused_lse_kernel(const float* __restrict__ zn, int R, int D, float inv_t,
                                                        float weight, float* __restrict__ lse_out,
                                                        float* __restrict__ loss) {
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
        if (loss) atomicAdd(loss, weight * (lse - srow[p]) / (float)R);
    }
}

// fused path, phase B: coef_ij = w/(R T) * [ (P_ij - [j==p_i]) + (P_ji - [i==p_j]) ], then as logits_bwd
template <typename T>
__global__ __launch_bounds__(256) void fused_bwd_kernel(const float* __restrict__ zn, const float* __restrict__ inv_norm,
                                                        const float* __restrict__ lse, int R, int D, float inv_t,
                                                        float weight, const float* __restrict__ dz_scale,
                                                        T* __restrict__ dz) {
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
//   loss += weight/Rl * (-S_ip/T + log sum_{j != self} exp(S_ij/T));   S_ij <- dloss/dS_ij (in place; 0 at self)
__global__ __launch_bounds__(256) void ntxent_rect_kernel(float* __restrict__ S, int Rl, int Rg, int off, float inv_t,
                                                          float weight, const float* __restrict__ dz_scale,
                                                          float* __restrict__ loss) {
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
    if (threadIdx.x == 0) atomicAdd(loss, k * (lse - row[pos] * inv_t));
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

extern "C" int sm3_ce_label0(const float* logits, int R, int Cc, float weight, float* loss, float* dlogits,
                             void* stream) {
    if (!logits || R <= 0 || Cc <= 0) return SM3_EINVAL;
    hipLaunchKernelGGL(ce_label0_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, R, Cc, weight, loss,
                       dlogits);
    SM3_CHECK_LAUNCH();
    return 0;
}

static int ntxent_fused_impl(int dtype, const float* z, int R, int D, float temperature, float weight,
                             const float* dz_scale, float* workspace, float* loss, void* dz, void* stream) {
    if (!z || !workspace || !dz || R < 2 || (R & 1) || D <= 0 || temperature <= 0) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const size_t lds = (size_t)(D + R + 8) * 4;
    if (lds > 60000) return SM3_EINVAL;
    float* zn = workspace;                    // [R][D]
    float* inv_norm = workspace + (size_t)R * D;  // [R]
    float* lse = inv_norm + R;                // [R]
    hipStream_t st = (hipStream_t)stream;
    const float inv_t = 1.f / temperature;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, st, z, R, D, zn, inv_norm);
    SM3_CHECK_LAUNCH();
    hipLaunchKernelGGL(fused_lse_kernel, dim3(R), dim3(256), lds, st, zn, R, D, inv_t, weight, lse, loss);
    SM3_CHECK_LAUNCH();
#define SM3_NTX(T)                                                                                                       \
    hipLaunchKernelGGL(fused_bwd_kernel<T>, dim3(R), dim3(256), lds, st, zn, inv_norm, lse, R, D, inv_t, weight, dz_scale, \
                       (T*)dz)
    SM3_DISPATCH_DTYPE(dtype, SM3_NTX);
#undef SM3_NTX
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
                               const float* dz_scale, float* loss, void* stream) {
    if (!S || !loss || Rl < 2 || (Rl & 1) || Rg < Rl || self_offset < 0 || self_offset + Rl > Rg || temperature <= 0)
        return SM3_EINVAL;
    hipLaunchKernelGGL(ntxent_rect_kernel, dim3(Rl), dim3(256), 0, (hipStream_t)stream, S, Rl, Rg, self_offset,
                       1.f / temperature, weight, dz_scale, loss);
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
