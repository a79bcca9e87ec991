// BatchNorm (2d / 1d as rows x C, channels contiguous): statistics finalisation, fused
// normalise(+residual)(+ReLU) apply, and the two-phase backward.  All HBM-bound: 16-byte vector
// accesses, a thread keeps one channel vector for its whole row walk so per-channel parameters
// are loaded once.  Statistics are combined in fp64 so E[x^2]-E[x]^2 loses nothing.
//
// Reference call sites replaced: nn.BatchNorm2d / BatchNorm1d in train mode
// (src/models/resnet.py:145-149,211,261; src/models/simclr.py:20-26), the bn->relu and
// bn->add->relu chains of Bottleneck.forward (resnet.py:154-174), SyncBatchNorm's two
// cross-rank reductions (tools/backbone_train.py:510) -- the reductions themselves are done by
// the host between sm3_bn_stats_reduce and sm3_bn_finalize / sm3_bn_bwd_apply.
#include "common.h"
#include <cstdlib>

namespace {

// 2-D work decomposition shared by the row-walk kernels.
struct RowWalk {
    int tbx, tby;   // threads along channel vectors / rows
    int gx, gy;     // blocks along channel vectors / row groups
};
static RowWalk make_walk(int64_t rows, int cvecs, int max_gy, int views = 1) {
    RowWalk w;
    if (cvecs >= 256) {
        w.tbx = 256;
    } else {
        w.tbx = cvecs;
    }
    w.tby = 256 / w.tbx;
    if (w.tby < 1) w.tby = 1;
    w.gx = (cvecs + w.tbx - 1) / w.tbx;
    int64_t gy = (rows + (int64_t)w.tby * 8 - 1) / ((int64_t)w.tby * 8);  // >= 8 rows per thread
    // At most 768 blocks = 3 per CU (env SM3_BN_GRID_CAP).  With 4-8 rows in flight per thread that already
    // saturates HBM (scratch/stream_bench.hip: 1024 >= 2048 blocks), and it leaves 20 of a CU's 32 wave slots to
    // the other execution lane's convolution: 2048 -> 768 was worth 1.2 % of the two-lane step in round 1.  Round 4
    // re-swept it against the one-stage convolutions -- 2048 / 1024 / 768 / 512 / 384 / 256 blocks: 4 263 / 4 302 /
    // 4 309 / 4 340 / 4 358* / 4 330* pairs/s (* another box, where 768 gave 4 370 and 512 gave 4 380): 512 is +0.2 ...
    // +0.7 % in the two-lane step but runs the pass itself 20 % slower when it has the chip alone (5.2 -> 4.2 TB/s, as
    // under a SyncBN exchange), so 768 stays.
    static const int grid_cap = getenv("SM3_BN_GRID_CAP") ? atoi(getenv("SM3_BN_GRID_CAP")) : 768;
    const int64_t cap = grid_cap / (w.gx * views) > 0 ? grid_cap / (w.gx * views) : 1;  // the cap is per launch, views included
    if (gy > cap) gy = cap;
    if (gy > max_gy) gy = max_gy;
    if (gy < 1) gy = 1;
    w.gy = (int)gy;
    return w;
}

// Stage A: grid (ceil(2C/64), G).  256 threads = 64 columns x 4 row lanes; a block sums its share of the
// partial rows in fp64 and writes one row of the [G][2C] fp64 workspace.  Stage B sums the G rows.  Two tiny
// launches, deterministic (no atomics); rows are read as 256-byte coalesced segments.
// "views" (blockIdx.z): the same kernel over V independent row ranges laid out back to back -- the two views of a
// branch go through the convolutions as one batch but keep separate BatchNorm statistics (simclr.py:58-59).
__global__ __launch_bounds__(256) void bn_stats_reduce_a(const float* __restrict__ partials, int rows, int C,
                                                         double* __restrict__ ws) {
    __shared__ double red[4][64];
    partials += (long)blockIdx.z * rows * 2 * C;
    ws += (long)blockIdx.z * gridDim.y * 2 * C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;  // in [0, 2C): stat-major within a partial row
    double a = 0.0;
    if (col < 2 * C) {
        // four independent chains: the loads of a chain of dependent fp64 adds would otherwise go out one at a time
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const int step = gridDim.y * 4;
        int r = blockIdx.y * 4 + ty;
        for (; r + 3 * step < rows; r += 4 * step) {
            a += (double)partials[(long)r * 2 * C + col];
            a1 += (double)partials[(long)(r + step) * 2 * C + col];
            a2 += (double)partials[(long)(r + 2 * step) * 2 * C + col];
            a3 += (double)partials[(long)(r + 3 * step) * 2 * C + col];
        }
        for (; r < rows; r += step) a += (double)partials[(long)r * 2 * C + col];
        a = (a + a1) + (a2 + a3);
    }
    red[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && col < 2 * C) ws[(long)blockIdx.y * 2 * C + col] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

// Sum over the G workspace rows of one column by the 4 lanes of a quad: lane q takes rows q, q+4, ... in four
// independent chains (G = 64: four loads in flight, four rounds), then the quad folds.  Fixed association, so every
// caller gets the same bits; every lane of the quad returns the total.
__device__ __forceinline__ double quad_sum_groups(const double* __restrict__ ws, long stride, long col, int G, int q) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int g = q;
    for (; g + 12 < G; g += 16) {
        a0 += ws[(long)g * stride + col];
        a1 += ws[(long)(g + 4) * stride + col];
        a2 += ws[(long)(g + 8) * stride + col];
        a3 += ws[(long)(g + 12) * stride + col];
    }
    for (; g < G; g += 4) a0 += ws[(long)g * stride + col];
    double a = (a0 + a1) + (a2 + a3);
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    return a;
}

__global__ void bn_stats_reduce_b(const double* __restrict__ ws, int G, int C, double* __restrict__ sums) {
    ws += (long)blockIdx.z * G * 2 * C;
    sums += (long)blockIdx.z * 2 * C;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, q = t & 3;
    const int col = t >> 2;
    const double a = quad_sum_groups(ws, 2L * C, col < 2 * C ? col : 0, G, q);
    if (col < 2 * C && q == 0) sums[col] = a;
}

// 16 lanes per channel: lane = (view v in {0,1}) x (statistic in {sum, sum of squares}) x (quad lane q); the four
// group sums of a channel (2 views x 2 statistics) run side by side instead of one after the other -- this kernel is
// pure latency (a chain of dependent L2 loads per sum) and sits on the critical path of every BatchNorm.  views > 2
// loop over pairs.  Running statistics are still updated view 0 first, then view 1, by one lane, as in the reference's
// sequential encoder(x1); encoder(x2).
// One channel's finalize, run by a 16-lane group: lane = (view-of-pair vl, statistic, quad lane q).  ws: [views][groups][2C]
// fp64 partial sums (groups == 1: the sums themselves).  Shared by bn_finalize_kernel and bn_stats_finalize_kernel: the two
// produce the same bits.  `lane16` = lane index inside the group, c < C or the group idles (it still takes part in shuffles).
struct BnFinalizeArgs {
    double count;
    int C;
    const float *gamma, *beta;
    float eps, momentum;
    float *running_mean, *running_var;
    float *scale, *shift, *save_mean, *save_invstd;
};
__device__ __forceinline__ void bn_finalize_channel(const double* __restrict__ sums, int groups, int views, int c, int lane16,
                                                    const BnFinalizeArgs& a) {
    const int C = a.C;
    const int q = lane16 & 3, stat = (lane16 >> 2) & 1, vl = (lane16 >> 3) & 1;
    const int cc = c < C ? c : 0;
    float rm = 0.f, rv = 0.f;
    const bool owner = (c < C) && (lane16 == 0);
    if (owner) {
        rm = a.running_mean ? a.running_mean[c] : 0.f;
        rv = a.running_var ? a.running_var[c] : 0.f;
    }
    for (int v0 = 0; v0 < views; v0 += 2) {
        const int v = v0 + vl;
        const bool vok = v < views;
        // groups > 1: stage B of the statistics reduction folded in here (same association as bn_stats_reduce_b)
        const double* sv = sums + (long)(vok ? v : v0) * groups * 2 * C;
        const double mine = quad_sum_groups(sv, 2L * C, (long)stat * C + cc, groups, q);
        // lanes (vl, stat, q=0) of the channel's 16-lane group hold the four sums; bring them to lane 0 of the group
        const double s2_same_view = __shfl_down(mine, 4, 64);   // from (vl, stat=1)
        const double s1_v1 = __shfl_down(mine, 8, 64);          // from (vl=1, stat=0)
        const double s2_v1 = __shfl_down(mine, 12, 64);         // from (vl=1, stat=1)
        if (!owner) continue;
        for (int k = 0; k < 2 && v0 + k < views; ++k) {
            const double s1 = k ? s1_v1 : mine, s2 = k ? s2_v1 : s2_same_view;
            const int vv = v0 + k;
            const double mean = s1 / a.count;
            double var = s2 / a.count - mean * mean;
            if (var < 0) var = 0;
            const double invstd = 1.0 / sqrt(var + (double)a.eps);
            const float g = a.gamma ? a.gamma[c] : 1.f, b = a.beta ? a.beta[c] : 0.f;
            a.scale[(long)vv * C + c] = (float)(g * invstd);
            a.shift[(long)vv * C + c] = (float)((double)b - mean * (double)g * invstd);
            if (a.save_mean) a.save_mean[(long)vv * C + c] = (float)mean;
            if (a.save_invstd) a.save_invstd[(long)vv * C + c] = (float)invstd;
            rm = (1.f - a.momentum) * rm + a.momentum * (float)mean;
            const double unbiased = var * (a.count / fmax(a.count - 1.0, 1.0));
            rv = (1.f - a.momentum) * rv + a.momentum * (float)unbiased;
        }
    }
    if (owner) {
        if (a.running_mean) a.running_mean[c] = rm;
        if (a.running_var) a.running_var[c] = rv;
    }
}

// 16 lanes per channel: lane = (view v in {0,1}) x (statistic in {sum, sum of squares}) x (quad lane q); the four
// group sums of a channel (2 views x 2 statistics) run side by side instead of one after the other -- this kernel is
// pure latency (a chain of dependent L2 loads per sum) and sits on the critical path of every BatchNorm.  views > 2
// loop over pairs.  Running statistics are still updated view 0 first, then view 1, by one lane, as in the reference's
// sequential encoder(x1); encoder(x2).
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int groups, int views, int64_t* nbt, BnFinalizeArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && nbt) nbt[0] += views;
    bn_finalize_channel(sums, groups, views, t >> 4, t & 15, a);
}

// Statistics reduction AND finalize of one BatchNorm in ONE launch (single rank; VERDICT r3 item 5: the two dependent
// 8 us launches per BatchNorm were 2.7 ms of a step).  Grid (ceil(C / 32), G, views): a block owns 32 channels x both
// statistics of G-th row group of one view -- stage A exactly as bn_stats_reduce_a (same row walk, same association), one
// fp64 row of the [views][G][2C] workspace per block.  Then an arrival ticket per channel block: the block that draws the
// last ticket finalizes those 32 channels for every view with bn_finalize_channel -- the same bits as the two-launch form.
// Hand-off (MI355X_MICROARCH.md, Valid forms): plain stores -> the storing wave's s_waitcnt vmcnt(0) -> workgroup barrier ->
// lane 0: agent-scope release fence, s_waitcnt vmcnt(0), relaxed agent-scope ticket add; last arriver: agent-scope acquire
// fence, s_waitcnt vmcnt(0), workgroup barrier, plain loads.  The last arriver zeroes the ticket word for the next launch
// (tickets live in a zero-initialised per-stream buffer).
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ partials, int rows, int views,
                                                                double* __restrict__ ws_all, unsigned* __restrict__ tickets,
                                                                int64_t* nbt, BnFinalizeArgs a) {
    __shared__ double red[4][64];
    __shared__ int s_last;
    const int C = a.C, G = gridDim.y, v = blockIdx.z;
    const float* part = partials + (long)v * rows * 2 * C;
    double* ws = ws_all + ((long)v * G + blockIdx.y) * 2 * C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int ch = blockIdx.x * 32 + (tx & 31);
    const int col = (tx >> 5) * C + ch;  // stat-major within a partial row
    double acc = 0.0;
    if (ch < C) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const int step = G * 4;
        int r = blockIdx.y * 4 + ty;
        for (; r + 3 * step < rows; r += 4 * step) {
            acc += (double)part[(long)r * 2 * C + col];
            a1 += (double)part[(long)(r + step) * 2 * C + col];
            a2 += (double)part[(long)(r + 2 * step) * 2 * C + col];
            a3 += (double)part[(long)(r + 3 * step) * 2 * C + col];
        }
        for (; r < rows; r += step) acc += (double)part[(long)r * 2 * C + col];
        acc = (acc + a1) + (a2 + a3);
    }
    red[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && ch < C) ws[col] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the storing wave's stores are acknowledged ...
    __syncthreads();                                   // ... before lane 0 publishes them
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned total = (unsigned)(G * views);
        const unsigned t = __hip_atomic_fetch_add(&tickets[blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == total - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(&tickets[blockIdx.x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
            if (blockIdx.x == 0 && nbt) nbt[0] += views;
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    // 32 channels x 16 lanes = 512 lane slots: two rounds of the 256 threads
#pragma unroll
    for (int half = 0; half < 2; ++half)
        bn_finalize_channel(ws_all, G, views, blockIdx.x * 32 + half * 16 + (threadIdx.x >> 4), threadIdx.x & 15, a);
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               int C, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - rm[c] * g * invstd;
}

// The three row-walk kernels below keep U rows in flight per thread (all loads issued before the first use) and
// mark their accesses non-temporal: every byte is touched once.  U=1 -> 5.05, U=8+nt -> 5.97 TB/s on a cold
// 1.2 GB working set (scratch/stream_bench.hip).
template <typename T, bool OUT_F32, int U, bool NT>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const T* __restrict__ res,
                                                     const float* __restrict__ res_scale,
                                                     const float* __restrict__ res_shift, int relu,
                                                     void* __restrict__ y, uint8_t* __restrict__ mask, int64_t rows,
                                                     int C, int tbx, int tby, float* __restrict__ colsum) {
    constexpr int E = ElemTraits<T>::kPer16B;
    __shared__ float scol[256 * 8];  // colsum only: per-thread column sums, folded over ty
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    const bool active = !(cv * E >= C || ty >= tby);
    if (!active && !colsum) return;
    // colsum (nullable): [views][gridDim.y][C] f32, the block's column sums of the STORED (rounded) outputs -- the first
    // moment of a convolution input that BatchNorm by linearity needs (linbn.hip sums the rows in a fixed order)
    float cs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) cs[e] = 0.f;
    if (active) {
    {  // view blockIdx.z: its own row range and scale/shift
        const int64_t vo = (int64_t)blockIdx.z * rows * C;
        x += vo;
        if (res) res += vo;
        y = OUT_F32 ? (void*)(reinterpret_cast<float*>(y) + vo) : (void*)(reinterpret_cast<T*>(y) + vo);
        if (mask) mask += (int64_t)blockIdx.z * rows * (C / E);
        scale += (int64_t)blockIdx.z * C;
        shift += (int64_t)blockIdx.z * C;
        if (res_scale) {
            res_scale += (int64_t)blockIdx.z * C;
            res_shift += (int64_t)blockIdx.z * C;
        }
    }
    // res_scale: the residual is itself a pre-BatchNorm tensor (the downsample branch, resnet.py:166-167): its
    // normalisation is applied here, y = relu(x*scale + shift + res*res_scale + res_shift), and the separate
    // bn_act pass over the downsample output (one read + one write of a block-output-sized tensor) disappears
    float sc[E], sh[E], rs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        sc[e] = scale[cv * E + e];
        sh[e] = shift[cv * E + e];
        rs[e] = 1.f;
        if (res_scale) {
            rs[e] = res_scale[cv * E + e];
            sh[e] += res_shift[cv * E + e];
        }
    }
    const int64_t rstep = (int64_t)gridDim.y * tby;
    auto finish = [&](int64_t r, const uint4& xu, const uint4& ru) {
        const int64_t off = r * C + (int64_t)cv * E;
        float v[E];
        unpack16<T>(xu, v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (res) {
            float q[E];
            unpack16<T>(ru, q);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += q[e] * rs[e];
        }
        if (relu) {
            if (mask) {  // one bit per element (y > 0), one byte per 16-byte vector: what backward needs of y
                unsigned m = 0;
#pragma unroll
                for (int e = 0; e < E; ++e) m |= (v[e] > 0.f ? 1u : 0u) << e;
                mask[r * (C / E) + cv] = (uint8_t)m;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = relu_f32(v[e]);
        }
        if constexpr (OUT_F32) {
            float* yo = reinterpret_cast<float*>(y) + off;
#pragma unroll
            for (int e = 0; e < E; e += 4)
                *reinterpret_cast<float4*>(yo + e) = make_float4(v[e], v[e + 1], v[e + 2], v[e + 3]);
        } else {
            const uint4 pk = pack16<T>(v);
            stg16<NT>(reinterpret_cast<T*>(y) + off, pk);
            if (colsum) {
                float q[E];
                unpack16<T>(pk, q);
#pragma unroll
                for (int e = 0; e < E; ++e) cs[e] += q[e];
            }
        }
    };
    int64_t r = (int64_t)blockIdx.y * tby + ty;
    for (; r + (U - 1) * rstep < rows; r += U * rstep) {
        uint4 xu[U], ru[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (r + u * rstep) * C + (int64_t)cv * E;
            xu[u] = ldg16<NT>(x + off);
            ru[u] = res ? ldg16<NT>(res + off) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) finish(r + u * rstep, xu[u], ru[u]);
    }
    for (; r < rows; r += rstep) {
        const int64_t off = r * C + (int64_t)cv * E;
        finish(r, ldg16<false>(x + off), res ? ldg16<false>(res + off) : make_uint4(0, 0, 0, 0));
    }
    }  // active
    if (colsum) {  // kernel-uniform: every thread of the block gets here
        if constexpr (E <= 8) {
#pragma unroll
            for (int e = 0; e < E; ++e) scol[threadIdx.x * E + e] = cs[e];
        }
        __syncthreads();
        if (active && ty == 0) {
            for (int j = 1; j < tby; ++j)
#pragma unroll
                for (int e = 0; e < E; ++e) cs[e] += scol[(j * tbx + tx) * E + e];
            float* o = colsum + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * C + (int64_t)cv * E;
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = cs[e];
        }
    }
}

// Input of a strided 1x1 convolution as a dense operand: y[n, oy, ox, :] = x[n, oy*stride, ox*stride, :] (y nullable) and the
// per-block column sums of those rows, as bn_act_kernel's colsum (same thread layout, same view-independent row cut).
template <typename T, int U>
__global__ __launch_bounds__(256) void subsample_colsum_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                               float* __restrict__ colsum, int64_t rows, int C, int Hs,
                                                               int Ws, int H, int W, int stride, FastDiv div_hw,
                                                               FastDiv div_w, int tbx, int tby) {
    constexpr int E = ElemTraits<T>::kPer16B;
    __shared__ float scol[256 * 8];
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    const bool active = !(cv * E >= C || ty >= tby);
    float cs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) cs[e] = 0.f;
    if (active) {
        const int64_t r_lo = (int64_t)blockIdx.z * rows;  // view blockIdx.z: its own range of OUTPUT rows
        const int64_t rstep = (int64_t)gridDim.y * tby;
        auto src_row = [&](int64_t r) {  // output row (n, oy, ox) -> input row
            const uint32_t g = (uint32_t)(r_lo + r);
            const uint32_t n = fdiv(g, div_hw), rem = g - n * (uint32_t)(Hs * Ws);
            const uint32_t oy = fdiv(rem, div_w), ox = rem - oy * (uint32_t)Ws;
            return ((int64_t)n * H + (int64_t)oy * stride) * W + (int64_t)ox * stride;
        };
        int64_t r = (int64_t)blockIdx.y * tby + ty;
        for (; r + (U - 1) * rstep < rows; r += U * rstep) {
            uint4 xu[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xu[u] = ldg16<false>(x + src_row(r + u * rstep) * C + (int64_t)cv * E);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (y) stg16<true>(y + (r_lo + r + u * rstep) * C + (int64_t)cv * E, xu[u]);
                float q[E];
                unpack16<T>(xu[u], q);
#pragma unroll
                for (int e = 0; e < E; ++e) cs[e] += q[e];
            }
        }
        for (; r < rows; r += rstep) {
            const uint4 xu = ldg16<false>(x + src_row(r) * C + (int64_t)cv * E);
            if (y) stg16<true>(y + (r_lo + r) * C + (int64_t)cv * E, xu);
            float q[E];
            unpack16<T>(xu, q);
#pragma unroll
            for (int e = 0; e < E; ++e) cs[e] += q[e];
        }
    }
    if constexpr (E <= 8) {
#pragma unroll
        for (int e = 0; e < E; ++e) scol[threadIdx.x * E + e] = cs[e];
    }
    __syncthreads();
    if (active && ty == 0) {
        for (int j = 1; j < tby; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) cs[e] += scol[(j * tbx + tx) * E + e];
        float* o = colsum + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * C + (int64_t)cv * E;
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = cs[e];
    }
}

// phase 1 of backward: relu mask, optional dz write-back, partial sums of dz and dz*xhat
template <typename T, int U, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                            const uint8_t* __restrict__ mask,
                                                            const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, T* __restrict__ dz,
                                                            int64_t rows, int C, float* __restrict__ partials,
                                                            int tbx, int tby) {
    constexpr int E = ElemTraits<T>::kPer16B;
    __shared__ float sred[256 * 2 * 8];
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    const bool active = (cv * E < C) && (ty < tby);
    {  // view blockIdx.z
        const int64_t vo = (int64_t)blockIdx.z * rows * C;
        dy += vo;
        if (x) {
            x += vo;
            mean += (int64_t)blockIdx.z * C;
            invstd += (int64_t)blockIdx.z * C;
        }
        if (y) y += vo;
        if (dz) dz += vo;
        if (mask) mask += (int64_t)blockIdx.z * rows * (C / E);
        partials += (int64_t)blockIdx.z * gridDim.y * 2 * C;
    }
    float s1[E], s2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s1[e] = s2[e] = 0.f;
    if (active) {
        // x == nullptr: mask and sum(dz) only, the sum(dz * xhat) slot stays 0 (a BatchNorm whose backward goes by
        // linearity never reads its input: linbn.hip)
        float mu[E], is[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            mu[e] = x ? mean[cv * E + e] : 0.f;
            is[e] = x ? invstd[cv * E + e] : 0.f;
        }
        const int64_t rstep = (int64_t)gridDim.y * tby;
        auto finish = [&](int64_t r, const uint4& gu, const uint4& xu, const uint4& yu, unsigned m) {
            const int64_t off = r * C + (int64_t)cv * E;
            float g[E], xv[E];
            unpack16<T>(gu, g);
            unpack16<T>(xu, xv);
            if (mask) {
#pragma unroll
                for (int e = 0; e < E; ++e) g[e] = keep_if_bit(g[e], m, e);
                if (dz) stg16<NT>(dz + off, pack16<T>(g));
            } else if (y) {
                float yv[E];
                unpack16<T>(yu, yv);
#pragma unroll
                for (int e = 0; e < E; ++e) g[e] = yv[e] > 0.f ? g[e] : 0.f;
                if (dz) stg16<NT>(dz + off, pack16<T>(g));
            } else if (dz && dz != dy) {
                stg16<NT>(dz + off, pack16<T>(g));
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                s1[e] += g[e];
                s2[e] += g[e] * (xv[e] - mu[e]) * is[e];
            }
        };
        int64_t r = (int64_t)blockIdx.y * tby + ty;
        for (; r + (U - 1) * rstep < rows; r += U * rstep) {
            uint4 gu[U], xu[U], yu[U];
            unsigned m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t rr = r + u * rstep, off = rr * C + (int64_t)cv * E;
                gu[u] = ldg16<NT>(dy + off);
                xu[u] = x ? ldg16<NT>(x + off) : make_uint4(0, 0, 0, 0);
                m[u] = mask ? mask[rr * (C / E) + cv] : 0u;
                yu[u] = (!mask && y) ? ldg16<NT>(y + off) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) finish(r + u * rstep, gu[u], xu[u], yu[u], m[u]);
        }
        for (; r < rows; r += rstep) {
            const int64_t off = r * C + (int64_t)cv * E;
            finish(r, ldg16<false>(dy + off), x ? ldg16<false>(x + off) : make_uint4(0, 0, 0, 0),
                   (!mask && y) ? ldg16<false>(y + off) : make_uint4(0, 0, 0, 0), mask ? mask[r * (C / E) + cv] : 0u);
        }
    }
    // reduce over ty within the block
    float* mine = sred + threadIdx.x * 2 * E;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        mine[e] = s1[e];
        mine[E + e] = s2[e];
    }
    __syncthreads();
    if (ty == 0 && cv * E < C) {
        for (int j = 1; j < tby; ++j) {
            const float* o = sred + (j * tbx + tx) * 2 * E;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                s1[e] += o[e];
                s2[e] += o[E + e];
            }
        }
        float* p1 = partials + ((int64_t)blockIdx.y * 2 + 0) * C + (int64_t)cv * E;
        float* p2 = partials + ((int64_t)blockIdx.y * 2 + 1) * C + (int64_t)cv * E;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            p1[e] = s1[e];
            p2[e] = s2[e];
        }
    }
}

template <typename T, int U, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ x,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const double* __restrict__ gsums, double inv_count,
                                                           const double* __restrict__ lsums, float* dgamma,
                                                           float* dbeta, T* __restrict__ dx, int64_t rows, int C,
                                                           int tbx, int tby) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    if (cv * E >= C || ty >= tby) return;
    {  // view blockIdx.z
        const int64_t vo = (int64_t)blockIdx.z * rows * C;
        dz += vo;
        x += vo;
        dx += vo;
        mean += (int64_t)blockIdx.z * C;
        invstd += (int64_t)blockIdx.z * C;
        gsums += (int64_t)blockIdx.z * 2 * C;
        if (lsums) lsums += (int64_t)blockIdx.z * 2 * C;
    }
    // dx = g*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) = k0*(dz - k1) - (x - mu)*q
    float mu[E], k0[E], k1[E], q[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int c = cv * E + e;
        mu[e] = mean[c];
        const float is = invstd[c];
        const float g = gamma ? gamma[c] : 1.f;
        k0[e] = g * is;
        k1[e] = (float)(gsums[c] * inv_count);              // (no fp64 divide per thread: 16 of them cost more
        q[e] = k0[e] * is * (float)(gsums[C + c] * inv_count);  //  than the rows a thread walks)
    }
    if (blockIdx.y == 0 && blockIdx.z == 0 && ty == 0 && lsums) {
        // parameter gradients from the LOCAL sums, once per channel: ONE thread adds up the views of the launch in view order
        // and issues ONE add per parameter (a fixed order: a step's gradients are a function of its inputs).  The add stays
        // an atomic because two launches may meet on a parameter from two streams (SM3_VIEW_LANES=1).  All loads first, then
        // the atomics: interleaved, each load would wait for the atomic before it (possible aliasing) -- 4 us per launch.
        float db[E], dg[E];
#pragma unroll
        for (int e = 0; e < E; ++e) db[e] = dg[e] = 0.f;
        for (int v = 0; v < (int)gridDim.z; ++v) {
            const double* l = lsums + (int64_t)v * 2 * C + cv * E;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                db[e] += (float)l[e];
                dg[e] += (float)l[C + e];
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (dbeta) atomicAdd(&dbeta[cv * E + e], db[e]);
            if (dgamma) atomicAdd(&dgamma[cv * E + e], dg[e]);
        }
    }
    const int64_t rstep = (int64_t)gridDim.y * tby;
    auto finish = [&](int64_t r, const uint4& gu, const uint4& xu) {
        float g[E], xv[E];
        unpack16<T>(gu, g);
        unpack16<T>(xu, xv);
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = k0[e] * (g[e] - k1[e]) - (xv[e] - mu[e]) * q[e];
        stg16<NT>(dx + r * C + (int64_t)cv * E, pack16<T>(g));
    };
    int64_t r = (int64_t)blockIdx.y * tby + ty;
    for (; r + (U - 1) * rstep < rows; r += U * rstep) {
        uint4 gu[U], xu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (r + u * rstep) * C + (int64_t)cv * E;
            gu[u] = ldg16<NT>(dz + off);
            xu[u] = ldg16<NT>(x + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) finish(r + u * rstep, gu[u], xu[u]);
    }
    for (; r < rows; r += rstep) {
        const int64_t off = r * C + (int64_t)cv * E;
        finish(r, ldg16<false>(dz + off), ldg16<false>(x + off));
    }
}

// Two BatchNorms that received the SAME output gradient: the bn3 / downsample-BN pair of a Bottleneck's join
// (out = relu(bn3(conv3) + bn_d(conv_d)), resnet.py:164-172).  One pass reads dz once and writes both input
// gradients; each BatchNorm's arithmetic is exactly bn_bwd_apply_kernel's.
struct BnApplySide {
    const void* x;
    const float *mean, *invstd, *gamma;
    const double *gsums, *lsums;
    float *dgamma, *dbeta;
    void* dx;
};
template <typename T, int U, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply2_kernel(const T* __restrict__ dz, BnApplySide a, BnApplySide b,
                                                            double inv_count, int64_t rows, int C, int tbx, int tby) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    if (cv * E >= C || ty >= tby) return;
    const int64_t vo = (int64_t)blockIdx.z * rows * C;
    dz += vo;
    const T* xa = reinterpret_cast<const T*>(a.x) + vo;
    const T* xb = reinterpret_cast<const T*>(b.x) + vo;
    T* dxa = reinterpret_cast<T*>(a.dx) + vo;
    T* dxb = reinterpret_cast<T*>(b.dx) + vo;
    float mua[E], k0a[E], k1a[E], qa[E], mub[E], k0b[E], k1b[E], qb[E];
    auto coeffs = [&](const BnApplySide& s, float* mu, float* k0, float* k1, float* q) {
        const int64_t vc = (int64_t)blockIdx.z * C, v2 = (int64_t)blockIdx.z * 2 * C;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = cv * E + e;
            mu[e] = s.mean[vc + c];
            const float is = s.invstd[vc + c];
            const float g = s.gamma ? s.gamma[c] : 1.f;
            k0[e] = g * is;
            k1[e] = (float)(s.gsums[v2 + c] * inv_count);
            q[e] = k0[e] * is * (float)(s.gsums[v2 + C + c] * inv_count);
        }
    };
    coeffs(a, mua, k0a, k1a, qa);
    coeffs(b, mub, k0b, k1b, qb);
    if (blockIdx.y == 0 && blockIdx.z == 0 && ty == 0) {  // one thread per channel, views in order, loads before atomics
        auto param_grads = [&](const BnApplySide& s) {   // (see bn_bwd_apply_kernel)
            if (!s.lsums) return;
            float db[E], dg[E];
#pragma unroll
            for (int e = 0; e < E; ++e) db[e] = dg[e] = 0.f;
            for (int v = 0; v < (int)gridDim.z; ++v) {
                const double* l = s.lsums + (int64_t)v * 2 * C + cv * E;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    db[e] += (float)l[e];
                    dg[e] += (float)l[C + e];
                }
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (s.dbeta) atomicAdd(&s.dbeta[cv * E + e], db[e]);
                if (s.dgamma) atomicAdd(&s.dgamma[cv * E + e], dg[e]);
            }
        };
        param_grads(a);
        param_grads(b);
    }
    const int64_t rstep = (int64_t)gridDim.y * tby;
    auto finish = [&](int64_t r, const uint4& gu, const uint4& xau, const uint4& xbu) {
        float g[E], xv[E], o[E];
        unpack16<T>(gu, g);
        unpack16<T>(xau, xv);
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = k0a[e] * (g[e] - k1a[e]) - (xv[e] - mua[e]) * qa[e];
        stg16<NT>(dxa + r * C + (int64_t)cv * E, pack16<T>(o));
        unpack16<T>(xbu, xv);
#pragma unroll
        for (int e = 0; e < E; ++e) o[e] = k0b[e] * (g[e] - k1b[e]) - (xv[e] - mub[e]) * qb[e];
        stg16<NT>(dxb + r * C + (int64_t)cv * E, pack16<T>(o));
    };
    int64_t r = (int64_t)blockIdx.y * tby + ty;
    for (; r + (U - 1) * rstep < rows; r += U * rstep) {
        uint4 gu[U], xau[U], xbu[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (r + u * rstep) * C + (int64_t)cv * E;
            gu[u] = ldg16<NT>(dz + off);
            xau[u] = ldg16<NT>(xa + off);
            xbu[u] = ldg16<NT>(xb + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) finish(r + u * rstep, gu[u], xau[u], xbu[u]);
    }
    for (; r < rows; r += rstep) {
        const int64_t off = r * C + (int64_t)cv * E;
        finish(r, ldg16<false>(dz + off), ldg16<false>(xa + off), ldg16<false>(xb + off));
    }
}

// Row-walk tuning knobs (environment, read once): SM3_BN_UNROLL[_ACT|_RED|_APP] in {1,2,4,8}, SM3_BN_NT in {0,1}.
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
static int bn_unroll(int kind) {  // 0 apply-forward, 1 backward reduce, 2 backward apply
    static const int u[3] = {env_int("SM3_BN_UNROLL_ACT", env_int("SM3_BN_UNROLL", 4)),
                             env_int("SM3_BN_UNROLL_RED", env_int("SM3_BN_UNROLL", 4)),
                             env_int("SM3_BN_UNROLL_APP", env_int("SM3_BN_UNROLL", 8))};
    return u[kind];
}
static bool bn_nt() {
    static const bool nt = env_int("SM3_BN_NT", 1) != 0;
    return nt;
}
// expands CALL(U, NT) for the configured (unroll, non-temporal) pair
#define SM3_BN_DISPATCH(KIND, CALL)                 \
    do {                                            \
        const int u_ = bn_unroll(KIND);             \
        if (bn_nt()) {                              \
            if (u_ >= 8) { CALL(8, true); }         \
            else if (u_ >= 4) { CALL(4, true); }    \
            else if (u_ >= 2) { CALL(2, true); }    \
            else { CALL(1, true); }                 \
        } else {                                    \
            if (u_ >= 8) { CALL(8, false); }        \
            else if (u_ >= 4) { CALL(4, false); }   \
            else if (u_ >= 2) { CALL(2, false); }   \
            else { CALL(1, false); }                \
        }                                           \
    } while (0)

}  // namespace

extern "C" int sm3_bn_reduce_groups(int rows) {
    if (rows <= 0) return SM3_EINVAL;
    int G = (rows + 31) / 32;
    return G > SM3_BN_REDUCE_GROUPS ? SM3_BN_REDUCE_GROUPS : G;
}

extern "C" int sm3_bn_stats_reduce(const float* partials, int rows, int C, double* sums, double* workspace,
                                   int views, void* stream) {
    if (!partials || !workspace || rows <= 0 || C <= 0 || views < 1) return SM3_EINVAL;
    const int G = sm3_bn_reduce_groups(rows);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_stats_reduce_a, dim3((2 * C + 63) / 64, G, views), dim3(256), 0, st, partials, rows, C, workspace);
    SM3_CHECK_LAUNCH();
    if (sums) {  // sums == NULL: the caller hands workspace + groups to sm3_bn_finalize instead (one launch fewer)
        hipLaunchKernelGGL(bn_stats_reduce_b, dim3((8 * C + 255) / 256, 1, views), dim3(256), 0, st, workspace, G, C, sums);
        SM3_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int sm3_bn_finalize(const double* sums, int groups, int views, double count, int C, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                               float* save_invstd, void* stream) {
    if (!sums || !scale || !shift || C <= 0 || count <= 0 || groups < 1 || views < 1) return SM3_EINVAL;
    const BnFinalizeArgs a{count, C, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd};
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((16 * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, groups, views,
                       num_batches_tracked, a);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_bn_stats_finalize(const float* partials, int rows, int C, int views, double* workspace, uint32_t* tickets,
                                     double count, const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, float* scale,
                                     float* shift, float* save_mean, float* save_invstd, void* stream) {
    if (!partials || !workspace || !tickets || !scale || !shift || rows <= 0 || C <= 0 || count <= 0 || views < 1)
        return SM3_EINVAL;
    const int G = sm3_bn_reduce_groups(rows);
    const BnFinalizeArgs a{count, C, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd};
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + 31) / 32, G, views), dim3(256), 0, (hipStream_t)stream, partials, rows,
                       views, workspace, tickets, num_batches_tracked, a);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                                       const float* running_var, float eps, int C, float* scale, float* shift,
                                       void* stream) {
    if (!running_mean || !running_var || !scale || !shift || C <= 0) return SM3_EINVAL;
    hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, C, scale, shift);
    SM3_CHECK_LAUNCH();
    return 0;
}

static int bn_act_impl(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
                       const float* res_scale, const float* res_shift, int relu, int out_f32, void* y,
                       uint8_t* relu_mask, int64_t rows, int C, int views, void* stream, float* colsum = nullptr) {
    if (!x || !scale || !shift || !y || rows <= 0 || C <= 0 || views < 1) return SM3_EINVAL;
    if ((res_scale != nullptr) != (res_shift != nullptr) || (res_scale && !residual)) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    if (colsum && out_f32) return SM3_EINVAL;
    // with column sums the row walk of a view is cut as if the launch always held two views: a view's partial rows -- and
    // so the bits of their sum -- are the same whether the two views of a branch share a launch or not
    const RowWalk w = make_walk(rows, C / E, 8192, (colsum && views < 2) ? 2 : views);
    dim3 grid(w.gx, w.gy, views), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SM3_ACT_T(T, U, NT)                                                                                          \
    if (out_f32 && sizeof(T) == 2)                                                                                      \
        hipLaunchKernelGGL((bn_act_kernel<T, true, U, NT>), grid, block, 0, st, (const T*)x, scale, shift,              \
                           (const T*)residual, res_scale, res_shift, relu, y, relu_mask, rows, C, w.tbx, w.tby,         \
                           (float*)nullptr);                                                                            \
    else /* f32 storage: the two output forms coincide */                                                               \
        hipLaunchKernelGGL((bn_act_kernel<T, false, U, NT>), grid, block, 0, st, (const T*)x, scale, shift,             \
                           (const T*)residual, res_scale, res_shift, relu, y, relu_mask, rows, C, w.tbx, w.tby, colsum)
#define SM3_ACT(U, NT)                                            \
    if (dtype == SM3_F32) { SM3_ACT_T(float, U, NT); }            \
    else if (dtype == SM3_BF16) { SM3_ACT_T(bf16_t, U, NT); }     \
    else { SM3_ACT_T(f16_t, U, NT); }
    SM3_BN_DISPATCH(0, SM3_ACT);
#undef SM3_ACT
#undef SM3_ACT_T
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_bn_act(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
                          int relu, int out_f32, void* y, uint8_t* relu_mask, int64_t rows, int C, int views,
                          void* stream) {
    return bn_act_impl(dtype, x, scale, shift, residual, nullptr, nullptr, relu, out_f32, y, relu_mask, rows, C, views,
                       stream);
}

extern "C" int sm3_bn_act_colsum_rows(int64_t rows, int C, int dtype) {
    if (rows <= 0 || C <= 0 || !SM3_DTYPE_OK(dtype)) return SM3_EINVAL;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    return make_walk(rows, C / E, 8192, 2).gy;
}

extern "C" int sm3_bn_act_colsum(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
                                 int relu, void* y, uint8_t* relu_mask, float* colsum_partials, int64_t rows, int C,
                                 int views, void* stream) {
    if (!colsum_partials) return SM3_EINVAL;
    return bn_act_impl(dtype, x, scale, shift, residual, nullptr, nullptr, relu, 0, y, relu_mask, rows, C, views, stream,
                       colsum_partials);
}

extern "C" int sm3_subsample_colsum_rows(int64_t rows, int C, int dtype) { return sm3_bn_act_colsum_rows(rows, C, dtype); }

extern "C" int sm3_subsample_colsum(int dtype, const void* x, void* y, float* colsum_partials, int N, int H, int W, int C,
                                    int stride, int views, void* stream) {
    if (!x || !colsum_partials || N <= 0 || H <= 0 || W <= 0 || C <= 0 || stride < 1 || views < 1 || N % views)
        return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const int Hs = (H - 1) / stride + 1, Ws = (W - 1) / stride + 1;
    const int64_t rows = (int64_t)(N / views) * Hs * Ws;  // output rows of ONE view
    if ((int64_t)N * Hs * Ws > 0x7fffffffL) return SM3_EINVAL;
    const RowWalk w = make_walk(rows, C / E, 8192, views < 2 ? 2 : views);  // as sm3_bn_act_colsum: view-independent cut
    dim3 grid(w.gx, w.gy, views), block(256);
    const FastDiv dhw = make_fastdiv((uint32_t)(Hs * Ws)), dw = make_fastdiv((uint32_t)Ws);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SM3_F32)
        hipLaunchKernelGGL((subsample_colsum_kernel<float, 4>), grid, block, 0, st, (const float*)x, (float*)y, colsum_partials, rows,
                           C, Hs, Ws, H, W, stride, dhw, dw, w.tbx, w.tby);
    else if (dtype == SM3_BF16)
        hipLaunchKernelGGL((subsample_colsum_kernel<bf16_t, 4>), grid, block, 0, st, (const bf16_t*)x, (bf16_t*)y, colsum_partials,
                           rows, C, Hs, Ws, H, W, stride, dhw, dw, w.tbx, w.tby);
    else
        hipLaunchKernelGGL((subsample_colsum_kernel<f16_t, 4>), grid, block, 0, st, (const f16_t*)x, (f16_t*)y, colsum_partials,
                           rows, C, Hs, Ws, H, W, stride, dhw, dw, w.tbx, w.tby);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_bn_add_bn_act(int dtype, const void* x, const float* scale, const float* shift, const void* x2,
                                 const float* scale2, const float* shift2, int relu, void* y, uint8_t* relu_mask,
                                 int64_t rows, int C, int views, void* stream) {
    if (!x2 || !scale2 || !shift2) return SM3_EINVAL;
    return bn_act_impl(dtype, x, scale, shift, x2, scale2, shift2, relu, 0, y, relu_mask, rows, C, views, stream);
}

static int bwd_gy(int64_t rows) {
    int64_t gy = (rows + 63) / 64;
    if (gy > 1024) gy = 1024;
    if (gy < 1) gy = 1;
    return (int)gy;
}

extern "C" int sm3_bn_bwd_partial_rows(int64_t rows, int C) {
    (void)C;
    return rows > 0 ? bwd_gy(rows) : SM3_EINVAL;
}

extern "C" int sm3_bn_bwd_reduce(int dtype, const void* dy, const void* y, const uint8_t* relu_mask, const void* x,
                                 const float* mean, const float* invstd, void* dz, int64_t rows, int C,
                                 float* partials, int views, void* stream) {
    if (!dy || (x && (!mean || !invstd)) || !partials || rows <= 0 || C <= 0 || views < 1) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    RowWalk w = make_walk(rows, C / E, 1 << 30);
    w.gy = bwd_gy(rows);
    dim3 grid(w.gx, w.gy, views), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SM3_RED_T(T, U, NT)                                                                                      \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, U, NT>), grid, block, 0, st, (const T*)dy, (const T*)y, relu_mask,  \
                       (const T*)x, mean, invstd, (T*)dz, rows, C, partials, w.tbx, w.tby)
#define SM3_RED(U, NT)                                        \
    if (dtype == SM3_F32) { SM3_RED_T(float, U, NT); }        \
    else if (dtype == SM3_BF16) { SM3_RED_T(bf16_t, U, NT); } \
    else { SM3_RED_T(f16_t, U, NT); }
    SM3_BN_DISPATCH(1, SM3_RED);
#undef SM3_RED
#undef SM3_RED_T
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_bn_bwd_apply(int dtype, const void* dz, const void* x, const float* mean, const float* invstd,
                                const float* gamma, const double* global_sums, double count,
                                const double* local_sums, float* dgamma, float* dbeta, void* dx, int64_t rows, int C,
                                int views, void* stream) {
    if (!dz || !x || !mean || !invstd || !global_sums || !dx || rows <= 0 || C <= 0 || count <= 0 || views < 1)
        return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const RowWalk w = make_walk(rows, C / E, 8192, views);
    dim3 grid(w.gx, w.gy, views), block(256);
    hipStream_t st = (hipStream_t)stream;
#define SM3_APP_T(T, U, NT)                                                                                       \
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, U, NT>), grid, block, 0, st, (const T*)dz, (const T*)x, mean, invstd, \
                       gamma, global_sums, 1.0 / count, local_sums, dgamma, dbeta, (T*)dx, rows, C, w.tbx, w.tby)
#define SM3_APP(U, NT)                                        \
    if (dtype == SM3_F32) { SM3_APP_T(float, U, NT); }        \
    else if (dtype == SM3_BF16) { SM3_APP_T(bf16_t, U, NT); } \
    else { SM3_APP_T(f16_t, U, NT); }
    SM3_BN_DISPATCH(2, SM3_APP);
#undef SM3_APP
#undef SM3_APP_T
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_bn_bwd_apply2(int dtype, const void* dz, double count, const sm3_bn_apply_side* a,
                                 const sm3_bn_apply_side* b, int64_t rows, int C, int views, void* stream) {
    if (!dz || !a || !b || rows <= 0 || C <= 0 || count <= 0 || views < 1) return SM3_EINVAL;
    for (const sm3_bn_apply_side* s : {a, b})
        if (!s->x || !s->mean || !s->invstd || !s->global_sums || !s->dx) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const RowWalk w = make_walk(rows, C / E, 8192, views);
    dim3 grid(w.gx, w.gy, views), block(256);
    hipStream_t st = (hipStream_t)stream;
    const BnApplySide sa = {a->x, a->mean, a->invstd, a->gamma, a->global_sums, a->local_sums, a->dgamma, a->dbeta, a->dx};
    const BnApplySide sb = {b->x, b->mean, b->invstd, b->gamma, b->global_sums, b->local_sums, b->dgamma, b->dbeta, b->dx};
    // half the unroll of the single form: three input streams per row instead of two
#define SM3_APP2_T(T, U, NT)                                                                                   \
    hipLaunchKernelGGL((bn_bwd_apply2_kernel<T, (U > 4 ? 4 : U), NT>), grid, block, 0, st, (const T*)dz, sa, sb,  \
                       1.0 / count, rows, C, w.tbx, w.tby)
#define SM3_APP2(U, NT)                                        \
    if (dtype == SM3_F32) { SM3_APP2_T(float, U, NT); }        \
    else if (dtype == SM3_BF16) { SM3_APP2_T(bf16_t, U, NT); } \
    else { SM3_APP2_T(f16_t, U, NT); }
    SM3_BN_DISPATCH(2, SM3_APP2);
#undef SM3_APP2
#undef SM3_APP2_T
    SM3_CHECK_LAUNCH();
    return 0;
}
