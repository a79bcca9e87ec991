// Stem im2col, max / average pooling (NHWC), weight layout preparation, casts.  HBM-bound
// helpers around the MFMA kernels: 16-byte vector accesses along the channel axis.
//
// Reference call sites replaced: stem nn.Conv2d(3,64,7,2,3) input gather (src/models/resnet.py:208-210,294),
// nn.MaxPool2d(3,2,1) (:213,297), nn.AdaptiveAvgPool2d(1)+flatten (:224,304-305) and their autograd.
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ void store_elem(T* p, float v);
template <>
__device__ __forceinline__ void store_elem<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_elem<bf16_t>(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
template <>
__device__ __forceinline__ void store_elem<f16_t>(f16_t* p, float v) { p->v = f32_to_f16(v); }

// cols[m][k], m = (n, oy, ox), k = (kh*7 + kw)*3 + c ; one thread per 16-byte chunk of a row
// Index decomposition of the element-wise kernels below: 32-bit with precomputed multipliers (the host rejects
// tensors of 2^31 vectors or more).  The 64-bit divisions by run-time values this replaces cost more than the
// memory traffic of these kernels (im2col: 3.3 -> 1.x ms per step).
struct PixDiv {
    FastDiv cv, w, hw;  // vectors per pixel, row width, pixels per image
};
static PixDiv make_pixdiv(int cv, int w, int h) {
    PixDiv d;
    d.cv = make_fastdiv((uint32_t)cv);
    d.w = make_fastdiv((uint32_t)w);
    d.hw = make_fastdiv((uint32_t)(w * h));
    return d;
}
__device__ __forceinline__ void split_index(uint32_t idx, const PixDiv& d, int& c, uint32_t& pix, int& n, int& y, int& x) {
    pix = fdiv(idx, d.cv);
    c = (int)(idx - pix * d.cv.d);
    n = (int)fdiv(pix, d.hw);
    const uint32_t rem = pix - (uint32_t)n * d.hw.d;
    y = (int)fdiv(rem, d.w);
    x = (int)(rem - (uint32_t)y * d.w.d);
}

// One block per output row (n, oy): the 7 input rows x 3 planes that row needs are staged in LDS with coalesced
// reads (zero padding included), then every thread composes 16-byte chunks of cols from LDS and writes them
// contiguously (the row's Wo x Kpad block of cols is one contiguous range).  The direct gather it replaces issued 8
// scattered 4-byte global loads per 16 bytes of output and ran at 1.7 TB/s.
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, T* __restrict__ cols, int H,
                                                          int W, int Ho, int Wo, int Kpad, const FastDiv div_cpr) {
    constexpr int E = ElemTraits<T>::kPer16B;
    extern __shared__ float tile[];  // [3][7][TW], TW = 2*Wo + 5: column j holds ix = j - 3
    const int TW = 2 * Wo + 5;
    const int oy = blockIdx.x % Ho, n = blockIdx.x / Ho;
    const float* xn = x + (int64_t)n * 3 * H * W;
    for (int r = threadIdx.x >> 6; r < 21; r += 4) {  // one wave per (plane, kh) row at a time
        const int c = r / 7, kh = r - c * 7;
        const int iy = oy * 2 - 3 + kh;
        const bool rowok = (unsigned)iy < (unsigned)H;
        const float* src = xn + ((int64_t)c * H + iy) * W;
        for (int j = threadIdx.x & 63; j < TW; j += 64) {
            const int ix = j - 3;
            tile[r * TW + j] = (rowok && (unsigned)ix < (unsigned)W) ? src[ix] : 0.f;
        }
    }
    __syncthreads();
    const int cpr = Kpad / E;
    T* out = cols + ((int64_t)blockIdx.x * Wo) * Kpad;
    for (uint32_t idx = threadIdx.x; idx < (uint32_t)(Wo * cpr); idx += 256) {
        const uint32_t ox = fdiv(idx, div_cpr);
        const int ch = (int)(idx - ox * (uint32_t)cpr);
        float v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = ch * E + e;
            float val = 0.f;
            if (k < 147) {
                const int tap = k / 3, c = k - tap * 3;
                const int kh = tap / 7, kw = tap - kh * 7;
                val = tile[(c * 7 + kh) * TW + 2 * (int)ox + kw];
            }
            v[e] = val;
        }
        stg16<true>(out + (int64_t)idx * E, pack16<T>(v));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ argmax, int N, int H, int W, int C,
                                                          int Ho, int Wo, uint32_t total, const PixDiv dv) {
    constexpr int E = ElemTraits<T>::kPer16B;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int c, n, oy, ox;
        uint32_t pix32;
        split_index(idx, dv, c, pix32, n, oy, ox);
        const int64_t pix = pix32;
        float best[E];
        int arg[E];  // window position kh*3+kw of the FIRST maximum (strict '>' in scan order, as ATen)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            best[e] = -INFINITY;
            arg[e] = -1;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy * 2 - 1 + kh;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = ox * 2 - 1 + kw;
                if ((unsigned)ix >= (unsigned)W) continue;
                float v[E];
                unpack16<T>(*reinterpret_cast<const uint4*>(x + (((int64_t)n * H + iy) * W + ix) * C + (int64_t)c * E), v);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (v[e] > best[e] || arg[e] < 0) {
                        best[e] = v[e];
                        arg[e] = kh * 3 + kw;
                    }
            }
        }
        *reinterpret_cast<uint4*>(y + pix * C + (int64_t)c * E) = pack16<T>(best);
        if (argmax) {
            uint8_t* a = argmax + pix * C + (int64_t)c * E;
            if constexpr (E == 8) {
                *reinterpret_cast<uint2*>(a) = make_uint2(
                    (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24),
                    (unsigned)arg[4] | ((unsigned)arg[5] << 8) | ((unsigned)arg[6] << 16) | ((unsigned)arg[7] << 24));
            } else {
                *reinterpret_cast<unsigned*>(a) =
                    (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
            }
        }
    }
}

// gather form of the backward: input pixel (iy,ix) receives dy of every window whose FIRST maximum (recorded by
// the forward pass as a window position 0..8) sits on it.  No atomics, deterministic; per input vector at most
// 4 windows x (8-byte argmax + 16-byte dy) instead of re-scanning 4 x 9 input vectors.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const uint8_t* __restrict__ argmax, const T* __restrict__ dy,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                          int Wo, uint32_t total, const PixDiv dv) {
    constexpr int E = ElemTraits<T>::kPer16B;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int c, n, iy, ix;
        uint32_t pix32;
        split_index(idx, dv, c, pix32, n, iy, ix);
        const int64_t pix = pix32;
        float g[E];
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = 0.f;
        const int oy_lo = iy / 2, oy_hi = min(Ho - 1, (iy + 1) / 2);
        const int ox_lo = ix / 2, ox_hi = min(Wo - 1, (ix + 1) / 2);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const int self = (iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1));
                const int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * C + (int64_t)c * E;
                unsigned a[2];
                if constexpr (E == 8) {
                    const uint2 q = *reinterpret_cast<const uint2*>(argmax + o);
                    a[0] = q.x;
                    a[1] = q.y;
                } else {
                    a[0] = *reinterpret_cast<const unsigned*>(argmax + o);
                    a[1] = 0;
                }
                float d[E];
                unpack16<T>(*reinterpret_cast<const uint4*>(dy + o), d);
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if ((int)((a[e >> 2] >> (8 * (e & 3))) & 0xff) == self) g[e] += d[e];
            }
        *reinterpret_cast<uint4*>(dx + pix * C + (int64_t)c * E) = pack16<T>(g);
    }
}

// Stem: BatchNorm + ReLU + max-pool in one pass (resnet.py:295-297 bn1 -> relu -> maxpool).  Reads the pre-BN stem
// output once (window overlap comes from L2) and writes only the pooled map and its argmax bytes: the post-ReLU
// stem activation -- the largest tensor of the network -- is never stored.  Every window value is rounded to T
// before the comparison, so maxima and tie-breaks are exactly those of sm3_bn_act followed by sm3_maxpool3x3s2_fwd.
// n_per_view: images per view (scale/shift are [views][C]).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ y,
                                                                  uint8_t* __restrict__ argmax, int N, int H, int W, int C,
                                                                  int Ho, int Wo, int n_per_view, uint32_t total,
                                                                  const PixDiv dv) {
    constexpr int E = ElemTraits<T>::kPer16B;
    for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        int c, n, oy, ox;
        uint32_t pix32;
        split_index(idx, dv, c, pix32, n, oy, ox);
        const int64_t pix = pix32;
        const int view = n / n_per_view;
        float sc[E], sh[E], best[E];
        int arg[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            sc[e] = scale[view * C + c * E + e];
            sh[e] = shift[view * C + c * E + e];
            best[e] = -INFINITY;
            arg[e] = -1;
        }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy * 2 - 1 + kh;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = ox * 2 - 1 + kw;
                if ((unsigned)ix >= (unsigned)W) continue;
                float v[E];
                unpack16<T>(*reinterpret_cast<const uint4*>(x + (((int64_t)n * H + iy) * W + ix) * C + (int64_t)c * E), v);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float a = ElemTraits<T>::round(fmaxf(v[e] * sc[e] + sh[e], 0.f));
                    if (a > best[e] || arg[e] < 0) {
                        best[e] = a;
                        arg[e] = kh * 3 + kw;
                    }
                }
            }
        }
        *reinterpret_cast<uint4*>(y + pix * C + (int64_t)c * E) = pack16<T>(best);
        if (argmax) {
            uint8_t* a = argmax + pix * C + (int64_t)c * E;
            if constexpr (E == 8) {
                *reinterpret_cast<uint2*>(a) = make_uint2(
                    (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24),
                    (unsigned)arg[4] | ((unsigned)arg[5] << 8) | ((unsigned)arg[6] << 16) | ((unsigned)arg[7] << 24));
            } else {
                *reinterpret_cast<unsigned*>(a) =
                    (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
            }
        }
    }
}

// Backward of that chain up to BatchNorm-backward phase 1, in one pass over the stem-sized tensors: gathers the
// pooled gradient through the argmax bytes (as maxpool_bwd_kernel), masks it with the ReLU recomputed from the
// pre-BN tensor (x*scale + shift > 0), stores dz and emits the per-block partial sums of (dz, dz * xhat) in the
// layout of sm3_bn_bwd_reduce.  Replaces maxpool_bwd + bn_bwd_reduce: one write and one read of the gradient of
// the largest activation fewer, and no ReLU mask was ever stored.
// grid (vectors per pixel / tbx, row groups, views); rows = pixels of ONE view.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_kernel(const uint8_t* __restrict__ argmax, const T* __restrict__ dy,
                                                             const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, T* __restrict__ dz,
                                                             float* __restrict__ partials, int H, int W, int C, int Ho,
                                                             int Wo, int n_per_view, int64_t rows, int tbx, int tby,
                                                             const FastDiv div_hw, const FastDiv div_w) {
    constexpr int E = ElemTraits<T>::kPer16B;
    __shared__ float sred[256 * 2 * 8];
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    const bool active = (cv * E < C) && (ty < tby);
    const int view = blockIdx.z;
    float s1[E], s2[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s1[e] = s2[e] = 0.f;
    if (active) {
        float sc[E], sh[E], mu[E], is[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            sc[e] = scale[view * C + cv * E + e];
            sh[e] = shift[view * C + cv * E + e];
            mu[e] = mean[view * C + cv * E + e];
            is[e] = invstd[view * C + cv * E + e];
        }
        // U = 2 rows per thread in flight: every load of both rows (the pre-BN vector and up to four (argmax, dy) pairs
        // each, a chain of dependent address arithmetic otherwise) is issued before the first use -- with one row per
        // iteration the kernel ran at the latency of that chain (3.6 TB/s).
        constexpr int U = 2;
        const int64_t rstep = (int64_t)gridDim.y * tby;
        for (int64_t r0 = (int64_t)blockIdx.y * tby + ty; r0 < rows; r0 += U * rstep) {
            uint4 xu[U], du[U][4];
            unsigned au[U][4][2];
            int selfs[U][4];
            int64_t pixs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = r0 + u * rstep;
                pixs[u] = -1;
                xu[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int w = 0; w < 4; ++w) selfs[u][w] = -1;
                if (r >= rows) continue;
                const uint32_t nl = fdiv((uint32_t)r, div_hw);
                const uint32_t rem = (uint32_t)r - nl * div_hw.d;
                const int iy = (int)fdiv(rem, div_w), ix = (int)(rem - (uint32_t)iy * div_w.d);
                const int64_t n = (int64_t)view * n_per_view + nl;
                pixs[u] = (int64_t)view * rows + r;
                xu[u] = ldg16<true>(x + pixs[u] * C + (int64_t)cv * E);
                const int oy_lo = iy / 2, oy_hi = min(Ho - 1, (iy + 1) / 2);
                const int ox_lo = ix / 2, ox_hi = min(Wo - 1, (ix + 1) / 2);
#pragma unroll
                for (int w = 0; w < 4; ++w) {  // the (at most) 2 x 2 pooling windows that contain this input pixel
                    const int oy = oy_lo + (w >> 1), ox = ox_lo + (w & 1);
                    if (oy > oy_hi || ox > ox_hi) continue;
                    selfs[u][w] = (iy - (oy * 2 - 1)) * 3 + (ix - (ox * 2 - 1));
                    const int64_t o = ((n * Ho + oy) * Wo + ox) * C + (int64_t)cv * E;
                    if constexpr (E == 8) {
                        const uint2 q = *reinterpret_cast<const uint2*>(argmax + o);
                        au[u][w][0] = q.x;
                        au[u][w][1] = q.y;
                    } else {
                        au[u][w][0] = *reinterpret_cast<const unsigned*>(argmax + o);
                        au[u][w][1] = 0;
                    }
                    du[u][w] = *reinterpret_cast<const uint4*>(dy + o);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pixs[u] < 0) continue;
                float g[E];
#pragma unroll
                for (int e = 0; e < E; ++e) g[e] = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (selfs[u][w] < 0) continue;
                    float d[E];
                    unpack16<T>(du[u][w], d);
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        if ((int)((au[u][w][e >> 2] >> (8 * (e & 3))) & 0xff) == selfs[u][w]) g[e] += d[e];
                }
                float xv[E];
                unpack16<T>(xu[u], xv);
#pragma unroll
                for (int e = 0; e < E; ++e) g[e] = (xv[e] * sc[e] + sh[e] > 0.f) ? g[e] : 0.f;
                const uint4 packed = pack16<T>(g);
                stg16<true>(dz + pixs[u] * C + (int64_t)cv * E, packed);
                float gr[E];
                unpack16<T>(packed, gr);  // sums of the STORED (rounded) dz: what the apply pass will read
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    s1[e] += gr[e];
                    s2[e] += gr[e] * (xv[e] - mu[e]) * is[e];
                }
            }
        }
    }
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
        float* pv = partials + (int64_t)view * gridDim.y * 2 * C;
        float* p1 = pv + ((int64_t)blockIdx.y * 2 + 0) * C + (int64_t)cv * E;
        float* p2 = pv + ((int64_t)blockIdx.y * 2 + 1) * C + (int64_t)cv * E;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            p1[e] = s1[e];
            p2[e] = s2[e];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, float* __restrict__ f32,
                                                          T* __restrict__ ft, int N, int HW, int C) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cv = C / E;
    const int64_t total = (int64_t)N * cv;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cv);
    const int n = (int)(idx / cv);
    float s[E];
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] = 0.f;
    for (int i = 0; i < HW; ++i) {
        float v[E];
        unpack16<T>(*reinterpret_cast<const uint4*>(x + ((int64_t)n * HW + i) * C + (int64_t)c * E), v);
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] += v[e];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < E; ++e) s[e] *= inv;
    if (f32) {
#pragma unroll
        for (int e = 0; e < E; ++e) f32[(int64_t)n * C + (int64_t)c * E + e] = s[e];
    }
    if (ft) *reinterpret_cast<uint4*>(ft + (int64_t)n * C + (int64_t)c * E) = pack16<T>(s);
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ df, T* __restrict__ dx, int N, int HW,
                                                          int C) {
    constexpr int E = ElemTraits<T>::kPer16B;
    const int cv = C / E;
    const int64_t total = (int64_t)N * HW * cv;
    const float inv = 1.f / (float)HW;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % cv);
        const int64_t pix = idx / cv;
        const int n = (int)(pix / HW);
        float v[E];
        unpack16<T>(*reinterpret_cast<const uint4*>(df + (int64_t)n * C + (int64_t)c * E), v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] *= inv;
        *reinterpret_cast<uint4*>(dx + pix * C + (int64_t)c * E) = pack16<T>(v);
    }
}

template <typename T>
__global__ void weight_prep_kernel(const float* __restrict__ w, int Co, int taps, int Ci, T* __restrict__ wf, int ld,
                                   T* __restrict__ wd) {
    const int64_t nf = wf ? (int64_t)Co * ld : 0;
    const int64_t nd = wd ? (int64_t)Co * taps * Ci : 0;
    const int K = taps * Ci;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nd;
         idx += (int64_t)gridDim.x * blockDim.x) {
        if (idx < nf) {
            const int co = (int)(idx / ld), k = (int)(idx % ld);
            store_elem<T>(wf + idx, k < K ? w[(int64_t)co * K + k] : 0.f);
        } else {
            // wd[ci][t][co] = w[co][t][ci]
            const int64_t j = idx - nf;
            const int co = (int)(j % Co);
            const int t = (int)((j / Co) % taps);
            const int ci = (int)(j / ((int64_t)Co * taps));
            store_elem<T>(wd + j, w[((int64_t)co * taps + t) * Ci + ci]);
        }
    }
}

// all filter banks of a model in one launch: blockIdx.y = bank.  Forward copy: grid-stride, coalesced both ways.
// Data-gradient bank wd[ci][t][co] = w[co][t][ci]: a transpose per tap, done through 32 x 33 LDS tiles so that both the
// fp32 reads (along ci) and the T writes (along co) are coalesced -- the element-wise form read w with a stride of
// taps*Ci floats per lane and ran at 0.8 TB/s (0.8 ms per step for 47 M weights).
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const sm3_wprep_item* __restrict__ items,
                                                                const int* __restrict__ only_if) {
    __shared__ float tile[32][33];
    if (only_if && *only_if == 0) return;  // masters unchanged since the banks were written (sm3_weights_changed)
    const sm3_wprep_item it = items[blockIdx.y];
    const float* __restrict__ w = it.w;
    T* __restrict__ wf = reinterpret_cast<T*>(it.w_fwd);
    T* __restrict__ wd = reinterpret_cast<T*>(it.w_dgrad);
    const int Co = it.Co, taps = it.taps, Ci = it.Ci, ld = it.ld_fwd, K = taps * Ci;
    // 16-bit banks of the usual shapes (no row padding, K a multiple of 4, Co even): 16-byte reads / 8-byte writes for the
    // forward copy, and the transpose through 64 x 32 tiles whose writes are PAIRS of output channels (128 B per half-wave
    // instead of 64) -- the element-wise forms ran at 0.85 TB/s, 0.66 ms per step for the model's 47 M weights
    const bool vec = sizeof(T) == 2 && ld == K && (K & 3) == 0 && (Co & 1) == 0 && ((uintptr_t)w & 15) == 0 &&
                     ((uintptr_t)wf & 7) == 0 && ((uintptr_t)wd & 3) == 0;
    if (wf && vec) {
        const int64_t nq = (int64_t)Co * K / 4;
        const float4* __restrict__ w4 = reinterpret_cast<const float4*>(w);
        uint2* __restrict__ o2 = reinterpret_cast<uint2*>(wf);
        for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
            const float4 v = w4[q];
            uint2 o;
            if constexpr (sizeof(T) == 2) {
                o.x = pack2<T>(v.x, v.y);
                o.y = pack2<T>(v.z, v.w);
            }
            o2[q] = o;
        }
    } else if (wf) {
        const int64_t nf = (int64_t)Co * ld;
        for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf; idx += (int64_t)gridDim.x * blockDim.x) {
            const int co = (int)(idx / ld), k = (int)(idx % ld);
            store_elem<T>(wf + idx, k < K ? w[(int64_t)co * K + k] : 0.f);
        }
    }
    if (wd && vec) {
        __shared__ float tile2[64][33];
        const int tco = (Co + 63) / 64, tci = (Ci + 31) / 32;
        const int ntiles = taps * tco * tci;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
        for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
            const int t = tl / (tco * tci), r = tl - t * tco * tci;
            const int co0 = (r / tci) * 64, ci0 = (r % tci) * 32;
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // rows co0 + ty + 8j, columns ci0 + tx (128 B along ci per half-wave)
                const int co = co0 + ty + 8 * j, ci = ci0 + tx;
                tile2[ty + 8 * j][tx] = (co < Co && ci < Ci) ? w[((int64_t)co * taps + t) * Ci + ci] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // rows ci0 + ty + 8j, channel pair co0 + 2 tx, + 1 (128 B along co per half-wave)
                const int ci = ci0 + ty + 8 * j, co = co0 + 2 * tx;
                if constexpr (sizeof(T) == 2) {
                    if (ci < Ci && co < Co)
                        *reinterpret_cast<uint32_t*>(wd + ((int64_t)ci * taps + t) * Co + co) =
                            pack2<T>(tile2[2 * tx][ty + 8 * j], tile2[2 * tx + 1][ty + 8 * j]);
                }
            }
            __syncthreads();
        }
    } else if (wd) {
        const int tco = (Co + 31) / 32, tci = (Ci + 31) / 32;
        const int ntiles = taps * tco * tci;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
        for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
            const int t = tl / (tco * tci), r = tl - t * tco * tci;
            const int co0 = (r / tci) * 32, ci0 = (r % tci) * 32;
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // read rows co0 + ty + 8j, columns ci0 + tx (contiguous along ci)
                const int co = co0 + ty + 8 * j, ci = ci0 + tx;
                tile[ty + 8 * j][tx] = (co < Co && ci < Ci) ? w[((int64_t)co * taps + t) * Ci + ci] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // write rows ci0 + ty + 8j, columns co0 + tx (contiguous along co)
                const int ci = ci0 + ty + 8 * j, co = co0 + tx;
                if (ci < Ci && co < Co) store_elem<T>(wd + ((int64_t)ci * taps + t) * Co + co, tile[tx][ty + 8 * j]);
            }
            __syncthreads();
        }
    }
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ s, T* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        store_elem<T>(d + i, s[i]);
}
template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ s, float* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = ElemTraits<T>::load(s + i);
}

inline unsigned grid_for(int64_t total, int per_block = 256, int64_t cap = 8192) {
    int64_t g = (total + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16, CALL_F16)  \
    if ((dtype) == SM3_F32) { CALL_F32; }       \
    else if ((dtype) == SM3_BF16) { CALL_BF16; } \
    else if ((dtype) == SM3_F16) { CALL_F16; } \
    else return SM3_EDTYPE;

extern "C" int sm3_stem_im2col(int dtype, const float* x_nchw, void* cols, int N, int H, int W, int Kpad,
                               void* stream) {
    if (!x_nchw || !cols || N <= 0 || H <= 0 || W <= 0 || Kpad < 147) return SM3_EINVAL;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (Kpad % E) return SM3_EALIGN;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if ((int64_t)N * Ho >= 0x7fffffffLL || (int64_t)Wo * (Kpad / E) >= 0x7fffffffLL) return SM3_EINVAL;
    const size_t lds = (size_t)21 * (2 * Wo + 5) * sizeof(float);
    if (lds > 64 * 1024) return SM3_EINVAL;  // image wider than ~1500 pixels
    hipStream_t st = (hipStream_t)stream;
    const FastDiv dc = make_fastdiv((uint32_t)(Kpad / E));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(N * Ho), dim3(256), lds, st, x_nchw, (float*)cols, H, W, Ho, Wo, Kpad, dc),
               hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(N * Ho), dim3(256), lds, st, x_nchw, (bf16_t*)cols, H, W, Ho, Wo, Kpad, dc),
               hipLaunchKernelGGL(stem_im2col_kernel<f16_t>, dim3(N * Ho), dim3(256), lds, st, x_nchw, (f16_t*)cols, H, W, Ho, Wo, Kpad, dc));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C,
                                    void* stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return SM3_EINVAL;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / E);
    if (total >= 0x7fffffffLL) return SM3_EINVAL;
    const unsigned g = grid_for(total, 256, 1 << 20);
    hipStream_t st = (hipStream_t)stream;
    const PixDiv dv = make_pixdiv(C / E, Wo, Ho);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, argmax, N, H, W, C, Ho, Wo, (uint32_t)total, dv),
               hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, argmax, N, H, W, C, Ho, Wo, (uint32_t)total, dv),
               hipLaunchKernelGGL(maxpool_fwd_kernel<f16_t>, dim3(g), dim3(256), 0, st, (const f16_t*)x, (f16_t*)y, argmax, N, H, W, C, Ho, Wo, (uint32_t)total, dv));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_maxpool3x3s2_bwd(int dtype, const uint8_t* argmax, const void* dy, void* dx, int N, int H, int W,
                                    int C, void* stream) {
    if (!argmax || !dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return SM3_EINVAL;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * H * W * (C / E);
    if (total >= 0x7fffffffLL) return SM3_EINVAL;
    const unsigned g = grid_for(total, 256, 1 << 20);
    hipStream_t st = (hipStream_t)stream;
    const PixDiv dv = make_pixdiv(C / E, W, H);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(g), dim3(256), 0, st, argmax, (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo, (uint32_t)total, dv),
               hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, argmax, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo, (uint32_t)total, dv),
               hipLaunchKernelGGL(maxpool_bwd_kernel<f16_t>, dim3(g), dim3(256), 0, st, argmax, (const f16_t*)dy, (f16_t*)dx, N, H, W, C, Ho, Wo, (uint32_t)total, dv));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_bn_relu_maxpool_fwd(int dtype, const void* x, const float* scale, const float* shift, void* y,
                                       uint8_t* argmax, int N, int H, int W, int C, int views, void* stream) {
    if (!x || !scale || !shift || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0 || views < 1 || N % views) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / E);
    if (total >= 0x7fffffffLL) return SM3_EINVAL;
    const unsigned g = grid_for(total, 256, 1 << 20);
    hipStream_t st = (hipStream_t)stream;
    const PixDiv dv = make_pixdiv(C / E, Wo, Ho);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, scale, shift, (float*)y, argmax, N, H, W, C, Ho, Wo, N / views, (uint32_t)total, dv),
               hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (bf16_t*)y, argmax, N, H, W, C, Ho, Wo, N / views, (uint32_t)total, dv),
               hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<f16_t>, dim3(g), dim3(256), 0, st, (const f16_t*)x, scale, shift, (f16_t*)y, argmax, N, H, W, C, Ho, Wo, N / views, (uint32_t)total, dv));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_maxpool_bn_bwd_partial_rows(int N, int H, int W, int views) {
    if (N <= 0 || H <= 0 || W <= 0 || views < 1 || N % views) return SM3_EINVAL;
    int64_t gy = ((int64_t)(N / views) * H * W + 63) / 64;
    return (int)(gy > 1024 ? 1024 : gy);
}

extern "C" int sm3_maxpool_bn_bwd(int dtype, const uint8_t* argmax, const void* dy, const void* x, const float* scale,
                                  const float* shift, const float* mean, const float* invstd, void* dz,
                                  float* partials, int N, int H, int W, int C, int views, void* stream) {
    if (!argmax || !dy || !x || !scale || !shift || !mean || !invstd || !dz || !partials) return SM3_EINVAL;
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || views < 1 || N % views) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(dtype)) return SM3_EDTYPE;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E || C / E > 256) return SM3_EALIGN;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t rows = (int64_t)(N / views) * H * W;
    if (rows * views >= 0x7fffffffLL) return SM3_EINVAL;
    const int cvecs = C / E;
    int tbx = 1;
    while (tbx < cvecs && tbx < 256) tbx <<= 1;  // power of two >= cvecs (idle lanes when C/E is not one)
    const int tby = 256 / tbx;
    const int gx = (cvecs + tbx - 1) / tbx;
    const int gy = sm3_maxpool_bn_bwd_partial_rows(N, H, W, views);
    hipStream_t st = (hipStream_t)stream;
    const FastDiv dhw = make_fastdiv((uint32_t)(H * W)), dw = make_fastdiv((uint32_t)W);
    dim3 grid(gx, gy, views);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(maxpool_bn_bwd_kernel<float>, grid, dim3(256), 0, st, argmax, (const float*)dy, (const float*)x, scale, shift, mean, invstd, (float*)dz, partials, H, W, C, Ho, Wo, N / views, rows, tbx, tby, dhw, dw),
               hipLaunchKernelGGL(maxpool_bn_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, argmax, (const bf16_t*)dy, (const bf16_t*)x, scale, shift, mean, invstd, (bf16_t*)dz, partials, H, W, C, Ho, Wo, N / views, rows, tbx, tby, dhw, dw),
               hipLaunchKernelGGL(maxpool_bn_bwd_kernel<f16_t>, grid, dim3(256), 0, st, argmax, (const f16_t*)dy, (const f16_t*)x, scale, shift, mean, invstd, (f16_t*)dz, partials, H, W, C, Ho, Wo, N / views, rows, tbx, tby, dhw, dw));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_avgpool_fwd(int dtype, const void* x, float* feat_f32, void* feat_t, int N, int HW, int C,
                               void* stream) {
    if (!x || N <= 0 || HW <= 0 || C <= 0 || (!feat_f32 && !feat_t)) return SM3_EINVAL;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const int64_t total = (int64_t)N * (C / E);
    const unsigned g = (unsigned)((total + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(avgpool_fwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, feat_f32, (float*)feat_t, N, HW, C),
               hipLaunchKernelGGL(avgpool_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, feat_f32, (bf16_t*)feat_t, N, HW, C),
               hipLaunchKernelGGL(avgpool_fwd_kernel<f16_t>, dim3(g), dim3(256), 0, st, (const f16_t*)x, feat_f32, (f16_t*)feat_t, N, HW, C));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_avgpool_bwd(int dtype, const void* dfeat, void* dx, int N, int HW, int C, void* stream) {
    if (!dfeat || !dx || N <= 0 || HW <= 0 || C <= 0) return SM3_EINVAL;
    const int E = dtype == SM3_F32 ? 4 : 8;
    if (C % E) return SM3_EALIGN;
    const unsigned g = grid_for((int64_t)N * HW * (C / E));
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(avgpool_bwd_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)dfeat, (float*)dx, N, HW, C),
               hipLaunchKernelGGL(avgpool_bwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)dfeat, (bf16_t*)dx, N, HW, C),
               hipLaunchKernelGGL(avgpool_bwd_kernel<f16_t>, dim3(g), dim3(256), 0, st, (const f16_t*)dfeat, (f16_t*)dx, N, HW, C));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_weight_prep(int dtype, const float* w, int Co, int taps, int Ci, void* w_fwd, int ld_fwd,
                               void* w_dgrad, void* stream) {
    if (!w || Co <= 0 || taps <= 0 || Ci <= 0 || (!w_fwd && !w_dgrad)) return SM3_EINVAL;
    if (w_fwd && ld_fwd < taps * Ci) return SM3_EINVAL;
    const int64_t total = (w_fwd ? (int64_t)Co * ld_fwd : 0) + (w_dgrad ? (int64_t)Co * taps * Ci : 0);
    const unsigned g = grid_for(total);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(weight_prep_kernel<float>, dim3(g), dim3(256), 0, st, w, Co, taps, Ci, (float*)w_fwd, ld_fwd, (float*)w_dgrad),
               hipLaunchKernelGGL(weight_prep_kernel<bf16_t>, dim3(g), dim3(256), 0, st, w, Co, taps, Ci, (bf16_t*)w_fwd, ld_fwd, (bf16_t*)w_dgrad),
               hipLaunchKernelGGL(weight_prep_kernel<f16_t>, dim3(g), dim3(256), 0, st, w, Co, taps, Ci, (f16_t*)w_fwd, ld_fwd, (f16_t*)w_dgrad));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_weight_prep_batch_if(int dtype, const sm3_wprep_item* items_device, int n, const int* only_if,
                                        void* stream) {
    if (!items_device || n <= 0 || n > 65535) return SM3_EINVAL;
    // blocks per bank: the ten layer-4 banks hold two thirds of the weights, so the launch ends with THEIR blocks: 96 -> 384 per
    // bank took the three launches of a step from 0.38 to 0.23 ms (round 6; they run on the main stream, alone)
    static const int kWprepGx = getenv("SM3_WPREP_GX") ? atoi(getenv("SM3_WPREP_GX")) : 384;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(weight_prep_batch_kernel<float>, dim3(kWprepGx, n), dim3(256), 0, st, items_device, only_if),
               hipLaunchKernelGGL(weight_prep_batch_kernel<bf16_t>, dim3(kWprepGx, n), dim3(256), 0, st, items_device, only_if),
               hipLaunchKernelGGL(weight_prep_batch_kernel<f16_t>, dim3(kWprepGx, n), dim3(256), 0, st, items_device, only_if));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_weight_prep_batch(int dtype, const sm3_wprep_item* items_device, int n, void* stream) {
    return sm3_weight_prep_batch_if(dtype, items_device, n, nullptr, stream);
}

namespace {

// h = sum_i (bits_i + c) * (2i + 1)  mod 2^64: any single changed word changes h (odd multiplier), and the masters'
// typical change (every word, by an optimizer step) cancelling exactly is a 2^-64 event.
__global__ __launch_bounds__(256) void weights_hash_kernel(const uint32_t* __restrict__ w, int64_t n,
                                                           unsigned long long* __restrict__ acc) {
    unsigned long long h = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        h += ((unsigned long long)w[i] + 0x9E3779B97F4A7C15ull) * (unsigned long long)(2 * i + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ void weights_changed_kernel(unsigned long long* __restrict__ state, int* __restrict__ changed) {
    *changed = state[0] != state[1];
    state[1] = state[0];
    state[0] = 0;  // accumulator of the next call
}

}  // namespace

extern "C" int sm3_weights_changed(const float* flat, int64_t n, uint64_t* state, int* changed, void* stream) {
    if (!flat || !state || !changed || n <= 0) return SM3_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(weights_hash_kernel, dim3(1024), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(flat), n,
                       reinterpret_cast<unsigned long long*>(state));
    SM3_CHECK_LAUNCH();
    hipLaunchKernelGGL(weights_changed_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long*>(state), changed);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_cast_from_f32(int dtype, const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return SM3_EINVAL;
    const unsigned g = grid_for(n);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(g), dim3(256), 0, st, src, (float*)dst, n),
               hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, dim3(g), dim3(256), 0, st, src, (bf16_t*)dst, n),
               hipLaunchKernelGGL(cast_from_f32_kernel<f16_t>, dim3(g), dim3(256), 0, st, src, (f16_t*)dst, n));
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst || n <= 0) return SM3_EINVAL;
    const unsigned g = grid_for(n);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(cast_to_f32_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)src, dst, n),
               hipLaunchKernelGGL(cast_to_f32_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)src, dst, n),
               hipLaunchKernelGGL(cast_to_f32_kernel<f16_t>, dim3(g), dim3(256), 0, st, (const f16_t*)src, dst, n));
    SM3_CHECK_LAUNCH();
    return 0;
}
