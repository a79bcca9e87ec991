// Design-space probe for the HBM-bound BatchNorm row-walk kernels (bf16): unroll depth, grid cap, non-temporal
// accesses.  Stand-alone: hipcc --offload-arch=gfx950 -O3 scratch/stream_bench.hip -o scratch/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint32_t pk(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 h = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ uint4 pack8(const float* f) { return make_uint4(pk(f[0], f[1]), pk(f[2], f[3]), pk(f[4], f[5]), pk(f[6], f[7])); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld(const uint4* p) {
    if constexpr (NT) { u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); return make_uint4(v.x, v.y, v.z, v.w); }
    else return *p;
}
template <bool NT> __device__ __forceinline__ void st(uint4* p, uint4 v) {
    if constexpr (NT) { u32x4 w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p)); }
    else *p = v;
}

// dx = a*dz + b*x + c per channel
template <int U, bool NT>
__global__ __launch_bounds__(256) void apply_k(const uint16_t* __restrict__ dz, const uint16_t* __restrict__ x,
                                               const float* __restrict__ ka, const float* __restrict__ kb,
                                               const float* __restrict__ kc, uint16_t* __restrict__ dx, int64_t rows,
                                               int C, int tbx, int tby) {
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    if (cv * 8 >= C || ty >= tby) return;
    float a[8], b[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = ka[cv * 8 + e]; b[e] = kb[cv * 8 + e]; c[e] = kc[cv * 8 + e]; }
    const int64_t rstep = (int64_t)gridDim.y * tby;
    int64_t r = (int64_t)blockIdx.y * tby + ty;
    for (; r + (U - 1) * rstep < rows; r += U * rstep) {
        uint4 g[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (r + u * rstep) * C + (int64_t)cv * 8;
            g[u] = ld<NT>(reinterpret_cast<const uint4*>(dz + off));
            v[u] = ld<NT>(reinterpret_cast<const uint4*>(x + off));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (r + u * rstep) * C + (int64_t)cv * 8;
            float gf[8], xf[8];
            unpack8(g[u], gf); unpack8(v[u], xf);
#pragma unroll
            for (int e = 0; e < 8; ++e) gf[e] = fmaf(a[e], gf[e], fmaf(b[e], xf[e], c[e]));
            st<NT>(reinterpret_cast<uint4*>(dx + off), pack8(gf));
        }
    }
    for (; r < rows; r += rstep) {
        const int64_t off = r * C + (int64_t)cv * 8;
        float gf[8], xf[8];
        unpack8(*reinterpret_cast<const uint4*>(dz + off), gf); unpack8(*reinterpret_cast<const uint4*>(x + off), xf);
#pragma unroll
        for (int e = 0; e < 8; ++e) gf[e] = fmaf(a[e], gf[e], fmaf(b[e], xf[e], c[e]));
        *reinterpret_cast<uint4*>(dx + off) = pack8(gf);
    }
}

// the production formula (five per-channel constants), U = 1
__global__ __launch_bounds__(256) void apply_prod_k(const uint16_t* __restrict__ dz, const uint16_t* __restrict__ x,
                                                    const float* __restrict__ ka, const float* __restrict__ kb,
                                                    const float* __restrict__ kc, uint16_t* __restrict__ dx,
                                                    int64_t rows, int C, int tbx, int tby) {
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    if (cv * 8 >= C || ty >= tby) return;
    float mu[8], is[8], k0[8], k1[8], k2[8];
    const double* gs = reinterpret_cast<const double*>(kc);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mu[e] = ka[cv * 8 + e]; is[e] = kb[cv * 8 + e]; k0[e] = kb[cv * 8 + e] * 1.5f;
        k1[e] = (float)(gs[cv * 8 + e] * 1e-3); k2[e] = (float)(gs[C + cv * 8 + e] * 1e-3);
    }
    const int64_t rstep = (int64_t)gridDim.y * tby;
    for (int64_t r = (int64_t)blockIdx.y * tby + ty; r < rows; r += rstep) {
        const int64_t off = r * C + (int64_t)cv * 8;
        float gf[8], xf[8];
        unpack8(*reinterpret_cast<const uint4*>(dz + off), gf); unpack8(*reinterpret_cast<const uint4*>(x + off), xf);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float xh = (xf[e] - mu[e]) * is[e]; gf[e] = k0[e] * (gf[e] - k1[e] - xh * k2[e]); }
        *reinterpret_cast<uint4*>(dx + off) = pack8(gf);
    }
}

// contiguous-chunk variant: a block owns a contiguous span of rows (better DRAM page locality?), threads stride inside it
template <int U, bool NT>
__global__ __launch_bounds__(256) void apply_chunk_k(const uint16_t* __restrict__ dz, const uint16_t* __restrict__ x,
                                                     const float* __restrict__ ka, const float* __restrict__ kb,
                                                     const float* __restrict__ kc, uint16_t* __restrict__ dx,
                                                     int64_t rows, int C, int tbx, int tby, int64_t rows_per_block) {
    const int tx = threadIdx.x % tbx, ty = threadIdx.x / tbx;
    const int cv = blockIdx.x * tbx + tx;
    if (cv * 8 >= C || ty >= tby) return;
    float a[8], b[8], c[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = ka[cv * 8 + e]; b[e] = kb[cv * 8 + e]; c[e] = kc[cv * 8 + e]; }
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block; if (r1 > rows) r1 = rows;
    int64_t r = r0 + ty;
    for (; r + (U - 1) * tby < r1; r += U * tby) {
        uint4 g[U], v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (r + u * tby) * C + (int64_t)cv * 8;
            g[u] = ld<NT>(reinterpret_cast<const uint4*>(dz + off));
            v[u] = ld<NT>(reinterpret_cast<const uint4*>(x + off));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t off = (r + u * tby) * C + (int64_t)cv * 8;
            float gf[8], xf[8];
            unpack8(g[u], gf); unpack8(v[u], xf);
#pragma unroll
            for (int e = 0; e < 8; ++e) gf[e] = fmaf(a[e], gf[e], fmaf(b[e], xf[e], c[e]));
            st<NT>(reinterpret_cast<uint4*>(dx + off), pack8(gf));
        }
    }
    for (; r < r1; r += tby) {
        const int64_t off = r * C + (int64_t)cv * 8;
        float gf[8], xf[8];
        unpack8(*reinterpret_cast<const uint4*>(dz + off), gf); unpack8(*reinterpret_cast<const uint4*>(x + off), xf);
#pragma unroll
        for (int e = 0; e < 8; ++e) gf[e] = fmaf(a[e], gf[e], fmaf(b[e], xf[e], c[e]));
        *reinterpret_cast<uint4*>(dx + off) = pack8(gf);
    }
}

__global__ void copy_k(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint4 u = a[i], v = b[i];
        o[i] = make_uint4(u.x ^ v.x, u.y ^ v.y, u.z ^ v.z, u.w ^ v.w);
    }
}

struct Walk { int tbx, tby, gx, gy; };
static Walk make_walk(int64_t rows, int cvecs, int cap_blocks, int U) {
    Walk w; w.tbx = cvecs >= 256 ? 256 : cvecs; w.tby = 256 / w.tbx; if (w.tby < 1) w.tby = 1;
    w.gx = (cvecs + w.tbx - 1) / w.tbx;
    int64_t gy = (rows + (int64_t)w.tby * 8 - 1) / ((int64_t)w.tby * 8);
    int64_t cap = cap_blocks / w.gx > 0 ? cap_blocks / w.gx : 1;
    if (gy > cap) gy = cap; if (gy < 1) gy = 1; w.gy = (int)gy; return w;
}

template <typename F> static float timeit(F f, int reps = 10) {
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(s)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(e)); CK(hipEventSynchronize(e));
    float ms; CK(hipEventElapsedTime(&ms, s, e)); return ms / reps;
}

int main() {
    const int64_t shapes[][2] = {{802816, 256}, {802816, 64}, {200704, 512}, {200704, 128}, {50176, 1024}, {50176, 256}, {12544, 2048}, {12544, 512}};
    const int64_t arena = 802816LL * 256 * 2;   // elements; tensors rotate through it so every launch reads cold data
    uint16_t *dzA, *xA, *dxA; float *ka, *kb, *kc;
    CK(hipMalloc(&dzA, arena * 2)); CK(hipMalloc(&xA, arena * 2)); CK(hipMalloc(&dxA, arena * 2));
    CK(hipMalloc(&ka, 8192)); CK(hipMalloc(&kb, 8192)); CK(hipMalloc(&kc, 65536));
    CK(hipMemset(dzA, 0x3c, arena * 2)); CK(hipMemset(xA, 0x3d, arena * 2)); CK(hipMemset(ka, 0, 8192)); CK(hipMemset(kb, 0, 8192)); CK(hipMemset(kc, 0, 65536));
    printf("%-16s %8s | %s\n", "rows x C", "MB", "GB/s (cold): copy | prod | fma3 U1 U2 U4 U4c1024 U4c4096 | nt U2 U4 U8 U8c1024 | chunk U4 U8nt");
    for (auto& s : shapes) {
        const int64_t rows = s[0]; const int C = (int)s[1]; const double bytes = 3.0 * rows * C * 2;
        const int cv = C / 8; const int64_t nel = rows * C; const int nslot = (int)(arena / nel);
        int slot = 0;
        auto next = [&](uint16_t*& dz, uint16_t*& x, uint16_t*& dx) { slot = (slot + 1) % nslot; dz = dzA + slot * nel; x = xA + slot * nel; dx = dxA + slot * nel; };
        auto run = [&](auto kern, int cap) {
            Walk w = make_walk(rows, cv, cap, 1);
            float ms = timeit([&] { uint16_t *dz, *x, *dx; next(dz, x, dx); hipLaunchKernelGGL(kern, dim3(w.gx, w.gy), dim3(256), 0, 0, dz, x, ka, kb, kc, dx, rows, C, w.tbx, w.tby); }, 20);
            return bytes / ms / 1e6;
        };
        auto runc = [&](auto kern, int cap) {
            Walk w = make_walk(rows, cv, cap, 1);
            int64_t rpb = (rows + w.gy - 1) / w.gy; rpb = (rpb + w.tby - 1) / w.tby * w.tby;
            int gy = (int)((rows + rpb - 1) / rpb);
            float ms = timeit([&] { uint16_t *dz, *x, *dx; next(dz, x, dx); hipLaunchKernelGGL(kern, dim3(w.gx, gy), dim3(256), 0, 0, dz, x, ka, kb, kc, dx, rows, C, w.tbx, w.tby, rpb); }, 20);
            return bytes / ms / 1e6;
        };
        const int64_t nvec = nel / 8;
        float cms = timeit([&] { uint16_t *dz, *x, *dx; next(dz, x, dx); hipLaunchKernelGGL(copy_k, dim3(256 * 8), dim3(256), 0, 0, (const uint4*)dz, (const uint4*)x, (uint4*)dx, nvec); }, 20);
        printf("%7lld x %-6d %8.1f | %6.0f | %6.0f | %6.0f %6.0f %6.0f %6.0f %6.0f | %6.0f %6.0f %6.0f %6.0f | %6.0f %6.0f\n", (long long)rows, C, bytes / 1e6, bytes / cms / 1e6,
               run(apply_prod_k, 2048),
               run(apply_k<1, false>, 2048), run(apply_k<2, false>, 2048), run(apply_k<4, false>, 2048), run(apply_k<4, false>, 1024), run(apply_k<4, false>, 4096),
               run(apply_k<2, true>, 2048), run(apply_k<4, true>, 2048), run(apply_k<8, true>, 2048), run(apply_k<8, true>, 1024),
               runc(apply_chunk_k<4, false>, 2048), runc(apply_chunk_k<8, true>, 2048));
        fflush(stdout);
    }
    return 0;
}
