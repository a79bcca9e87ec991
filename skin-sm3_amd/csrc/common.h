// Shared device/host helpers for the SM3 HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sm3_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

struct bf16_t {
    uint16_t v;
};
// IEEE half: the reference's own AMP storage type (torch.cuda.amp.autocast(), tools/backbone_train.py:27,98) -- 11 bits of
// significand against bf16's 8, at the price of a narrow exponent (dynamic loss scaling: sm3_loss_scale_update)
struct f16_t {
    uint16_t v;
};
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even; NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}

// both halves in ONE v_cvt_pk_bf16_f32 (a vector conversion: written as two scalar casts + shift/or the compiler emits two
// conversions and a v_or_sdwa per pair -- three instructions where one does; same round-to-nearest-even per element)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

__device__ __forceinline__ float f16_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) {  // round-to-nearest-even (v_cvt_f16_f32); overflow -> inf
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{lo, hi}, f16x2_t));
}

// ReLU as ONE v_max_f32: fmaxf(v, 0) costs a canonicalising v_max in front of the v_max wherever the compiler cannot prove
// its operand quiet (values unpacked from 16-bit words), and it folds v_med3(v, 0, inf) back into the same pair -- so the
// instruction is written out.  max(0, NaN) = 0, as fmaxf(NaN, 0) is.
__device__ __forceinline__ float relu_f32(float v) {
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
// v where bit e of m is set, else +0: v_bfe_i32 (0 / -1) + v_and instead of v_and + v_cmp + v_cndmask
__device__ __forceinline__ float keep_if_bit(float v, unsigned m, int e) {
    return __uint_as_float(__float_as_uint(v) & (uint32_t)__builtin_amdgcn_sbfe((int)m, (unsigned)e, 1u));
}

template <typename T>
struct ElemTraits;
template <>
struct ElemTraits<float> {
    static constexpr int kPer16B = 4;
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ float round(float v) { return v; }
};
template <>
struct ElemTraits<bf16_t> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(p->v); }
    __device__ static __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

template <>
struct ElemTraits<f16_t> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float load(const f16_t* p) { return f16_to_f32(p->v); }
    __device__ static __forceinline__ float round(float v) { return f16_to_f32(f32_to_f16(v)); }
};

// two floats -> one dword of T (16-bit types), and v_dot2 on such dwords: what the lean convolution epilogue needs
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <>
__device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <>
__device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }
template <typename T>
__device__ __forceinline__ float dot2acc(uint32_t a, uint32_t b, float acc);
template <>
__device__ __forceinline__ float dot2acc<bf16_t>(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}
template <>
__device__ __forceinline__ float dot2acc<f16_t>(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), acc, false);
}
template <typename T>
__device__ __forceinline__ uint32_t ones2();  // (1, 1) as a dword of T
template <>
__device__ __forceinline__ uint32_t ones2<bf16_t>() { return 0x3f803f80u; }
template <>
__device__ __forceinline__ uint32_t ones2<f16_t>() { return 0x3c003c00u; }

// 16 bytes of T -> floats
template <typename T>
__device__ __forceinline__ void unpack16(const uint4& u, float* f);
template <>
__device__ __forceinline__ void unpack16<float>(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x);
    f[1] = __uint_as_float(u.y);
    f[2] = __uint_as_float(u.z);
    f[3] = __uint_as_float(u.w);
}
template <>
__device__ __forceinline__ void unpack16<bf16_t>(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16);
    f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16);
    f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16);
    f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16);
    f[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <>
__device__ __forceinline__ void unpack16<f16_t>(const uint4& u, float* f) {
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = f16_to_f32((uint16_t)(w[i] & 0xffffu));
        f[2 * i + 1] = f16_to_f32((uint16_t)(w[i] >> 16));
    }
}
template <typename T>
__device__ __forceinline__ uint4 pack16(const float* f);
template <>
__device__ __forceinline__ uint4 pack16<float>(const float* f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <>
__device__ __forceinline__ uint4 pack16<bf16_t>(const float* f) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                      pack_bf16x2(f[6], f[7]));
}

template <>
__device__ __forceinline__ uint4 pack16<f16_t>(const float* f) {
    return make_uint4(pack_f16x2(f[0], f[1]), pack_f16x2(f[2], f[3]), pack_f16x2(f[4], f[5]), pack_f16x2(f[6], f[7]));
}

// 16-byte global accesses for the streaming kernels.  NT = non-temporal: the data is touched once per kernel,
// and the hint is worth +10..18 % on cold >100 MB tensors (scratch/stream_bench.hip: 5.05 -> 5.97 TB/s with 8 in flight).
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ldg16(const void* p) {
    if constexpr (NT) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const uint4*>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void stg16(void* p, const uint4& v) {
    if constexpr (NT) {
        const u32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<u32x4_t*>(p));
    } else {
        *reinterpret_cast<uint4*>(p) = v;
    }
}

// n / d for 0 <= n < 2^31 with a precomputed multiplier (host: make_fastdiv)
struct FastDiv {
    uint32_t mul, shr, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) {
        f.mul = 0;
        f.shr = 0;
        return f;
    }
    uint32_t l = 0;
    while ((1u << l) < d) ++l;
    uint32_t p = 31 + l;
    uint64_t m = ((1ull << p) + d - 1) / d;
    f.mul = (uint32_t)m;
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Fixed-order sum of n fp32 terms by ONE 256-thread workgroup (every thread must call it): thread t adds terms t, t + 256, ...
// in fp64, the 64 lanes of a wave fold by the xor butterfly, the four wave sums are added in wave order.  The result is a
// function of the terms alone -- no atomics, no arrival order.  `sh4` = 4 doubles of LDS.  (The reference's loss is one
// fixed-order reduction as well: nn.CrossEntropyLoss over a materialised logits tensor, tools/backbone_train.py:531.)
__device__ __forceinline__ double block256_ordered_sum(const float* __restrict__ terms, int n, double* sh4) {
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)terms[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = a;
    __syncthreads();
    const double s = ((sh4[0] + sh4[1]) + sh4[2]) + sh4[3];
    __syncthreads();
    return s;
}

// expands CALL(T) for the element type of `dtype`; returns SM3_EDTYPE for anything else
#define SM3_DISPATCH_DTYPE(dtype, CALL)          \
    do {                                         \
        if ((dtype) == SM3_F32) { CALL(float); } \
        else if ((dtype) == SM3_BF16) { CALL(bf16_t); } \
        else if ((dtype) == SM3_F16) { CALL(f16_t); }   \
        else return SM3_EDTYPE;                  \
    } while (0)
#define SM3_DTYPE_OK(dtype) ((dtype) == SM3_F32 || (dtype) == SM3_BF16 || (dtype) == SM3_F16)

#define SM3_CHECK_LAUNCH()                          \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)
