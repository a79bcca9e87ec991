// Does the packed conversion (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32) round every float like the scalar one?  All 2^32 bit patterns.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned long long* bad, uint32_t* first) {
    const uint64_t n = 1ull << 32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const float f = __uint_as_float((uint32_t)i);
        const uint16_t hs = __builtin_bit_cast(uint16_t, (_Float16)f);
        const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{f, f}, f16x2));
        const uint16_t bs = __builtin_bit_cast(uint16_t, (__bf16)f);
        const uint32_t bp = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{f, f}, bf16x2));
        const bool nanf_ = (f != f);
        if (!nanf_ && ((hp & 0xffff) != hs || (hp >> 16) != hs)) { if (atomicAdd(&bad[0], 1ull) == 0) first[0] = (uint32_t)i; }
        if (!nanf_ && ((bp & 0xffff) != bs || (bp >> 16) != bs)) { if (atomicAdd(&bad[1], 1ull) == 0) first[1] = (uint32_t)i; }
    }
}
int main() {
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, 16); hipMalloc(&first, 8); hipMemset(bad, 0, 16); hipMemset(first, 0, 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, first);
    unsigned long long hb[2]; uint32_t hf[2];
    hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 8, hipMemcpyDeviceToHost);
    printf("f16: %llu mismatches (first bits 0x%08x)   bf16: %llu mismatches (first bits 0x%08x)\n", hb[0], hf[0], hb[1], hf[1]);
    return 0;
}
