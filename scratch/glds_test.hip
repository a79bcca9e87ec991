// Probe: does buffer_load ... lds write zeros for out-of-range lanes, or leave LDS untouched?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const char* src, unsigned bytes, unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    auto r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    unsigned voff = (threadIdx.x & 1) ? 0xC0000000u : threadIdx.x * 16u;   // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    char* d; unsigned* o; unsigned h[256];
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    unsigned init[1024]; for (int i = 0; i < 1024; ++i) init[i] = 0x1000 + i;
    hipMemcpy(d, init, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1024u, o);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 6; ++l) printf("lane %d: %08x %08x %08x %08x\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
