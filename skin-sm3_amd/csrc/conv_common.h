// Pieces shared by the gather-GEMM convolution kernels (conv_igemm.hip: 128-row tiles, 4 waves; conv_wgrad.hip).
#pragma once
#include "common.h"

namespace sm3conv {

struct ConvParams {
    const char* x;
    const char* w;
    char* y;
    const char* addend;
    float* partials;
    int M, Hi, Wi, Ci, Co;
    int sy, sx, ntaps;
    int dy[SM3_MAX_TAPS], dx[SM3_MAX_TAPS], wtap[SM3_MAX_TAPS];
    int w_row_stride;
    int Wout, HWout, osy, osx, ooy, oox;
    int HoWo, Wo;
    FastDiv div_HoWo, div_Wo;
    int kchunks;  // K-steps per tap = Ci*sizeof(T)/128
    int tilesM, tilesN;
    FastDiv div_tilesN;  // conv_igemm_v2 only
    uint32_t x_bytes, w_bytes;  // buffer-descriptor extents (< 3 GB)
    // optional fusion of the NEXT BatchNorm-backward's first phase into this (data-gradient) epilogue
    const uint8_t* fz_mask;     // relu bits of that BN's output (1 byte per 16-byte vector), or null
    const char* fz_x;           // that BN's input (its conv's output), same indexing as y; null = fusion off
    const float* fz_mean;
    const float* fz_invstd;
    float* fz_partials;         // [fz_row_off + tilesM][2][Co]
    int fz_row_off;
    int fz_view_tiles;          // 0: one BatchNorm batch; else tiles (of 128 rows) per view, two views back to back
    int fz_row_off1;            // first partial row of view 1
    // addend given only at the even (y, x) positions of the output, as a compact [N, add_sp_h, add_sp_w, Co] tensor
    // (the data gradient of a stride-2 1x1 downsample convolution); 0 = dense addend
    int add_sp_h, add_sp_w;
    // optional inference epilogue: y = relu?(acc * ep_scale[co] + ep_shift[co] (+ addend))  (eval-mode BatchNorm)
    const float* ep_scale;
    const float* ep_shift;
    int ep_relu;
    // ... or the BatchNorm's own tensors, scale/shift derived per channel in the epilogue (ep_rv != null)
    const float* ep_gamma;
    const float* ep_beta;
    const float* ep_rm;
    const float* ep_rv;
    float ep_eps;
    // train-mode form of the same epilogue (sm3_conv_bn_act_fused): ep_scale / ep_shift are [views][Co] (a tile's view as
    // for the fused BN-backward: fz_view_tiles), and the ReLU bits of the output go to ep_mask (1 byte per 16-byte vector)
    uint8_t* ep_mask;
    // ---- second K segment (SEG kernels, "BatchNorm backward by linearity": csrc/linbn.hip) ----------------------------
    // taps with tap_src[t] = 1 take their A rows from x1 ([pixels][Ci1], same pixel geometry) and their B rows from w1
    // ([Co][w1_row_stride]); nsteps_seg = sum over taps of that tap's K-steps.  Two views in one launch (fz_view_tiles > 0)
    // may use different banks: view 1's tiles read w + w_view_bytes / w1 + w1_view_bytes.
    const char* x1;
    const char* w1;
    uint32_t x1_bytes, w1_bytes;
    int Ci1, kchunks1, w1_row_stride, nsteps_seg;
    uint8_t tap_src[SM3_MAX_TAPS];
    uint32_t w_view_bytes, w1_view_bytes;
    const float* col_bias;  // [views][Co] f32 added to the accumulators (before the addend), or null
    int halo_rows, halo_a_bytes, halo_stat_off;  // kVarHalo (conv_igemm.hip): rows of the A image, its bytes, statistics scratch
    // kVarHaloBn (sm3_conv3x3_bnin): x is the PRODUCER convolution's raw output; its train-mode BatchNorm + ReLU
    // (in_scale / in_shift [views][Ci]) is applied to the staged A image in LDS, and the activation (in_act, same layout as
    // x) + its ReLU bits (in_mask, 1 byte per 16-byte vector) are written for the rows of this tile
    const float* in_scale;
    const float* in_shift;
    char* in_act;
    uint8_t* in_mask;
    // K order of a 16-bit stride-1 3 x 3 launch: 1 = chunk outer, tap inner -- the order of kVarHalo, taken by EVERY lean kernel
    // such a launch can reach (general gather, 4-stage deep, 8-wave; every epilogue), so that the bits of a layer's output do
    // not depend on which of them the grid size selects (B vs 2B, both views in one batch vs per-view passes); 0 = tap outer
    int kord;
    int dbg;  // SM3_CONV_DBG: experiment switches of the split K loop (bit 0: consumer waves at s_setprio 1)
};

// eval-mode BatchNorm as y = x * scale + shift, the arithmetic of bn_eval_kernel (bn.hip)
__device__ __forceinline__ void ep_affine(const ConvParams& p, int c, float& scale, float& shift) {
    if (p.ep_rv) {
        const float invstd = 1.f / sqrtf(p.ep_rv[c] + p.ep_eps);
        const float g = p.ep_gamma ? p.ep_gamma[c] : 1.f, b = p.ep_beta ? p.ep_beta[c] : 0.f;
        scale = g * invstd;
        shift = b - p.ep_rm[c] * g * invstd;
    } else {
        scale = p.ep_scale[c];
        shift = p.ep_shift[c];
    }
}

template <typename T>
__device__ __forceinline__ void mma_frag(const uint4& a, const uint4& b, f32x16& c);

template <>
__device__ __forceinline__ void mma_frag<bf16_t>(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_frag<f16_t>(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_frag<float>(const uint4& a, const uint4& b, f32x16& c) {
    // lane (r, h) holds k = 8*kk + 4*h + {0,1,2,3}: the j-th MFMA uses element j of both fragments,
    // so A and B agree on k and the four instructions together cover 8 consecutive k.
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
}

// 16 x 16 x 32 form of the 16-bit MFMA (kVarM16 of conv_igemm.hip): A / B fragments of 8 consecutive k per lane
template <typename T>
__device__ __forceinline__ void mma_frag16(const uint4& a, const uint4& b, f32x4& c);
template <>
__device__ __forceinline__ void mma_frag16<bf16_t>(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_frag16<f16_t>(const uint4& a, const uint4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_frag16<float>(const uint4&, const uint4&, f32x4&) {}  // (never instantiated for f32)

// byte offset of 16-byte chunk `c` of row `r` inside a [rows][128 B] LDS tile.  Two rows share a
// 256-byte bank row, so the swizzle key is the row pair: the 16 lanes of every ds_read_b128 group
// then hit 16 distinct 16-byte slots.
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + (((c ^ (r >> 1)) & 7) << 4); }

constexpr uint32_t kOOB = 0xC0000000u;  // voffset beyond any tensor (< 3 GB, checked on the host): reads zeros

// LDS-DMA: buffer_load_dwordx4 ... lds.  One wave-instruction moves 8 tile rows x 128 B = 1 KiB straight from
// global memory into LDS (destination = wave-uniform base + lane*16, source = per-lane offset), with no VGPR
// staging and no ds_write (whose VGPR->LDS transfer, ~13 cycles per KiB, made the register-staged version of this
// kernel LDS-bound: 830 write + 512 read LDS cycles against 1024 MFMA cycles per K-step pair).  The XOR swizzle
// is applied on the SOURCE side: lane (row, pos) fetches chunk pos ^ key(row), so it lands where lds_off(row,
// chunk) expects it.  Rows outside the image (padding), beyond M or beyond Cout use an out-of-range offset: the
// buffer range check makes the DMA write zeros (verified on gfx950, scratch/glds_test.hip) -- no branches.
//
// The DMA is issued from inline asm on purpose: through the builtin, hipcc orders every later ds_read behind the
// DMA with s_waitcnt vmcnt(0) (it cannot see that the DMA fills the OTHER stage), which serialises load and
// compute.  In asm the compiler does not track it; we drain it ourselves (vmcnt(0)) right before the barrier that
// publishes the stage.  M0 (LDS destination base) is saved/restored inside the statement (hipcc reserves it).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t lds_wave_base, uint32_t voff, uint32_t soff) {
    uint32_t keep;
    soff = __builtin_amdgcn_readfirstlane(soff);                    // wave-uniform by construction: keep them in
    lds_wave_base = __builtin_amdgcn_readfirstlane(lds_wave_base);  // SGPRs whatever the divergence analysis thinks
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_wave_base), "v"(voff), "s"(rsrc), "s"(soff)
        : "memory");
}
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


}  // namespace sm3conv
