// Direct 7x7 / stride 2 / pad 3 stem convolution on MFMA (gfx950), forward and weight gradient, straight from the
// NCHW fp32 image batch the loader delivers -- no im2col matrix in HBM (the round-1 path wrote and re-read a
// [N*Ho*Wo, 192] matrix: 1.2 GB per view at B = 256, the largest tensor of the step and the one that capped the
// two-views-in-one-batch mode at B <= 256).
//
//   y[n, oy, ox, co] = sum_{kh, kw, c} x[n, c, 2*oy - 3 + kh, 2*ox - 3 + kw] * w[co][kh][kw][c]
//
// One tile = up to 128 consecutive output pixels of one output row (n, oy).  Its 7 x 3 input rows ("groups"
// g = kh*3 + c) are staged in LDS as fp32 with zero padding; group g's row holds columns ix = 2*x0 - 3 + j.
// K is ordered (g, kw) with kw padded to 8: the 8 k-values of one MFMA operand fragment are then 8 CONSECUTIVE
// floats of a staged row (A[pixel][8h + j] = row[g = 2*ks + h][2*px + j]), read with four ds_read_b64 and rounded
// to bf16 in registers -- the im2col matrix exists only as those fragments.  The filter bank (64 x 176 bf16 in the
// same K order, the 8th tap and the 22nd group zero) lives in registers for the whole kernel.
// Forward epilogue: bf16 NHWC rows + per-tile BatchNorm partial sums, as the gather-GEMM's.
//
// Weight gradient: dW[co][k] = sum_pixels dxo[pix][co] * patch[pix][k] with the pixel axis as the MFMA K; the
// dxo operand is the stem BatchNorm's input gradient, computed on the fly from dz and the saved pre-BN output
// (dx = g*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)): sm3_bn_bwd_apply's arithmetic) while the tile is staged --
// the stem has no data gradient, so that tensor has exactly one consumer and is never written to HBM.
//
// Reference call sites replaced: nn.Conv2d(3, 64, 7, 2, 3) forward (src/models/resnet.py:208-210,294) and its
// autograd weight gradient, plus phase 2 of bn1's backward (resnet.py:211,295).
#include <atomic>

#include "conv_common.h"

namespace {

constexpr int PW = 264;   // floats per staged row: 2*127 + 8 = 262 needed
constexpr int NG = 21;    // (kh, c) groups
constexpr int KS = 11;    // MFMA K-steps of 16 = two groups; group 21 does not exist (zero fragment)
constexpr int KPAD = KS * 16;  // 176
constexpr int OUT_PITCH = 144;  // bytes per pixel row of the staged output tile (64 bf16 + 16: bank spread)

struct StemTile {
    int n, oy, x0;
};
__device__ __forceinline__ StemTile decode_tile(long t, int xblocks, int Ho) {
    StemTile s;
    const int xb = (int)(t % xblocks);
    const long r = t / xblocks;
    s.oy = (int)(r % Ho);
    s.n = (int)(r / Ho);
    s.x0 = xb * 128;
    return s;
}

// 7 x 3 input rows of a tile -> LDS (fp32, zero padded); row g = kh*3 + c, column j <-> ix = 2*x0 - 3 + j.
// Thread t owns column t of every row (and, for t < 168, one of the 8 tail columns 256..263 of row t/8).  Split into
// a load half (22 independent global loads into registers) and a store half: the kernels issue the loads of tile
// i+1 before the MFMA loop of tile i and store them after its epilogue, so the memory latency of a tile (15 us when
// the staging was a plain load-store loop: hipcc keeps ONE load in flight per thread there) hides under the
// previous tile's compute.
struct PatchRegs {
    float v[NG];
    float vt;
};
__device__ __forceinline__ void load_patch(PatchRegs& r, const float* __restrict__ x, const StemTile& t, int H, int W) {
    const int tid = threadIdx.x;
    const float* xn = x + (long)t.n * 3 * H * W;
    const int ix = 2 * t.x0 - 3 + tid;
    const bool okx = (unsigned)ix < (unsigned)W;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int kh = g / 3, c = g - 3 * kh;  // compile-time after unrolling
        const int iy = 2 * t.oy - 3 + kh;
        const bool ok = okx && (unsigned)iy < (unsigned)H;
        r.v[g] = ok ? xn[((long)c * H + iy) * W + ix] : 0.f;
    }
    r.vt = 0.f;
    if (tid < NG * 8) {
        const int gt = tid >> 3, jt = 256 + (tid & 7);
        const int kh = gt / 3, c = gt - 3 * kh;
        const int iy = 2 * t.oy - 3 + kh, ixt = 2 * t.x0 - 3 + jt;
        if ((unsigned)iy < (unsigned)H && (unsigned)ixt < (unsigned)W) r.vt = xn[((long)c * H + iy) * W + ixt];
    }
}
__device__ __forceinline__ void store_patch(float* patch, const PatchRegs& r) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int g = 0; g < NG; ++g) patch[g * PW + tid] = r.v[g];
    if (tid < NG * 8) patch[(tid >> 3) * PW + 256 + (tid & 7)] = r.vt;
}

template <typename T>
__global__ __launch_bounds__(256, 3) void stem_fwd_kernel(const float* __restrict__ x, const uint4* __restrict__ w,
                                                       T* __restrict__ y, float* __restrict__ partials, int N,
                                                       int H, int W, int Ho, int Wo, int xblocks, long tiles) {
    __shared__ __attribute__((aligned(16))) float patch[NG * PW];  // reused as the bf16 output tile
    __shared__ float sStat[4][64][2];
    // The filter bank (64 x 176 values, 22.5 KB) lives in LDS for the whole (persistent) workgroup: as 88 registers per
    // lane it held the kernel at 184 VGPRs = two workgroups per CU, and the tile loop is bound by the latency of its
    // one-tile-ahead patch loads.  Row pitch 184 values (368 B): the 16 rows of a ds_read_b128 group fall on 16
    // different 16-byte bank slots.
    constexpr int WP = KPAD + 8;
    __shared__ __attribute__((aligned(16))) uint16_t sW[64 * WP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    for (int i = tid; i < 64 * (KPAD / 8); i += 256) {
        const int co = i / (KPAD / 8), c8 = i - co * (KPAD / 8);
        *reinterpret_cast<uint4*>(&sW[co * WP + c8 * 8]) = w[i];
    }
    // lane (col, h) reads w[nb*32 + col][ks*16 + 8h .. +8) for every (ks, nb)
    const uint16_t* wlane = &sW[col * WP + 8 * h];

    PatchRegs pre;
    if ((long)blockIdx.x < tiles) {
        load_patch(pre, x, decode_tile(blockIdx.x, xblocks, Ho), H, W);
        store_patch(patch, pre);
    }
    __syncthreads();
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const StemTile tl = decode_tile(t, xblocks, Ho);
        const long tn = t + gridDim.x;
        if (tn < tiles) load_patch(pre, x, decode_tile(tn, xblocks, Ho), H, W);  // lands during this tile's compute
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const int px = 32 * wave + col;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int g = 2 * ks + h;
            uint4 a = make_uint4(0, 0, 0, 0);
            if (g < NG) {
                const float2* src = reinterpret_cast<const float2*>(patch + g * PW + 2 * px);
                const float2 v0 = src[0], v1 = src[1], v2 = src[2], v3 = src[3];
                a = make_uint4(pack2<T>(v0.x, v0.y), pack2<T>(v1.x, v1.y), pack2<T>(v2.x, v2.y), pack2<T>(v3.x, v3.y));
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                sm3conv::mma_frag<T>(a, *reinterpret_cast<const uint4*>(wlane + nb * 32 * WP + ks * 16), acc[nb]);
        }
        __syncthreads();  // everyone is done reading the patch: its LDS becomes the output tile
        char* sOut = reinterpret_cast<char*>(patch);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                const uint16_t b = (uint16_t)pack2<T>(acc[nb][r], 0.f);
                *reinterpret_cast<uint16_t*>(sOut + row * OUT_PITCH + (nb * 32 + col) * 2) = b;
                if (tl.x0 + row < Wo) {  // statistics of the stored (rounded) values of real pixels only
                    const float v = ElemTraits<T>::round(acc[nb][r]);
                    s1 += v;
                    s2 += v * v;
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                sStat[wave][nb * 32 + lane][0] = s1;
                sStat[wave][nb * 32 + lane][1] = s2;
            }
        }
        __syncthreads();
        if (partials && tid < 128) {
            const int c = tid & 63, st = tid >> 6;
            partials[(t * 2 + st) * 64 + c] = (sStat[0][c][st] + sStat[1][c][st]) + (sStat[2][c][st] + sStat[3][c][st]);
        }
        {
            const int ch = tid & 7, r0 = tid >> 3;
            const long pix0 = ((long)tl.n * Ho + tl.oy) * Wo + tl.x0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = r0 + 32 * k;
                if (tl.x0 + row < Wo)
                    stg16<true>(y + (pix0 + row) * 64 + ch * 8,
                                *reinterpret_cast<const uint4*>(sOut + row * OUT_PITCH + ch * 16));
            }
        }
        __syncthreads();  // the output tile has been read: the next tile's patch may overwrite it
        if (tn < tiles) store_patch(patch, pre);
        __syncthreads();
    }
}

// ---- weight gradient with the BatchNorm-backward apply fused into the dxo operand -----------------------------
struct StemWgradParams {
    const float* x;
    const char* dz;
    const char* xo;
    const float *mean, *invstd, *gamma;       // [views][64], [views][64], [64] (nullable)
    const double *gsums, *lsums;              // [views][128]
    float *dgamma, *dbeta;                    // nullable
    float* dw;                                // [64][147] fp32, accumulated into (float atomics) -- or, dw_slabs given,
    float* dw_slabs;                          // workgroup b STORES its partial product to dw_slabs + b * 64 * 147 (no atomics)
    double inv_count;
    int N, H, W, Ho, Wo, xblocks, n_per_view, views;
    long tiles;
};

__device__ __forceinline__ uint32_t swz128(int row) { return (uint32_t)((row >> 1) & 1) << 6; }

template <typename T>
__global__ __launch_bounds__(256, 3) void stem_wgrad_kernel(const StemWgradParams p) {
    __shared__ __attribute__((aligned(16))) float patch[NG * PW];
    __shared__ __attribute__((aligned(16))) char sD[128 * 128];  // dxo tile [pixel][64 co] bf16, swizzled for tr reads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 1, kb0 = 3 * (wave >> 1);  // this wave: co block cb, k blocks kb0 .. kb0+2

    if (blockIdx.x == 0 && tid < 64 && p.lsums) {  // parameter gradients of bn1 from the LOCAL sums, once
        for (int v = 0; v < p.views; ++v) {
            if (p.dbeta) atomicAdd(&p.dbeta[tid], (float)p.lsums[v * 128 + tid]);
            if (p.dgamma) atomicAdd(&p.dgamma[tid], (float)p.lsums[v * 128 + 64 + tid]);
        }
    }

    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // staging role: 16-byte channel vector ch of rows r0 + 32k
    const int ch = tid & 7, r0 = tid >> 3;
    // per-channel coefficients of the fused BN-backward apply, [A | B | C][64], refreshed when the view changes
    // (kept in LDS: as 32 registers per lane they helped pin the kernel at two workgroups per CU)
    __shared__ __attribute__((aligned(16))) float sCoef[3][64];
    int cur_view = -1;

    // transposing-read addressing of the dxo tile (as conv_wgrad.hip): 16-lane group g16 reads a 4(k) x 16(co) block
    const int g16 = lane >> 4, ii = lane & 15, qq = ii >> 2, pq = ii & 3, hh = g16 >> 1;
    const uint32_t ra_off = (uint32_t)(8 * hh + qq) * 128u + ((((uint32_t)(cb * 32 + 16 * (g16 & 1) + 4 * pq)) * 2u) ^ swz128(qq));
    // B operand: lane (n = lane & 31, half h) of k block kb supplies k = 32*kb + n -> (group, kw)
    const int bn = lane & 31, bh = lane >> 5;

    // operand loads of a tile: its patch and the (dz, xo) rows of this thread's channel vector
    // Only the patch is prefetched across the MFMA loop; the (dz, xo) vectors are fetched when the tile is staged.  Holding
    // them too (32 more registers) pinned the kernel at two workgroups per CU; with three, the other workgroups cover
    // that fetch.
    PatchRegs pre;
    auto load_tile = [&](const StemTile& tl) { load_patch(pre, p.x, tl, p.H, p.W); };
    auto store_tile = [&](const StemTile& tl) {
        const int view = tl.n / p.n_per_view;
        if (view != cur_view) {  // tiles are view-major: happens once or twice per workgroup (block-uniform branch)
            cur_view = view;
            if (tid < 64) {
                // dx = k0*(dz - k1) - (x - mu)*q  =  A*dz - B*x + C   with A = k0, B = q, C = q*mu - k0*k1
                const int c = tid;
                const float is = p.invstd[view * 64 + c];
                const float g = p.gamma ? p.gamma[c] : 1.f;
                const float k0c = g * is;
                const float k1c = (float)(p.gsums[view * 128 + c] * p.inv_count);
                const float qc = k0c * is * (float)(p.gsums[view * 128 + 64 + c] * p.inv_count);
                sCoef[0][c] = k0c;
                sCoef[1][c] = qc;
                sCoef[2][c] = qc * p.mean[view * 64 + c] - k0c * k1c;
            }
            __syncthreads();
        }
        store_patch(patch, pre);
        float cA[8], cB[8], cC[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            *reinterpret_cast<float4*>(cA + e) = *reinterpret_cast<const float4*>(&sCoef[0][ch * 8 + e]);
            *reinterpret_cast<float4*>(cB + e) = *reinterpret_cast<const float4*>(&sCoef[1][ch * 8 + e]);
            *reinterpret_cast<float4*>(cC + e) = *reinterpret_cast<const float4*>(&sCoef[2][ch * 8 + e]);
        }
        const long pix0 = ((long)tl.n * p.Ho + tl.oy) * p.Wo + tl.x0;
#pragma unroll
        for (int kb = 0; kb < 4; kb += 2) {  // two rows of this thread's channel vector at a time (register budget)
            uint4 gu[2], xu[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = r0 + 32 * (kb + k);
                const bool ok = tl.x0 + row < p.Wo;
                gu[k] = ok ? ldg16<true>(p.dz + ((pix0 + row) * 64 + ch * 8) * 2) : make_uint4(0, 0, 0, 0);
                xu[k] = ok ? ldg16<true>(p.xo + ((pix0 + row) * 64 + ch * 8) * 2) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = r0 + 32 * (kb + k);
                uint4 out = make_uint4(0, 0, 0, 0);
                if (tl.x0 + row < p.Wo) {
                    float g[8], xv[8];
                    unpack16<T>(gu[k], g);
                    unpack16<T>(xu[k], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = cA[e] * g[e] - cB[e] * xv[e] + cC[e];
                    out = pack16<T>(g);
                }
                *reinterpret_cast<uint4*>(sD + row * 128 + (((uint32_t)ch * 16u) ^ swz128(row))) = out;
            }
        }
    };
    if ((long)blockIdx.x < p.tiles) {
        const StemTile t0 = decode_tile(blockIdx.x, p.xblocks, p.Ho);
        load_tile(t0);
        store_tile(t0);
    }
    for (long t = blockIdx.x; t < p.tiles; t += gridDim.x) {
        const long tn = t + gridDim.x;
        StemTile nx = {0, 0, 0};
        if (tn < p.tiles) {
            nx = decode_tile(tn, p.xblocks, p.Ho);
            load_tile(nx);  // in flight during this tile's MFMA loop
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) {  // 16 pixels per step
            const char* a0 = sD + ra_off + s * 16 * 128;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * 128));
            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
            const uint4 fa = make_uint4(l2.x, l2.y, h2.x, h2.y);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = 32 * (kb0 + j) + bn;
                const int g = k >> 3, kw = k & 7;
                uint4 fb = make_uint4(0, 0, 0, 0);
                if (g < NG) {
                    const float* src = patch + g * PW + 2 * (16 * s + 8 * bh) + kw;
                    fb = make_uint4(pack2<T>(src[0], src[2]), pack2<T>(src[4], src[6]), pack2<T>(src[8], src[10]),
                                    pack2<T>(src[12], src[14]));
                }
                sm3conv::mma_frag<T>(fa, fb, acc[j]);
            }
        }
        __syncthreads();  // everyone is done reading this tile's operands
        if (tn < p.tiles) store_tile(nx);
    }

    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = 32 * (kb0 + j) + frow;
        const int g = k >> 3, kw = k & 7;
        if (g >= NG || kw >= 7) continue;
        const int kh = g / 3, c = g - 3 * kh;
        const int kcol = (kh * 7 + kw) * 3 + c;  // master layout [co][kh][kw][c]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (p.dw_slabs) p.dw_slabs[(long)blockIdx.x * (64 * 147) + co * 147 + kcol] = acc[j][r];
            else atomicAdd(p.dw + co * 147 + kcol, acc[j][r]);
        }
    }
}


// ==== the 16-bit image path (round 6) ===================================================================================
// The kernels above convert the fp32 images on the fly, per tile: 22 global loads and 22 LDS stores per thread to stage a
// patch, fp32 -> 16-bit in front of every MFMA fragment, forward and again in the weight gradient.  Here the images are
// rounded ONCE per step (sm3_stem_image_prep: the same round-to-nearest-even, so every output bit stays what it was) into
//     ximg [N][3][H][Wp] 16-bit,  ximg[.., q] = x[.., q - 3] for 3 <= q < W + 3, else 0,   Wp = round_up(W + 6, 8)
// i.e. with the convolution's left / right zero padding materialised.  The staged row of tile x0 is then the 264
// CONSECUTIVE values ximg[.., 2*x0 .. 2*x0 + 264): 33 aligned 16-byte chunks that go global -> LDS by LDS-DMA (no VGPRs,
// no ds_write; rows above / below the image and chunks beyond Wp read as zero through the buffer range check), the NEXT
// tile's patch while this one is computed.  An MFMA A fragment (8 consecutive k of one pixel) is 16 bytes at a 4-byte
// aligned LDS address: four dword reads, no conversion.  sm3_stem_image_prep also replaces the torch.cat of the two views
// of a branch (it reads them from two pointers), so the step moves fewer bytes than before.
constexpr int PCH = 33;               // 16-byte chunks per staged row
constexpr int PROW16 = PCH * 16;      // bytes per staged row (264 values)
constexpr int PDMA = 11;              // wave-instructions per patch: 11 x 64 lanes >= 21 x 33 = 693 chunks
constexpr int PATCH16 = PDMA * 1024;  // bytes per patch buffer

template <typename T>
__global__ __launch_bounds__(256) void stem_image_prep_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                              uint4* __restrict__ out, int n_per_view, int H, int W, int Wp8,
                                                              long chunks) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunks) return;
    const int j = (int)(i % Wp8);
    const long r = i / Wp8;                 // row (n, c, iy)
    const long rows_img = 3L * H;
    const int n = (int)(r / rows_img);
    const long rr = r - (long)n * rows_img;
    const float* src = (n < n_per_view ? x0 + (long)n * rows_img * W : x1 + (long)(n - n_per_view) * rows_img * W) + rr * W;
    float v[8];
    const int q0 = 8 * j - 4;  // the chunk's values are src[q0 + 1 .. q0 + 8]
    if ((W & 3) == 0 && q0 >= 0 && q0 + 8 < W && (((uintptr_t)src) & 15) == 0) {
        // interior chunk of a 16-byte aligned row: two aligned 16-byte loads + one float instead of eight scalar loads at a
        // 32-byte stride (which ran this pass at half the HBM rate)
        const float4 a = *reinterpret_cast<const float4*>(src + q0), b = *reinterpret_cast<const float4*>(src + q0 + 4);
        v[0] = a.y; v[1] = a.z; v[2] = a.w; v[3] = b.x; v[4] = b.y; v[5] = b.z; v[6] = b.w; v[7] = src[q0 + 8];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int q = 8 * j + e - 3;
            v[e] = (unsigned)q < (unsigned)W ? src[q] : 0.f;
        }
    }
    out[i] = make_uint4(pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7]));
}

// tile index -> (n, oy, x0) with 32-bit multiply-high divisions (the 64-bit % and / of decode_tile cost ~600 scalar
// instructions per tile, twice per tile: current + next)
struct TileDiv {
    FastDiv xb, ho;
};
__device__ __forceinline__ StemTile decode_tile32(uint32_t t, const TileDiv& d) {
    StemTile s;
    const uint32_t r = fdiv(t, d.xb), xb = t - r * d.xb.d;
    const uint32_t n = fdiv(r, d.ho);
    s.oy = (int)(r - n * d.ho.d);
    s.n = (int)n;
    s.x0 = (int)xb * 128;
    return s;
}

// the PDMA wave-instructions of one patch, dealt round-robin to the 4 waves; lane (row g, chunk j) of instruction k
__device__ __forceinline__ void patch_dma(__amdgpu_buffer_rsrc_t rs, uint32_t lds_patch, const StemTile& t, int H, int Wp,
                                          int wave, int lane) {
    for (int k = wave; k < PDMA; k += 4) {
        const int idx = k * 64 + lane;
        const int g = idx / PCH, j = idx - g * PCH;
        const int kh = g / 3, c = g - 3 * kh;
        const int iy = 2 * t.oy - 3 + kh, q = 2 * t.x0 + 8 * j;
        const bool ok = g < NG && (unsigned)iy < (unsigned)H && q < Wp;
        const uint32_t off = ok ? (uint32_t)((((long)t.n * 3 + c) * H + iy) * Wp + q) * 2u : sm3conv::kOOB;
        sm3conv::dma16(rs, lds_patch + (uint32_t)k * 1024u, off, 0u);
    }
}

constexpr int FWD16_LDS = 2 * PATCH16 + 128 * OUT_PITCH + 4 * 64 * 2 * 4 + 64 * (KPAD + 8) * 2;

template <typename T>
__global__ __launch_bounds__(256, 2) void stem_fwd16_kernel(const char* __restrict__ ximg, uint32_t ximg_bytes,
                                                          const uint4* __restrict__ w, T* __restrict__ y,
                                                          float* __restrict__ partials, int H, int Wp, int Ho, int Wo,
                                                          const TileDiv td, uint32_t tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem16[];
    char* sPatch = smem16;                                   // [2][PATCH16]
    char* sOut = smem16 + 2 * PATCH16;                       // [128][OUT_PITCH]
    float(*sStat)[64][2] = reinterpret_cast<float(*)[64][2]>(sOut + 128 * OUT_PITCH);
    constexpr int WP = KPAD + 8;
    uint16_t* sW = reinterpret_cast<uint16_t*>(sOut + 128 * OUT_PITCH + 4 * 64 * 2 * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    for (int i = tid; i < 64 * (KPAD / 8); i += 256) {
        const int co = i / (KPAD / 8), c8 = i - co * (KPAD / 8);
        *reinterpret_cast<uint4*>(&sW[co * WP + c8 * 8]) = w[i];
    }
    const uint16_t* wlane = &sW[col * WP + 8 * h];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)ximg, 0, ximg_bytes, 0x00020000);
    const uint32_t lds_patch = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)sPatch);

    int buf = 0;
    if (blockIdx.x < tiles) patch_dma(rs, lds_patch, decode_tile32(blockIdx.x, td), H, Wp, wave, lane);
    sm3conv::dma_drain();
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x, buf ^= 1) {
        const StemTile tl = decode_tile32(t, td);
        const uint32_t tn = t + gridDim.x;
        __syncthreads();       // this tile's patch has landed for everyone (every wave drained its pieces before it got here);
                               // everyone is done with the previous tile's patch and output tile
        if (tn < tiles)        // the next tile's patch goes into the other buffer while this one is computed
            patch_dma(rs, lds_patch + (uint32_t)((buf ^ 1) * PATCH16), decode_tile32(tn, td), H, Wp, wave, lane);
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const int px = 32 * wave + col;
        const char* prow = sPatch + buf * PATCH16 + 4 * px;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int g = 2 * ks + h;
            uint4 a = make_uint4(0, 0, 0, 0);
            if (g < NG) {
                const uint32_t* src = reinterpret_cast<const uint32_t*>(prow + g * PROW16);  // 8 consecutive k of pixel px
                a = make_uint4(src[0], src[1], src[2], src[3]);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                sm3conv::mma_frag<T>(a, *reinterpret_cast<const uint4*>(wlane + nb * 32 * WP + ks * 16), acc[nb]);
        }
        // epilogue: the arithmetic of stem_fwd_kernel (same rounding, same partial sums)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                const uint16_t b = (uint16_t)pack2<T>(acc[nb][r], 0.f);
                *reinterpret_cast<uint16_t*>(sOut + row * OUT_PITCH + (nb * 32 + col) * 2) = b;
                if (tl.x0 + row < Wo) {
                    const float v = ElemTraits<T>::round(acc[nb][r]);
                    s1 += v;
                    s2 += v * v;
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                sStat[wave][nb * 32 + lane][0] = s1;
                sStat[wave][nb * 32 + lane][1] = s2;
            }
        }
        // the next patch (issued before the MFMA loop) has had the loop and the staging above to land; waiting HERE, in front
        // of this tile's global stores, keeps the wait from ever covering a store's round trip
        sm3conv::dma_drain();
        __syncthreads();
        if (partials && tid < 128) {
            const int c = tid & 63, st = tid >> 6;
            partials[((long)t * 2 + st) * 64 + c] = (sStat[0][c][st] + sStat[1][c][st]) + (sStat[2][c][st] + sStat[3][c][st]);
        }
        {
            const int ch = tid & 7, r0 = tid >> 3;
            const long pix0 = ((long)tl.n * Ho + tl.oy) * Wo + tl.x0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = r0 + 32 * k;
                if (tl.x0 + row < Wo)
                    stg16<true>(y + (pix0 + row) * 64 + ch * 8,
                                *reinterpret_cast<const uint4*>(sOut + row * OUT_PITCH + ch * 16));
            }
        }
    }
}

struct StemWgrad16Params {
    StemWgradParams q;      // q.x unused
    const char* ximg;
    uint32_t ximg_bytes;
    int Wp;
    TileDiv td;
};

template <typename T>
__global__ __launch_bounds__(256, 3) void stem_wgrad16_kernel(const StemWgrad16Params pp) {
    const StemWgradParams& p = pp.q;
    __shared__ __attribute__((aligned(16))) char sPatch[2 * PATCH16];
    __shared__ __attribute__((aligned(16))) char sD[128 * 128];  // dxo tile [pixel][64 co], swizzled for tr reads
    __shared__ __attribute__((aligned(16))) float sCoef[3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave & 1, kb0 = 3 * (wave >> 1);

    if (blockIdx.x == 0 && tid < 64 && p.lsums) {  // parameter gradients of bn1 from the LOCAL sums, once, views in order
        float db = 0.f, dg = 0.f;
        for (int v = 0; v < p.views; ++v) {
            db += (float)p.lsums[v * 128 + tid];
            dg += (float)p.lsums[v * 128 + 64 + tid];
        }
        if (p.dbeta) atomicAdd(&p.dbeta[tid], db);
        if (p.dgamma) atomicAdd(&p.dgamma[tid], dg);
    }

    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    const int ch = tid & 7, r0 = tid >> 3;
    int cur_view = -1;
    const int g16 = lane >> 4, ii = lane & 15, qq = ii >> 2, pq = ii & 3, hh = g16 >> 1;
    const uint32_t ra_off = (uint32_t)(8 * hh + qq) * 128u + ((((uint32_t)(cb * 32 + 16 * (g16 & 1) + 4 * pq)) * 2u) ^ swz128(qq));
    const int bn = lane & 31, bh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pp.ximg, 0, pp.ximg_bytes, 0x00020000);
    const uint32_t lds_patch = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)sPatch);
    // B operand: lane (n = bn, half bh) of k block kb supplies k = 32 kb + bn -> (group g, kw): the 8 pixels of a fragment sit
    // at every other 16-bit value of the staged row -- dwords (16 s + 8 bh) + (kw >> 1) + i, low or high half by kw's parity
    uint32_t b_off[3], b_sel[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = 32 * (kb0 + j) + bn, g = k >> 3, kw = k & 7;
        b_off[j] = g < NG ? (uint32_t)(g * PROW16 + 4 * (8 * bh + (kw >> 1))) : 0xffffffffu;
        b_sel[j] = (kw & 1) ? 0x07060302u : 0x05040100u;
    }

    auto stage_dxo = [&](const StemTile& tl) {  // dx = A dz - B x + C per channel (the arithmetic of stem_wgrad_kernel)
        const int view = tl.n / p.n_per_view;
        if (view != cur_view) {
            cur_view = view;
            if (tid < 64) {
                const int c = tid;
                const float is = p.invstd[view * 64 + c];
                const float g = p.gamma ? p.gamma[c] : 1.f;
                const float k0c = g * is;
                const float k1c = (float)(p.gsums[view * 128 + c] * p.inv_count);
                const float qc = k0c * is * (float)(p.gsums[view * 128 + 64 + c] * p.inv_count);
                sCoef[0][c] = k0c;
                sCoef[1][c] = qc;
                sCoef[2][c] = qc * p.mean[view * 64 + c] - k0c * k1c;
            }
            __syncthreads();
        }
        float cA[8], cB[8], cC[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
            *reinterpret_cast<float4*>(cA + e) = *reinterpret_cast<const float4*>(&sCoef[0][ch * 8 + e]);
            *reinterpret_cast<float4*>(cB + e) = *reinterpret_cast<const float4*>(&sCoef[1][ch * 8 + e]);
            *reinterpret_cast<float4*>(cC + e) = *reinterpret_cast<const float4*>(&sCoef[2][ch * 8 + e]);
        }
        const long pix0 = ((long)tl.n * p.Ho + tl.oy) * p.Wo + tl.x0;
#pragma unroll
        for (int kb = 0; kb < 4; kb += 2) {
            uint4 gu[2], xu[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = r0 + 32 * (kb + k);
                const bool ok = tl.x0 + row < p.Wo;
                gu[k] = ok ? ldg16<true>(p.dz + ((pix0 + row) * 64 + ch * 8) * 2) : make_uint4(0, 0, 0, 0);
                xu[k] = ok ? ldg16<true>(p.xo + ((pix0 + row) * 64 + ch * 8) * 2) : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int row = r0 + 32 * (kb + k);
                uint4 out = make_uint4(0, 0, 0, 0);
                if (tl.x0 + row < p.Wo) {
                    float g[8], xv[8];
                    unpack16<T>(gu[k], g);
                    unpack16<T>(xu[k], xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) g[e] = cA[e] * g[e] - cB[e] * xv[e] + cC[e];
                    out = pack16<T>(g);
                }
                *reinterpret_cast<uint4*>(sD + row * 128 + (((uint32_t)ch * 16u) ^ swz128(row))) = out;
            }
        }
    };

    int buf = 0;
    const uint32_t ntiles = (uint32_t)p.tiles;
    if (blockIdx.x < ntiles) {
        const StemTile t0 = decode_tile32(blockIdx.x, pp.td);
        patch_dma(rs, lds_patch, t0, p.H, pp.Wp, wave, lane);
        stage_dxo(t0);
    }
    sm3conv::dma_drain();
    for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x, buf ^= 1) {
        const uint32_t tn = t + gridDim.x;
        __syncthreads();  // this tile's patch (every wave drained its pieces) and dxo rows are in LDS for everyone
        StemTile nx = {0, 0, 0};
        if (tn < ntiles) {
            nx = decode_tile32(tn, pp.td);
            patch_dma(rs, lds_patch + (uint32_t)((buf ^ 1) * PATCH16), nx, p.H, pp.Wp, wave, lane);
        }
        const char* patch = sPatch + buf * PATCH16;
#pragma unroll
        for (int s = 0; s < 8; ++s) {  // 16 pixels per step
            const char* a0 = sD + ra_off + s * 16 * 128;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * 128));
            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
            const uint4 fa = make_uint4(l2.x, l2.y, h2.x, h2.y);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                uint4 fb = make_uint4(0, 0, 0, 0);
                if (b_off[j] != 0xffffffffu) {
                    const uint32_t* d = reinterpret_cast<const uint32_t*>(patch + b_off[j] + s * 64);  // 16 pixels = 16 dwords
                    fb = make_uint4(__builtin_amdgcn_perm(d[1], d[0], b_sel[j]), __builtin_amdgcn_perm(d[3], d[2], b_sel[j]),
                                    __builtin_amdgcn_perm(d[5], d[4], b_sel[j]), __builtin_amdgcn_perm(d[7], d[6], b_sel[j]));
                }
                sm3conv::mma_frag<T>(fa, fb, acc[j]);
            }
        }
        __syncthreads();  // everyone is done reading this tile's dxo rows
        if (tn < ntiles) stage_dxo(nx);
        sm3conv::dma_drain();  // the next patch, in flight since before the MFMA loop
    }

    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = 32 * (kb0 + j) + frow;
        const int g = k >> 3, kw = k & 7;
        if (g >= NG || kw >= 7) continue;
        const int kh = g / 3, c = g - 3 * kh;
        const int kcol = (kh * 7 + kw) * 3 + c;  // master layout [co][kh][kw][c]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (p.dw_slabs) p.dw_slabs[(long)blockIdx.x * (64 * 147) + co * 147 + kcol] = acc[j][r];
            else atomicAdd(p.dw + co * 147 + kcol, acc[j][r]);
        }
    }
}

// master [64][kh][kw][c] fp32 -> bf16 [64][176], k = (kh*3 + c)*8 + kw, zero where kw == 7 or k >= 168
template <typename T>
__global__ void stem_weight_prep_kernel(const float* __restrict__ w, T* __restrict__ out, const int* __restrict__ only_if) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * KPAD || (only_if && *only_if == 0)) return;
    const int co = i / KPAD, k = i - co * KPAD;
    const int g = k >> 3, kw = k & 7;
    float v = 0.f;
    if (g < NG && kw < 7) {
        const int kh = g / 3, c = g - 3 * kh;
        v = w[co * 147 + (kh * 7 + kw) * 3 + c];
    }
    out[i].v = (uint16_t)pack2<T>(v, 0.f);
}


// ---- exact-f32 mode: the same two kernels on v_mfma_f32_32x32x2_f32 ------------------------------------------------
// K is still ordered (group g, kw padded to 8); one sm3conv::mma_frag<float> call (four MFMAs, K = 8) covers exactly one
// group: lane (pixel px, half h) supplies patch row g, columns 2 px + 4 h + {0..3}; the filter bank is fp32 [64][176] in
// the same order, kept in LDS at a pitch of 172 floats (the 16 rows of a ds_read_b128 group fall on 16 distinct 16-byte
// slots).  The output tile goes from the accumulators straight to HBM (32 consecutive channels = 128 B per lane group):
// 128 pixels x 64 fp32 do not fit the patch's LDS.
constexpr int WPF = 172;

__global__ __launch_bounds__(256, 2) void stem_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ y, float* __restrict__ partials, int N,
                                                              int H, int W, int Ho, int Wo, int xblocks, long tiles) {
    __shared__ __attribute__((aligned(16))) float patch[NG * PW];
    __shared__ float sStat[4][64][2];
    __shared__ __attribute__((aligned(16))) float sW[64 * WPF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    for (int i = tid; i < 64 * (NG * 8 / 4); i += 256) {  // 168 floats per filter, 16 bytes at a time
        const int co = i / (NG * 2), c4 = i - co * (NG * 2);
        *reinterpret_cast<float4*>(&sW[co * WPF + c4 * 4]) = *reinterpret_cast<const float4*>(w + (long)co * KPAD + c4 * 4);
    }
    const float* wlane = &sW[col * WPF + 4 * h];

    PatchRegs pre;
    if ((long)blockIdx.x < tiles) {
        load_patch(pre, x, decode_tile(blockIdx.x, xblocks, Ho), H, W);
        store_patch(patch, pre);
    }
    __syncthreads();
    for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
        const StemTile tl = decode_tile(t, xblocks, Ho);
        const long tn = t + gridDim.x;
        if (tn < tiles) load_patch(pre, x, decode_tile(tn, xblocks, Ho), H, W);  // lands during this tile's compute
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const int px = 32 * wave + col;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float2* src = reinterpret_cast<const float2*>(patch + g * PW + 2 * px + 4 * h);
            const float2 v0 = src[0], v1 = src[1];
            const uint4 a = make_uint4(__float_as_uint(v0.x), __float_as_uint(v0.y), __float_as_uint(v1.x), __float_as_uint(v1.y));
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                sm3conv::mma_frag<float>(a, *reinterpret_cast<const uint4*>(wlane + nb * 32 * WPF + g * 8), acc[nb]);
        }
        const long pix0 = ((long)tl.n * Ho + tl.oy) * Wo + tl.x0;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (tl.x0 + row < Wo) {
                    const float v = acc[nb][r];
                    y[(pix0 + row) * 64 + nb * 32 + col] = v;
                    s1 += v;
                    s2 += v * v;
                }
            }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                sStat[wave][nb * 32 + lane][0] = s1;
                sStat[wave][nb * 32 + lane][1] = s2;
            }
        }
        __syncthreads();  // statistics visible; everyone is done reading the patch
        if (partials && tid < 128) {
            const int c = tid & 63, st = tid >> 6;
            partials[(t * 2 + st) * 64 + c] = (sStat[0][c][st] + sStat[1][c][st]) + (sStat[2][c][st] + sStat[3][c][st]);
        }
        if (tn < tiles) store_patch(patch, pre);
        __syncthreads();
    }
}

// dW[co][k] = sum_pixels dxo[pix][co] patch[pix][k], pixel axis as the MFMA K (8 pixels per mma_frag<float>); the dxo
// tile is fp32 [128 pixels][64 co] in LDS, written with the BatchNorm-backward apply on the fly as in the 16-bit kernel.
__global__ __launch_bounds__(256, 2) void stem_wgrad_f32_kernel(const StemWgradParams p) {
    __shared__ __attribute__((aligned(16))) float patch[NG * PW];
    __shared__ __attribute__((aligned(16))) float sD[128 * 64];
    __shared__ __attribute__((aligned(16))) float sCoef[3][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = wave & 1, kb0 = 3 * (wave >> 1);  // this wave: co block cb, k blocks kb0 .. kb0+2
    const int r = lane & 31, h = lane >> 5;

    if (blockIdx.x == 0 && tid < 64 && p.lsums) {  // parameter gradients of bn1 from the LOCAL sums, once
        for (int v = 0; v < p.views; ++v) {
            if (p.dbeta) atomicAdd(&p.dbeta[tid], (float)p.lsums[v * 128 + tid]);
            if (p.dgamma) atomicAdd(&p.dgamma[tid], (float)p.lsums[v * 128 + 64 + tid]);
        }
    }
    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

    const int v16 = tid & 15, r0 = tid >> 4;  // staging role: channels 4 v16 .. +3 of rows r0 + 16 k
    int cur_view = -1;
    const float* dzf = reinterpret_cast<const float*>(p.dz);
    const float* xof = reinterpret_cast<const float*>(p.xo);
    PatchRegs pre;
    auto store_tile = [&](const StemTile& tl) {
        const int view = tl.n / p.n_per_view;
        if (view != cur_view) {  // block-uniform
            cur_view = view;
            if (tid < 64) {
                const int c = tid;
                const float is = p.invstd[view * 64 + c];
                const float g = p.gamma ? p.gamma[c] : 1.f;
                const float k0c = g * is;
                const float k1c = (float)(p.gsums[view * 128 + c] * p.inv_count);
                const float qc = k0c * is * (float)(p.gsums[view * 128 + 64 + c] * p.inv_count);
                sCoef[0][c] = k0c;
                sCoef[1][c] = qc;
                sCoef[2][c] = qc * p.mean[view * 64 + c] - k0c * k1c;
            }
            __syncthreads();
        }
        store_patch(patch, pre);
        const float4 cA = *reinterpret_cast<const float4*>(&sCoef[0][v16 * 4]);
        const float4 cB = *reinterpret_cast<const float4*>(&sCoef[1][v16 * 4]);
        const float4 cC = *reinterpret_cast<const float4*>(&sCoef[2][v16 * 4]);
        const long pix0 = ((long)tl.n * p.Ho + tl.oy) * p.Wo + tl.x0;
#pragma unroll
        for (int kb = 0; kb < 8; kb += 4) {  // four rows in flight
            float4 gv[4], xv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = r0 + 16 * (kb + k);
                const bool ok = tl.x0 + row < p.Wo;
                gv[k] = ok ? *reinterpret_cast<const float4*>(dzf + (pix0 + row) * 64 + v16 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                xv[k] = ok ? *reinterpret_cast<const float4*>(xof + (pix0 + row) * 64 + v16 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = r0 + 16 * (kb + k);
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tl.x0 + row < p.Wo) {
                    o.x = cA.x * gv[k].x - cB.x * xv[k].x + cC.x;
                    o.y = cA.y * gv[k].y - cB.y * xv[k].y + cC.y;
                    o.z = cA.z * gv[k].z - cB.z * xv[k].z + cC.z;
                    o.w = cA.w * gv[k].w - cB.w * xv[k].w + cC.w;
                }
                *reinterpret_cast<float4*>(&sD[row * 64 + v16 * 4]) = o;
            }
        }
    };
    if ((long)blockIdx.x < p.tiles) {
        const StemTile t0 = decode_tile(blockIdx.x, p.xblocks, p.Ho);
        load_patch(pre, p.x, t0, p.H, p.W);
        store_tile(t0);
    }
    for (long t = blockIdx.x; t < p.tiles; t += gridDim.x) {
        const long tn = t + gridDim.x;
        StemTile nx = {0, 0, 0};
        if (tn < p.tiles) {
            nx = decode_tile(tn, p.xblocks, p.Ho);
            load_patch(pre, p.x, nx, p.H, p.W);  // in flight during this tile's MFMA loop
        }
        __syncthreads();
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {  // 8 pixels per step: lane (r, h) holds pixels 8 s + 4 h + {0..3}
            const int pb = 8 * s + 4 * h;
            const float* ap = sD + pb * 64 + cb * 32 + r;
            const uint4 fa = make_uint4(__float_as_uint(ap[0]), __float_as_uint(ap[64]), __float_as_uint(ap[128]),
                                        __float_as_uint(ap[192]));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int k = 32 * (kb0 + j) + r;
                const int g = k >> 3, kw = k & 7;
                uint4 fb = make_uint4(0, 0, 0, 0);
                if (g < NG) {
                    const float* src = patch + g * PW + 2 * pb + kw;
                    fb = make_uint4(__float_as_uint(src[0]), __float_as_uint(src[2]), __float_as_uint(src[4]),
                                    __float_as_uint(src[6]));
                }
                sm3conv::mma_frag<float>(fa, fb, acc[j]);
            }
        }
        __syncthreads();  // everyone is done reading this tile's operands
        if (tn < p.tiles) store_tile(nx);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = 32 * (kb0 + j) + r;
        const int g = k >> 3, kw = k & 7;
        if (g >= NG || kw >= 7) continue;
        const int kh = g / 3, c = g - 3 * kh;
        const int kcol = (kh * 7 + kw) * 3 + c;  // master layout [co][kh][kw][c]
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = cb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (p.dw_slabs) p.dw_slabs[(long)blockIdx.x * (64 * 147) + co * 147 + kcol] = acc[j][q];
            else atomicAdd(p.dw + co * 147 + kcol, acc[j][q]);
        }
    }
}

__global__ void stem_weight_prep_f32_kernel(const float* __restrict__ w, float* __restrict__ out, const int* __restrict__ only_if) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * KPAD || (only_if && *only_if == 0)) return;
    const int co = i / KPAD, k = i - co * KPAD;
    const int g = k >> 3, kw = k & 7;
    float v = 0.f;
    if (g < NG && kw < 7) {
        const int kh = g / 3, c = g - 3 * kh;
        v = w[co * 147 + (kh * 7 + kw) * 3 + c];
    }
    out[i] = v;
}

int stem_geometry(int N, int H, int W, int& Ho, int& Wo, int& xblocks, long& tiles) {
    if (N <= 0 || H <= 0 || W <= 0) return SM3_EINVAL;
    Ho = (H - 1) / 2 + 1;
    Wo = (W - 1) / 2 + 1;
    xblocks = (Wo + 127) / 128;
    tiles = (long)N * Ho * xblocks;
    if ((long)N * 3 * H * W >= 0x7fffffffL || (long)N * Ho * Wo >= 0x7fffffffL / 64) return SM3_EINVAL;
    return 0;
}

}  // namespace

extern "C" int sm3_stem_partial_rows(int N, int H, int W) {
    int Ho, Wo, xb;
    long tiles;
    if (stem_geometry(N, H, W, Ho, Wo, xb, tiles)) return SM3_EINVAL;
    return tiles > 0x7fffffffL ? SM3_EINVAL : (int)tiles;
}

extern "C" int sm3_stem_weight_prep_if(int dtype, const float* w_master, void* w_stem, const int* only_if, void* stream) {
    if (!w_master || !w_stem) return SM3_EINVAL;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(stem_weight_prep_kernel<bf16_t>, dim3((64 * KPAD + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           w_master, (bf16_t*)w_stem, only_if);
    else if (dtype == SM3_F16)
        hipLaunchKernelGGL(stem_weight_prep_kernel<f16_t>, dim3((64 * KPAD + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           w_master, (f16_t*)w_stem, only_if);
    else if (dtype == SM3_F32)
        hipLaunchKernelGGL(stem_weight_prep_f32_kernel, dim3((64 * KPAD + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           w_master, (float*)w_stem, only_if);
    else
        return SM3_EDTYPE;
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_stem_weight_prep(int dtype, const float* w_master, void* w_stem, void* stream) {
    return sm3_stem_weight_prep_if(dtype, w_master, w_stem, nullptr, stream);
}

extern "C" int sm3_stem_conv_fwd(int dtype, const float* x_nchw, const void* w_stem, void* y, float* stat_partials, int N,
                                 int H, int W, void* stream) {
    if (!x_nchw || !w_stem || !y) return SM3_EINVAL;
    if (dtype != SM3_BF16 && dtype != SM3_F16 && dtype != SM3_F32) return SM3_EDTYPE;
    int Ho, Wo, xb;
    long tiles;
    if (int rc = stem_geometry(N, H, W, Ho, Wo, xb, tiles)) return rc;
    if (dtype == SM3_F32) {  // exact-f32 parity mode: two workgroups per CU (68 KB of LDS each)
        const unsigned gridf = (unsigned)(tiles < 512 ? tiles : 512);
        hipLaunchKernelGGL(stem_fwd_f32_kernel, dim3(gridf), dim3(256), 0, (hipStream_t)stream, x_nchw, (const float*)w_stem,
                           (float*)y, stat_partials, N, H, W, Ho, Wo, xb, tiles);
        SM3_CHECK_LAUNCH();
        return 0;
    }
    const unsigned grid = (unsigned)(tiles < 768 ? tiles : 768);  // persistent: 3 workgroups per CU (136 VGPRs)
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(stem_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nchw,
                           (const uint4*)w_stem, (bf16_t*)y, stat_partials, N, H, W, Ho, Wo, xb, tiles);
    else
        hipLaunchKernelGGL(stem_fwd_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nchw,
                           (const uint4*)w_stem, (f16_t*)y, stat_partials, N, H, W, Ho, Wo, xb, tiles);
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_stem_wgrad_bn(int dtype, const float* x_nchw, const void* dz, const void* xo, const float* mean,
                                 const float* invstd, const float* gamma, const double* global_sums, double count,
                                 const double* local_sums, float* dgamma, float* dbeta, float* dw, float* dw_slabs,
                                 int N, int H, int W, int views, void* stream) {
    if (!x_nchw || !dz || !xo || !mean || !invstd || !global_sums || !dw || count <= 0 || views < 1 || N % views)
        return SM3_EINVAL;
    if (dw_slabs && (((uintptr_t)dw_slabs | (uintptr_t)dw) & 15)) return SM3_EALIGN;
    if (dtype != SM3_BF16 && dtype != SM3_F16 && dtype != SM3_F32) return SM3_EDTYPE;
    StemWgradParams p;
    if (int rc = stem_geometry(N, H, W, p.Ho, p.Wo, p.xblocks, p.tiles)) return rc;
    p.x = x_nchw; p.dz = (const char*)dz; p.xo = (const char*)xo;
    p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.gsums = global_sums; p.lsums = local_sums;
    p.dgamma = dgamma; p.dbeta = dbeta; p.dw = dw; p.dw_slabs = dw_slabs; p.inv_count = 1.0 / count;
    p.N = N; p.H = H; p.W = W; p.n_per_view = N / views; p.views = views;
    if (dtype == SM3_F32) {
        const unsigned gridf = (unsigned)(p.tiles < 512 ? p.tiles : 512);
        hipLaunchKernelGGL(stem_wgrad_f32_kernel, dim3(gridf), dim3(256), 0, (hipStream_t)stream, p);
        SM3_CHECK_LAUNCH();
        return dw_slabs ? sm3_slab_reduce(dw_slabs, (int)gridf, 64 * 147, dw, 1, stream) : 0;
    }
    const unsigned grid = (unsigned)(p.tiles < SM3_STEM_WGRAD_SLABS ? p.tiles : SM3_STEM_WGRAD_SLABS);
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(stem_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(stem_wgrad_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    SM3_CHECK_LAUNCH();
    // the persistent grid and its tile walk are functions of the geometry: the slabs add up to the same bits in every run
    return dw_slabs ? sm3_slab_reduce(dw_slabs, (int)grid, 64 * 147, dw, 1, stream) : 0;
}

extern "C" int sm3_stem_image_cols(int W) { return W > 0 ? (W + 6 + 7) / 8 * 8 : SM3_EINVAL; }

extern "C" int sm3_stem_image_prep(int dtype, const float* x_view0, const float* x_view1, void* ximg, int n_per_view,
                                   int views, int H, int W, void* stream) {
    if (!x_view0 || !ximg || n_per_view <= 0 || H <= 0 || W <= 0 || views < 1 || views > 2 || (views == 2 && !x_view1))
        return SM3_EINVAL;
    if (dtype != SM3_BF16 && dtype != SM3_F16) return SM3_EDTYPE;
    if ((uintptr_t)ximg & 15) return SM3_EALIGN;
    const int Wp = sm3_stem_image_cols(W);
    const long chunks = (long)n_per_view * views * 3 * H * (Wp / 8);
    if (chunks * 16 >= 0xC0000000L) return SM3_EINVAL;  // 32-bit buffer offsets in the kernels that read it
    const long blocks = (chunks + 255) / 256;
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(stem_image_prep_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_view0,
                           x_view1, (uint4*)ximg, n_per_view, H, W, Wp / 8, chunks);
    else
        hipLaunchKernelGGL(stem_image_prep_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x_view0,
                           x_view1, (uint4*)ximg, n_per_view, H, W, Wp / 8, chunks);
    SM3_CHECK_LAUNCH();
    return 0;
}

template <typename K>
static int stem16_allow_lds(K kern, int bytes) {
    // dynamic LDS above 64 KB needs the function attribute: per (instantiation, device), set once each
    static std::atomic<int> done[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    if (!done[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        done[dev].store(1, std::memory_order_release);
    }
    return 0;
}

extern "C" int sm3_stem_conv_fwd16(int dtype, const void* ximg, const void* w_stem, void* y, float* stat_partials, int N,
                                   int H, int W, void* stream) {
    if (!ximg || !w_stem || !y) return SM3_EINVAL;
    if (dtype != SM3_BF16 && dtype != SM3_F16) return SM3_EDTYPE;
    int Ho, Wo, xb;
    long tiles;
    if (int rc = stem_geometry(N, H, W, Ho, Wo, xb, tiles)) return rc;
    const int Wp = sm3_stem_image_cols(W);
    const long bytes = (long)N * 3 * H * Wp * 2;
    if (bytes >= 0xC0000000L) return SM3_EINVAL;
    if (tiles > 0x7fffffffL) return SM3_EINVAL;
    const TileDiv td{make_fastdiv((uint32_t)xb), make_fastdiv((uint32_t)Ho)};
    const unsigned grid = (unsigned)(tiles < 512 ? tiles : 512);  // persistent: 2 workgroups per CU (65 KB of LDS each)
    if (dtype == SM3_BF16) {
        if (int rc = stem16_allow_lds(stem_fwd16_kernel<bf16_t>, FWD16_LDS)) return rc;
        hipLaunchKernelGGL(stem_fwd16_kernel<bf16_t>, dim3(grid), dim3(256), FWD16_LDS, (hipStream_t)stream, (const char*)ximg,
                           (uint32_t)bytes, (const uint4*)w_stem, (bf16_t*)y, stat_partials, H, Wp, Ho, Wo, td, (uint32_t)tiles);
    } else {
        if (int rc = stem16_allow_lds(stem_fwd16_kernel<f16_t>, FWD16_LDS)) return rc;
        hipLaunchKernelGGL(stem_fwd16_kernel<f16_t>, dim3(grid), dim3(256), FWD16_LDS, (hipStream_t)stream, (const char*)ximg,
                           (uint32_t)bytes, (const uint4*)w_stem, (f16_t*)y, stat_partials, H, Wp, Ho, Wo, td, (uint32_t)tiles);
    }
    SM3_CHECK_LAUNCH();
    return 0;
}

extern "C" int sm3_stem_wgrad_bn16(int dtype, const void* ximg, const void* dz, const void* xo, const float* mean,
                                   const float* invstd, const float* gamma, const double* global_sums, double count,
                                   const double* local_sums, float* dgamma, float* dbeta, float* dw, float* dw_slabs,
                                   int N, int H, int W, int views, void* stream) {
    if (!ximg || !dz || !xo || !mean || !invstd || !global_sums || !dw || count <= 0 || views < 1 || N % views)
        return SM3_EINVAL;
    if (dtype != SM3_BF16 && dtype != SM3_F16) return SM3_EDTYPE;
    if (dw_slabs && (((uintptr_t)dw_slabs | (uintptr_t)dw) & 15)) return SM3_EALIGN;
    StemWgrad16Params pp;
    StemWgradParams& p = pp.q;
    if (int rc = stem_geometry(N, H, W, p.Ho, p.Wo, p.xblocks, p.tiles)) return rc;
    pp.Wp = sm3_stem_image_cols(W);
    const long bytes = (long)N * 3 * H * pp.Wp * 2;
    if (bytes >= 0xC0000000L) return SM3_EINVAL;
    if (p.tiles > 0x7fffffffL) return SM3_EINVAL;
    pp.ximg = (const char*)ximg; pp.ximg_bytes = (uint32_t)bytes;
    pp.td = TileDiv{make_fastdiv((uint32_t)p.xblocks), make_fastdiv((uint32_t)p.Ho)};
    p.x = nullptr; p.dz = (const char*)dz; p.xo = (const char*)xo;
    p.mean = mean; p.invstd = invstd; p.gamma = gamma; p.gsums = global_sums; p.lsums = local_sums;
    p.dgamma = dgamma; p.dbeta = dbeta; p.dw = dw; p.dw_slabs = dw_slabs; p.inv_count = 1.0 / count;
    p.N = N; p.H = H; p.W = W; p.n_per_view = N / views; p.views = views;
    const unsigned grid = (unsigned)(p.tiles < SM3_STEM_WGRAD_SLABS ? p.tiles : SM3_STEM_WGRAD_SLABS);
    if (dtype == SM3_BF16)
        hipLaunchKernelGGL(stem_wgrad16_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pp);
    else
        hipLaunchKernelGGL(stem_wgrad16_kernel<f16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pp);
    SM3_CHECK_LAUNCH();
    return dw_slabs ? sm3_slab_reduce(dw_slabs, (int)grid, 64 * 147, dw, 1, stream) : 0;
}
