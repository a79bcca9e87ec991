// 3x3 / stride 1 / pad 1 convolution (forward and data gradient) with INPUT-PATCH REUSE, bf16 MFMA on gfx950.
//
// The 128-row gather-GEMM (conv_igemm.hip) stages, for every one of the 9 taps and every 64-channel chunk, a fresh
// 128 x 128 B activation tile and a 128 x 128 B filter tile: 32 KB of LDS-DMA per 64 MFMAs, i.e. 64 B per MFMA-cycle
// per CU -- about twice what a CU's vector-memory path delivers from L2 (MI355X_MICROARCH.md, "Indexed rows": 66-73
// GB/s per CU), which is why its 3x3 layers stop at ~800 TFLOP/s whatever the schedule (round 1: K-loop issue-bound
// on staging).  For stride 1 the output pixel index m = (n, y, x) IS the input pixel index, so tap (dy, dx) of a run
// of consecutive output pixels is the same run of the input shifted by dy*W + dx rows.  This kernel therefore stages
// ONE patch per 64-channel chunk -- input rows [m0 - W - 1, m0 + 255 + W + 1] of a 256-pixel tile -- and reads the 9
// taps as shifted windows of it; only the filter tile changes per tap.  Bytes staged per MFMA drop 3.3x (256-row
// tile: the filter tile is shared by twice the rows; patch: 370 rows instead of 9 x 256), so the loop becomes
// MFMA-paced.  Pixels whose tap falls outside the image (padding; the patch holds the neighbouring row or image
// there) are zeroed on the A fragment with a per-row 9-bit validity mask.
//
// 512 threads = 4 (M) x 2 (N) waves, each a 64 x 64 (BN = 128) or 64 x 32 (BN = 64) sub-tile; LDS: two patch stages
// (channel chunk c+1 streams in, one piece per tap step, while chunk c is consumed) + two filter stages.
// Epilogues = conv_igemm.hip's: lean bf16 store + BatchNorm partial sums (forward), or addend + fused
// BatchNorm-backward phase 1 (data gradient); a 256-row tile writes TWO partial rows (one per 128 rows) so every
// consumer of the partial-row layout is unchanged.
//
// Reference call sites replaced: conv3x3 (src/models/resnet.py:49-62) as used by Bottleneck.conv2 (:146) with
// stride 1, forward and autograd data gradient.
#include <stdlib.h>

#include "conv_common.h"

using namespace sm3conv;

namespace {

constexpr int PBM = 256, PNT = 512, PWM = 4, PWN = 2;
constexpr int PROWS_MAX = 376;                 // 256 + 2*56 + 2 = 370, rounded up to whole 8-row DMA pieces
constexpr int PST = PROWS_MAX * 128;           // bytes per patch stage

template <int BN, bool LEAN>
__global__ __launch_bounds__(PNT, 2) void conv3x3_patch_kernel(const ConvParams p, const int prows) {
    constexpr int WTM = PBM / PWM, WTN = BN / PWN, TM = WTM / 32, TN = WTN / 32;
    constexpr int BST = BN * 128;                       // bytes per filter stage
    constexpr int BSTAGES = 3;                          // filter ring: tiles of steps s+1 and s+2 in flight / landed
    constexpr int MAIN_BYTES = 2 * PST + BSTAGES * BST;
    constexpr int BPIECES = BN / 8;                     // 1 KiB DMA pieces per filter tile
    constexpr int BI = (BPIECES + 7) / 8;               // per wave
    static_assert(TM == 2 && TN >= 1, "tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / PWN, wn = wave % PWN;

    int bid = blockIdx.x;
    {  // XCD-aware block remap (bijective): the N-tiles of one patch run on one XCD's L2
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int bn = bid % p.tilesN, bm = bid / p.tilesN;
    const int m0 = bm * PBM, n0 = bn * BN;
    const int W = p.Wi, H = p.Hi;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const uint32_t row_bytes = (uint32_t)p.Ci * 2u;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int lr = lane >> 3, pos = lane & 7;

    // ---- patch loader: piece q = rows 8q .. 8q+7 of the patch, patch row r <-> input pixel m0 - (W+1) + r ----
    const int npieces = prows >> 3;
    auto dma_patch_piece = [&](int stage, int piece, uint32_t kc_off) {
        const int r = piece * 8 + lr;
        const int pix = m0 - (W + 1) + r;
        const uint32_t off = ((unsigned)pix < (unsigned)p.M)
                                 ? (uint32_t)pix * row_bytes + (uint32_t)((pos ^ (r >> 1)) & 7) * 16u
                                 : kOOB;
        dma16(rx, smem_lds + (uint32_t)(stage * PST + piece * 1024), off, kc_off);
    };
    // ---- filter loader ----
    uint32_t b_off[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int piece = wave + 8 * i;
        const int r = piece * 8 + lr;
        const int co = n0 + r;
        b_off[i] = (piece < BPIECES && co < p.Co)
                       ? (uint32_t)co * (uint32_t)(p.w_row_stride * 2) + (uint32_t)((pos ^ (r >> 1)) & 7) * 16u
                       : kOOB;
    }
    auto dma_filter = [&](int stage, uint32_t soff) {
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int piece = wave + 8 * i;
            if (piece < BPIECES) dma16(rw, smem_lds + (uint32_t)(2 * PST + stage * BST + piece * 1024), b_off[i], soff);
        }
    };

    // ---- per-lane fragment rows and their tap-validity masks ----
    const int frow = lane & 31, fh = lane >> 5;
    int prow_base[TM];      // patch row of tap (0,0) for this lane's fragment row
    unsigned vmask[TM];     // bit t: tap t reads a real input pixel
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * WTM + i * 32 + frow;
        prow_base[i] = r + W + 1;
        const int m = m0 + r;
        unsigned vm = 0;
        if (m < p.M) {
            const int n = fdiv(m, p.div_HoWo);
            const int rem = m - n * p.HoWo;
            const int y = fdiv(rem, p.div_Wo);
            const int x = rem - y * p.Wo;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + p.dy[t], ix = x + p.dx[t];
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) vm |= 1u << t;
            }
        }
        vmask[i] = vm;
    }
    uint32_t fb_base[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) fb_base[j] = (uint32_t)lds_off(wn * WTN + j * 32 + frow, fh);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kchunks = p.kchunks, nsteps = 9 * kchunks;
    auto filter_soff = [&](int step) {  // scalar offset of the filter tile of step `step` = (chunk, tap)
        const int c = step / 9, tp = step - 9 * c;
        return (uint32_t)p.wtap[tp] * row_bytes + (uint32_t)c * 128u;
    };
    // prologue: the whole patch of chunk 0 and the filter tiles of steps 0 and 1
    for (int q = wave; q < npieces; q += 8) dma_patch_piece(0, q, 0u);
    dma_filter(0, filter_soff(0));
    if (nsteps > 1) dma_filter(1, filter_soff(1));
    dma_drain();
    __syncthreads();

    // Two steps of prefetch: the DMAs issued in step s (filter tile of step s+2, one patch piece of the next channel
    // chunk) are NOT waited for at the end of step s -- only those of step s-1 are (counted vmcnt: DMAs complete in
    // issue order) -- so a tile has a whole step of compute plus a barrier to land instead of none.
    int kc = 0, t = 0;
    for (int s = 0; s < nsteps; ++s) {
        int issued = 0;
        if (s + 2 < nsteps) {
            dma_filter((s + 2) % BSTAGES, filter_soff(s + 2));
            issued += (wave < BPIECES ? 1 : 0) + (BI > 1 && wave + 8 < BPIECES ? 1 : 0);
        }
        {
            const int q = wave + 8 * t;
            if (kc + 1 < kchunks && q < npieces) {
                dma_patch_piece((kc + 1) & 1, q, (uint32_t)(kc + 1) * 128u);
                ++issued;
            }
        }
        // compute: tap t of chunk kc
        {
            const char* sP = smem + (kc & 1) * PST;
            const char* sB = smem + 2 * PST + (s % BSTAGES) * BST;
            const int sh = p.dy[t] * W + p.dx[t];
            uint32_t fa_base[TM];
            bool ok[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int pr = prow_base[i] + sh;
                fa_base[i] = (uint32_t)pr * 128u + (uint32_t)(((fh ^ (pr >> 1)) & 7) << 4);
                ok[i] = (vmask[i] >> t) & 1u;
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                uint4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    fa[i] = *reinterpret_cast<const uint4*>(sP + (fa_base[i] ^ (kk << 5)));
                    if (!ok[i]) fa[i] = make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(sB + (fb_base[j] ^ (kk << 5)));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mma_frag<bf16_t>(fa[i], fb[j], acc[i][j]);
            }
        }
        // everything this wave issued BEFORE this step has landed (the filter tile of step s+1, older patch pieces)
        issued = __builtin_amdgcn_readfirstlane(issued);
        if (issued >= 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (issued == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (issued == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // everyone's have; and everyone is done reading this step's stages
        if (++t == 9) { t = 0; ++kc; }
    }

    // =====================================================================================================
    // lean epilogue (forward): bf16 pairs in registers, BatchNorm sums with v_dot2, whole tile staged once
    // =====================================================================================================
    if constexpr (LEAN) {
        constexpr int LEAN_PITCH = BN * 2 + 16;
        constexpr int CPR = BN / 8, RSTEP = PNT / CPR, NPASS = PBM / RSTEP;
        char* sC = smem;
        float* sStat = reinterpret_cast<float*>(smem + MAIN_BYTES);  // [PWM][BN][2]
        const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
            char* colp = sC + (wn * WTN + j * 32 + frow) * 2 + (wm * WTM + 4 * fh) * LEAN_PITCH;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t pk = pack_bf16x2(acc[i][j][2 * q], acc[i][j][2 * q + 1]);
                    const bf16x2_t pv = __builtin_bit_cast(bf16x2_t, pk);
                    s1 = __builtin_amdgcn_fdot2_f32_bf16(pv, ones, s1, false);
                    s2 = __builtin_amdgcn_fdot2_f32_bf16(pv, pv, s2, false);
                    const int R = i * 32 + (q & 1) * 2 + 8 * (q >> 1);
                    *reinterpret_cast<uint16_t*>(colp + R * LEAN_PITCH) = (uint16_t)pk;
                    *reinterpret_cast<uint16_t*>(colp + (R + 1) * LEAN_PITCH) = (uint16_t)(pk >> 16);
                }
            if (p.partials) {
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (lane < 32) {
                    const int col = wn * WTN + j * 32 + lane;
                    sStat[(wm * BN + col) * 2 + 0] = s1;
                    sStat[(wm * BN + col) * 2 + 1] = s2;
                }
            }
        }
        __syncthreads();
        if (p.partials && tid < 2 * BN) {  // one partial row per 128 rows, as the 128-row kernel writes them
            const int half = tid / BN, col = tid % BN;
            if (n0 + col < p.Co && m0 + half * 128 < p.M) {
                const float s1 = sStat[((2 * half) * BN + col) * 2 + 0] + sStat[((2 * half + 1) * BN + col) * 2 + 0];
                const float s2 = sStat[((2 * half) * BN + col) * 2 + 1] + sStat[((2 * half + 1) * BN + col) * 2 + 1];
                const long prow = 2L * bm + half;
                p.partials[(prow * 2 + 0) * p.Co + n0 + col] = s1;
                p.partials[(prow * 2 + 1) * p.Co + n0 + col] = s2;
            }
        }
        const int cc = tid % CPR, r0 = tid / CPR;
        const int ncol = n0 + cc * 8;
        if (ncol < p.Co) {
            char* yp = p.y + ((long)(m0 + r0) * p.Co + ncol) * 2;
            const long ystep = (long)RSTEP * p.Co * 2;
            const char* sp = sC + r0 * LEAN_PITCH + cc * 16;
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                if (m0 + r0 + k * RSTEP < p.M)
                    stg16<true>(yp + k * ystep, *reinterpret_cast<const uint4*>(sp + k * RSTEP * LEAN_PITCH));
            }
        }
        return;
    } else {
        // =================================================================================================
        // general epilogue (data gradient): optional addend, optional fused BatchNorm-backward phase 1
        // =================================================================================================
        constexpr int EPC = 8, CPR = BN / EPC, RSTEP = PNT / CPR, NROW = WTM / RSTEP;
        static_assert(PNT % CPR == 0 && WTM % RSTEP == 0, "a thread must keep one channel vector across its rows");
        const bool fz = p.fz_x != nullptr;
        const int cc = tid % CPR;
        const int ncol = n0 + cc * EPC;
        float* sC = reinterpret_cast<float*>(smem);
        float f_mu[EPC], f_is[EPC], f_s1[EPC], f_s2[EPC];
#pragma unroll
        for (int hh = 0; hh < PWM; ++hh) {
            const int hb = 2 * bm + (hh >> 1);  // 128-row block of this wave-row: the unit of partial rows and views
            if (fz && (hh & 1) == 0) {
                const int fz_view = (p.fz_view_tiles > 0 && hb >= p.fz_view_tiles) ? 1 : 0;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    f_s1[e] = 0.f;
                    f_s2[e] = 0.f;
                    f_mu[e] = (ncol < p.Co) ? p.fz_mean[fz_view * p.Co + ncol + e] : 0.f;
                    f_is[e] = (ncol < p.Co) ? p.fz_invstd[fz_view * p.Co + ncol + e] : 0.f;
                }
            }
            long e_off[NROW];
            uint4 pre_add[NROW], pre_x[NROW];
            unsigned pre_mk[NROW];
#pragma unroll
            for (int k = 0; k < NROW; ++k) {
                const int m = m0 + hh * WTM + tid / CPR + k * RSTEP;
                e_off[k] = -1;
                pre_add[k] = make_uint4(0, 0, 0, 0);
                pre_x[k] = make_uint4(0, 0, 0, 0);
                pre_mk[k] = 0xffu;
                if (m < p.M && ncol < p.Co) {
                    e_off[k] = (long)m * p.Co + ncol;
                    if (p.addend) pre_add[k] = ldg16<true>(p.addend + e_off[k] * 2);
                    if (fz) {
                        pre_x[k] = ldg16<true>(p.fz_x + e_off[k] * 2);
                        if (p.fz_mask) pre_mk[k] = p.fz_mask[e_off[k] / EPC];
                    }
                }
            }
            __syncthreads();  // the main loop's stages / the previous wave-row have been read out of sC
            if (wm == hh) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                            const int col = wn * WTN + j * 32 + frow;
                            sC[row * BN + col] = acc[i][j][r];
                        }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NROW; ++k) {
                if (e_off[k] < 0) continue;
                const int r = tid / CPR + k * RSTEP;
                float v[EPC];
#pragma unroll
                for (int e = 0; e < EPC; e += 4) {
                    const float4 q = *reinterpret_cast<const float4*>(&sC[r * BN + cc * EPC + e]);
                    v[e] = q.x;
                    v[e + 1] = q.y;
                    v[e + 2] = q.z;
                    v[e + 3] = q.w;
                }
                const long boff = e_off[k] * 2;
                if (p.addend) {
                    float a[EPC];
                    unpack16<bf16_t>(pre_add[k], a);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] += a[e];
                }
                if (fz) {
                    float xv[EPC];
                    unpack16<bf16_t>(pre_x[k], xv);
                    const unsigned mk = pre_mk[k];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = ((mk >> e) & 1u) ? v[e] : 0.f;
                    const uint4 packed = pack16<bf16_t>(v);
                    float dzr[EPC];
                    unpack16<bf16_t>(packed, dzr);  // sums of the STORED (rounded) dz, as the standalone kernel's
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        f_s1[e] += dzr[e];
                        f_s2[e] += dzr[e] * (xv[e] - f_mu[e]) * f_is[e];
                    }
                    stg16<true>(p.y + boff, packed);
                } else {
                    stg16<true>(p.y + boff, pack16<bf16_t>(v));
                }
            }
            if (fz && (hh & 1) == 1) {  // the 128-row block is complete: reduce its sums over threads, one partial row
                __syncthreads();
                float* sRed = reinterpret_cast<float*>(smem);  // [PNT][2*EPC]
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    sRed[tid * 2 * EPC + e] = f_s1[e];
                    sRed[tid * 2 * EPC + EPC + e] = f_s2[e];
                }
                __syncthreads();
                const int fz_view = (p.fz_view_tiles > 0 && hb >= p.fz_view_tiles) ? 1 : 0;
                const int fz_prow = fz_view ? p.fz_row_off1 + hb - p.fz_view_tiles : p.fz_row_off + hb;
                if ((long)hb * 128 < p.M) {
                    for (int o = tid; o < 2 * BN; o += PNT) {
                        const int stat = o / BN, col = o % BN;
                        float a = 0.f;
                        for (int rl = 0; rl < PNT / CPR; ++rl)
                            a += sRed[(rl * CPR + col / EPC) * 2 * EPC + stat * EPC + col % EPC];
                        if (n0 + col < p.Co) p.fz_partials[((long)fz_prow * 2 + stat) * p.Co + n0 + col] = a;
                    }
                }
            }
        }
    }
}

template <int BN, bool LEAN>
int launch_patch(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    constexpr int MAIN = 2 * PST + 3 * BN * 128;
    constexpr int LDS = MAIN + PWM * BN * 2 * 4;
    static_assert(MAIN >= PBM * (BN * 2 + 16) && MAIN >= PNT * 16 * 4, "epilogue staging must fit");
    p.tilesM = (p.M + PBM - 1) / PBM;
    p.tilesN = (p.Co + BN - 1) / BN;
    auto kern = conv3x3_patch_kernel<BN, LEAN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long nblocks = (long)p.tilesM * p.tilesN;
    if (nblocks <= 0 || nblocks > 0x7fffffffL) return SM3_EINVAL;
    const int prows = (PBM + 2 * p.Wi + 2 + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(PNT), LDS, st, p, prows);
    SM3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

namespace sm3conv {

// 3x3 / stride 1 / pad 1, dense, bf16, W <= 56, at least 4 row tiles: the shapes of Bottleneck.conv2 at 224x224 / 448x448
bool conv_patch_eligible(const sm3_conv_desc* d, const ConvParams& p) {
    static const int enabled = getenv("SM3_CONV_PATCH") ? atoi(getenv("SM3_CONV_PATCH")) : 1;
    if (!enabled || d->dtype != SM3_BF16 || d->ntaps != 9) return false;
    if (d->sy != 1 || d->sx != 1 || d->Hi != d->Ho || d->Wi != d->Wo) return false;
    if (d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0 || d->Hout != d->Ho || d->Wout != d->Wo) return false;
    if (d->Wi > 56 || d->Wi < 2 || d->Hi < 2 || p.M < 4 * PBM) return false;
    static const int min_chunks = getenv("SM3_CONV_PATCH_MIN_CHUNKS") ? atoi(getenv("SM3_CONV_PATCH_MIN_CHUNKS")) : 2;
    if (p.kchunks < min_chunks) return false;  // 64 input channels = 9 K-steps per tile: the epilogue dominates, 1 workgroup per CU loses
    if (p.ep_scale) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {  // the nine taps of a pad-1 3x3 window, in any order
        const int a = d->dy[t] + 1, b = d->dx[t] + 1;
        if (a < 0 || a > 2 || b < 0 || b > 2) return false;
        seen |= 1u << (a * 3 + b);
    }
    return seen == 0x1ffu;
}

int launch_conv_patch(const ConvParams& p, hipStream_t st) {
    const bool lean = !p.addend && !p.fz_x;
    if (p.Co <= 64) return lean ? launch_patch<64, true>(p, st) : launch_patch<64, false>(p, st);
    return lean ? launch_patch<128, true>(p, st) : launch_patch<128, false>(p, st);
}

}  // namespace sm3conv
