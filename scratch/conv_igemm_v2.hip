// Gather-GEMM convolution, second structure (bf16): persistent 256 x 128 tiles, 8 waves, 3-stage LDS-DMA ring.
//
// Same arithmetic and the same operand layouts as conv_igemm.hip; what changes is how the machine is kept busy:
//   * one 512-thread workgroup per CU (160 KB of LDS) walks a list of output tiles; its (tile, K-step) units form
//     one flat sequence, so the DMA for the first K-steps of the NEXT tile is in flight while the epilogue of the
//     current one stores (the 128-row kernel relies on 2-3 co-resident workgroups for that overlap);
//   * three 48 KB stages; the DMA of unit u+2 is issued right after the single barrier of unit u and waited for
//     with a COUNTED s_waitcnt vmcnt(6) (6 DMA instructions per wave per unit), so two units are always in
//     flight and a K-step never waits a full memory latency;
//   * the two waves of every SIMD run the unit schedule one barrier apart: one is in its MFMA phase while the
//     other issues DMAs / LDS reads (workgroups of the 128-row kernel drift apart by themselves; inside one
//     workgroup the stagger has to be built);
//   * 8 waves = 4 (M) x 2 (N), each a 64 x 64 sub-tile (2 x 2 v_mfma_f32_32x32x16_bf16 accumulators);
//   * epilogue per wave through a private 2 KB LDS window (8 rows x 64 columns f32 at a time): no block barrier,
//     128-byte row segments to HBM; BatchNorm statistics (forward) / fused BN-backward sums (data gradient) are
//     accumulated in the store loop and combined per 128-row half -> the same partial-row granularity as the
//     128-row kernel.
//
// Reference call sites replaced: the same as conv_igemm.hip (src/models/resnet.py:49-67 and autograd).
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"

namespace sm3conv {
namespace {

constexpr int BM = 256, BN = 128, NT = 512;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES, NSTAGE = 3;
constexpr int EPI_OFF = NSTAGE * STAGE, EPI_WAVE = 2048, LDS_BYTES = EPI_OFF + 8 * EPI_WAVE;  // 163840
static_assert(LDS_BYTES == 160 * 1024, "uses the whole LDS of a CU");

#define SM3_WAIT_DMA_AND_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")

// DIAG: per-wave s_memtime totals of the schedule's segments -> dbg[(block*8+wave)*8 + segment] (diagnostic launches
// only, SM3_CONV_V2_DIAG=1; the production instantiation carries no stamp).
template <typename T, bool DIAG>
__global__ __launch_bounds__(NT) void conv_igemm_v2_kernel(const ConvParams p, const int ntiles, long long* dbg) {
    static_assert(sizeof(T) == 2, "bf16 only");
    constexpr int SZ = 2, EPC = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int G = gridDim.x;
    int vb = blockIdx.x;  // XCD-aware (bijective): the 32 workgroups of one XCD take 32 consecutive tiles of a round
    {
        const int q = G >> 3, r = G & 7, xcd = vb & 7, slot = vb >> 3;
        vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int my_tiles = (ntiles - vb + G - 1) / G;
    const int nsteps = p.ntaps * p.kchunks;
    const int total_units = my_tiles * nsteps;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // ---- loader: runs two units ahead of the MFMA side, with its own (tile, tap, k-chunk) position ----------
    const int lrow = tid >> 3;                                              // tile row of this lane, + 64 per DMA
    const uint32_t src_chunk = (uint32_t)(((tid & 7) ^ (lrow >> 1)) & 7) * 16u;  // swizzle on the source side
    const uint32_t row_bytes = (uint32_t)p.Ci * SZ;
    int a_pix[4], a_iy0[4], a_ix0[4];
    uint32_t a_off[4], b_off[2];
    int l_tile = vb, l_tap = 0, l_kc = 0;
    uint32_t l_wtap_off = (uint32_t)p.wtap[0] * row_bytes;

    auto loader_tile = [&](int tile) {
        const int bm = fdiv((uint32_t)tile, p.div_tilesN), bn = tile - bm * p.tilesN;
        const int m0 = bm * BM, n0 = bn * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lrow + i * 64;
            if (m < p.M) {
                const int n = fdiv(m, p.div_HoWo);
                const int rem = m - n * p.HoWo;
                const int oy = fdiv(rem, p.div_Wo);
                const int ox = rem - oy * p.Wo;
                a_iy0[i] = oy * p.sy;
                a_ix0[i] = ox * p.sx;
                a_pix[i] = (n * p.Hi + a_iy0[i]) * p.Wi + a_ix0[i];
            } else {
                a_iy0[i] = -(1 << 20);
                a_ix0[i] = 0;
                a_pix[i] = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = n0 + lrow + i * 64;
            b_off[i] = (co < p.Co) ? (uint32_t)co * (uint32_t)(p.w_row_stride * SZ) + src_chunk : kOOB;
        }
    };
    auto set_tap = [&](int t) {
        const int ddy = p.dy[t], ddx = p.dx[t];
        const int dpix = ddy * p.Wi + ddx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int iy = a_iy0[i] + ddy, ix = a_ix0[i] + ddx;
            const bool ok = ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
            a_off[i] = ok ? (uint32_t)(a_pix[i] + dpix) * row_bytes + src_chunk : kOOB;
        }
    };
    const uint32_t smem_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    auto loader_issue = [&](int stage) {  // exactly 6 DMA instructions per wave: the vmcnt(6) below counts them
        const uint32_t sA = smem_lds + (uint32_t)(stage * STAGE + wave * 1024);
        const uint32_t ka = (uint32_t)l_kc * 128u;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(rx, sA + i * 8192, a_off[i], ka);
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16(rw, sA + A_BYTES + i * 8192, b_off[i], l_wtap_off + ka);
    };
    auto loader_advance = [&]() {
        if (++l_kc == p.kchunks) {
            l_kc = 0;
            if (++l_tap == p.ntaps) {
                l_tap = 0;
                l_tile += G;
                if (l_tile < ntiles) loader_tile(l_tile);
            }
            set_tap(l_tap);
            l_wtap_off = (uint32_t)p.wtap[l_tap] * row_bytes;
        }
    };

    // ---- MFMA side ------------------------------------------------------------------------------------------
    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();
    const int frow = lane & 31, fh = lane >> 5;
    uint32_t fa_base[2], fb_base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa_base[i] = lds_off(wm * 64 + i * 32 + frow, fh);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb_base[j] = A_BYTES + lds_off(wn * 64 + j * 32 + frow, fh);

    uint4 fa[4][2], fb[4][2];  // all fragments of one unit: read in phase 1, consumed by the 16 MFMAs of phase 2
    auto read_frags = [&](int stage) {
        const char* sS = smem + stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[kk][i] = *reinterpret_cast<const uint4*>(sS + (fa_base[i] ^ (kk << 5)));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[kk][j] = *reinterpret_cast<const uint4*>(sS + (fb_base[j] ^ (kk << 5)));
        }
    };
    auto mma_block = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma_frag<T>(fa[kk][i], fb[kk][j], acc[i][j]);
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- epilogue of one tile (per wave, private LDS window) ---------------------------------------------------
    const bool dense = (p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.HWout == p.HoWo);
    const bool fz = p.fz_x != nullptr;
    const bool ep = p.ep_scale != nullptr;
    const bool want_sums = fz || p.partials != nullptr;
    float* stg = reinterpret_cast<float*>(smem + EPI_OFF + wave * EPI_WAVE);  // [8 rows][64 columns] f32
    const int rrow = lane >> 3, cc = lane & 7;

    auto epilogue = [&](int tile) {
        const int bm = fdiv((uint32_t)tile, p.div_tilesN), bn = tile - bm * p.tilesN;
        const int m0 = bm * BM + wm * 64, n0 = bn * BN + wn * 64;
        const int ncol = n0 + cc * EPC;
        const bool colok = ncol < p.Co;
        // f_mu/f_is double as the inference epilogue's per-channel shift/scale (the two modes are exclusive)
        float f_mu[EPC], f_is[EPC], f_s1[EPC], f_s2[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            f_s1[e] = 0.f;
            f_s2[e] = 0.f;
            f_mu[e] = colok ? (fz ? p.fz_mean[ncol + e] : (ep ? p.ep_shift[ncol + e] : 0.f)) : 0.f;
            f_is[e] = colok ? (fz ? p.fz_invstd[ncol + e] : (ep ? p.ep_scale[ncol + e] : 0.f)) : 0.f;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            long e_off[4];
            uint4 pre_add[4], pre_x[4];
            unsigned pre_mk[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // operands of the store loop: requested before the LDS round trips
                const int m = m0 + (half * 4 + k) * 8 + rrow;
                e_off[k] = -1;
                pre_add[k] = make_uint4(0, 0, 0, 0);
                pre_x[k] = make_uint4(0, 0, 0, 0);
                pre_mk[k] = 0xffu;
                if (m < p.M && colok) {
                    long opix = m;
                    if (!dense) {
                        const int nn = fdiv(m, p.div_HoWo);
                        const int rem = m - nn * p.HoWo;
                        const int oy = fdiv(rem, p.div_Wo);
                        const int ox = rem - oy * p.Wo;
                        opix = (long)nn * p.HWout + (long)(oy * p.osy + p.ooy) * p.Wout + (ox * p.osx + p.oox);
                    }
                    e_off[k] = opix * p.Co + ncol;
                    if (p.addend) pre_add[k] = *reinterpret_cast<const uint4*>(p.addend + e_off[k] * SZ);
                    if (fz) {
                        pre_x[k] = *reinterpret_cast<const uint4*>(p.fz_x + e_off[k] * SZ);
                        if (p.fz_mask) pre_mk[k] = p.fz_mask[e_off[k] / EPC];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                constexpr int dummy = 0;
                (void)dummy;
                const int q = half * 4 + k;  // rows 8q .. 8q+7 of the wave's 64: accumulator (q>>2), registers 4*(q&3)..+3
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) stg[(rr + 4 * fh) * 64 + j * 32 + frow] = acc[q >> 2][j][(q & 3) * 4 + rr];
                float v[EPC];
                {
                    const float4 q0 = *reinterpret_cast<const float4*>(&stg[rrow * 64 + cc * EPC]);
                    const float4 q1 = *reinterpret_cast<const float4*>(&stg[rrow * 64 + cc * EPC + 4]);
                    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
                    v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
                }
                if (e_off[k] < 0) continue;
                const long boff = e_off[k] * SZ;
                if (ep) {  // eval-mode BatchNorm folded into the epilogue (f32, before the residual add)
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = v[e] * f_is[e] + f_mu[e];
                }
                if (p.addend) {
                    float a[EPC];
                    unpack16<T>(pre_add[k], a);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] += a[e];
                }
                if (ep && p.ep_relu) {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (fz) {  // BN-backward phase 1 of the producer BatchNorm (see conv_igemm.hip)
                    float xv[EPC];
                    unpack16<T>(pre_x[k], xv);
                    const unsigned mk = pre_mk[k];
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = ((mk >> e) & 1u) ? v[e] : 0.f;
                    const uint4 packed = pack16<T>(v);
                    float dzr[EPC];
                    unpack16<T>(packed, dzr);  // sums are those of the STORED (rounded) dz
#pragma unroll
                    for (int e = 0; e < EPC; ++e) {
                        f_s1[e] += dzr[e];
                        f_s2[e] += dzr[e] * (xv[e] - f_mu[e]) * f_is[e];
                    }
                    *reinterpret_cast<uint4*>(p.y + boff) = packed;
                } else {
                    const uint4 packed = pack16<T>(v);
                    if (p.partials) {  // train-mode BN statistics of the stored (rounded) outputs
                        float yr[EPC];
                        unpack16<T>(packed, yr);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            f_s1[e] += yr[e];
                            f_s2[e] += yr[e] * yr[e];
                        }
                    }
                    *reinterpret_cast<uint4*>(p.y + boff) = packed;
                }
            }
        }
        if (want_sums) {
            // rows live in lane bits 3..5: fold them and publish [2][64] in this wave's window; combine() adds the
            // two waves of a 128-row half later, behind a barrier
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    f_s1[e] += __shfl_xor(f_s1[e], o, 64);
                    f_s2[e] += __shfl_xor(f_s2[e], o, 64);
                }
            }
            if (rrow == 0) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    stg[cc * EPC + e] = f_s1[e];
                    stg[64 + cc * EPC + e] = f_s2[e];
                }
            }
        }
    };
    // Sums of one 128-row half (= one partial row, the granularity of the 128-row kernel): the even-wm wave adds its
    // own and its odd-wm neighbour's [2][64]; both are in the same wave group, see the schedule below.
    auto combine = [&](int tile) {
        if (!want_sums || (wm & 1)) return;
        const int bm = fdiv((uint32_t)tile, p.div_tilesN), bn = tile - bm * p.tilesN;
        const int prow = bm * 2 + (wm >> 1);
        const float* o = reinterpret_cast<const float*>(smem + EPI_OFF + (wave + 2) * EPI_WAVE);
        const float a1 = stg[lane] + o[lane], a2 = stg[64 + lane] + o[64 + lane];
        const int col = bn * BN + wn * 64 + lane;
        if (col < p.Co && prow * 128 < p.M) {
            float* dst = fz ? p.fz_partials + (long)(p.fz_row_off + prow) * 2 * p.Co : p.partials + (long)prow * 2 * p.Co;
            dst[col] = a1;
            dst[p.Co + col] = a2;
        }
    };

    // ---- the unit loop ------------------------------------------------------------------------------------------
    // Two wave groups (A = waves 0-3, B = waves 4-7; SIMD s hosts waves s and s+4) run the same code ONE BARRIER
    // apart, so on every SIMD one wave is in its MFMA phase while the other issues DMAs and LDS reads:
    //   B1 | phase 1(u): issue DMA(u+2); ds_read all fragments of unit u; wait until my DMA(u+1) landed | B2 |
    //      | phase 2(u): 16 MFMAs (+ epilogue after the last K-step of a tile)
    // Barrier instances, numbered globally: A passes B1(u), B2(u) as #2u, #2u+1; B as #2u+1, #2u+2.
    //   read-after-DMA: stage(u) is read after B1(u); every wave confirmed its part of DMA(u) before its B2(u-1)
    //     (prologue for u = 0), which precedes both groups' B1(u).
    //   DMA-after-read: DMA(u+2) overwrites stage(u-1) after B1(u) (>= #2u); its last reads (group B, phase 1(u-1))
    //     completed (lgkmcnt(0)) before B's B2(u-1) = #2u.
    //   sums: staged in phase 2 of the tile's last unit; combined in the next phase 1 (after that group's next
    //     barrier), overwritten no earlier than the phase 2 after that.
    const bool groupB = wave >= 4;
    loader_tile(vb);
    set_tap(0);
    loader_issue(0);
    loader_advance();
    if (total_units > 1) {
        loader_issue(1);
        loader_advance();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (groupB) asm volatile("s_barrier" ::: "memory");
    int c_tile = vb, cs = 0, st = 0, pending = -1;
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    auto stamp = [&](int seg) {
        if constexpr (DIAG) {
            const long long t1 = __builtin_amdgcn_s_memtime();
            tacc[seg] += t1 - t0;
            t0 = t1;
        }
    };
    if constexpr (DIAG) t0 = __builtin_amdgcn_s_memtime();
    for (int u = 0; u < total_units; ++u) {
        asm volatile("s_barrier" ::: "memory");  // B1
        stamp(0);
        if (u + 2 < total_units) {
            loader_issue(st == 0 ? 2 : st - 1);
            loader_advance();
        }
        stamp(1);
        read_frags(st);
        if (pending >= 0) {
            combine(pending);
            pending = -1;
        }
        stamp(2);
        if (u + 2 < total_units)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // all but the six just issued: DMA(u+1) has landed
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp(4);
        asm volatile("s_barrier" ::: "memory");  // B2
        stamp(5);
        mma_block();
        if constexpr (DIAG) asm volatile("s_nop 0" ::"v"(acc[0][0][0]), "v"(acc[1][1][15]));
        stamp(6);
        st = (st == 2) ? 0 : st + 1;
        if (++cs == nsteps) {
            cs = 0;
            epilogue(c_tile);
            zero_acc();
            pending = c_tile;
            c_tile += G;
            stamp(7);
        }
    }
    if constexpr (DIAG) {
        if (lane == 0)
            for (int i = 0; i < 8; ++i) dbg[((long)blockIdx.x * 8 + wave) * 8 + i] = tacc[i];
    }
    // both groups pass the same number of barriers: A 2U+2, B 1+2U+1
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (pending >= 0) combine(pending);
    if (!groupB) asm volatile("s_barrier" ::: "memory");
}

int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n = v;
    }
    return n;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

}  // namespace

bool conv_v2_eligible(const sm3_conv_desc* d) {
    const int enabled = env_int("SM3_CONV_V2", 0);  // read per call: lets one process A/B the two structures
    const int min_tiles = env_int("SM3_CONV_V2_MIN_TILES", 128);
    if (!enabled || d->dtype != SM3_BF16 || d->Co < 128) return false;
    const long M = (long)d->N * d->Ho * d->Wo;
    const long tiles = ((M + BM - 1) / BM) * ((d->Co + BN - 1) / BN);
    return tiles >= min_tiles;
}

int launch_conv_v2(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    p.tilesM = (p.M + BM - 1) / BM;
    p.tilesN = (p.Co + BN - 1) / BN;
    p.div_tilesN = make_fastdiv((uint32_t)p.tilesN);
    const long ntiles = (long)p.tilesM * p.tilesN;
    if (ntiles <= 0 || ntiles > 0x7fffffffL) return SM3_EINVAL;
    auto kern = conv_igemm_v2_kernel<bf16_t, false>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ncu = cu_count();
    const int grid = ntiles < ncu ? (int)ntiles : ncu;
    if (env_int("SM3_CONV_V2_DIAG", 0)) {  // diagnostic: synchronous, prints the segment averages (cycles per unit)
        auto dk = conv_igemm_v2_kernel<bf16_t, true>;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dk), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        long long* dbg = nullptr;
        const size_t n = (size_t)grid * 64;
        if (hipMalloc(&dbg, n * 8) != hipSuccess) return SM3_EINVAL;
        hipLaunchKernelGGL(dk, dim3(grid), dim3(NT), LDS_BYTES, st, p, (int)ntiles, dbg);
        (void)hipStreamSynchronize(st);
        long long* h = (long long*)malloc(n * 8);
        (void)hipMemcpy(h, dbg, n * 8, hipMemcpyDeviceToHost);
        double a[2][8] = {};
        for (int b = 0; b < grid; ++b)
            for (int w = 0; w < 8; ++w)
                for (int i = 0; i < 8; ++i) a[w >= 4][i] += (double)h[((size_t)b * 8 + w) * 8 + i];
        const double units = (double)ntiles * p.ntaps * p.kchunks * 4;  // per group: 4 waves per workgroup
        static const char* names[8] = {"B1", "dma_issue", "ds_read", "vmcnt", "lgkmcnt", "B2", "mfma", "epilogue"};
        fprintf(stderr, "[v2 diag] M=%d Co=%d taps=%d kchunks=%d tiles=%ld grid=%d: cycles per unit per wave (group A | B)\n", p.M, p.Co, p.ntaps, p.kchunks, ntiles, grid);
        for (int i = 0; i < 8; ++i) fprintf(stderr, "   %-10s %8.0f | %8.0f\n", names[i], a[0][i] / units, a[1][i] / units);
        free(h);
        (void)hipFree(dbg);
        return 0;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_BYTES, st, p, (int)ntiles, (long long*)nullptr);
    SM3_CHECK_LAUNCH();
    return 0;
}

}  // namespace sm3conv
