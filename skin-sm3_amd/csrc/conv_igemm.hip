// Gather-GEMM convolution on MFMA (gfx950): forward conv, data gradient, bias-free Linear.
//
//   Y[m, co] = sum_t sum_ci X[pix(m,t), ci] * W[co][wtap[t]][ci]        m = (n, oy, ox)
//
// One 256-thread workgroup computes a BM x BN tile of Y.  Per K-step every row of the A tile
// (activations, gathered per tap) and of the B tile (filters) contributes 128 contiguous bytes
// of K (64 bf16 / 32 f32), staged global -> registers -> LDS (XOR-swizzled 16-byte chunks) and
// consumed as 16-byte MFMA fragments: one v_mfma_f32_32x32x16_bf16 per fragment pair in bf16,
// four v_mfma_f32_32x32x2_f32 (exact f32) in f32 mode.  The byte-level data movement is the same
// for both dtypes.  Epilogue: accumulators -> LDS (f32) -> 16-byte coalesced NHWC stores, plus
// per-row-block partial sums (sum, sum of squares of the stored values) for train-mode BatchNorm.
//
// Reference call sites replaced: src/models/resnet.py:49-67 (conv3x3 / conv1x1) as used at
// :144-148,:260; nn.Linear(bias=False) in src/models/simclr.py:17-27; and their autograd
// data-gradients.
#include "common.h"

namespace {

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
};

template <typename T>
__device__ __forceinline__ void mma_frag(const uint4& a, const uint4& b, f32x16& c);

template <>
__device__ __forceinline__ void mma_frag<bf16_t>(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
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

// byte offset of 16-byte chunk `c` of row `r` inside a [rows][128 B] LDS tile.  Two rows share a
// 256-byte bank row, so the swizzle key is the row pair: the 16 lanes of every ds_read_b128 group
// then hit 16 distinct 16-byte slots.
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + (((c ^ (r >> 1)) & 7) << 4); }

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(const ConvParams p) {
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;  // rows covered per loader pass
    constexpr int AI = BM / RPP, BI = BN / RPP;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int SZ = sizeof(T);
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int C_BYTES = BM * BN * 4;
    constexpr int MAIN_BYTES = (2 * STAGE > C_BYTES) ? 2 * STAGE : C_BYTES;
    static_assert(AI >= 1 && BI >= 1 && TM >= 1 && TN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware block remap (bijective): blocks that share an A row-panel run on one XCD's L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int bn = bid % p.tilesN, bm = bid / p.tilesN;
    const int m0 = bm * BM, n0 = bn * BN;

    // ---- loader state -------------------------------------------------------------------
    const int c16 = tid & 7;
    int a_n[AI], a_iy0[AI], a_ix0[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int m = m0 + (tid >> 3) + i * RPP;
        if (m < p.M) {
            const int n = fdiv(m, p.div_HoWo);
            const int rem = m - n * p.HoWo;
            const int oy = fdiv(rem, p.div_Wo);
            const int ox = rem - oy * p.Wo;
            a_n[i] = n;
            a_iy0[i] = oy * p.sy;
            a_ix0[i] = ox * p.sx;
        } else {
            a_n[i] = -1;
            a_iy0[i] = 0;
            a_ix0[i] = 0;
        }
    }
    const char* a_ptr[AI];
    const char* b_ptr[BI];
    uint4 ra[AI], rb[BI];

    auto set_tap = [&](int t) {
        const int ddy = p.dy[t], ddx = p.dx[t];
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int iy = a_iy0[i] + ddy, ix = a_ix0[i] + ddx;
            const bool ok = (a_n[i] >= 0) && ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
            const long pix = ((long)a_n[i] * p.Hi + iy) * p.Wi + ix;
            a_ptr[i] = ok ? p.x + pix * (long)p.Ci * SZ + c16 * 16 : nullptr;
        }
        const long woff = (long)p.wtap[t] * p.Ci * SZ + c16 * 16;
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int co = n0 + (tid >> 3) + i * RPP;
            b_ptr[i] = (co < p.Co) ? p.w + (long)co * p.w_row_stride * SZ + woff : nullptr;
        }
    };
    auto load_regs = [&](int kc) {
        const long koff = (long)kc * 128;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            ra[i] = a_ptr[i] ? *reinterpret_cast<const uint4*>(a_ptr[i] + koff) : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BI; ++i)
            rb[i] = b_ptr[i] ? *reinterpret_cast<const uint4*>(b_ptr[i] + koff) : make_uint4(0, 0, 0, 0);
    };
    auto store_lds = [&](int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int r = (tid >> 3) + i * RPP;
            *reinterpret_cast<uint4*>(sA + lds_off(r, c16)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int r = (tid >> 3) + i * RPP;
            *reinterpret_cast<uint4*>(sB + lds_off(r, c16)) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nsteps = p.ntaps * p.kchunks;
    int t = 0, kc = 0;
    set_tap(0);
    load_regs(0);
    store_lds(0);
    __syncthreads();

    const int frow = lane & 31, fh = lane >> 5;
    for (int s = 0; s < nsteps; ++s) {
        const bool more = (s + 1 < nsteps);
        if (more) {
            if (++kc == p.kchunks) {
                kc = 0;
                ++t;
                set_tap(t);
            }
            load_regs(kc);
        }
        const char* sA = smem + (s & 1) * STAGE;
        const char* sB = sA + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int cc = 2 * kk + fh;
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm * WTM + i * 32 + frow;
                fa[i] = *reinterpret_cast<const uint4*>(sA + lds_off(r, cc));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn * WTN + j * 32 + frow;
                fb[j] = *reinterpret_cast<const uint4*>(sB + lds_off(r, cc));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_frag<T>(fa[i], fb[j], acc[i][j]);
        }
        if (more) store_lds((s + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue -----------------------------------------------------------------------
    float* sC = reinterpret_cast<float*>(smem);
    float* sStat = reinterpret_cast<float*>(smem + MAIN_BYTES);  // [WM][BN][2]
    if (p.partials) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = ElemTraits<T>::round(acc[i][j][r]);
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (lane < 32) {
                const int col = wn * WTN + j * 32 + lane;
                sStat[(wm * BN + col) * 2 + 0] = s1;
                sStat[(wm * BN + col) * 2 + 1] = s2;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int col = wn * WTN + j * 32 + frow;
                sC[row * BN + col] = acc[i][j][r];
            }
    __syncthreads();

    if (p.partials && tid < BN && n0 + tid < p.Co) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
            s1 += sStat[(w * BN + tid) * 2 + 0];
            s2 += sStat[(w * BN + tid) * 2 + 1];
        }
        p.partials[((long)bm * 2 + 0) * p.Co + n0 + tid] = s1;
        p.partials[((long)bm * 2 + 1) * p.Co + n0 + tid] = s2;
    }

    constexpr int EPC = 16 / SZ;     // elements per 16-byte store
    constexpr int CPR = BN / EPC;    // stores per tile row
    const bool dense = (p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.HWout == p.HoWo);
    for (int idx = tid; idx < BM * CPR; idx += NT) {
        const int r = idx / CPR, cc = idx % CPR;
        const int m = m0 + r, n = n0 + cc * EPC;
        if (m >= p.M || n >= p.Co) continue;
        long opix = m;
        if (!dense) {
            const int nn = fdiv(m, p.div_HoWo);
            const int rem = m - nn * p.HoWo;
            const int oy = fdiv(rem, p.div_Wo);
            const int ox = rem - oy * p.Wo;
            opix = (long)nn * p.HWout + (long)(oy * p.osy + p.ooy) * p.Wout + (ox * p.osx + p.oox);
        }
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
            const float4 q = *reinterpret_cast<const float4*>(&sC[r * BN + cc * EPC + e]);
            v[e] = q.x;
            v[e + 1] = q.y;
            v[e + 2] = q.z;
            v[e + 3] = q.w;
        }
        const long boff = (opix * p.Co + n) * SZ;
        if (p.addend) {
            float a[EPC];
            unpack16<T>(*reinterpret_cast<const uint4*>(p.addend + boff), a);
#pragma unroll
            for (int e = 0; e < EPC; ++e) v[e] += a[e];
        }
        *reinterpret_cast<uint4*>(p.y + boff) = pack16<T>(v);
    }
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_conv(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int C_BYTES = BM * BN * 4;
    constexpr int MAIN = (2 * STAGE > C_BYTES) ? 2 * STAGE : C_BYTES;
    constexpr int LDS = MAIN + WM * BN * 2 * 4;
    p.tilesM = (p.M + BM - 1) / BM;
    p.tilesN = (p.Co + BN - 1) / BN;
    auto kern = conv_igemm_kernel<T, BM, BN, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const long nblocks = (long)p.tilesM * p.tilesN;
    if (nblocks <= 0 || nblocks > 0x7fffffffL) return SM3_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(WM * WN * 64), LDS, st, p);
    SM3_CHECK_LAUNCH();
    return 0;
}

constexpr int kBM = 128;

int fill_params(const sm3_conv_desc* d, ConvParams& p, int sz) {
    if (!d) return SM3_EINVAL;
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ci <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Co <= 0) return SM3_EINVAL;
    if (d->ntaps < 1 || d->ntaps > SM3_MAX_TAPS) return SM3_EINVAL;
    if ((d->Ci * sz) % 128 != 0) return SM3_EALIGN;
    if ((d->Co * sz) % 16 != 0) return SM3_EALIGN;
    const long M = (long)d->N * d->Ho * d->Wo;
    if (M > 0x7fffffffL) return SM3_EINVAL;
    if ((long)d->N * d->Hi * d->Wi > 0x7fffffffL) return SM3_EINVAL;
    p.M = (int)M;
    p.Hi = d->Hi; p.Wi = d->Wi; p.Ci = d->Ci; p.Co = d->Co;
    p.sy = d->sy; p.sx = d->sx; p.ntaps = d->ntaps;
    for (int t = 0; t < SM3_MAX_TAPS; ++t) { p.dy[t] = d->dy[t]; p.dx[t] = d->dx[t]; p.wtap[t] = d->wtap[t]; }
    p.w_row_stride = d->w_row_stride;
    p.Wout = d->Wout; p.HWout = d->Hout * d->Wout;
    p.osy = d->osy; p.osx = d->osx; p.ooy = d->ooy; p.oox = d->oox;
    p.HoWo = d->Ho * d->Wo; p.Wo = d->Wo;
    p.div_HoWo = make_fastdiv((uint32_t)p.HoWo);
    p.div_Wo = make_fastdiv((uint32_t)p.Wo);
    p.kchunks = d->Ci * sz / 128;
    if ((d->Ho - 1) * d->osy + d->ooy >= d->Hout || (d->Wo - 1) * d->osx + d->oox >= d->Wout) return SM3_EINVAL;
    return 0;
}

}  // namespace

extern "C" int sm3_conv_partial_rows(const sm3_conv_desc* d) {
    if (!d) return SM3_EINVAL;
    const long M = (long)d->N * d->Ho * d->Wo;
    return (int)((M + kBM - 1) / kBM);
}

extern "C" int sm3_conv_gather_gemm(const sm3_conv_desc* d, const void* x, const void* w, void* y,
                                    const void* addend, float* stat_partials, void* stream) {
    if (!d || !x || !w || !y) return SM3_EINVAL;
    if (d->dtype != SM3_F32 && d->dtype != SM3_BF16) return SM3_EDTYPE;
    ConvParams p;
    const int sz = d->dtype == SM3_F32 ? 4 : 2;
    int rc = fill_params(d, p, sz);
    if (rc) return rc;
    p.x = (const char*)x; p.w = (const char*)w; p.y = (char*)y; p.addend = (const char*)addend;
    p.partials = stat_partials;
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = d->Co <= 64;
    if (d->dtype == SM3_BF16)
        return narrow ? launch_conv<bf16_t, kBM, 64, 2, 2>(p, st) : launch_conv<bf16_t, kBM, 128, 2, 2>(p, st);
    return narrow ? launch_conv<float, kBM, 64, 2, 2>(p, st) : launch_conv<float, kBM, 128, 2, 2>(p, st);
}
