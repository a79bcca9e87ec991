// Weight gradient of the gather-GEMM convolution on MFMA (gfx950).
//
//   dW[co][wtap[t]][ci] += sum_{m=(n,oy,ox)} dY[m, co] * X[pix(m,t), ci]
//
// GEMM with the pixel index as K: both operands arrive "K-major" ([pixel][channel], channel
// contiguous), which is the transposed form of what an MFMA fragment wants (8 consecutive k per
// lane).  bf16: tiles are staged [pixel][channel] in LDS (row pitch padded by 64 B) and fragments
// are fetched with ds_read_b64_tr_b16, the hardware transposing read.  f32: v_mfma_f32_32x32x2_f32
// takes one k per lane, so plain ds_read_b32 along the channel axis is already conflict-free.
// Split over pixels across workgroups (grid.y); partial tiles are combined with f32 atomics
// straight into the (already accumulating) fp32 gradient buffer, one 128-byte row segment per
// half-wave.
//
// Reference call sites replaced: autograd weight-gradients of nn.Conv2d (src/models/resnet.py:49-67)
// and nn.Linear (src/models/simclr.py:17-27) inside loss.backward() (tools/backbone_train.py:125).
#include "common.h"

namespace {

struct WgradParams {
    const char* x;
    const char* dy;
    float* dw;
    int M, Hi, Wi, Ci, Co;
    int sy, sx, ntaps;
    int dyt[SM3_MAX_TAPS], dxt[SM3_MAX_TAPS], wtap[SM3_MAX_TAPS];
    int w_row_stride;
    int HoWo, Wo;
    FastDiv div_HoWo, div_Wo;
    int tilesCo, tilesCi;   // grid.x = tilesCo * ntaps * tilesCi
    int k_per_split;        // pixels per grid.y slice (multiple of KP)
};

constexpr int KP = 32;  // pixels per K-step

template <typename T, int BMW, int BNW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int SZ = sizeof(T);
    constexpr bool kBf16 = (SZ == 2);
    constexpr int PAD = kBf16 ? 64 : 0;
    constexpr int PITCH_A = BMW * SZ + PAD, PITCH_B = BNW * SZ + PAD;
    constexpr int A_BYTES = KP * PITCH_A, B_BYTES = KP * PITCH_B, STAGE = A_BYTES + B_BYTES;
    constexpr int CPR_A = BMW * SZ / 16, CPR_B = BNW * SZ / 16;  // 16-byte chunks per tile row
    constexpr int AI = KP * CPR_A / 256, BI = KP * CPR_B / 256;
    constexpr int WTM = BMW / 2, WTN = BNW / 2, TM = WTM / 32, TN = WTN / 32;
    static_assert(AI >= 1 && BI >= 1 && TM >= 1 && TN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int bx = blockIdx.x;
    const int tci = bx % p.tilesCi;
    bx /= p.tilesCi;
    const int tap = bx % p.ntaps;
    const int tco = bx / p.ntaps;
    const int co0 = tco * BMW, ci0 = tci * BNW;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.M, kbeg + p.k_per_split);
    if (kbeg >= kend) return;
    const int ddy = p.dyt[tap], ddx = p.dxt[tap];

    uint4 ra[AI], rb[BI];
    const int a_c = tid % CPR_A, a_r0 = tid / CPR_A;
    const int b_c = tid % CPR_B, b_r0 = tid / CPR_B;
    constexpr int A_RPP = 256 / CPR_A, B_RPP = 256 / CPR_B;
    const bool a_col_ok = (co0 + a_c * (16 / SZ)) < p.Co;
    const bool b_col_ok = (ci0 + b_c * (16 / SZ)) < p.Ci;

    auto load_regs = [&](int k0) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int m = k0 + a_r0 + i * A_RPP;
            ra[i] = make_uint4(0, 0, 0, 0);
            if (m < kend && a_col_ok)
                ra[i] = *reinterpret_cast<const uint4*>(p.dy + ((long)m * p.Co + co0) * SZ + a_c * 16);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int m = k0 + b_r0 + i * B_RPP;
            rb[i] = make_uint4(0, 0, 0, 0);
            if (m < kend && b_col_ok) {
                const int n = fdiv(m, p.div_HoWo);
                const int rem = m - n * p.HoWo;
                const int oy = fdiv(rem, p.div_Wo);
                const int ox = rem - oy * p.Wo;
                const int iy = oy * p.sy + ddy, ix = ox * p.sx + ddx;
                if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
                    const long pix = ((long)n * p.Hi + iy) * p.Wi + ix;
                    rb[i] = *reinterpret_cast<const uint4*>(p.x + (pix * p.Ci + ci0) * SZ + b_c * 16);
                }
            }
        }
    };
    auto store_lds = [&](int stage) {
        char* sA = smem + stage * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < AI; ++i)
            *reinterpret_cast<uint4*>(sA + (a_r0 + i * A_RPP) * PITCH_A + a_c * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < BI; ++i)
            *reinterpret_cast<uint4*>(sB + (b_r0 + i * B_RPP) * PITCH_B + b_c * 16) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nsteps = (kend - kbeg + KP - 1) / KP;
    load_regs(kbeg);
    store_lds(0);
    __syncthreads();

    for (int s = 0; s < nsteps; ++s) {
        const bool more = (s + 1 < nsteps);
        if (more) load_regs(kbeg + (s + 1) * KP);
        const char* sA = smem + (s & 1) * STAGE;
        const char* sB = sA + A_BYTES;
        if constexpr (kBf16) {
            // transposing read: 16-lane group g reads a 4(k) x 16(channel) block; lane 4q+p supplies the
            // address of k-row q, channels 4p..4p+3, and receives the 4 k values of channel (lane & 15).
            const int g = lane >> 4, ii = lane & 15, q = ii >> 2, pq = ii & 3, h = g >> 1;
            const int colsel = 16 * (g & 1) + 4 * pq;
#pragma unroll
            for (int ks = 0; ks < KP / 16; ++ks) {
                const int krow = ks * 16 + 8 * h + q;
                uint4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int col = wm * WTM + i * 32 + colsel;
                    const char* a0 = sA + krow * PITCH_A + col * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0 + 4 * PITCH_A));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = wn * WTN + j * 32 + colsel;
                    const char* b0 = sB + krow * PITCH_B + col * 2;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(b0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(b0 + 4 * PITCH_B));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, fa[i]), __builtin_bit_cast(bf16x8, fb[j]), acc[i][j], 0, 0, 0);
            }
        } else {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll 4
            for (int ks = 0; ks < KP / 2; ++ks) {
                const int krow = ks * 2 + h;
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[i] = *reinterpret_cast<const float*>(sA + krow * PITCH_A + (wm * WTM + i * 32 + r) * 4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[j] = *reinterpret_cast<const float*>(sB + krow * PITCH_B + (wn * WTN + j * 32 + r) * 4);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) store_lds((s + 1) & 1);
        __syncthreads();
    }

    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int ci = ci0 + wn * WTN + j * 32 + frow;
                const long kcol = (long)p.wtap[tap] * p.Ci + ci;  // column inside the dw row
                if (co < p.Co && ci < p.Ci && kcol < p.w_row_stride)
                    atomicAdd(p.dw + (long)co * p.w_row_stride + kcol, acc[i][j][r]);
            }
}

template <typename T, int BMW, int BNW>
int launch_wgrad(WgradParams p, hipStream_t st) {
    constexpr int SZ = sizeof(T);
    constexpr int PAD = (SZ == 2) ? 64 : 0;
    constexpr int LDS = 2 * KP * ((BMW * SZ + PAD) + (BNW * SZ + PAD));
    p.tilesCo = (p.Co + BMW - 1) / BMW;
    p.tilesCi = (p.Ci + BNW - 1) / BNW;
    const long gx = (long)p.tilesCo * p.ntaps * p.tilesCi;
    // split the pixel axis so that ~1024 workgroups are in flight, >= 8 K-steps each
    long splits = (1024 + gx - 1) / gx;
    const long max_splits = (p.M + KP * 8 - 1) / (KP * 8);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long kps = (p.M + splits - 1) / splits;
    kps = (kps + KP - 1) / KP * KP;
    splits = (p.M + kps - 1) / kps;
    p.k_per_split = (int)kps;
    auto kern = conv_wgrad_kernel<T, BMW, BNW>;
    static bool attr_set = false;
    if (!attr_set && LDS > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    if (gx > 0x7fffffffL || splits > 65535) return SM3_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (unsigned)splits), dim3(256), LDS, st, p);
    SM3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int sm3_conv_wgrad(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, void* stream) {
    if (!d || !x || !dy || !dw) return SM3_EINVAL;
    if (d->dtype != SM3_F32 && d->dtype != SM3_BF16) return SM3_EDTYPE;
    const int sz = d->dtype == SM3_F32 ? 4 : 2;
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ci <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Co <= 0) return SM3_EINVAL;
    if (d->ntaps < 1 || d->ntaps > SM3_MAX_TAPS) return SM3_EINVAL;
    if ((d->Ci * sz) % 16 != 0 || (d->Co * sz) % 16 != 0) return SM3_EALIGN;
    if (d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0 || d->Hout != d->Ho || d->Wout != d->Wo)
        return SM3_EINVAL;  // dy must be the dense output of the forward conv
    const long M = (long)d->N * d->Ho * d->Wo;
    if (M > 0x7fffffffL || (long)d->N * d->Hi * d->Wi > 0x7fffffffL) return SM3_EINVAL;
    WgradParams p;
    p.x = (const char*)x; p.dy = (const char*)dy; p.dw = dw;
    p.M = (int)M; p.Hi = d->Hi; p.Wi = d->Wi; p.Ci = d->Ci; p.Co = d->Co;
    p.sy = d->sy; p.sx = d->sx; p.ntaps = d->ntaps;
    for (int t = 0; t < SM3_MAX_TAPS; ++t) { p.dyt[t] = d->dy[t]; p.dxt[t] = d->dx[t]; p.wtap[t] = d->wtap[t]; }
    p.w_row_stride = d->w_row_stride;
    p.HoWo = d->Ho * d->Wo; p.Wo = d->Wo;
    p.div_HoWo = make_fastdiv((uint32_t)p.HoWo);
    p.div_Wo = make_fastdiv((uint32_t)p.Wo);
    hipStream_t st = (hipStream_t)stream;
    const bool nco = d->Co <= 64, nci = d->Ci <= 64;
    if (d->dtype == SM3_BF16) {
        if (nco && nci) return launch_wgrad<bf16_t, 64, 64>(p, st);
        if (nco) return launch_wgrad<bf16_t, 64, 128>(p, st);
        if (nci) return launch_wgrad<bf16_t, 128, 64>(p, st);
        return launch_wgrad<bf16_t, 128, 128>(p, st);
    }
    if (nco && nci) return launch_wgrad<float, 64, 64>(p, st);
    if (nco) return launch_wgrad<float, 64, 128>(p, st);
    if (nci) return launch_wgrad<float, 128, 64>(p, st);
    return launch_wgrad<float, 128, 128>(p, st);
}
