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
#include <stdlib.h>

#include <atomic>

#include "conv_common.h"

namespace {

struct WgradParams {
    const char* x;
    const char* dy;
    float* dw;
    int M, Hi, Wi, Ci, Co;
    int sy, sx, ntaps;
    int dyt[SM3_MAX_TAPS], dxt[SM3_MAX_TAPS], wtap[SM3_MAX_TAPS];
    int w_row_stride;
    int HoWo, Wo, Ho;
    int adv_n, adv_oy, adv_ox;  // one K-step (KP pixels) = adv_n images + adv_oy rows + adv_ox pixels
    FastDiv div_HoWo, div_Wo;
    int tilesCo, tilesCi;   // tiles per pixel slice: gx = tilesCo * ntaps * tilesCi
    int k_per_split;        // pixels per slice (multiple of KP)
    int gx, splits;         // grid = gx * round_up(splits, 8), see the block -> (tile, slice) map in the kernel
    uint32_t x_bytes;
    // sm3_conv_wgrad_cat: dY is the channel concatenation [dy (Co) | dy1 (Co1)] of two tensors over the same pixels, the
    // rows of the second part go to dw1; `views` independent row ranges of Mv pixels each (slices never straddle them),
    // view v accumulating into dw + v * dw_view_stride / dw1 + v * dw1_view_stride.  Plain wgrad: Co1 = 0, views = 1.
    const char* dy1;
    float* dw1;
    int Co1, views, Mv, splits_view;
    long dw_view_stride, dw1_view_stride;
    // sm3_conv_wgrad_slabs: no atomics -- slice j of view v STORES its tiles into dw + (v * splits_view + j) * slab_stride
    // (every slab a full [Co][w_row_stride] matrix); the caller sums the slabs in a fixed order (deterministic results)
    long slab_stride;  // 0: accumulate with float atomics
    int slab_cap;      // slabs available per view
    // sm3_conv_wgrad_det: the slabs are an intermediate of ONE gradient (all views add into the same dw), so the pixel axis is
    // cut for the launch as it is (pair_rule = 0: no "as if two views" rule), and a launch that ends up with a single
    // slice adds its tiles to dw_direct itself (rmw: plain read-modify-write, one writer per element) -- no slab, no reduce
    float* dw_direct;
    int pair_rule, rmw;
};

#ifdef SM3_STAMP
// Diagnostic build only (scratch/build_stamp.sh, scratch/stamp_wgrad.py; in the product library no stamp executes): s_memtime
// at the segment boundaries of the ring loop, summed per wave -- wait (counted vmcnt), barrier, issue (DMA of the stage NST - 1
// ahead), compute (transposing fragment reads + MFMA issue) -- plus entry / loop / exit marks; one 16-word record per wave.
__device__ unsigned long long* g_wstamp_buf = nullptr;
__device__ long g_wstamp_cap = 0;
#define SM3_WSTAMP_NOW() __builtin_amdgcn_s_memtime()
#endif

using sm3conv::dma16;       // LDS-DMA from inline asm, zero-fill by the buffer range check: see conv_common.h
using sm3conv::dma_drain;
using sm3conv::kOOB;

// bf16 tiles are read with ds_read_b64_tr_b16, whose 16-lane groups fetch 4 pixel rows x 32 bytes: with the
// unpadded pitch LDS-DMA needs (a wave-instruction writes 1 KiB contiguously) those 4 rows would share banks, so
// bits 6-7 (256-byte rows) or bit 6 (128-byte rows) of the byte offset are XORed with a key of the pixel row --
// applied on the DMA *source* chunk and on the read address alike.  f32 tiles are read along the channel axis
// with ds_read_b32 and need no swizzle.
template <int ROW_BYTES, bool BF16>
__device__ __forceinline__ uint32_t swz_bytes(int row) {
    if constexpr (!BF16) return 0u;
    else if constexpr (ROW_BYTES >= 256) return (uint32_t)(row & 3) << 6;  // (512-byte rows: the same four 64-byte slots)
    else return (uint32_t)((row >> 1) & 1) << 6;
}

// DENSE: 1x1 / stride 1 / no offset forward conv (X row of output pixel m is simply row m): every DMA offset is
// a per-lane constant plus a scalar that advances by one K-step -- no VALU at all in the loop.
// KG = 2: two groups of 4 waves share a workgroup, each walks half of the workgroup's pixel slice with its own LDS ring
// and accumulators; group 1 hands its tile to group 0 through LDS and ONE set of f32 atomics leaves the workgroup --
// the same waves per CU as two 4-wave workgroups at half the atomic traffic (the 1x1 layers paid 20 % for atomics).
template <typename T, int BMW, int BNW, int KP, bool DENSE, int NST, int KG = 1>  // KP = pixels per K-step, NST = LDS stages
__global__ __launch_bounds__(256 * KG, NST == 1 ? 4 : 1) void conv_wgrad_kernel(const WgradParams p) {
    constexpr int SZ = sizeof(T);
    constexpr bool kBf16 = (SZ == 2);
    constexpr int RA = BMW * SZ, RB = BNW * SZ;                 // bytes per tile row (one pixel)
    constexpr int A_BYTES = KP * RA, B_BYTES = KP * RB, STAGE = A_BYTES + B_BYTES;
    constexpr int RPI_A = 1024 / RA, RPI_B = 1024 / RB;        // pixel rows per DMA wave-instruction
    constexpr int AI = A_BYTES / 4096, BI = B_BYTES / 4096;    // DMA instructions per wave per K-step
    constexpr int WTM = BMW / 2, WTN = BNW / 2, TM = WTM / 32, TN = WTN / 32;
    static_assert(AI >= 1 && BI >= 1 && TM >= 1 && TN >= 1 && RA <= 512 && RB <= 512, "unsupported tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all & 3, grp = wave_all >> 2;  // role inside the group of 4 waves; K-group
#ifdef SM3_STAMP
    const unsigned long long st_entry = SM3_WSTAMP_NOW(), st_rentry = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_seg[4] = {0, 0, 0, 0}, st_loop0 = 0, st_loop1 = 0;
#endif
    const int wm = wave >> 1, wn = wave & 1;

    // Block -> (tile, pixel slice).  Workgroups are dealt round-robin over the 8 XCDs, so id & 7 labels the
    // XCD group; inside a group consecutive workgroups walk all gx tiles of ONE pixel slice before moving to the
    // next slice: the tiles of a slice re-read the same dY / X rows (18x re-read for a 3x3 256->256 layer) and
    // now find them in their XCD's L2 instead of fetching them once per XCD.  Speed only, never correctness.
    // With fewer than 8 slices that map would idle XCDs, so tiles simply go round-robin.
    int bx, slice;
    if (p.splits >= 8) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        bx = q % p.gx;
        slice = (q / p.gx) * 8 + xcd;
    } else {
        bx = blockIdx.x % p.gx;
        slice = blockIdx.x / p.gx;
    }
    if (slice >= p.splits) return;
    const int tci = bx % p.tilesCi;
    bx /= p.tilesCi;
    const int tap = bx % p.ntaps;
    const int tco = bx / p.ntaps;
    const int ci0 = tci * BNW;
    // channel tile of dY: in the first tensor, or (sm3_conv_wgrad_cat) in the second one
    const bool second = tco * BMW >= p.Co;
    const int co0 = second ? tco * BMW - p.Co : tco * BMW;
    const int dyC = second ? p.Co1 : p.Co;  // channels (row pitch) of the tensor this tile reads
    const char* const dy_base = second ? p.dy1 : p.dy;
    const int view = slice / p.splits_view;
    const int kbeg0 = view * p.Mv + (slice - view * p.splits_view) * p.k_per_split;
    const int kend = min((view + 1) * p.Mv, kbeg0 + p.k_per_split);
    if (kbeg0 >= kend) return;
    // K-steps per group (both groups run the same number: rows past kend read as zero), and this group's first pixel
    const int nsteps = ((kend - kbeg0 + KP - 1) / KP + KG - 1) / KG;
    const int kbeg = kbeg0 + grp * nsteps * KP;
    const int ddy = p.dyt[tap], ddx = p.dxt[tap];

    // descriptors: dY rows end at kend (rows of the next slice must read as zero); X is the whole tensor
    const __amdgpu_buffer_rsrc_t rdy =
        __builtin_amdgcn_make_buffer_rsrc((void*)dy_base, 0, (uint32_t)kend * (uint32_t)(dyC * SZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.x, 0, DENSE ? (uint32_t)kend * (uint32_t)(p.Ci * SZ) : p.x_bytes, 0x00020000);

    // lane -> (row inside the DMA instruction, 16-byte chunk inside the row)
    const int a_rin = (lane * 16) / RA, a_pos = ((lane * 16) % RA) / 16;
    const int b_rin = (lane * 16) / RB, b_pos = ((lane * 16) % RB) / 16;
    uint32_t a_off[AI], b_off[BI];
    int b_row[BI];
    uint32_t b_cho[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int r = (wave + 4 * i) * RPI_A + a_rin;  // pixel row inside the K-step
        const uint32_t lchunk = ((uint32_t)a_pos * 16u) ^ swz_bytes<RA, kBf16>(r);
        const bool col_ok = co0 + (int)(lchunk / SZ) < dyC;
        a_off[i] = col_ok ? (uint32_t)(kbeg + r) * (uint32_t)(dyC * SZ) + (uint32_t)(co0 * SZ) + lchunk : kOOB;
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int r = (wave + 4 * i) * RPI_B + b_rin;
        const uint32_t lchunk = ((uint32_t)b_pos * 16u) ^ swz_bytes<RB, kBf16>(r);
        const bool col_ok = ci0 + (int)(lchunk / SZ) < p.Ci;
        b_row[i] = col_ok ? r : -(1 << 28);
        b_cho[i] = (uint32_t)(ci0 * SZ) + lchunk;
        b_off[i] = col_ok ? (uint32_t)(kbeg + r) * (uint32_t)(p.Ci * SZ) + b_cho[i] : kOOB;  // DENSE form
    }
    const uint32_t smem_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    int g_m[BI], g_n[BI], g_oy[BI], g_ox[BI];  // non-DENSE: output pixel of this lane's X rows in the next step to stage
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        g_m[i] = kbeg + b_row[i];  // negative for lanes whose channel chunk lies beyond Ci
        const int m = g_m[i] < 0 ? 0 : g_m[i];
        g_n[i] = fdiv(m, p.div_HoWo);
        const int rem = m - g_n[i] * p.HoWo;
        g_oy[i] = fdiv(rem, p.div_Wo);
        g_ox[i] = rem - g_oy[i] * p.Wo;
    }

    auto dma_stage = [&](int stage, int step) {
        const uint32_t sA = smem_lds + (uint32_t)((grp * NST + stage) * STAGE) + (uint32_t)wave * 1024u;
        const uint32_t sB = sA + A_BYTES;
        const uint32_t soff_a = (uint32_t)(step * KP) * (uint32_t)(dyC * SZ);
#pragma unroll
        for (int i = 0; i < AI; ++i) dma16(rdy, sA + i * 4096, a_off[i], soff_a);
        if constexpr (DENSE) {
            const uint32_t soff_b = (uint32_t)(step * KP) * (uint32_t)(p.Ci * SZ);
#pragma unroll
            for (int i = 0; i < BI; ++i) dma16(rx, sB + i * 4096, b_off[i], soff_b);
        } else {
            // X rows follow the tap-shifted pixel of each output pixel.  (n, oy, ox) of the BI rows this lane stages
            // are carried from step to step (one K-step = adv_n images + adv_oy rows + adv_ox pixels, single carries)
            // instead of two divisions per row per step: that index arithmetic, not the MFMAs, was setting the
            // K-step time of the 3x3 layers.  dma_stage is called with step = 0, 1, 2, ... in order.
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                uint32_t off = kOOB;
                if (g_m[i] >= 0 && g_m[i] < kend) {
                    const int iy = g_oy[i] * p.sy + ddy, ix = g_ox[i] * p.sx + ddx;
                    if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                        off = (uint32_t)((g_n[i] * p.Hi + iy) * p.Wi + ix) * (uint32_t)(p.Ci * SZ) + b_cho[i];
                }
                dma16(rx, sB + i * 4096, off, 0u);
                g_m[i] += KP;
                g_n[i] += p.adv_n;
                g_oy[i] += p.adv_oy;
                g_ox[i] += p.adv_ox;
                if (g_ox[i] >= p.Wo) {
                    g_ox[i] -= p.Wo;
                    ++g_oy[i];
                }
                if (g_oy[i] >= p.Ho) {
                    g_oy[i] -= p.Ho;
                    ++g_n[i];
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane fragment read offsets
    uint32_t ra_off[TM], rb_off[TN];
    if constexpr (kBf16) {
        // transposing read: 16-lane group g reads a 4(k) x 16(channel) block; lane 4q+pq supplies the address of
        // k-row q, channels 4pq..4pq+3, and receives the 4 k values of channel (lane & 15).
        const int g = lane >> 4, ii = lane & 15, q = ii >> 2, pq = ii & 3, h = g >> 1;
        const int colsel = 16 * (g & 1) + 4 * pq;
        const int krow0 = 8 * h + q;  // + 16*ks (+4 for the second half): the swizzle key of the row is that of q
#pragma unroll
        for (int i = 0; i < TM; ++i)
            ra_off[i] = (uint32_t)krow0 * RA + (((uint32_t)(wm * WTM + i * 32 + colsel) * 2u) ^ swz_bytes<RA, true>(q));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            rb_off[j] = A_BYTES + (uint32_t)krow0 * RB + (((uint32_t)(wn * WTN + j * 32 + colsel) * 2u) ^ swz_bytes<RB, true>(q));
    } else {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < TM; ++i) ra_off[i] = (uint32_t)h * RA + (uint32_t)(wm * WTM + i * 32 + r) * 4u;
#pragma unroll
        for (int j = 0; j < TN; ++j) rb_off[j] = A_BYTES + (uint32_t)h * RB + (uint32_t)(wn * WTN + j * 32 + r) * 4u;
    }

    // Ring of NST stages, NST - 1 of them in flight: a K-step's MFMAs (0.1-0.25 us) are far shorter than the round
    // trip of its DMA, so with one stage in flight (NST = 2) the loop runs at the DMA latency; each step waits (counted
    // vmcnt) only for its own stage and passes one barrier.
    constexpr int PER = AI + BI;  // DMA instructions per stage per wave
    static_assert(NST >= 1 && NST <= 4 && 2 * PER <= 63, "vmcnt immediates below cover NST <= 4");
    // NST = 1 (round 4, the lesson of the forward kernel): ONE stage of KP pixels, nothing in flight while it is computed --
    // the workgroups of a CU (four at 32 KB and 128 registers) overlap each other instead of a ring overlapping itself
    for (int s = 0; s < NST - 1 && s < nsteps; ++s) dma_stage(s, s);
#ifdef SM3_STAMP
    st_loop0 = SM3_WSTAMP_NOW();
#endif

    for (int s = 0; s < nsteps; ++s) {
#ifdef SM3_STAMP
        const unsigned long long q0 = SM3_WSTAMP_NOW();
        unsigned long long q1 = q0, q2 = q0, q3 = q0;
#endif
        if constexpr (NST == 1) {
            if (s > 0) __syncthreads();  // everyone is done computing stage s - 1
            dma_stage(0, s);
            dma_drain();
            __syncthreads();  // stage s has landed for everyone
        } else {
            const int younger = min(NST - 2, nsteps - 1 - s);  // stages issued after stage s
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else dma_drain();
#ifdef SM3_STAMP
            q1 = SM3_WSTAMP_NOW();
#endif
            __syncthreads();  // stage s has landed for everyone; everyone is done computing stage s - 1
#ifdef SM3_STAMP
            q2 = SM3_WSTAMP_NOW();
#endif
            if (s + NST - 1 < nsteps) dma_stage((s + NST - 1) % NST, s + NST - 1);
        }
#ifdef SM3_STAMP
        q3 = SM3_WSTAMP_NOW();
#endif
        const char* sS = smem + (grp * NST + s % NST) * STAGE;
        if constexpr (kBf16) {
#pragma unroll
            for (int ks = 0; ks < KP / 16; ++ks) {
                uint4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const char* a0 = sS + ra_off[i] + ks * 16 * RA;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a0 + 4 * RA));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fa[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const char* b0 = sS + rb_off[j] + ks * 16 * RB;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(b0 + 4 * RB));
                    uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    fb[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        sm3conv::mma_frag<T>(fa[i], fb[j], acc[i][j]);
            }
        } else {
#pragma unroll 4
            for (int ks = 0; ks < KP / 2; ++ks) {
                float fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float*>(sS + ra_off[i] + ks * 2 * RA);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float*>(sS + rb_off[j] + ks * 2 * RB);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
#ifdef SM3_STAMP
        const unsigned long long q4 = SM3_WSTAMP_NOW();
        st_seg[0] += q1 - q0; st_seg[1] += q2 - q1; st_seg[2] += q3 - q2; st_seg[3] += q4 - q3;
#endif
    }
#ifdef SM3_STAMP
    st_loop1 = SM3_WSTAMP_NOW();
    struct WStampOut {
        unsigned long long e, re, l0, l1, *seg; int ns, grp, wave;
        __device__ ~WStampOut() {
            const long w = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
            if ((threadIdx.x & 63) == 0 && g_wstamp_buf && w < g_wstamp_cap) {
                unsigned long long* o = g_wstamp_buf + w * 16;
                o[0] = e; o[1] = l0; o[2] = l1; o[3] = __builtin_amdgcn_s_memtime();
                o[4] = seg[0]; o[5] = seg[1]; o[6] = seg[2]; o[7] = seg[3];
                o[8] = (unsigned long long)ns; o[9] = blockIdx.x; o[10] = re; o[11] = __builtin_amdgcn_s_memrealtime();
                o[12] = (unsigned long long)grp; o[13] = (unsigned long long)wave;
            }
        }
    } st_out{st_entry, st_rentry, st_loop0, st_loop1, st_seg, nsteps, grp, wave};
#endif

    if constexpr (KG == 2) {
        static_assert(KG * NST * STAGE >= BMW * BNW * 4, "group 1's tile must fit the workgroup's LDS");
        __syncthreads();  // both groups are done with their rings
        // hand-over area [wave][register][lane]: group 1's own ring where that is large enough, else the whole LDS
        float* xch = reinterpret_cast<float*>(smem + (NST * STAGE >= BMW * BNW * 4 ? NST * STAGE : 0));
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((wave * TM * TN + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += xch[((wave * TM * TN + i * TN + j) * 16 + r) * 64 + lane];
    }
    const int frow = lane & 31, fh = lane >> 5;
    float* const dw_out = p.slab_stride ? p.dw + (long)slice * p.slab_stride
                          : second    ? p.dw1 + (long)view * p.dw1_view_stride
                                      : p.dw + (long)view * p.dw_view_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int ci = ci0 + wn * WTN + j * 32 + frow;
                const long kcol = (long)p.wtap[tap] * p.Ci + ci;  // column inside the dw row
                if (co < dyC && ci < p.Ci && kcol < p.w_row_stride) {
                    if (p.slab_stride) dw_out[(long)co * p.w_row_stride + kcol] = acc[i][j][r];
                    else if (p.rmw) dw_out[(long)co * p.w_row_stride + kcol] += acc[i][j][r];
                    else atomicAdd(dw_out + (long)co * p.w_row_stride + kcol, acc[i][j][r]);
                }
            }
}

// ---- nine-tap owner: the weight gradient of a stride-1 "same" 3x3 convolution (round 6) ----------------------------------
// The tap-shifted kernel above gives every (Co tile, tap, Ci tile) its own workgroup: dY and X cross L2 -> LDS nine times,
// 15.6 KB per MFLOP.  Here a workgroup owns a 64 x 64 (co, ci) tile pair and ALL nine taps.  A stage is R whole rows of ONE
// image (R | H): dY rows [R*W, padded with zero rows to a multiple of 16] and the X "halo image" of the same rows plus one
// image row above and below plus one pixel either side ((R + 2) * W + 2 rows), each staged ONCE by LDS-DMA; tap (dy, dx) of
// a 16-pixel K block is a transposing read of the X image at the row shift W + 1 + dy * W + dx -- a per-lane base address
// per tap, fixed for the whole kernel -- so nine MFMAs share one dY fragment: 3.3 ... 5 KB per MFLOP.  Validity:
//   * dy: a stage never leaves its image, so the halo rows above row 0 / below row H - 1 are zero-filled by the DMA's range
//     check (a wave-uniform decision per stage), as are the two corner rows of the image;
//   * dx: the pixel at column 0 (W - 1) must not see its left (right) neighbour, which in the flattened image is the last
//     (first) pixel of the adjacent row: because a stage starts at a row start, "column of K row r" = r mod W is the same in
//     every stage, and two masks per K block (a small LDS table built once) zero those K rows of the dY FRAGMENT (4 v_and
//     each) -- a masked product is an exact zero, the sums are those of the tap-shifted kernel.
// Four waves, each a 32 x 32 tile of all nine taps (144 accumulator registers): two workgroups per CU, a 2-stage ring.
struct Wgrad9Params {
    const char* x;
    const char* dy;
    float* dw;
    int H, W, Ci, Co, R, RW;          // RW = R * W pixels per stage
    int kblocks, XR;                  // 16-pixel K blocks per stage (dY rows = 16 * kblocks); X image rows (multiple of 8)
    int spi;                          // stages per image = H / R
    int stages_view, stages_per_split, splits_view, splits, gx, tilesCi;
    int wcol[9];                      // column offset of tap t = 3 * (dy + 1) + (dx + 1) in a dw row (canonical tap order)
    int w_row_stride;
    uint32_t x_bytes, dy_bytes;
    long slab_stride, dw_view_stride;
    int rmw;
    int img_view;                     // images per view
};

template <typename T, int NST, int KB>  // KB: K blocks per stage at compile time (full unroll, immediate offsets); 0 = p.kblocks
__global__ __launch_bounds__(256, 2) void conv_wgrad9_kernel(const Wgrad9Params p) {
    static_assert(sizeof(T) == 2 && (NST == 1 || NST == 2), "16-bit types, one or two stages");
    const int kblocks = KB ? KB : p.kblocks;
    constexpr int ROWB = 128;  // 64 channels x 2 bytes: one LDS row of either operand
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int bx, slice;  // the block -> (tile, slice) map of conv_wgrad_kernel: a slice's tiles side by side on one XCD
    if (p.splits >= 8) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        bx = q % p.gx;
        slice = (q / p.gx) * 8 + xcd;
    } else {
        bx = blockIdx.x % p.gx;
        slice = blockIdx.x / p.gx;
    }
    if (slice >= p.splits) return;
    const int tci = bx % p.tilesCi, tco = bx / p.tilesCi;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int view = slice / p.splits_view;
    const int sbeg = (slice - view * p.splits_view) * p.stages_per_split;       // stages of this view
    const int nst = min(p.stages_per_split, p.stages_view - sbeg);
    if (nst <= 0) return;
    const int stage0 = view * p.stages_view + sbeg;  // global stage index = image * spi + row group

    const int A_BYTES = kblocks * 16 * ROWB, STAGE = A_BYTES + p.XR * ROWB;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);

    // dx masks of the dY fragment, [K block][lane half h][variant: dx = -1, dx = +1][4 dwords]: dword j of a fragment holds
    // the K rows 16 * ks + 8 * h + 2 * j (low half) and + 1 (high half)
    uint32_t* const mtab = reinterpret_cast<uint32_t*>(smem + NST * STAGE);
    for (int e = tid; e < kblocks * 16; e += 256) {
        const int j = e & 3, var = (e >> 2) & 1, h = (e >> 3) & 1, ks = e >> 4;
        const int r0 = ks * 16 + 8 * h + 2 * j;
        const int c0 = r0 % p.W, c1 = (r0 + 1) % p.W;
        const int bad = var ? p.W - 1 : 0;
        mtab[e] = (c0 == bad ? 0u : 0x0000ffffu) | (c1 == bad ? 0u : 0xffff0000u);
    }

    // DMA lane roles: a wave-instruction stages 8 rows x 128 B; lane -> (row lane >> 3, 16-byte chunk lane & 7)
    const int rin = lane >> 3;
    const uint32_t lchunk = ((uint32_t)(lane & 7) * 16u) ^ ((uint32_t)((rin >> 1) & 1) << 6);  // source-side swizzle
    const uint32_t rowbA = (uint32_t)p.Co * 2u, rowbB = (uint32_t)p.Ci * 2u;
    const uint32_t laneA = (uint32_t)co0 * 2u + lchunk, laneB = (uint32_t)ci0 * 2u + lchunk;
    const int nA = kblocks * 2, nB = p.XR >> 3;  // wave-instructions per stage

    auto dma_stage = [&](int buf, int gs) {  // gs: global stage index
        const int img = gs / p.spi, oy0 = (gs - img * p.spi) * p.R;
        const int f0 = (img * p.H + oy0) * p.W;  // flattened pixel of the stage's first dY row
        const uint32_t sA = smem_lds + (uint32_t)(buf * STAGE), sB = sA + (uint32_t)A_BYTES;
        for (int i = wave; i < nA; i += 4) {
            const int r = i * 8 + rin;
            const uint32_t off = r < p.RW ? (uint32_t)(f0 + r) * rowbA + laneA : kOOB;
            dma16(rdy, sA + (uint32_t)i * 1024u, off, 0u);
        }
        // X image row j <-> flattened pixel f0 - W - 1 + j; real pixels j in [jlo, jhi], zeros outside
        const int jlo = 1 + (oy0 == 0 ? p.W : 0), jhi = (p.R + 2) * p.W - (oy0 + p.R == p.H ? p.W : 0);
        const int fb = f0 - p.W - 1;
        for (int i = wave; i < nB; i += 4) {
            const int j = i * 8 + rin;
            const uint32_t off = (j >= jlo && j <= jhi) ? (uint32_t)(fb + j) * rowbB + laneB : kOOB;
            dma16(rx, sB + (uint32_t)i * 1024u, off, 0u);
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposing reads (see conv_wgrad_kernel): 16-lane group g reads 4 K rows x 16 channels; lane 4q + pq supplies K row q
    const int g = lane >> 4, ii = lane & 15, q = ii >> 2, pq = ii & 3, h = g >> 1;
    const int colsel = 16 * (g & 1) + 4 * pq;
    const int krow0 = 8 * h + q;
    const uint32_t ra_off = (uint32_t)krow0 * ROWB + (((uint32_t)(wm * 32 + colsel) * 2u) ^ ((uint32_t)((q >> 1) & 1) << 6));
    uint32_t rb_off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int row = krow0 + (t / 3) * p.W + (t % 3);  // shift W + 1 + dy * W + dx; + 16 * ks (+ 4) keeps the swizzle key
        rb_off[t] = (uint32_t)row * ROWB + (((uint32_t)(wn * 32 + colsel) * 2u) ^ ((uint32_t)((row >> 1) & 1) << 6));
    }
    const uint32_t m_off = (uint32_t)(NST * STAGE) + (uint32_t)h * 32u;

    if constexpr (NST == 2) dma_stage(0, stage0);
    for (int s = 0; s < nst; ++s) {
        if constexpr (NST == 1) {
            if (s > 0) __syncthreads();
            dma_stage(0, stage0 + s);
            dma_drain();
            __syncthreads();
        } else {
            dma_drain();
            __syncthreads();  // stage s has landed for everyone (and the mask table, s = 0); everyone is done with s - 1
            if (s + 1 < nst) dma_stage((s + 1) & 1, stage0 + s + 1);
        }
        const char* sA = smem + (NST == 2 ? (s & 1) * STAGE : 0);
        const char* sB = sA + A_BYTES;
        auto kblock = [&](int ks) {
            const char* a0 = sA + ra_off + ks * 16 * ROWB;
            const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
            const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * ROWB));
            const uint4 mm = *reinterpret_cast<const uint4*>(smem + m_off + ks * 64);
            const uint4 mp = *reinterpret_cast<const uint4*>(smem + m_off + ks * 64 + 16);
            const uint2 l2 = __builtin_bit_cast(uint2, alo), h2 = __builtin_bit_cast(uint2, ahi);
            uint4 fa[3];
            fa[1] = make_uint4(l2.x, l2.y, h2.x, h2.y);
            fa[0] = make_uint4(l2.x & mm.x, l2.y & mm.y, h2.x & mm.z, h2.y & mm.w);
            fa[2] = make_uint4(l2.x & mp.x, l2.y & mp.y, h2.x & mp.z, h2.y & mp.w);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const char* b0 = sB + rb_off[t] + ks * 16 * ROWB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0 + 4 * ROWB));
                const uint2 bl = __builtin_bit_cast(uint2, lo), bh = __builtin_bit_cast(uint2, hi);
                const uint4 fb = make_uint4(bl.x, bl.y, bh.x, bh.y);
                sm3conv::mma_frag<T>(fa[t % 3], fb, acc[t]);
            }
        };
        if constexpr (KB > 0) {
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) kblock(ks);
        } else {
#pragma unroll 1
            for (int ks = 0; ks < kblocks; ++ks) kblock(ks);
        }
    }

    const int frow = lane & 31, fh = lane >> 5;
    float* const dw_out = p.slab_stride ? p.dw + (long)slice * p.slab_stride : p.dw + (long)view * p.dw_view_stride;
    float* const o0 = dw_out + (long)(co0 + wm * 32 + 4 * fh) * p.w_row_stride + ci0 + wn * 32 + frow;
    auto emit = [&](auto&& put) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) put(o0 + (long)((r & 3) + 8 * (r >> 2)) * p.w_row_stride + p.wcol[t], acc[t][r]);
    };
    if (p.slab_stride) emit([](float* o, float v) { *o = v; });
    else if (p.rmw) emit([](float* o, float v) { *o += v; });
    else emit([](float* o, float v) { atomicAdd(o, v); });
}

// slabs per view of this host thread's most recent launch (sm3_conv_wgrad_slabs returns it right after its own launch;
// thread_local: launches from another host thread -- a second engine, an evaluation thread -- cannot get in between)
static thread_local int g_last_slabs = 0;

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

static int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    return dev;
}

// CUs per XCD (every device of a node is the same part: looked up once)
static int cus_per_xcd() {
    static std::atomic<int> v{0};
    int r = v.load(std::memory_order_relaxed);
    if (!r) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || cus < 8)
            cus = 256;
        r = cus / 8;
        v.store(r, std::memory_order_relaxed);
    }
    return r;
}

template <typename T, int BMW, int BNW, int KP, bool DENSE, int NST, int KG = 1>
int launch_wgrad_kp(WgradParams p, hipStream_t st) {
    constexpr int SZ = sizeof(T);
    constexpr int LDS = KG * NST * KP * (BMW * SZ + BNW * SZ);
    if (p.Co1 > 0 && p.Co % BMW) return SM3_EALIGN;  // a channel tile reads ONE of the two dY tensors
    p.tilesCo = (p.Co + BMW - 1) / BMW + (p.Co1 + BMW - 1) / BMW;
    p.tilesCi = (p.Ci + BNW - 1) / BNW;
    p.adv_n = KP / p.HoWo;
    p.adv_oy = (KP % p.HoWo) / p.Wo;
    p.adv_ox = (KP % p.HoWo) % p.Wo;
    const long gx = (long)p.tilesCo * p.ntaps * p.tilesCi;
    auto kern = conv_wgrad_kernel<T, BMW, BNW, KP, DENSE, NST, KG>;
    // resident workgroups per CU, and the dynamic-LDS limit of the function: per (instantiation, device), set once each
    static std::atomic<int> per_cu_dev[32];
    const int dev = current_device();
    int per_cu = per_cu_dev[dev].load(std::memory_order_acquire);
    if (!per_cu) {
        if (LDS > 65536) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) return (int)e;
        }
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256 * KG, LDS) != hipSuccess || n < 1) n = 1;
        per_cu = n;
        per_cu_dev[dev].store(n, std::memory_order_release);
    }
    // Split the pixel axis so that the launch is ONE full wave of workgroups: every XCD holds whole pixel slices (all
    // gx tiles of a slice side by side, re-reading the slice's dY / X rows from that XCD's L2 while they walk it in step)
    // and as many slices as fit its CUs.  One workgroup too many per XCD and the launch takes a second, nearly empty
    // round: 3x3 256->256 at 72 workgroups for 64 slots ran at 515 TFLOP/s, at 108 for 128 slots at 749.  Each
    // workgroup keeps >= 8 K-steps, and ends with BMW*BNW f32 atomics (~1.3 TB/s chip-wide), which is what keeps the
    // 64 KB-per-workgroup variant (fewer, longer workgroups) ahead for the 1x1 layers.  SM3_WGRAD_TARGET_CTAS overrides.
    const long slots_xcd = (long)cus_per_xcd() * per_cu;
    const long target = env_int("SM3_WGRAD_TARGET_CTAS", 0);
    long splits;  // of the whole launch; each of the `views` row ranges gets splits / views slices of its own
    if (target > 0) splits = (target + gx - 1) / gx;
    else if (gx <= slots_xcd) splits = 8 * (slots_xcd / gx);
    else splits = 8 * slots_xcd / gx;  // < 8 slices: tiles go round-robin over the XCDs (see the kernel)
    // per view from here on.  Slab mode: as if the launch always held two views, so that a view's pixels are cut into the
    // same slices -- and its slabs add up to the same bits -- whether the two views of a branch share a launch or not
    splits = (p.slab_stride && p.pair_rule) ? (splits + 1) / 2 : (splits + p.views - 1) / p.views;
    if (p.slab_stride && splits > p.slab_cap) splits = p.slab_cap;
    const long max_splits = (p.Mv + KP * 8 * KG - 1) / (KP * 8 * KG);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long kps = (p.Mv + splits - 1) / splits;
    kps = (kps + KP - 1) / KP * KP;
    splits = (p.Mv + kps - 1) / kps;
    p.k_per_split = (int)kps;
    p.gx = (int)gx;
    p.splits_view = (int)splits;
    g_last_slabs = (int)splits;
    splits *= p.views;
    p.splits = (int)splits;
    if (p.slab_stride && p.dw_direct && splits == 1) {  // one slice: its tiles are the whole product
        p.slab_stride = 0;
        p.rmw = 1;
        p.dw = p.dw_direct;
        g_last_slabs = 0;
    }
    const long nblocks = gx * (splits >= 8 ? (splits + 7) / 8 * 8 : splits);
    if (gx > 0x7fffffffL || nblocks > 0x7fffffffL) return SM3_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256 * KG), LDS, st, p);
    SM3_CHECK_LAUNCH();
    return 0;
}

template <typename T, int BMW, int BNW>
int launch_wgrad(WgradParams p, hipStream_t st) {
    // K-step: 32 pixels in bf16, 16 in f32 (8-16 KB of tile bytes per stage).
    // 1x1 stride-1 layers: two K-groups of 4 waves on a ring of 4 stages each (3 in flight), one 128 KB workgroup per CU.
    // Tap-shifted layers (3x3, strided): 16-bit with Ci >= 128: ONE stage of 64 pixels (32 KB, 124 registers: four workgroups
    // per CU overlap each other -- the forward kernel's lesson of round 4, 1.5 - 3.5 % here); otherwise a 2-stage ring of 32
    // (32 KB, four to five workgroups per CU: measured per shape, profiles/r02b).
    // Measured slower and removed in round 6 (scratch/r6_pruned_variants.patch keeps the code): one K-group / 2-stage / 1-stage
    // dense rings (+3 ... +40 % time), 2-stage dense rings at two workgroups per CU with 32- or 64-pixel steps (+8 ... +9 % over
    // the class, profiles/r05_wgrad_dense_variants_ab.txt), two K-groups on the tap-shifted kernel (+4 ... +67 %), the
    // 128-register cap of the tap-shifted kernel (+-0), and a 128 x 256 tile for the layer-3 / layer-4 dense shapes (3-stage
    // ring x 2 K-groups, 256 registers with 8 spilled: +3 % over the class; its table was lost with that session's container, the code is
    // in the patch).  Round 6 also measured ONE stage of 128 pixels at two workgroups per CU: faster alone on two layer-4 shapes,
    // nothing in the step (profiles/r06_wgrad_dense_stage_ab.txt).
    constexpr int KP = sizeof(T) == 2 ? 32 : 16;
    const bool dense = p.ntaps == 1 && p.sy == 1 && p.sx == 1 && p.dyt[0] == 0 && p.dxt[0] == 0 &&
                       p.HoWo == p.Hi * p.Wi;
    if (dense) return launch_wgrad_kp<T, BMW, BNW, KP, true, 4, 2>(p, st);
    if constexpr (sizeof(T) == 2) {
        if (p.Ci >= 128) return launch_wgrad_kp<T, BMW, BNW, 64, false, 1>(p, st);
    }
    return launch_wgrad_kp<T, BMW, BNW, KP, false, 2>(p, st);
}

// Nine-tap owner launch.  Returns 1 when the launch does not qualify (the caller falls back to the tap-shifted kernel).
// R = the largest divisor of H whose stage (R rows of an image) fits: <= 13 K blocks and a 2-stage ring within 80 KB (two
// workgroups per CU); a function of the geometry alone, like the slice partition below (a function of geometry and device).
template <typename T>
int launch_wgrad9(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, int views, long dw_view_stride,
                  int slab_cap, float* dw_direct, hipStream_t st) {
    if (!env_int("SM3_WGRAD9", 1)) return 1;  // read per launch: the tests A/B the two kernels in one process
    if (d->ntaps != 9 || d->sy != 1 || d->sx != 1 || d->Hi != d->Ho || d->Wi != d->Wo) return 1;
    if (d->Co % 64 || d->Ci % 64 || d->N % views) return 1;
    for (int t = 0; t < 9; ++t)
        if (d->dy[t] != t / 3 - 1 || d->dx[t] != t % 3 - 1 || d->wtap[t] < 0 || (d->wtap[t] + 1) * d->Ci > d->w_row_stride)
            return 1;
    const int H = d->Ho, W = d->Wo;
    const int force_r = env_int("SM3_WGRAD9_R", 0), force_nst = env_int("SM3_WGRAD9_NST", 0);  // experiments only
    // (R, ring depth): the candidate with the fewest padded K rows (R * W against its multiple of 16) wins -- measured on the
    // step's shapes (profiles/r06_wgrad9_ab.txt), padding costs more than the second stage buys --, then two stages over
    // one, then the larger R; below 85 % useful K rows (7 x 7 maps: 49 of 64) the tap-shifted kernel is faster
    int R = 0, kblocks = 0, XR = 0, nstg = 0;
    long best_num = 0, best_den = 1;
    for (int r = H; r >= 1; --r) {
        if (H % r || (force_r && r != force_r)) continue;
        const int kb = (r * W + 15) / 16, xr = (16 * kb + 2 * W + 2 + 7) / 8 * 8;
        const long stage = (long)(16 * kb + xr) * 128;
        if (kb > 13) continue;
        const int n = force_nst ? force_nst : (2 * stage + kb * 64 <= 81920 ? 2 : stage + kb * 64 <= 81920 ? 1 : 0);
        if (!n || n * stage + kb * 64 > 160 * 1024) continue;
        const long num = (long)r * W, den = 16L * kb;  // efficiency num / den
        const bool better = !R || num * best_den > best_num * den || (num * best_den == best_num * den && n > nstg);
        if (better) { R = r; kblocks = kb; XR = xr; nstg = n; best_num = num; best_den = den; }
    }
    if (R && !force_r && best_num * 100 < best_den * 85) return 1;
    if (!R) return 1;
    Wgrad9Params p;
    p.x = (const char*)x; p.dy = (const char*)dy; p.dw = dw;
    p.H = H; p.W = W; p.Ci = d->Ci; p.Co = d->Co; p.R = R; p.RW = R * W; p.kblocks = kblocks; p.XR = XR;
    p.spi = H / R;
    p.img_view = d->N / views;
    p.stages_view = p.img_view * p.spi;
    p.tilesCi = d->Ci / 64;
    p.gx = (d->Co / 64) * p.tilesCi;
    for (int t = 0; t < 9; ++t) p.wcol[t] = d->wtap[t] * d->Ci;
    p.w_row_stride = d->w_row_stride;
    p.x_bytes = (uint32_t)((long)d->N * H * W * d->Ci * 2);
    p.dy_bytes = (uint32_t)((long)d->N * H * W * d->Co * 2);
    p.slab_stride = slab_cap > 0 ? (long)d->Co * d->w_row_stride : 0;
    p.dw_view_stride = dw_view_stride;
    p.rmw = 0;
    const int LDS = nstg * (16 * kblocks + XR) * 128 + kblocks * 64;
    const int kbi = kblocks == 7 ? 1 : kblocks == 13 ? 2 : 0;  // unrolled instantiations: the step's shapes at 224 and 448
    void (*kern)(const Wgrad9Params) =
        nstg == 2 ? (kbi == 1 ? conv_wgrad9_kernel<T, 2, 7> : kbi == 2 ? conv_wgrad9_kernel<T, 2, 13> : conv_wgrad9_kernel<T, 2, 0>)
                  : (kbi == 1 ? conv_wgrad9_kernel<T, 1, 7> : kbi == 2 ? conv_wgrad9_kernel<T, 1, 13> : conv_wgrad9_kernel<T, 1, 0>);
    static std::atomic<int> attr_dev[6][32];
    const int dev = current_device(), ki = (nstg - 1) * 3 + kbi;
    if (!attr_dev[ki][dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_dev[ki][dev].store(1, std::memory_order_release);
    }
    // one full wave of workgroups, whole slices per XCD (the rule of launch_wgrad_kp), two workgroups per CU by registers
    const int per_cu = LDS > 81920 ? 1 : 2;
    const long slots_xcd = (long)cus_per_xcd() * per_cu, gx = p.gx;
    const long target = env_int("SM3_WGRAD_TARGET_CTAS", 0);
    long splits;
    if (target > 0) splits = (target + gx - 1) / gx;
    else if (gx <= slots_xcd) splits = 8 * (slots_xcd / gx);
    else splits = 8 * slots_xcd / gx;
    const bool pair_rule = p.slab_stride && !dw_direct;
    splits = pair_rule ? (splits + 1) / 2 : (splits + views - 1) / views;
    if (p.slab_stride && splits > slab_cap) splits = slab_cap;
    const long max_splits = (p.stages_view + 3) / 4;  // >= 4 stages per slice
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const long sps = (p.stages_view + splits - 1) / splits;
    splits = (p.stages_view + sps - 1) / sps;
    p.stages_per_split = (int)sps;
    p.splits_view = (int)splits;
    g_last_slabs = (int)splits;
    splits *= views;
    p.splits = (int)splits;
    if (p.slab_stride && dw_direct && splits == 1) {
        p.slab_stride = 0;
        p.rmw = 1;
        p.dw = dw_direct;
        g_last_slabs = 0;
    }
    const long nblocks = gx * (splits >= 8 ? (splits + 7) / 8 * 8 : splits);
    if (nblocks > 0x7fffffffL) return SM3_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), LDS, st, p);
    SM3_CHECK_LAUNCH();
    return 0;
}

}  // namespace

static int wgrad_impl(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, const void* dy1, int Co1,
                      float* dw1, int views, long dw_view_stride, long dw1_view_stride, void* stream, int slab_cap = 0,
                      float* dw_direct = nullptr) {
    if (!d || !x || !dy || !dw) return SM3_EINVAL;
    if (!SM3_DTYPE_OK(d->dtype)) return SM3_EDTYPE;
    const int sz = d->dtype == SM3_F32 ? 4 : 2;
    if (d->N <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ci <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Co <= 0) return SM3_EINVAL;
    if (d->ntaps < 1 || d->ntaps > SM3_MAX_TAPS) return SM3_EINVAL;
    if ((d->Ci * sz) % 16 != 0 || (d->Co * sz) % 16 != 0 || (Co1 * sz) % 16 != 0) return SM3_EALIGN;
    if (d->osy != 1 || d->osx != 1 || d->ooy != 0 || d->oox != 0 || d->Hout != d->Ho || d->Wout != d->Wo)
        return SM3_EINVAL;  // dy must be the dense output of the forward conv
    const long M = (long)d->N * d->Ho * d->Wo;
    if (M > 0x7fffffffL || (long)d->N * d->Hi * d->Wi > 0x7fffffffL) return SM3_EINVAL;
    if (views < 1 || M % views || Co1 < 0 || (Co1 > 0 && (!dy1 || !dw1))) return SM3_EINVAL;
    WgradParams p;
    p.x = (const char*)x; p.dy = (const char*)dy; p.dw = dw;
    p.dy1 = (const char*)dy1; p.dw1 = dw1; p.Co1 = Co1;
    p.views = views; p.Mv = (int)(M / views);
    p.dw_view_stride = dw_view_stride; p.dw1_view_stride = dw1_view_stride;
    p.slab_cap = slab_cap;
    p.slab_stride = slab_cap > 0 ? (long)d->Co * d->w_row_stride : 0;
    p.dw_direct = dw_direct;
    p.pair_rule = dw_direct ? 0 : 1;
    p.rmw = 0;
    p.M = (int)M; p.Hi = d->Hi; p.Wi = d->Wi; p.Ci = d->Ci; p.Co = d->Co;
    p.sy = d->sy; p.sx = d->sx; p.ntaps = d->ntaps;
    for (int t = 0; t < SM3_MAX_TAPS; ++t) { p.dyt[t] = d->dy[t]; p.dxt[t] = d->dx[t]; p.wtap[t] = d->wtap[t]; }
    p.w_row_stride = d->w_row_stride;
    p.HoWo = d->Ho * d->Wo; p.Wo = d->Wo; p.Ho = d->Ho;
    p.div_HoWo = make_fastdiv((uint32_t)p.HoWo);
    p.div_Wo = make_fastdiv((uint32_t)p.Wo);
    const long xb = (long)d->N * d->Hi * d->Wi * d->Ci * sz, yb = M * d->Co * sz, y1b = M * Co1 * sz;
    if (xb >= 0xC0000000L || yb >= 0xC0000000L || y1b >= 0xC0000000L) return SM3_EINVAL;  // 32-bit buffer offsets
    p.x_bytes = (uint32_t)xb;
    hipStream_t st = (hipStream_t)stream;
    if (d->dtype != SM3_F32 && Co1 == 0) {  // stride-1 3x3: the nine-tap owner when the geometry fits it
        const int rc = d->dtype == SM3_BF16
                           ? launch_wgrad9<bf16_t>(d, x, dy, dw, views, dw_view_stride, slab_cap, dw_direct, st)
                           : launch_wgrad9<f16_t>(d, x, dy, dw, views, dw_view_stride, slab_cap, dw_direct, st);
        if (rc != 1) return rc;
    }
    const bool nco = d->Co + Co1 <= 64, nci = d->Ci <= 64;
    if (d->dtype == SM3_BF16) {
        if (nco && nci) return launch_wgrad<bf16_t, 64, 64>(p, st);
        if (nco) return launch_wgrad<bf16_t, 64, 128>(p, st);
        if (nci) return launch_wgrad<bf16_t, 128, 64>(p, st);
        return launch_wgrad<bf16_t, 128, 128>(p, st);
    }
    if (d->dtype == SM3_F16) {
        if (nco && nci) return launch_wgrad<f16_t, 64, 64>(p, st);
        if (nco) return launch_wgrad<f16_t, 64, 128>(p, st);
        if (nci) return launch_wgrad<f16_t, 128, 64>(p, st);
        return launch_wgrad<f16_t, 128, 128>(p, st);
    }
    if (nco && nci) return launch_wgrad<float, 64, 64>(p, st);
    if (nco) return launch_wgrad<float, 64, 128>(p, st);
    if (nci) return launch_wgrad<float, 128, 64>(p, st);
    return launch_wgrad<float, 128, 128>(p, st);
}

#ifdef SM3_STAMP
extern "C" int sm3_debug_set_wgrad_stamps(void* buf, long capacity_waves) {
    unsigned long long* b = (unsigned long long*)buf;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wstamp_buf), &b, sizeof(b)) != hipSuccess) return SM3_EINVAL;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wstamp_cap), &capacity_waves, sizeof(long)) != hipSuccess) return SM3_EINVAL;
    return 0;
}
#endif

extern "C" int sm3_conv_wgrad(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, void* stream) {
    return wgrad_impl(d, x, dy, dw, nullptr, 0, nullptr, 1, 0, 0, stream);
}

extern "C" int sm3_conv_wgrad_slabs(const sm3_conv_desc* d, const void* x, const void* dy, float* slabs, int slab_capacity,
                                    int views, int* slabs_used, void* stream) {
    if (!slabs_used || slab_capacity < 1) return SM3_EINVAL;
    if (d && d->ntaps * d->Ci != d->w_row_stride) return SM3_EINVAL;  // every slab a dense [Co][taps * Ci] matrix
    const int rc = wgrad_impl(d, x, dy, slabs, nullptr, 0, nullptr, views, 0, 0, stream, slab_capacity);
    if (rc == 0) *slabs_used = g_last_slabs;
    return rc;
}

// ---- sm3_slab_reduce ------------------------------------------------------------------------------------------------
// out[e] (+)= sum_j slabs[j][e], j = 0 .. nslabs - 1, in an order that depends on nothing but (nslabs, e): a workgroup owns
// 128 consecutive elements (32 float4 columns) x 8 slab lanes; lane l adds slabs l, l + 8, ... on four interleaved
// accumulators, the eight lane sums are added in lane order through LDS.  One writer per element: plain stores.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, long n,
                                                          float* __restrict__ out, int accumulate) {
    __shared__ float4 red[8][32];
    const int col = threadIdx.x & 31, jl = threadIdx.x >> 5;
    const long e4 = ((long)blockIdx.x * 32 + col) * 4;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    auto add = [](float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    if (e4 < n) {
        const float* src = slabs + e4;
        int j = jl;
        for (; j + 24 < nslabs; j += 32) {
            const float4 v0 = *reinterpret_cast<const float4*>(src + (long)j * n);
            const float4 v1 = *reinterpret_cast<const float4*>(src + (long)(j + 8) * n);
            const float4 v2 = *reinterpret_cast<const float4*>(src + (long)(j + 16) * n);
            const float4 v3 = *reinterpret_cast<const float4*>(src + (long)(j + 24) * n);
            add(a0, v0); add(a1, v1); add(a2, v2); add(a3, v3);
        }
        for (; j < nslabs; j += 8) add(a0, *reinterpret_cast<const float4*>(src + (long)j * n));
        add(a0, a1);
        add(a2, a3);
        add(a0, a2);
    }
    red[jl][col] = a0;
    __syncthreads();
    if (jl == 0 && e4 < n) {
        float4 t = red[0][col];
#pragma unroll
        for (int l = 1; l < 8; ++l) add(t, red[l][col]);
        float4* o = reinterpret_cast<float4*>(out + e4);
        if (accumulate) {
            float4 g = *o;
            add(g, t);
            t = g;
        }
        *o = t;
    }
}

extern "C" int sm3_slab_reduce(const float* slabs, int nslabs, int64_t n, float* out, int accumulate, void* stream) {
    if (!slabs || !out || nslabs < 1 || n < 4 || n % 4) return SM3_EINVAL;
    if (((uintptr_t)slabs | (uintptr_t)out) & 15) return SM3_EALIGN;
    const long blocks = (n / 4 + 31) / 32;
    if (blocks > 0x7fffffffL) return SM3_EINVAL;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslabs, (long)n,
                       out, accumulate);
    SM3_CHECK_LAUNCH();
    return 0;
}

// dw += dY^T X as a function of its inputs: plain-store split-K slabs (one per pixel slice, the partition a function of the
// geometry and the device) and ONE fixed-order sum into dw -- no float atomics, so two runs of a step give the same bits.
extern "C" int sm3_conv_wgrad_det(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, float* slabs,
                                  int slab_capacity, void* stream) {
    if (!slabs || !dw || slab_capacity < 1) return SM3_EINVAL;
    if (d && d->ntaps * d->Ci != d->w_row_stride) return SM3_EINVAL;  // every slab a dense [Co][taps * Ci] matrix
    if (((uintptr_t)slabs | (uintptr_t)dw) & 15) return SM3_EALIGN;
    const int rc = wgrad_impl(d, x, dy, slabs, nullptr, 0, nullptr, 1, 0, 0, stream, slab_capacity, dw);
    if (rc != 0 || g_last_slabs == 0) return rc;  // 0 slabs: a single slice added its tiles to dw itself
    return sm3_slab_reduce(slabs, g_last_slabs, (int64_t)d->Co * d->w_row_stride, dw, 1, stream);
}

extern "C" int sm3_conv_wgrad_cat(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, const void* dy1,
                                  int Co1, float* dw1, int views, int64_t dw_view_stride, int64_t dw1_view_stride,
                                  void* stream) {
    if (Co1 > 0 && d && d->Co % 128) return SM3_EALIGN;  // tiles of 64 or 128 channels never straddle the two tensors
    return wgrad_impl(d, x, dy, dw, dy1, Co1, dw1, views, (long)dw_view_stride, (long)dw1_view_stride, stream);
}
