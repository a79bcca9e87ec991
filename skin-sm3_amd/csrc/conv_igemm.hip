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
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "conv_common.h"

using namespace sm3conv;


namespace {

// STAGES = 2: K-loop with the DMA of step s+1 in flight while step s is computed (LDS 64 KB + 2 KB, 2 workgroups
// per CU).  STAGES = 1 (host picks it for <= 8 K-steps, the memory-bound 1x1 layers; measured: scratch/bench_kernels.py): one 32 KB stage, no
// intra-workgroup overlap, but 34 KB of LDS lets 4 workgroups share a CU and overlap each other's load latency,
// K-step and store tail -- which is what those short workgroups need.
// SEG: the K loop may draw taps from a SECOND (A, B) operand pair (ConvParams::x1 / w1, tap_src): Y = sum_t A_t B_t^T
// over two tensors with different channel counts -- the data gradient of "BatchNorm backward by linearity" (linbn.hip),
// [dz | y_in] x [diag(a) W ; -H]^T.  Only the general epilogue has it; the production forward kernels are untouched.
// EPI: 0 = general epilogue (f32 staging, one wave-row at a time: every feature, every type); 16-bit types only:
//      1 = lean (whole tile staged once as 16-bit, stored with no arithmetic: the train-mode forward convolutions),
//      2 = lean + per-row work on the read-back: addend (dense or compact), BatchNorm affine + ReLU + ReLU bits
//          (sm3_conv_bn_act_fused / the inference forms with a residual), strided outputs,
//      3 = 2 + the fused BatchNorm-backward phase 1 (ReLU mask, sum dz, sum dz*xhat) of the data-gradient launches.
//      2 / 3 round the GEMM result to 16 bits BEFORE the addend is added in fp32 -- what autocast does with a convolution
//      output that is then accumulated -- and keep twice the bytes of 0 in flight per workgroup with half its barriers.
#ifdef SM3_STAMP
// Diagnostic build only (scratch/stamp_conv.py / stamp_phases.py build a second library with -DSM3_STAMP; in the product
// library no stamp executes): s_memtime at the phase boundaries of the kernel (marks) and at the segment boundaries of the
// 2-stage K loop (summed per wave), one 24-word record per wave written to a buffer of its own (sm3_debug_set_stamps) --
// no output value depends on a stamp.
__device__ unsigned long long* g_stamp_buf = nullptr;
__device__ long g_stamp_cap = 0;
struct StampRec {
    unsigned long long t_entry, r_entry, t_loop0 = 0, t_loop1 = 0, seg[4] = {0, 0, 0, 0}, mark[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int nsteps = 0;
    __device__ StampRec() : t_entry(__builtin_amdgcn_s_memtime()), r_entry(__builtin_amdgcn_s_memrealtime()) {}
    __device__ ~StampRec() {
        const unsigned long long t_exit = __builtin_amdgcn_s_memtime();
        const long wave = ((long)blockIdx.y * gridDim.x + blockIdx.x) * (blockDim.x / 64) + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && g_stamp_buf && wave < g_stamp_cap) {
            unsigned long long* o = g_stamp_buf + wave * 24;
            o[0] = t_entry; o[1] = t_loop0; o[2] = t_loop1; o[3] = t_exit;
            o[4] = seg[0]; o[5] = seg[1]; o[6] = seg[2]; o[7] = seg[3];
            o[8] = (unsigned long long)nsteps; o[9] = blockIdx.x; o[10] = r_entry; o[11] = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < 8; ++i) o[12 + i] = mark[i];
            o[20] = threadIdx.x >> 6;
        }
    }
};
#define SM3_STAMP_NOW() __builtin_amdgcn_s_memtime()
#define SM3_MARK(i) (stamp.mark[i] = __builtin_amdgcn_s_memtime())
#else
#define SM3_MARK(i) ((void)0)
#endif

// VAR (bit mask of variants that change instructions, never results):
//   kVarPw     POINTWISE launches (one tap at (0, 0) -- or two K segments over the same pixels --, stride 1, dense output, no
//              compact addend): a tile row IS pixel m0 + r, so the loader state and the epilogue's row addresses are one
//              multiply-add per row instead of the general gather's divisions and bounds tests.  These launches are the
//              HBM-bound half of the step and were VALU-issue-bound: ~1 170 vector instructions per thread and tile, 217 of
//              them set-up and ~280 row addressing (profiles/r04a_smemtime_phases_1x1.txt);
//   kVarNoX    EPI 3 without the producer's x (the linear BatchNorm forms: ReLU mask + sum(dz) only): no x operand stream,
//              no sum(dz * xhat) arithmetic, and with the registers that frees ALL rows' operands are requested before the
//              staging instead of half of them after it.
//   kVarHalo   stride-1 3 x 3 launches (forward and data gradient): a tile's rows are 128 CONSECUTIVE pixels, so the nine taps
//              of a channel chunk read the same W + 1 pixels either side of them.  The A image of a chunk (tile rows + that
//              halo + one zero row) is staged ONCE and the nine K-steps of the chunk read it at a per-tap row shift (rows whose
//              tap falls outside the image read the zero row); only the B tile is staged per K-step.  18 - 22 KB of A per
//              nine K-steps instead of 144: these launches drew ~15 TB/s through L2 -> LDS, 85 % of what LDS-DMA delivers
//              from L2 (MI355X_MICROARCH.md, "Indexed rows: gather into LDS").  K order: chunk outer, tap inner.
//   kVarHaloBn (with kVarHalo, plain epilogue): the A tensor is the RAW output of the producer convolution; every thread
//              applies the producer's BatchNorm affine + ReLU to the LDS-DMA pieces IT staged (after its own vmcnt wait,
//              before the barrier that publishes the image: no extra barrier), in the arithmetic of bn_act_kernel, and
//              writes the activation and its ReLU bits for the tile's own 128 rows -- the producer's separate apply pass
//              (one read of that tensor and one launch per Bottleneck) disappears (forward half).
// (bits 1 and 8 were the loader / consumer split and the 16 x 16 x 32 MFMA shape of round 4: measured slower two rounds
// running and removed in round 6 -- profiles/r04a_split_ab_*.txt, r04a_mfma_16x16x32_ab_single_lane.txt, scratch/r6_pruned_variants.patch)
constexpr int kVarPw = 2, kVarNoX = 4, kVarHalo = 16, kVarHaloBn = 32;
template <typename T, int BM, int BN, int WM, int WN, int STAGES, int EPI, bool SEG = false, int VAR = 0>
// registers: the 1-stage kernels (34 KB of LDS) run 4 workgroups per CU = 4 waves per SIMD, so their epilogues must fit 128
// registers; the 2- and 4-stage kernels are limited to 2 / 1 workgroups per CU by their LDS and may use 256
__global__ __launch_bounds__(WM* WN * 64, (EPI >= 1 && STAGES == 1) ? 4 : 2) void conv_igemm_kernel(const ConvParams p) {
    static_assert(STAGES == 1 || STAGES == 4, "one-stage loop, or the 4-stage ring of the small-grid launches");
    constexpr bool PW = (VAR & kVarPw) != 0, NOX = (VAR & kVarNoX) != 0;
    constexpr bool HALO = (VAR & kVarHalo) != 0, HALOBN = (VAR & kVarHaloBn) != 0;
    static_assert(!HALOBN || (HALO && EPI == 1), "kVarHaloBn: the plain forward epilogue of the halo kernel");
    static_assert(!HALO || (EPI >= 1 && STAGES == 1 && !SEG && !(VAR & kVarPw) && sizeof(T) == 2 && WM * WN == 4),
                  "kVarHalo: 16-bit lean one-stage kernels, 4 waves");
    static_assert(!NOX || EPI == 3, "kVarNoX: the fused BN-backward epilogue");
#ifdef SM3_STAMP
    StampRec stamp;
#endif
    constexpr bool LEAN = EPI >= 1;
    static_assert(!(SEG && EPI == 1) && !(SEG && STAGES > 2), "segments: data-gradient epilogues, 1 or 2 stages");
    constexpr int NT = WM * WN * 64;
    constexpr int RPP = NT / 8;  // rows covered per loader pass
    constexpr int AI = BM / RPP, BI = BN / RPP;
    constexpr int WTM = BM / WM, WTN = BN / WN, TM = WTM / 32, TN = WTN / 32;
    constexpr int SZ = sizeof(T);
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    // general epilogue: one wave-row (WTM rows) of the tile at a time in f32; lean epilogue: the whole tile in bf16
    constexpr int LEAN_PITCH = BN * 2 + 16;  // bytes; +16 puts rows r and r+4 (the lane halves of a C register) on different banks
    // general epilogue staging: rows of BN floats + 4 (16 B): the 16-byte read-back of 16 lanes x 32 B per row no longer
    // puts channel vectors cc and cc + 8 of neighbouring rows on the same banks (29 % LDS conflict cycles measured)
    constexpr int CP = BN + 4;
    constexpr int C_BYTES = LEAN ? BM * LEAN_PITCH : WTM * CP * 4;
    constexpr int MAIN_BYTES = (STAGES * STAGE > C_BYTES) ? STAGES * STAGE : C_BYTES;
    static_assert(AI >= 1 && BI >= 1 && TM >= 1 && TN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // provably wave-uniform (LDS-DMA base goes to M0)
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware block remap (bijective): blocks that share an A row-panel run on one XCD's L2.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int bn = bid % p.tilesN, bm = bid / p.tilesN;
    const int m0 = bm * BM, n0 = bn * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    // second segment's operands (SEG only; otherwise aliases that the optimiser drops)
    const __amdgpu_buffer_rsrc_t rx1 =
        SEG ? __builtin_amdgcn_make_buffer_rsrc((void*)p.x1, 0, p.x1_bytes, 0x00020000) : rx;
    const __amdgpu_buffer_rsrc_t rw1 =
        SEG ? __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, p.w1_bytes, 0x00020000) : rw;
    // two views in one launch: a tile belongs to exactly one of them (view rows are a multiple of BM)
    const int tile_view = (p.fz_view_tiles > 0 && bm >= p.fz_view_tiles) ? 1 : 0;

    // ---- loader state -------------------------------------------------------------------
    // lane (row = (tid>>3) + i*RPP, pos = tid&7) of DMA instruction i; rows 8*wave + 32*i .. +7 per instruction
    const int pos = tid & 7;
    int a_pix[AI], a_iy0[AI], a_ix0[AI];  // pixel index of tap (0,0) and its coordinates; a_iy0 = -2^20 if row >= M
    uint32_t a_ch[AI], a_off[AI], b_off[BI];
    uint32_t b_off1[SEG ? BI : 1];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int r = (tid >> 3) + i * RPP;
        a_ch[i] = (uint32_t)((pos ^ (r >> 1)) & 7) * 16u;  // source chunk of this lane (swizzle on the source side)
        const int m = m0 + r;
        if constexpr (HALO) {  // (own loader state below)
            a_iy0[i] = a_ix0[i] = a_pix[i] = 0;
            continue;
        }
        if constexpr (PW) {  // the row is pixel m of the source
            a_iy0[i] = a_ix0[i] = 0;
            a_pix[i] = m < p.M ? m : -1;
            continue;
        }
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
    for (int i = 0; i < BI; ++i) {
        const int r = (tid >> 3) + i * RPP;
        const int co = n0 + r;
        b_off[i] = (co < p.Co) ? (uint32_t)co * (uint32_t)(p.w_row_stride * SZ) + (uint32_t)((pos ^ (r >> 1)) & 7) * 16u
                               : kOOB;
        if constexpr (SEG) {
            if (co < p.Co) b_off[i] += (uint32_t)tile_view * p.w_view_bytes;
            b_off1[i] = (co < p.Co) ? (uint32_t)co * (uint32_t)(p.w1_row_stride * SZ) + (uint32_t)((pos ^ (r >> 1)) & 7) * 16u +
                                          (uint32_t)tile_view * p.w1_view_bytes
                                    : kOOB;
        }
    }
    const uint32_t row_bytes = (uint32_t)p.Ci * SZ;
    const uint32_t row_bytes1 = SEG ? (uint32_t)p.Ci1 * SZ : row_bytes;
    int cur_src = 0;  // segment of the tap being staged (wave-uniform)

    auto set_tap = [&](int t) {
        if constexpr (SEG) cur_src = __builtin_amdgcn_readfirstlane((int)p.tap_src[t]);
        const uint32_t rb = (SEG && cur_src) ? row_bytes1 : row_bytes;
        if constexpr (PW) {
#pragma unroll
            for (int i = 0; i < AI; ++i) a_off[i] = a_pix[i] >= 0 ? (uint32_t)a_pix[i] * rb + a_ch[i] : kOOB;
            return;
        }
        const int ddy = p.dy[t], ddx = p.dx[t];
        const int dpix = ddy * p.Wi + ddx;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int iy = a_iy0[i] + ddy, ix = a_ix0[i] + ddx;
            const bool ok = ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
            a_off[i] = ok ? (uint32_t)(a_pix[i] + dpix) * rb + a_ch[i] : kOOB;
        }
    };
    const uint32_t smem_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    auto dma_stage = [&](int stage, uint32_t soff_a, uint32_t soff_b) {
        const uint32_t sA = smem_lds + (uint32_t)(stage * STAGE + wave * (8 * 128));
        const uint32_t sB = sA + A_BYTES;
        if (SEG && cur_src) {
#pragma unroll
            for (int i = 0; i < AI; ++i) dma16(rx1, sA + i * (RPP * 128), a_off[i], soff_a);
#pragma unroll
            for (int i = 0; i < BI; ++i) dma16(rw1, sB + i * (RPP * 128), b_off1[SEG ? i : 0], soff_b);
            return;
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) dma16(rx, sA + i * (RPP * 128), a_off[i], soff_a);
#pragma unroll
        for (int i = 0; i < BI; ++i) dma16(rw, sB + i * (RPP * 128), b_off[i], soff_b);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets: chunk (2*kk + fh) of row r sits at r*128 + (((2*kk+fh) ^ (r>>1)) & 7) * 16
    //   = base(r, fh) ^ (kk << 5)   (bits 5-6 of the offset carry kk; r*128 leaves them clear)
    const int frow = lane & 31, fh = lane >> 5;
    uint32_t fa_base[TM], fb_base[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa_base[i] = lds_off(wm * WTM + i * 32 + frow, fh);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb_base[j] = A_BYTES + lds_off(wn * WTN + j * 32 + frow, fh);

    auto compute = [&](int stage) {
        const char* sS = smem + stage * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(sS + (fa_base[i] ^ (kk << 5)));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(sS + (fb_base[j] ^ (kk << 5)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mma_frag<T>(fa[i], fb[j], acc[i][j]);
        }
    };

    const int nsteps = SEG ? p.nsteps_seg : p.ntaps * p.kchunks;
    int t = 0, kc = 0;
    SM3_MARK(0);
    if constexpr (!HALO) set_tap(0);
    int kch = (SEG && cur_src) ? p.kchunks1 : p.kchunks;  // K-steps of the current tap
    uint32_t wtap_off = (uint32_t)p.wtap[0] * ((SEG && cur_src) ? row_bytes1 : row_bytes);
    auto advance = [&]() {
        if constexpr (!SEG && !PW && !HALO && sizeof(T) == 2 && EPI >= 1) {
            if (p.kord) {  // chunk outer, tap inner: the summation order of kVarHalo (ConvParams::kord)
                if (++t == p.ntaps) {
                    t = 0;
                    ++kc;
                }
                set_tap(t);
                wtap_off = (uint32_t)p.wtap[t] * row_bytes;
                return;
            }
        }
        if (++kc == kch) {
            kc = 0;
            ++t;
            set_tap(t);
            if constexpr (SEG) kch = cur_src ? p.kchunks1 : p.kchunks;
            wtap_off = (uint32_t)p.wtap[t] * ((SEG && cur_src) ? row_bytes1 : row_bytes);
        }
    };
    if constexpr (HALO) {
        // A image: halo rows j = 0 .. HR-1 hold pixels m0 - (W+1) + j (zeros where that is no pixel of the tensor), rows HR,
        // HR + 1 (and the rest of the last 32-row pass) zeros; B tile behind it.  Both swizzled like the general stage.
        const int HR = p.halo_rows, hoff = p.Wi + 1, q0 = m0 - hoff;
        const uint32_t a_chunk = (uint32_t)((pos ^ (tid >> 4)) & 7) * 16u;  // (j >> 1) & 7 of row j = (tid >> 3) + 32 i: the same for every i
        const uint32_t sAw = smem_lds + (uint32_t)(wave * (8 * 128));
        const uint32_t sBw = sAw + (uint32_t)p.halo_a_bytes;
        const int npass = p.halo_a_bytes >> 12;  // 32 rows x 128 B per pass
        // (the chunk's byte offset rides in the per-lane offset, not the scalar one: the eight offsets are then no loop
        // invariants the compiler would keep in registers across the K loop)
        auto halo_issue = [&](uint32_t chunk_off) {
            const uint32_t base = (uint32_t)(q0 + (tid >> 3)) * row_bytes + a_chunk + chunk_off;
            constexpr int kMaxPass = BM == 256 ? 12 : 8;  // 32-row passes of the image: BM + 2 (W + 1) + 2 rows
#pragma unroll
            for (int i = 0; i < kMaxPass; ++i) {
                if (i < npass) {
                    const int j = (tid >> 3) + i * 32, q = q0 + j;
                    const bool ok = j < HR && (unsigned)q < (unsigned)p.M;
                    dma16(rx, sAw + i * 4096, ok ? base + (uint32_t)(i * 32) * row_bytes : kOOB, 0);
                }
            }
        };
        // which of the nine taps exist for this lane's fragment rows: bit (dy + 1) * 3 + (dx + 1)
        int frag_row[TM];
        unsigned tap_ok[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            frag_row[i] = wm * WTM + i * 32 + frow;
            const int m = m0 + frag_row[i];
            tap_ok[i] = 0;
            if (m < p.M) {
                const int n = fdiv(m, p.div_HoWo);
                const int rem = m - n * p.HoWo;
                const int oy = fdiv(rem, p.div_Wo);
                const int ox = rem - oy * p.Wo;
                const unsigned ym = (oy > 0 ? 1u : 0u) | 2u | (oy + 1 < p.Hi ? 4u : 0u);
                const unsigned xm = (ox > 0 ? 1u : 0u) | 2u | (ox + 1 < p.Wi ? 4u : 0u);
                tap_ok[i] = ((ym & 1u) ? xm : 0u) | ((ym & 2u) ? xm << 3 : 0u) | ((ym & 4u) ? xm << 6 : 0u);
            }
        }
        uint32_t fbh[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) fbh[j] = (uint32_t)p.halo_a_bytes + lds_off(wn * WTN + j * 32 + frow, fh);
        const int kchunks = p.kchunks;
#ifdef SM3_STAMP
        stamp.mark[1] = stamp.t_loop0 = SM3_STAMP_NOW();
        stamp.nsteps = 9 * kchunks;
#endif
#pragma unroll 1
        for (int c = 0; c < kchunks; ++c) {
#pragma unroll 1
            for (int tt = 0; tt < 9; ++tt) {
#ifdef SM3_STAMP
                const unsigned long long q0 = SM3_STAMP_NOW();
#endif
                if ((c | tt) != 0) __syncthreads();  // everyone is done reading what this step overwrites
#ifdef SM3_STAMP
                const unsigned long long q1 = SM3_STAMP_NOW();
#endif
                if (tt == 0) halo_issue((uint32_t)c * 128u);
                // kVarHaloBn: the producer's scale / shift of this lane's 8 channels of the chunk, requested before the wait
                float4 bsc0 = make_float4(0.f, 0.f, 0.f, 0.f), bsc1 = bsc0, bsh0 = bsc0, bsh1 = bsc0;
                if constexpr (HALOBN) {
                    if (tt == 0) {
                        const int ch0 = tile_view * p.Ci + c * (128 / SZ) + (int)(a_chunk >> 4) * (16 / SZ);
                        bsc0 = *reinterpret_cast<const float4*>(p.in_scale + ch0);
                        bsc1 = *reinterpret_cast<const float4*>(p.in_scale + ch0 + 4);
                        bsh0 = *reinterpret_cast<const float4*>(p.in_shift + ch0);
                        bsh1 = *reinterpret_cast<const float4*>(p.in_shift + ch0 + 4);
                    }
                }
                const uint32_t soff_b = (uint32_t)p.wtap[tt] * row_bytes + (uint32_t)c * 128u;
#pragma unroll
                for (int i = 0; i < BI; ++i) dma16(rw, sBw + i * (RPP * 128), b_off[i], soff_b);
                const int ddy = p.dy[tt], ddx = p.dx[tt];
                const int bit = (ddy + 1) * 3 + ddx + 1, shift = hoff + ddy * p.Wi + ddx;
                uint32_t fah[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    // a tap outside the image reads zeros AT THE BANKS its own row would have used (rows HR, HR + 1 are one
                    // 256-byte bank row of zeros): the 16 lanes of a read group stay on 16 distinct slots (all redirected
                    // lanes on ONE slot of row HR cost a 2-way conflict per group: 20 % LDS conflict cycles measured)
                    const uint32_t own = lds_off(frag_row[i] + shift, fh);
                    fah[i] = ((tap_ok[i] >> bit) & 1u) ? own : (uint32_t)HR * 128u + (own & 255u);
                }
#ifdef SM3_STAMP
                const unsigned long long q2 = SM3_STAMP_NOW();
#endif
                dma_drain();
                if constexpr (HALOBN) {
                    if (tt == 0) {
                        // y = relu(x * scale + shift) on the pieces THIS thread staged (they have landed: vmcnt(0) above), in
                        // place, before the barrier publishes the image.  Rows outside the tensor keep their zero fill (no tap
                        // that is inside its image ever reads them); rows HR.. are the zero rows.
                        const float sc[8] = {bsc0.x, bsc0.y, bsc0.z, bsc0.w, bsc1.x, bsc1.y, bsc1.z, bsc1.w};
                        const float sh[8] = {bsh0.x, bsh0.y, bsh0.z, bsh0.w, bsh1.x, bsh1.y, bsh1.z, bsh1.w};
                        char* const lbase = smem + wave * (8 * 128) + lane * 16;
                        const long gch = (long)c * 128 + (long)a_chunk;  // byte offset of the lane's vector inside a row
                        // (one pass at a time: unrolled, the eight passes' vectors and products are all live at once and the
                        // kernel spills 57 registers into its K loop's neighbourhood)
#pragma unroll 1
                        for (int i = 0; i < npass; ++i) {
                            {
                                const int j = (tid >> 3) + i * 32, q = q0 + j;
                                if (j < HR && (unsigned)q < (unsigned)p.M) {
                                    uint4* lp = reinterpret_cast<uint4*>(lbase + i * 4096);
                                    float v[8];
                                    unpack16<T>(*lp, v);
#pragma unroll
                                    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
                                    // the activation / mask of a row leave from ONE workgroup: the tile of column block 0 (every
                                    // column block transforms its own LDS copy; ADVICE r5: 2-4 x the stores otherwise)
                                    const bool own = n0 == 0 && j >= hoff && j < hoff + BM;
                                    const long goff = (long)q * row_bytes + gch;
                                    if (own) {
                                        unsigned mk = 0;
#pragma unroll
                                        for (int e = 0; e < 8; ++e) mk |= (v[e] > 0.f ? 1u : 0u) << e;
                                        p.in_mask[goff >> 4] = (uint8_t)mk;
                                    }
#pragma unroll
                                    for (int e = 0; e < 8; ++e) v[e] = relu_f32(v[e]);
                                    const uint4 pk = pack16<T>(v);
                                    *lp = pk;
                                    if (own) stg16<true>(p.in_act + goff, pk);
                                }
                            }
                        }
                    }
                }
#ifdef SM3_STAMP
                const unsigned long long q3 = SM3_STAMP_NOW();
#endif
                __syncthreads();
#ifdef SM3_STAMP
                const unsigned long long q4 = SM3_STAMP_NOW();
#endif
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    uint4 fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(smem + (fah[i] ^ (kk << 5)));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(smem + (fbh[j] ^ (kk << 5)));
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) mma_frag<T>(fa[i], fb[j], acc[i][j]);
                }
#ifdef SM3_STAMP
                const unsigned long long q5 = SM3_STAMP_NOW();
                stamp.seg[0] += q2 - q1; stamp.seg[1] += q5 - q4; stamp.seg[2] += q3 - q2; stamp.seg[3] += (q1 - q0) + (q4 - q3);
#endif
            }
        }
#ifdef SM3_STAMP
        stamp.t_loop1 = SM3_STAMP_NOW();
#endif
        __syncthreads();  // the epilogue re-uses the image
    } else {
    dma_stage(0, 0, wtap_off);
    if constexpr (STAGES > 2) {
        // Deep pipeline for launches of at most one workgroup per CU (projector Linears: M = 256..512 rows, 32 K-steps):
        // with one stage in flight a K-step costs a full DMA round trip (~1.1 us); here STAGES - 1 stages are in
        // flight, each step waits (counted vmcnt) only for its own stage and passes one barrier.
        constexpr int PER = AI + BI;  // DMA instructions per stage per wave
        for (int s = 1; s < STAGES - 1 && s < nsteps; ++s) {
            advance();
            dma_stage(s, (uint32_t)kc * 128u, wtap_off + (uint32_t)kc * 128u);
        }
        for (int s = 0; s < nsteps; ++s) {
            const int younger = min(STAGES - 2, nsteps - 1 - s);  // stages issued after stage s
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else dma_drain();
            __syncthreads();  // stage s is complete for everyone; everyone has finished computing stage s - 1
            if (s + STAGES - 1 < nsteps) {
                advance();
                dma_stage((s + STAGES - 1) % STAGES, (uint32_t)kc * 128u, wtap_off + (uint32_t)kc * 128u);
            }
            compute(s % STAGES);
        }
        __syncthreads();  // the epilogue re-uses the stage memory
    } else {
        dma_drain();
        __syncthreads();  // publishes the stage
        SM3_MARK(1);
#ifdef SM3_STAMP
        stamp.nsteps = nsteps;
#endif
        for (int s = 0; s < nsteps; ++s) {
#ifdef SM3_STAMP
            const unsigned long long q0 = SM3_STAMP_NOW();
#endif
            compute(0);
#ifdef SM3_STAMP
            const unsigned long long q1 = SM3_STAMP_NOW();
#endif
            __syncthreads();  // everyone is done reading the stage
#ifdef SM3_STAMP
            const unsigned long long q2 = SM3_STAMP_NOW();
            stamp.seg[1] += q1 - q0; stamp.seg[3] += q2 - q1;
#endif
            if (s + 1 < nsteps) {
                advance();
                dma_stage(0, (uint32_t)kc * 128u, wtap_off + (uint32_t)kc * 128u);
#ifdef SM3_STAMP
                const unsigned long long q3 = SM3_STAMP_NOW();
#endif
                dma_drain();
#ifdef SM3_STAMP
                const unsigned long long q4 = SM3_STAMP_NOW();
#endif
                __syncthreads();
#ifdef SM3_STAMP
                const unsigned long long q5 = SM3_STAMP_NOW();
                stamp.seg[0] += q3 - q2; stamp.seg[2] += q4 - q3; stamp.seg[3] += q5 - q4;
#endif
            }
        }
#ifdef SM3_STAMP
        stamp.t_loop0 = stamp.mark[1];
        stamp.t_loop1 = SM3_STAMP_NOW();
#endif
    }
    }
    SM3_MARK(2);

    // ---- lean epilogue: bf16 forward convolution, dense output, no addend / BN-backward fusion ----------------
    // What the train-mode forward launches need, at a quarter of the general epilogue's VALU work (which, with 3-4
    // workgroups per CU, was as long as the HBM time of a K<=256 tile): accumulators -> bf16 pairs in registers
    // (v_cvt_pk_bf16_f32); BatchNorm sums of the ROUNDED values with two v_dot2c_f32_bf16 per pair; the whole
    // tile staged once in LDS as bf16; read back as 16-byte vectors and stored with no arithmetic in between.
    if constexpr (LEAN) {
        static_assert(sizeof(T) == 2, "lean epilogue is for the 16-bit types");
        constexpr int CPR = BN / 8, RSTEP = NT / CPR, NPASS = BM / RSTEP;
        char* sC = smem;
        float* sStat = reinterpret_cast<float*>(smem + (HALO ? p.halo_stat_off : MAIN_BYTES));  // [WM][BN][2]
        const uint32_t ones = ones2<T>();
        const int cc = tid % CPR, r0 = tid / CPR;
        const int ncol = n0 + cc * 8;
        const bool fz = EPI == 3 && p.fz_partials != nullptr;
        const bool fzx = !NOX && fz && p.fz_x != nullptr;
        // ---- EPI >= 2: what the read-back loop needs from global memory, requested NOW (all NPASS rows of this thread:
        // 16-32 KB per workgroup in flight while the accumulators are converted and staged)
        uint32_t eoff[EPI >= 2 ? NPASS : 1];           // byte offset of the thread's vector in row k of the output; ~0 = none
        uint4 pre_add[EPI >= 2 ? NPASS : 1], pre_x[(EPI == 3 && !NOX) ? NPASS : 1];
        unsigned pre_mk[EPI == 3 ? NPASS : 1];
        // EPI 3 asks for the first half of its rows here and for the second half right after the barrier (while the first
        // half is being finished): two operand streams per row would otherwise hold 64 registers beside the accumulators
        constexpr int NEARLY = (EPI == 3 && !NOX) ? NPASS / 2 : NPASS;
        const bool dense_out = PW || (p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.HWout == p.HoWo);
        // pointwise: row k of this thread is output row m0 + r0 + k * RSTEP, RSTEP rows further down each time
        const uint32_t pw_base = (uint32_t)(((long)(m0 + r0) * p.Co + ncol) * 2), pw_step = (uint32_t)RSTEP * (uint32_t)p.Co * 2u;
        auto request = [&](int k) {
            const int m = m0 + r0 + k * RSTEP;
            eoff[k] = ~0u;
            pre_add[k] = make_uint4(0, 0, 0, 0);
            if constexpr (EPI == 3) {
                if constexpr (!NOX) pre_x[k] = make_uint4(0, 0, 0, 0);
                pre_mk[k] = 0xffu;
            }
            if constexpr (PW) {
                if (m < p.M && ncol < p.Co) {
                    eoff[k] = pw_base + (uint32_t)k * pw_step;  // tensors stay below 3 GB (host check)
                    if (p.addend) pre_add[k] = ldg16<true>(p.addend + eoff[k]);
                    if constexpr (EPI == 3) {
                        if constexpr (!NOX) {
                            if (fzx) pre_x[k] = ldg16<true>(p.fz_x + eoff[k]);
                        }
                        if (fz && p.fz_mask) pre_mk[k] = p.fz_mask[eoff[k] >> 4];
                    }
                }
                return;
            }
            if (m < p.M && ncol < p.Co) {
                long opix = m;
                int nn = 0, oy = 0, ox = 0;
                if (!dense_out || p.add_sp_h) {
                    nn = fdiv(m, p.div_HoWo);
                    const int rem = m - nn * p.HoWo;
                    oy = fdiv(rem, p.div_Wo);
                    ox = rem - oy * p.Wo;
                    if (!dense_out) opix = (long)nn * p.HWout + (long)(oy * p.osy + p.ooy) * p.Wout + (ox * p.osx + p.oox);
                }
                eoff[k] = (uint32_t)((opix * p.Co + ncol) * 2);  // tensors stay below 3 GB (host check)
                if (p.addend) {
                    if (p.add_sp_h) {  // compact stride-2 addend: present at even (y, x) only
                        if (((oy | ox) & 1) == 0)
                            pre_add[k] = ldg16<true>(p.addend + ((((long)nn * p.add_sp_h + (oy >> 1)) * p.add_sp_w + (ox >> 1)) * p.Co + ncol) * 2);
                    } else {
                        pre_add[k] = ldg16<true>(p.addend + eoff[k]);
                    }
                }
                if constexpr (EPI == 3) {
                    if constexpr (!NOX) {
                        if (fzx) pre_x[k] = ldg16<true>(p.fz_x + eoff[k]);
                    }
                    if (fz && p.fz_mask) pre_mk[k] = p.fz_mask[eoff[k] >> 4];
                }
            }
        };
        // per-column BatchNorm affine / constant term of this lane's TN columns (accumulator layout), requested BEFORE the
        // row operands: their latency (two dependent L2 round trips per column block, 4 - 6 k cycles of the 7 k the staging
        // phase of the fused-BatchNorm forward launches took) now hides behind the issue of the row requests
        const bool epl = p.ep_scale != nullptr || p.ep_rv != nullptr;
        const bool aff = epl || (SEG && p.col_bias);
        constexpr int NCB = TN, CBW = 32;  // column blocks of this lane, their width
        const int lcol = frow;
        float esc_j[NCB], esh_j[NCB];
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            esc_j[j] = 1.f;
            esh_j[j] = 0.f;
            const int gcol = n0 + wn * WTN + j * CBW + lcol;
            if (epl && gcol < p.Co) {
                if (p.ep_rv) {
                    ep_affine(p, gcol, esc_j[j], esh_j[j]);
                } else {
                    esc_j[j] = p.ep_scale[tile_view * p.Co + gcol];
                    esh_j[j] = p.ep_shift[tile_view * p.Co + gcol];
                }
            }
            if constexpr (SEG) {  // constant term of the linear BatchNorm backward (per output channel and view)
                if (p.col_bias && gcol < p.Co) esh_j[j] += p.col_bias[tile_view * p.Co + gcol];
            }
        }
        if constexpr (EPI >= 2) {
#pragma unroll
            for (int k = 0; k < NEARLY; ++k) request(k);
        }
        SM3_MARK(6);
        // The staging loop in four compile-time flavours (statistics wanted or not, affine or not): the launch-level flags
        // are tested once, not per accumulator pair -- EPI 2 / 3 launches almost never want the BatchNorm sums of their
        // output and used to pay 2 v_dot2 per pair for them.
        const bool early_relu = epl && p.ep_relu && (EPI == 1 || !p.addend) && !p.ep_mask;
        auto stage_tile = [&](auto stats_c, auto aff_c, auto relu_c) {
            constexpr bool STATS = decltype(stats_c)::value, AFF = decltype(aff_c)::value, RELU = decltype(relu_c)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float s1 = 0.f, s2 = 0.f;
                char* colp = sC + (wn * WTN + j * 32 + frow) * 2 + (wm * WTM + 4 * fh) * LEAN_PITCH;
                const uint32_t colp_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)colp);
                // BatchNorm affine (+ReLU when nothing is added afterwards) of this lane's column, in the accumulator layout;
                // per view in the train-mode fused form
                const float esc = esc_j[j], esh = esh_j[j];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {  // registers 2q, 2q+1 = rows R, R+1 of this lane's column
                        float v0 = acc[i][j][2 * q], v1 = acc[i][j][2 * q + 1];
                        if constexpr (AFF) {  // (the ReLU only where nothing is added afterwards: a flavour, not a max with -inf)
                            // both rows share the column's scale / shift: one v_pk_fma_f32 (the same fma, per element)
                            f32x2_t vv = {v0, v1};
                            vv = __builtin_elementwise_fma(vv, f32x2_t{esc, esc}, f32x2_t{esh, esh});
                            v0 = vv.x;
                            v1 = vv.y;
                            if constexpr (RELU) {
                                v0 = fmaxf(v0, 0.f);
                                v1 = fmaxf(v1, 0.f);
                            }
                        }
                        const uint32_t pk = pack2<T>(v0, v1);
                        if constexpr (STATS) {
                            s1 = dot2acc<T>(pk, ones, s1);
                            s2 = dot2acc<T>(pk, pk, s2);
                        }
                        const int R = i * 32 + (q & 1) * 2 + 8 * (q >> 1);
                        // ONE v_cvt_pk per pair: the high half leaves through ds_write_b16_d16_hi (left to itself the compiler
                        // converts each value on its own to keep both in low halves: 64 conversions per tile instead of 32)
                        asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3"
                                     :
                                     : "v"(colp_lds), "v"(pk), "n"(R * LEAN_PITCH), "n"((R + 1) * LEAN_PITCH)
                                     : "memory");
                    }
                if constexpr (STATS) {
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (lane < 32) {
                        const int col = wn * WTN + j * 32 + lane;
                        sStat[(wm * BN + col) * 2 + 0] = s1;
                        sStat[(wm * BN + col) * 2 + 1] = s2;
                    }
                }
            }
        };
        using std::true_type;
        using std::false_type;
        if (p.partials) {
            if (aff && early_relu) stage_tile(true_type{}, true_type{}, true_type{});
            else if (aff) stage_tile(true_type{}, true_type{}, false_type{});
            else stage_tile(true_type{}, false_type{}, false_type{});
        } else {
            if (aff && early_relu) stage_tile(false_type{}, true_type{}, true_type{});
            else if (aff) stage_tile(false_type{}, true_type{}, false_type{});
            else stage_tile(false_type{}, false_type{}, false_type{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging writes above are issued from asm
        __syncthreads();
        SM3_MARK(3);
        if (p.partials && tid < BN && n0 + tid < p.Co) {
            // one partial row per 128 tile rows, summed over their wave rows in wave order: a 256-row tile (kVarHalo, tall)
            // leaves the same two rows, bit for bit, as the two 128-row tiles it replaces
            constexpr int G = BM / 128 > 0 ? BM / 128 : 1, WPG = WM / G;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int w = 0; w < WPG; ++w) {
                    s1 += sStat[((g * WPG + w) * BN + tid) * 2 + 0];
                    s2 += sStat[((g * WPG + w) * BN + tid) * 2 + 1];
                }
                const long prow = (long)bm * G + g;
                if (G == 1 || prow * 128 < p.M) {
                    p.partials[(prow * 2 + 0) * p.Co + n0 + tid] = s1;
                    p.partials[(prow * 2 + 1) * p.Co + n0 + tid] = s2;
                }
            }
        }
        const char* sp = sC + r0 * LEAN_PITCH + cc * 16;
        if constexpr (EPI == 3) {
#pragma unroll
            for (int k = NEARLY; k < NPASS; ++k) request(k);
        }
        if constexpr (EPI == 1) {
            if (ncol < p.Co) {
                char* yp = p.y + ((long)(m0 + r0) * p.Co + ncol) * 2;
                const long ystep = (long)RSTEP * p.Co * 2;
#pragma unroll
                for (int k = 0; k < NPASS; ++k) {
                    if (m0 + r0 + k * RSTEP < p.M)
                        stg16<true>(yp + k * ystep, *reinterpret_cast<const uint4*>(sp + k * RSTEP * LEAN_PITCH));
                }
            }
            SM3_MARK(4);
            return;
        } else {
            const bool epl2 = p.ep_scale != nullptr || p.ep_rv != nullptr;
            const bool late_relu = p.ep_relu && !(epl2 && !p.addend && !p.ep_mask);  // i.e. not applied before the staging
            float f_mu[8], f_is[8], f_s1[8], f_s2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f_s1[e] = f_s2[e] = f_mu[e] = f_is[e] = 0.f;
                if (EPI == 3 && fzx && ncol < p.Co) {
                    f_mu[e] = p.fz_mean[tile_view * p.Co + ncol + e];
                    f_is[e] = p.fz_invstd[tile_view * p.Co + ncol + e];
                }
            }
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                if (eoff[k] == ~0u) continue;
                uint4 packed = *reinterpret_cast<const uint4*>(sp + k * RSTEP * LEAN_PITCH);
                if (p.addend || late_relu || (EPI == 3 && fz)) {
                    float v[8];
                    unpack16<T>(packed, v);
                    if (p.addend) {
                        float a[8];
                        unpack16<T>(pre_add[k], a);
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {  // v_pk_add_f32
                            const f32x2_t t = f32x2_t{v[e], v[e + 1]} + f32x2_t{a[e], a[e + 1]};
                            v[e] = t.x;
                            v[e + 1] = t.y;
                        }
                    }
                    if (late_relu) {
                        if (p.ep_mask) {  // what the backward pass needs of the output: one bit per element
                            unsigned mbits = 0;
#pragma unroll
                            for (int e = 0; e < 8; ++e) mbits |= (v[e] > 0.f ? 1u : 0u) << e;
                            p.ep_mask[eoff[k] >> 4] = (uint8_t)mbits;
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = relu_f32(v[e]);
                    }
                    if constexpr (EPI == 3) {
                        if (fz) {
                            const unsigned mk = pre_mk[k];
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = keep_if_bit(v[e], mk, e);
                        }
                    }
                    packed = pack16<T>(v);
                    if constexpr (EPI == 3) {
                        if (fz) {  // sums of the STORED (rounded) dz, as the standalone kernel's
                            float dzr[8];
                            unpack16<T>(packed, dzr);
                            if constexpr (NOX) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) f_s1[e] += dzr[e];
                            } else {
                                float xv[8];
                                unpack16<T>(pre_x[k], xv);
#pragma unroll
                                for (int e = 0; e < 8; ++e) {
                                    f_s1[e] += dzr[e];
                                    f_s2[e] += dzr[e] * (xv[e] - f_mu[e]) * f_is[e];  // f_mu = f_is = 0 without x: stays 0
                                }
                            }
                        }
                    }
                }
                stg16<true>(p.y + eoff[k], packed);
            }
            SM3_MARK(4);
            if constexpr (EPI == 3) {
                if (fz) {
                    const int fz_prow = tile_view ? p.fz_row_off1 + bm - p.fz_view_tiles : p.fz_row_off + bm;
                    __syncthreads();  // everyone is done reading sC: reuse it for the cross-thread reduction
                    // sRed[value e][thread], rows of NT + 8 floats: a wave writes 64 consecutive banks, and the 64 readers of
                    // a wave (value col % 8 of thread rl * CPR + col / 8) hit bank 8 * (col % 8) + col / 8 + const -- all
                    // different.  (The [thread][16] layout of round 3 was an 8-way conflict on every write: most of the
                    // 28 % LDS bank-conflict cycles of this instantiation.)
                    constexpr int RP = NT + 8;
                    constexpr int NSTAT = NOX ? 1 : 2;  // without x the sum(dz * xhat) slot is written as 0
                    float* sRed = reinterpret_cast<float*>(smem);  // [8 * NSTAT][RP] floats <= 16.5 KB
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sRed[e * RP + tid] = f_s1[e];
                        if constexpr (!NOX) sRed[(8 + e) * RP + tid] = f_s2[e];
                    }
                    __syncthreads();
                    for (int o = tid; o < 2 * BN; o += NT) {
                        const int stat = o / BN, col = o % BN;  // channel n0+col lives in threads with cc == col/8
                        float a = 0.f;
                        if (stat < NSTAT) {
                            const float* src = sRed + (stat * 8 + col % 8) * RP + col / 8;
                            for (int rl = 0; rl < NT / CPR; ++rl) a += src[rl * CPR];
                        }
                        if (n0 + col < p.Co) p.fz_partials[((long)fz_prow * 2 + stat) * p.Co + n0 + col] = a;
                    }
                    SM3_MARK(5);
                }
            }
            return;
        }
    }

    // ---- epilogue -----------------------------------------------------------------------
    // One wave-row of the tile (WTM rows) at a time: its waves write their accumulators to LDS (f32), then all
    // threads read rows back as 16-byte vectors (a thread keeps one channel vector for all its rows) and store
    // NHWC.  Operands the store loop needs from global memory (addend; fused BN-backward x and ReLU bits) are
    // requested BEFORE the staging so their latency hides under it.
    constexpr int EPC = 16 / SZ;       // elements per 16-byte store
    constexpr int CPR = BN / EPC;      // stores per tile row
    constexpr int RSTEP = NT / CPR;    // rows between two stores of one thread
    constexpr int NROW = WTM / RSTEP;  // stores per thread per wave-row
    static_assert(NT % CPR == 0 && WTM % RSTEP == 0, "a thread must keep one channel vector across its rows");
    const bool dense = (p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.HWout == p.HoWo);
    const bool fz = p.fz_partials != nullptr;  // BatchNorm-backward phase 1 of the producer unit in this epilogue
    const bool fzx = p.fz_x != nullptr;        // ... including sum(dz * xhat); without x only the mask and sum(dz)
    const int cc = tid % CPR;  // fixed per thread
    const int ncol = n0 + cc * EPC;
    float* sC = reinterpret_cast<float*>(smem);
    float* sStat = reinterpret_cast<float*>(smem + MAIN_BYTES);  // [WM][BN][2]

    if (p.partials) {  // train-mode BN statistics of the stored (rounded) outputs
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

    // f_mu/f_is double as the inference epilogue's per-channel scale/shift (the two modes are exclusive)
    const bool ep = p.ep_scale != nullptr || p.ep_rv != nullptr;
    const int fz_view = tile_view;
    const int fz_prow = fz_view ? p.fz_row_off1 + bm - p.fz_view_tiles : p.fz_row_off + bm;
    float f_mu[EPC], f_is[EPC], f_s1[EPC], f_s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        f_s1[e] = 0.f;
        f_s2[e] = 0.f;
        f_mu[e] = 0.f;
        f_is[e] = 0.f;
        if (ncol < p.Co) {
            if (fzx) {
                f_mu[e] = p.fz_mean[fz_view * p.Co + ncol + e];
                f_is[e] = p.fz_invstd[fz_view * p.Co + ncol + e];
            } else if (ep) {
                if (p.ep_rv) {
                    ep_affine(p, ncol + e, f_is[e], f_mu[e]);
                } else {  // per-view vectors (train-mode fused form); view 0 = the only one at inference
                    f_is[e] = p.ep_scale[tile_view * p.Co + ncol + e];
                    f_mu[e] = p.ep_shift[tile_view * p.Co + ncol + e];
                }
            }
        }
    }

#pragma unroll
    for (int hh = 0; hh < WM; ++hh) {
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
                long opix = m;
                if (!dense) {
                    const int nn = fdiv(m, p.div_HoWo);
                    const int rem = m - nn * p.HoWo;
                    const int oy = fdiv(rem, p.div_Wo);
                    const int ox = rem - oy * p.Wo;
                    opix = (long)nn * p.HWout + (long)(oy * p.osy + p.ooy) * p.Wout + (ox * p.osx + p.oox);
                }
                e_off[k] = opix * p.Co + ncol;
                if (p.addend) {
                    if (p.add_sp_h) {  // compact stride-2 addend: present at even (y, x) only (dense output)
                        const int nn = fdiv(m, p.div_HoWo);
                        const int rem = m - nn * p.HoWo;
                        const int oy = fdiv(rem, p.div_Wo);
                        const int ox = rem - oy * p.Wo;
                        if (((oy | ox) & 1) == 0)
                            pre_add[k] = ldg16<true>(p.addend + ((((long)nn * p.add_sp_h + (oy >> 1)) * p.add_sp_w + (ox >> 1)) * p.Co + ncol) * SZ);
                    } else {
                        pre_add[k] = ldg16<true>(p.addend + e_off[k] * SZ);
                    }
                }
                if (fz) {
                    if (fzx) pre_x[k] = ldg16<true>(p.fz_x + e_off[k] * SZ);
                    if (p.fz_mask) pre_mk[k] = p.fz_mask[e_off[k] / EPC];
                }
            }
        }
        if (hh > 0) __syncthreads();  // the previous wave-row has been read out of sC
        if (wm == hh) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        const int col = wn * WTN + j * 32 + frow;
                        sC[row * CP + col] = acc[i][j][r];
                    }
        }
        __syncthreads();
        if (hh == 0 && p.partials && tid < BN && n0 + tid < p.Co) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                s1 += sStat[(w * BN + tid) * 2 + 0];
                s2 += sStat[(w * BN + tid) * 2 + 1];
            }
            p.partials[((long)bm * 2 + 0) * p.Co + n0 + tid] = s1;
            p.partials[((long)bm * 2 + 1) * p.Co + n0 + tid] = s2;
        }
#pragma unroll
        for (int k = 0; k < NROW; ++k) {
            if (e_off[k] < 0) continue;
            const int r = tid / CPR + k * RSTEP;  // row inside this wave-row
            float v[EPC];
#pragma unroll
            for (int e = 0; e < EPC; e += 4) {
                const float4 q = *reinterpret_cast<const float4*>(&sC[r * CP + cc * EPC + e]);
                v[e] = q.x;
                v[e + 1] = q.y;
                v[e + 2] = q.z;
                v[e + 3] = q.w;
            }
            const long boff = e_off[k] * SZ;
            if (ep) {  // eval-mode BatchNorm folded into the epilogue (f32, before the residual add)
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = v[e] * f_is[e] + f_mu[e];
            }
            if constexpr (SEG) {
                if (p.col_bias) {  // constant term of the linear BatchNorm backward (per output channel and view)
                    const float* cb = p.col_bias + (long)tile_view * p.Co + ncol;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] += cb[e];
                }
            }
            if (p.addend) {
                float a[EPC];
                unpack16<T>(pre_add[k], a);
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] += a[e];
            }
            if (p.ep_relu) {
                if (p.ep_mask) {  // what the backward pass needs of the output: one bit per element
                    unsigned m = 0;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) m |= (v[e] > 0.f ? 1u : 0u) << e;
                    p.ep_mask[e_off[k] / EPC] = (uint8_t)m;
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = relu_f32(v[e]);
            }
            if (fz) {
                // BN-backward phase 1 fused here (data-gradient launches): the value just computed is dy of the
                // producer BatchNorm's output; mask it with that BN's ReLU bits, store dz instead of dy, and
                // accumulate this tile's sum(dz), sum(dz * xhat) per channel.
                float xv[EPC];
                unpack16<T>(pre_x[k], xv);
                const unsigned mk = pre_mk[k];
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = keep_if_bit(v[e], mk, e);
                const uint4 packed = pack16<T>(v);
                float dzr[EPC];
                unpack16<T>(packed, dzr);  // sums are those of the STORED (rounded) dz, as the standalone kernel's
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    f_s1[e] += dzr[e];
                    f_s2[e] += dzr[e] * (xv[e] - f_mu[e]) * f_is[e];  // f_mu = f_is = 0 without x: the slot stays 0
                }
                stg16<true>(p.y + boff, packed);
            } else {
                stg16<true>(p.y + boff, pack16<T>(v));
            }
        }
    }
    if (fz) {
        __syncthreads();  // everyone is done reading sC: reuse it for the cross-thread reduction
        float* sRed = reinterpret_cast<float*>(smem);  // [NT][2*EPC] floats = 16 KB
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            sRed[tid * 2 * EPC + e] = f_s1[e];
            sRed[tid * 2 * EPC + EPC + e] = f_s2[e];
        }
        __syncthreads();
        for (int o = tid; o < 2 * BN; o += NT) {
            const int stat = o / BN, col = o % BN;  // channel n0+col lives in threads with cc == col/EPC
            float a = 0.f;
            for (int rl = 0; rl < NT / CPR; ++rl) a += sRed[(rl * CPR + col / EPC) * 2 * EPC + stat * EPC + col % EPC];
            if (n0 + col < p.Co) p.fz_partials[((long)fz_prow * 2 + stat) * p.Co + n0 + col] = a;
        }
    }
}

constexpr int kHaloLdsMax = 40960;  // a quarter of a CU's LDS: the halo kernels keep 4 workgroups per CU
constexpr int kHaloLdsMaxTall = 53248;  // 256-row tiles (SM3_CONV_HALO_TALL): a third of a CU's LDS, 3 workgroups per CU

// SM3_CONV_HALO (A/B switch, read at every launch): the halo-resident A image (kVarHalo) for the launches it fits:
// nine taps at (-1..1, -1..1) over one tensor, stride 1, same geometry in and out, dense output and addend.
// Default on: 3x3 forward 887 -> 967 (256 ch), 906 -> 1 004 (128 ch), 718 -> 802 (64 ch) TFLOP/s, data gradient + fused
// BN-backward phase 1 +7 ... +14 %, whole step 4 321 -> 4 432 pairs/s (profiles/r04c_halo_ab.txt).  The sums are the same,
// their order is not (chunk outer, tap inner): results agree with the general gather to one rounding of the 16-bit output.
static int conv_halo_mode() {
    const char* v = getenv("SM3_CONV_HALO");
    return v ? atoi(v) : 1;
}
// the launches whose K order is "chunk outer, tap inner" (ConvParams::kord): a property of the LAYER, never of the grid
static bool conv_halo_geometry(const ConvParams& p) {
    if (!conv_halo_mode() || p.ntaps != 9 || p.x1 || p.sy != 1 || p.sx != 1 || p.add_sp_h) return false;
    const bool dense = p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.HWout == p.HoWo;
    if (!dense || p.Hi * p.Wi != p.HoWo || p.Wi != p.Wo) return false;
    unsigned seen = 0;
    for (int t = 0; t < 9; ++t) {
        if (p.dy[t] < -1 || p.dy[t] > 1 || p.dx[t] < -1 || p.dx[t] > 1) return false;
        seen |= 1u << ((p.dy[t] + 1) * 3 + p.dx[t] + 1);
    }
    return seen == 0x1ffu;
}
// ... and of those the ones whose A image fits a quarter of a CU's LDS run on the halo kernel itself
template <int BM, int BN>
static bool conv_halo_ok(const ConvParams& p) {
    if (!p.kord) return false;
    const int rows = BM + 2 * (p.Wi + 1) + 2;
    return ((rows + 31) / 32) * 4096 + BN * 128 <= (BM == 256 ? kHaloLdsMaxTall : kHaloLdsMax);
}
// SM3_CONV_HALO_TALL (A/B switch, read at every launch): the plain-epilogue 3x3 forward launches on 256 x 64 tiles (4 x 1
// waves of 64 x 64: the same wave tile, accumulators and fragment traffic as 128 x 128) -- per nine K-steps a workgroup stages
// an A image of 256 + 2 (W + 1) rows once and nine 8 KB B tiles, 108 - 112 KB for the FLOPs the 128 x 128 tile stages 168 KB
// for (the B tiles are the larger stream since the image became resident, and L2 -> LDS bytes are what these launches and
// their co-runners contend for: profiles/r05_corun_regs.txt), at three workgroups per CU instead of four.
static int conv_halo_tall_mode() {
    const char* v = getenv("SM3_CONV_HALO_TALL");
    return v ? atoi(v) : 0;
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES, int EPI, bool SEG = false, int VAR = 0>
int launch_conv_st(const ConvParams& p0, hipStream_t st) {
    ConvParams p = p0;
    {
        const char* dv = getenv("SM3_CONV_DBG");
        p.dbg = dv ? atoi(dv) : 0;
    }
    constexpr bool LEAN = EPI >= 1;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int C_BYTES = LEAN ? BM * (BN * 2 + 16) : (BM / WM) * (BN + 4) * 4;
    constexpr int MAIN = (STAGES * STAGE > C_BYTES) ? STAGES * STAGE : C_BYTES;
    constexpr int STAT = WM * BN * 2 * 4;
    constexpr int LDS_STATIC = MAIN + STAT;
    static_assert(MAIN >= 256 * 2 * 8 * 4, "reduction scratch of the fused BN-backward epilogue must fit");
    int LDS = LDS_STATIC;
    if constexpr ((VAR & kVarHalo) != 0) {
        // halo image: 128 tile rows + (W + 1) either side + a zero row, in whole 32-row passes; the B tile behind it; the
        // statistics scratch of the epilogue sits at the end (inside the by then dead image when that is the larger)
        p.halo_rows = BM + 2 * (p.Wi + 1);
        p.halo_a_bytes = ((p.halo_rows + 2 + 31) / 32) * 4096;  // + two zero rows (one 256-byte bank row)
        const int image = p.halo_a_bytes + BN * 128;
        LDS = image > C_BYTES + STAT ? image : C_BYTES + STAT;
        if (LDS > (BM == 256 ? kHaloLdsMaxTall : kHaloLdsMax)) return SM3_EINVAL;  // (conv_halo_ok() is asked first)
        p.halo_stat_off = LDS - STAT;
    }
    p.tilesM = (p.M + BM - 1) / BM;
    p.tilesN = (p.Co + BN - 1) / BN;
    auto kern = conv_igemm_kernel<T, BM, BN, WM, WN, STAGES, EPI, SEG, VAR>;
    // the dynamic-LDS limit is a per-device attribute of the function: set it once per (instantiation, device)
    static std::atomic<uint32_t> attr_set{0};  // bit d: done on device d
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    if (!(attr_set.load(std::memory_order_acquire) & (1u << dev))) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (VAR & kVarHalo) ? (BM == 256 ? kHaloLdsMaxTall : kHaloLdsMax) : LDS_STATIC);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(1u << dev, std::memory_order_release);
    }
    const long nblocks = (long)p.tilesM * p.tilesN;
    if (nblocks <= 0 || nblocks > 0x7fffffffL) return SM3_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(WM * WN * 64), LDS, st, p);
    SM3_CHECK_LAUNCH();
    return 0;
}

// SM3_CONV_PW (A/B switch, read at every launch; default 3): bit 0 = the pointwise variants (kVarPw) for the launches that
// qualify, bit 1 = the no-x variant of the fused BN-backward epilogue (kVarNoX) where no x is given.
static int conv_pw_mode() {
    const char* v = getenv("SM3_CONV_PW");
    return v ? atoi(v) : 3;
}
// a tile row is pixel m0 + r of the source AND of the output: one tap at (0, 0) (or two segments, which the host only builds
// over the same pixels), stride 1, same geometry in and out, dense output, dense addend
static bool conv_is_pointwise(const ConvParams& p) {
    const bool dense = p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.HWout == p.HoWo;
    const bool taps0 = (p.x1 != nullptr) || (p.ntaps == 1 && p.dy[0] == 0 && p.dx[0] == 0);
    return dense && taps0 && p.sy == 1 && p.sx == 1 && p.Hi * p.Wi == p.HoWo && p.add_sp_h == 0;
}

// 16-bit lean epilogues on the one-stage K loop (34 KB of LDS, 4 workgroups per CU overlapping each other: it won at every
// K length measured in round 4, profiles/r04a_stage_choice_*; the double-buffered loop of rounds 1-3 was removed in round 6)
template <typename T, int BM, int BN, int WM, int WN, int EPI, bool SEG>
int launch_conv_lean(const ConvParams& p, hipStream_t st) {
    if constexpr (!SEG && EPI == 1 && BM == 128 && WM == 2 && WN == 2) {
        // 256 x 64 tiles: whole 64-column tiles only, and at least two rounds of them (3 workgroups x 256 CUs)
        if (conv_halo_tall_mode() && p.Co % 64 == 0 && conv_halo_ok<256, 64>(p) &&
            (long)((p.M + 255) / 256) * (p.Co / 64) >= 1536) {  // (784 tiles on 768 slots: +16 % alone, profiles/r05_tall_tiles.txt)
            ConvParams q = p;
            if (q.fz_view_tiles) q.fz_view_tiles = (q.fz_view_tiles % 2) ? 0 : q.fz_view_tiles / 2;  // (unused by this epilogue)
            return launch_conv_st<T, 256, 64, 4, 1, 1, 1, false, kVarHalo>(q, st);
        }
    }
    if constexpr (!SEG && (EPI == 1 || EPI == 3) && WM * WN == 4) {
        if (conv_halo_ok<BM, BN>(p)) return launch_conv_st<T, BM, BN, WM, WN, 1, EPI, false, kVarHalo>(p, st);
    }
    const int mode = conv_pw_mode();
    if ((mode & 1) && conv_is_pointwise(p)) {
        if constexpr (EPI == 3 && !SEG) {
            if ((mode & 2) && p.fz_partials && !p.fz_x) return launch_conv_st<T, BM, BN, WM, WN, 1, 3, false, kVarPw | kVarNoX>(p, st);
        }
        return launch_conv_st<T, BM, BN, WM, WN, 1, EPI, SEG, kVarPw>(p, st);
    }
    return launch_conv_st<T, BM, BN, WM, WN, 1, EPI, SEG>(p, st);
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
int launch_conv_epi(const ConvParams& p, hipStream_t st, bool deep) {
    if (deep) return launch_conv_st<T, BM, BN, WM, WN, 4, EPI>(p, st);
    if constexpr (EPI >= 1 && sizeof(T) == 2) return launch_conv_lean<T, BM, BN, WM, WN, EPI, false>(p, st);
    return launch_conv_st<T, BM, BN, WM, WN, 1, EPI>(p, st);
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_conv(const ConvParams& p_in, hipStream_t st) {
    ConvParams p = p_in;
    p.kord = 0;
    const char* lv = getenv("SM3_CONV_LEAN");
    const bool lean = !(lv && atoi(lv) == 0);  // SM3_CONV_LEAN=0: everything through the general epilogue (A/B, debugging)
    if (p.x1) {  // two K segments (16-bit types only: the exact-f32 parity mode never takes the linear BatchNorm backward)
        if constexpr (sizeof(T) == 2) {
            if (lean && p.fz_partials) return launch_conv_lean<T, BM, BN, WM, WN, 3, true>(p, st);
            if (lean) return launch_conv_lean<T, BM, BN, WM, WN, 2, true>(p, st);
            return launch_conv_st<T, BM, BN, WM, WN, 1, 0, true>(p, st);
        } else {
            return SM3_EDTYPE;
        }
    }
    // at most one workgroup per CU and a K-loop worth pipelining: 3 stages in flight (128 KB of LDS, see the kernel)
    const long nblocks = (long)((p.M + BM - 1) / BM) * ((p.Co + BN - 1) / BN);
    const char* dv = getenv("SM3_CONV_DEEP");
    const bool deep = !(dv && atoi(dv) == 0) && nblocks <= 256 && p.ntaps * p.kchunks >= 6;
    if constexpr (sizeof(T) == 2) {
        if (lean) {
            const bool dense = p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 && p.HWout == p.HoWo;
            // 16-bit stride-1 3 x 3 launches sum chunk outer, tap inner in whichever kernel and with whichever epilogue the
            // tests below pick (the K order is a property of the layer, not of the grid)
            p.kord = conv_halo_geometry(p) ? 1 : 0;
            if (p.fz_partials) return launch_conv_epi<T, BM, BN, WM, WN, 3>(p, st, deep);  // data gradient + BN-backward phase 1
            // per-row work on the read-back: an addend, ReLU bits, per-view affine, a strided output
            if (p.addend || p.ep_mask || !dense || (p.ep_scale && p.fz_view_tiles))
                return launch_conv_epi<T, BM, BN, WM, WN, 2>(p, st, deep);
            return launch_conv_epi<T, BM, BN, WM, WN, 1>(p, st, deep);  // train-mode forward, conv + evalBN (+ReLU)
        }
    }
    return launch_conv_epi<T, BM, BN, WM, WN, 0>(p, st, deep);
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
    // buffer descriptors address each operand with a 32-bit offset; 0xC0000000 must stay out of range
    const long xb = (long)d->N * d->Hi * d->Wi * d->Ci * sz;
    long wmax = 0;
    for (int t = 0; t < d->ntaps; ++t) wmax = d->wtap[t] > wmax ? d->wtap[t] : wmax;
    const long wb = ((long)(d->Co - 1) * d->w_row_stride + (wmax + 1) * d->Ci) * sz;
    if (xb >= 0xC0000000L || wb >= 0xC0000000L) return SM3_EINVAL;
    p.x_bytes = (uint32_t)xb;
    p.w_bytes = (uint32_t)wb;
    return 0;
}

}  // namespace

extern "C" int sm3_conv_partial_rows(const sm3_conv_desc* d) {
    if (!d) return SM3_EINVAL;
    const long M = (long)d->N * d->Ho * d->Wo;
    return (int)((M + kBM - 1) / kBM);
}

struct EvalBn {  // BatchNorm folded into the epilogue: precomputed vectors ([views][Co]), or an eval BatchNorm's tensors
    const float *scale, *shift, *gamma, *beta, *rm, *rv;
    float eps;
    int relu;
    uint8_t* mask = nullptr;  // train-mode fused form: ReLU bits of the output
    int views = 1;
};

struct BnIn {  // sm3_conv3x3_bnin: the producer's BatchNorm + ReLU applied to the A image in LDS (kVarHaloBn)
    const float *scale, *shift;
    int views;
    void* act;
    uint8_t* mask;
    bool probe;  // true: launch nothing, return 0 iff the launch would take the halo kernel
};

// the launches sm3_conv3x3_bnin can take: what launch_conv would route to the halo-resident kernel with the plain epilogue
template <typename T>
static int launch_bnin(const ConvParams& p_in, int Co, bool probe, hipStream_t st) {
    if constexpr (sizeof(T) != 2) {
        return SM3_EDTYPE;
    } else {
        ConvParams p = p_in;
        const char* lv = getenv("SM3_CONV_LEAN");
        if (lv && atoi(lv) == 0) return SM3_EINVAL;
        p.kord = conv_halo_geometry(p) ? 1 : 0;
        const long tiles128 = (long)((p.M + kBM - 1) / kBM) * ((Co + 127) / 128);
        const bool narrow = Co <= 64 || (tiles128 <= 96 && p.ntaps * p.kchunks >= 6);
        const long nblocks = (long)((p.M + kBM - 1) / kBM) * ((Co + (narrow ? 63 : 127)) / (narrow ? 64 : 128));
        const char* dv = getenv("SM3_CONV_DEEP");
        const bool deep = !(dv && atoi(dv) == 0) && nblocks <= 256 && p.ntaps * p.kchunks >= 6;
        if (deep || !p.kord) return SM3_EINVAL;
        // SM3_CONV_BNIN (A/B, read at every launch): 2 = only the 64-column launches (layer 1: no spills, the largest tensors),
        // 3 = only the 128-column ones
        const char* bv = getenv("SM3_CONV_BNIN");
        const int bmode = bv ? atoi(bv) : 1;
        if ((bmode == 2 && !narrow) || (bmode == 3 && narrow)) return SM3_EINVAL;
        if (narrow) {
            if (!conv_halo_ok<kBM, 64>(p)) return SM3_EINVAL;
            return probe ? 0 : launch_conv_st<T, kBM, 64, 2, 2, 1, 1, false, kVarHalo | kVarHaloBn>(p, st);
        }
        if (!conv_halo_ok<kBM, 128>(p)) return SM3_EINVAL;
        return probe ? 0 : launch_conv_st<T, kBM, 128, 2, 2, 1, 1, false, kVarHalo | kVarHaloBn>(p, st);
    }
}

static int conv_gather_gemm_impl(const sm3_conv_desc* d, const void* x, const void* w, void* y, const void* addend,
                                 float* stat_partials, const sm3_bn_bwd_fuse* fuse, void* stream,
                                 const EvalBn* ebn = nullptr, const sm3_conv_seg* seg = nullptr, const BnIn* bnin = nullptr) {
    if (!d || ((!x || !w || !y) && !(bnin && bnin->probe))) return SM3_EINVAL;
    // fuse->x NULL: ReLU mask + sum(dz) only (the sum(dz * xhat) slot of the partial rows is written as 0)
    if (fuse && ((fuse->x && (!fuse->mean || !fuse->invstd)) || !fuse->partials || fuse->partial_row_offset < 0))
        return SM3_EINVAL;
    if (!SM3_DTYPE_OK(d->dtype)) return SM3_EDTYPE;
    ConvParams p;
    const int sz = d->dtype == SM3_F32 ? 4 : 2;
    int rc = fill_params(d, p, sz);
    if (rc) return rc;
    p.x = (const char*)x; p.w = (const char*)w; p.y = (char*)y; p.addend = (const char*)addend;
    p.partials = stat_partials;
    p.fz_mask = fuse ? fuse->relu_mask : nullptr;
    p.fz_x = fuse ? (const char*)fuse->x : nullptr;
    p.fz_mean = fuse ? fuse->mean : nullptr;
    p.fz_invstd = fuse ? fuse->invstd : nullptr;
    p.fz_partials = fuse ? fuse->partials : nullptr;
    p.fz_row_off = fuse ? fuse->partial_row_offset : 0;
    p.fz_view_tiles = 0;
    p.fz_row_off1 = 0;
    p.add_sp_h = p.add_sp_w = 0;
    if (fuse && fuse->addend_sp_h > 0) {
        const bool dense = d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->Hout == d->Ho && d->Wout == d->Wo;
        if (!addend || !dense || fuse->addend_sp_h != (d->Ho + 1) / 2 || fuse->addend_sp_w != (d->Wo + 1) / 2) return SM3_EINVAL;
        p.add_sp_h = fuse->addend_sp_h;
        p.add_sp_w = fuse->addend_sp_w;
    }
    if (fuse && fuse->views > 1) {
        if (fuse->views != 2 || (p.M % 2) || ((p.M / 2) % kBM) || fuse->partial_row_offset_view1 < 0) return SM3_EALIGN;
        p.fz_view_tiles = p.M / 2 / kBM;
        p.fz_row_off1 = fuse->partial_row_offset_view1;
    }
    p.ep_scale = ebn ? ebn->scale : nullptr;
    p.ep_shift = ebn ? ebn->shift : nullptr;
    p.ep_gamma = ebn ? ebn->gamma : nullptr;
    p.ep_beta = ebn ? ebn->beta : nullptr;
    p.ep_rm = ebn ? ebn->rm : nullptr;
    p.ep_rv = ebn ? ebn->rv : nullptr;
    p.ep_eps = ebn ? ebn->eps : 0.f;
    p.ep_relu = ebn ? ebn->relu : 0;
    p.ep_mask = ebn ? ebn->mask : nullptr;
    if (ebn && ebn->views > 1) {
        if (ebn->views != 2 || fuse || (p.M % 2) || ((p.M / 2) % kBM)) return SM3_EALIGN;
        p.fz_view_tiles = p.M / 2 / kBM;
    }
    p.x1 = p.w1 = nullptr;
    p.col_bias = nullptr;
    p.x1_bytes = p.w1_bytes = p.w_view_bytes = p.w1_view_bytes = 0;
    p.Ci1 = p.kchunks1 = p.w1_row_stride = p.nsteps_seg = 0;
    for (int t = 0; t < SM3_MAX_TAPS; ++t) p.tap_src[t] = 0;
    if (seg) {
        // Y = X W^T + X1 W1^T (+ col_bias): a 1x1 / stride-1 / dense product over the same pixels
        const bool plain = d->ntaps == 1 && d->sy == 1 && d->sx == 1 && d->dy[0] == 0 && d->dx[0] == 0 && d->wtap[0] == 0 &&
                           d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->Hout == d->Ho &&
                           d->Wout == d->Wo && d->Ho == d->Hi && d->Wo == d->Wi;
        if (!plain || !seg->x1 || !seg->w1 || seg->Ci1 <= 0 || sz != 2 || (ebn && (ebn->scale || ebn->rv))) return SM3_EINVAL;
        if ((seg->Ci1 * sz) % 128 != 0) return SM3_EALIGN;
        const int nviews = ((fuse && fuse->views > 1) || seg->views > 1) ? 2 : 1;
        if (!fuse && seg->views > 1) {
            if (seg->views != 2 || (p.M % 2) || ((p.M / 2) % kBM)) return SM3_EALIGN;
            p.fz_view_tiles = p.M / 2 / kBM;
        }
        if (seg->w_view_stride < 0 || seg->w1_view_stride < 0) return SM3_EINVAL;
        const long x1b = (long)p.M * seg->Ci1 * sz;
        const long w0b = ((long)(nviews - 1) * seg->w_view_stride + (long)(d->Co - 1) * d->w_row_stride + d->Ci) * sz;
        const long w1b = ((long)(nviews - 1) * seg->w1_view_stride + (long)d->Co * seg->Ci1) * sz;
        if (x1b >= 0xC0000000L || w0b >= 0xC0000000L || w1b >= 0xC0000000L) return SM3_EINVAL;
        p.x1 = (const char*)seg->x1;
        p.w1 = (const char*)seg->w1;
        p.x1_bytes = (uint32_t)x1b;
        p.w1_bytes = (uint32_t)w1b;
        p.w_bytes = (uint32_t)w0b;
        p.Ci1 = seg->Ci1;
        p.kchunks1 = seg->Ci1 * sz / 128;
        p.w1_row_stride = seg->Ci1;
        p.ntaps = 2;
        p.dy[1] = p.dx[1] = 0;
        p.wtap[1] = 0;
        p.tap_src[1] = 1;
        p.nsteps_seg = p.kchunks + p.kchunks1;
        p.w_view_bytes = (uint32_t)(seg->w_view_stride * sz);
        p.w1_view_bytes = (uint32_t)(seg->w1_view_stride * sz);
        p.col_bias = seg->col_bias;
    }
    hipStream_t st = (hipStream_t)stream;
    p.in_scale = p.in_shift = nullptr;
    p.in_act = nullptr;
    p.in_mask = nullptr;
    if (bnin) {
        if (addend || fuse || ebn || seg) return SM3_EINVAL;
        if (bnin->views > 1) {
            if (bnin->views != 2 || (p.M % 2) || ((p.M / 2) % kBM)) return SM3_EALIGN;
            p.fz_view_tiles = p.M / 2 / kBM;
        }
        p.in_scale = bnin->scale;
        p.in_shift = bnin->shift;
        p.in_act = (char*)bnin->act;
        p.in_mask = bnin->mask;
        if (d->dtype == SM3_BF16) return launch_bnin<bf16_t>(p, d->Co, bnin->probe, st);
        if (d->dtype == SM3_F16) return launch_bnin<f16_t>(p, d->Co, bnin->probe, st);
        return SM3_EDTYPE;
    }
    // 64-column tiles for Co <= 64, and for the small-M Linears whose 128-column grid would leave most CUs idle
    const long tiles128 = (long)((p.M + kBM - 1) / kBM) * ((d->Co + 127) / 128);
    // SM3_CONV_FORCE_NARROW=1 (experiment switch, scratch/r5_corun_regs.py): 64-column tiles everywhere -- 74-83 registers per
    // wave instead of 110-127, to measure whether a lighter gather-GEMM shares a CU better with the other lane's kernels
    const char* fnv = getenv("SM3_CONV_FORCE_NARROW");
    const bool narrow = d->Co <= 64 || (tiles128 <= 96 && p.ntaps * p.kchunks >= 6) || (fnv && atoi(fnv) == 1);
    if (d->dtype == SM3_BF16)
        return narrow ? launch_conv<bf16_t, kBM, 64, 2, 2>(p, st) : launch_conv<bf16_t, kBM, 128, 2, 2>(p, st);
    if (d->dtype == SM3_F16)
        return narrow ? launch_conv<f16_t, kBM, 64, 2, 2>(p, st) : launch_conv<f16_t, kBM, 128, 2, 2>(p, st);
    return narrow ? launch_conv<float, kBM, 64, 2, 2>(p, st) : launch_conv<float, kBM, 128, 2, 2>(p, st);
}

extern "C" int sm3_conv_gather_gemm(const sm3_conv_desc* d, const void* x, const void* w, void* y,
                                    const void* addend, float* stat_partials, void* stream) {
    return conv_gather_gemm_impl(d, x, w, y, addend, stat_partials, nullptr, stream);
}

extern "C" int sm3_conv_dgrad_bnfuse(const sm3_conv_desc* d, const void* dy_in, const void* w_dgrad, void* dz_out,
                                     const void* addend, const sm3_bn_bwd_fuse* fuse, void* stream) {
    if (!fuse) return SM3_EINVAL;
    return conv_gather_gemm_impl(d, dy_in, w_dgrad, dz_out, addend, nullptr, fuse, stream);
}

extern "C" int sm3_conv_dgrad_seg_bnfuse(const sm3_conv_desc* d, const void* x0, const void* w0, const sm3_conv_seg* seg,
                                         void* dz_out, const void* addend, const sm3_bn_bwd_fuse* fuse, void* stream) {
    if (!seg) return SM3_EINVAL;
    return conv_gather_gemm_impl(d, x0, w0, dz_out, addend, nullptr, fuse, stream, nullptr, seg);
}

extern "C" int sm3_conv_gather_gemm_seg(const sm3_conv_desc* d, const void* x0, const void* w0, const sm3_conv_seg* seg,
                                        void* y, const void* addend, void* stream) {
    if (!seg) return SM3_EINVAL;
    return conv_gather_gemm_impl(d, x0, w0, y, addend, nullptr, nullptr, stream, nullptr, seg);
}

// conv2 of a Bottleneck reading conv1's RAW output: y = conv3x3(relu(x_raw * in_scale[v] + in_shift[v])) with the activation
// and its ReLU bits written on the side (kVarHaloBn).  Only for the launches the halo-resident kernel takes
// (sm3_conv3x3_bnin_ok: stride-1 3 x 3, 16-bit, more than 256 workgroups, the A image within a quarter of a CU's LDS);
// everything else returns SM3_EINVAL and the caller keeps the two-pass form (sm3_bn_act, then sm3_conv_gather_gemm).
extern "C" int sm3_conv3x3_bnin_ok(const sm3_conv_desc* d, int views) {
    const BnIn b{nullptr, nullptr, views, nullptr, nullptr, true};
    return conv_gather_gemm_impl(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &b) == 0 ? 1 : 0;
}
extern "C" int sm3_conv3x3_bnin(const sm3_conv_desc* d, const void* x_raw, const float* in_scale, const float* in_shift,
                                int views, void* act_out, uint8_t* mask_out, const void* w, void* y, float* stat_partials,
                                void* stream) {
    if (!in_scale || !in_shift || !act_out || !mask_out || views < 1) return SM3_EINVAL;
    const BnIn b{in_scale, in_shift, views, act_out, mask_out, false};
    return conv_gather_gemm_impl(d, x_raw, w, y, nullptr, stat_partials, nullptr, stream, nullptr, nullptr, &b);
}

#ifdef SM3_STAMP
extern "C" int sm3_debug_set_stamps(void* buf, long capacity_waves) {
    unsigned long long* b = (unsigned long long*)buf;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &b, sizeof(b)) != hipSuccess) return SM3_EINVAL;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_cap), &capacity_waves, sizeof(long)) != hipSuccess) return SM3_EINVAL;
    return 0;
}
#endif

extern "C" int sm3_conv_seg_act(const sm3_conv_desc* d, const void* x0, const void* w0, const sm3_conv_seg* seg, int relu,
                                void* y, uint8_t* relu_mask, void* stream) {
    if (!seg || (relu_mask && !relu)) return SM3_EINVAL;
    EvalBn e{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, relu};
    e.mask = relu_mask;
    e.views = seg->views > 1 ? seg->views : 1;
    return conv_gather_gemm_impl(d, x0, w0, y, nullptr, nullptr, nullptr, stream, &e, seg);
}

extern "C" int sm3_conv_bn_act_eval(const sm3_conv_desc* d, const void* x, const void* w, const float* scale,
                                    const float* shift, const void* residual, int relu, void* y, void* stream) {
    if (!scale || !shift) return SM3_EINVAL;
    const EvalBn e{scale, shift, nullptr, nullptr, nullptr, nullptr, 0.f, relu};
    return conv_gather_gemm_impl(d, x, w, y, residual, nullptr, nullptr, stream, &e);
}

extern "C" int sm3_conv_bn_act_fused(const sm3_conv_desc* d, const void* x, const void* w, const float* scale,
                                     const float* shift, const void* residual, int relu, void* y, uint8_t* relu_mask,
                                     int views, void* stream) {
    if (!scale || !shift || views < 1 || (relu_mask && !relu)) return SM3_EINVAL;
    if (d && !(d->osy == 1 && d->osx == 1 && d->ooy == 0 && d->oox == 0 && d->Hout == d->Ho && d->Wout == d->Wo))
        return SM3_EINVAL;  // the mask is indexed like a dense output
    EvalBn e{scale, shift, nullptr, nullptr, nullptr, nullptr, 0.f, relu};
    e.mask = relu_mask;
    e.views = views;
    return conv_gather_gemm_impl(d, x, w, y, residual, nullptr, nullptr, stream, &e);
}

extern "C" int sm3_conv_bn_eval(const sm3_conv_desc* d, const void* x, const void* w, const float* gamma,
                                const float* beta, const float* running_mean, const float* running_var, float eps,
                                const void* residual, int relu, void* y, void* stream) {
    if (!running_mean || !running_var || !(eps >= 0.f)) return SM3_EINVAL;
    const EvalBn e{nullptr, nullptr, gamma, beta, running_mean, running_var, eps, relu};
    return conv_gather_gemm_impl(d, x, w, y, residual, nullptr, nullptr, stream, &e);
}
