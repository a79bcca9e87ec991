/*
 * sm3_hip.h -- C ABI of libsm3hip.so: the MI355X (gfx950) kernels of the SM3 pre-training hot path.
 *
 * The reference (Dylan-H-Wang/skin-sm3) is pure Python and has no FFI of its own: every entry point
 * below replaces the ATen/cuDNN/cuBLAS work behind one call site of the reference, cited as
 * file:line under /root/reference.  The library has no PyTorch types in its signatures: device
 * pointers, sizes, a hipStream_t passed as void*.  All tensors are dense; activations are NHWC
 * ("rows x channels", channels contiguous), weights are [Cout][taps][Cin] (= the memory order of a
 * torch channels_last OIHW tensor).  `dtype` selects the storage/MFMA type of activations and
 * weight copies: SM3_F32 (exact-f32 MFMA, parity mode), SM3_BF16 or SM3_F16 (16-bit MFMA, fp32 accumulate; SM3_F16
 * needs the caller's loss scaling: sm3_loss_scale_update).
 * Statistics, master weights, gradients of weights and optimizer state are always fp32 (BN sums fp64).
 *
 * Every function returns 0 on success, a positive hipError_t on a runtime failure, or a negative
 * SM3_E* code on a rejected argument; nothing is launched when an argument is rejected.
 * Nothing here allocates, frees or synchronises: safe to capture into a hipGraph.
 */
#ifndef SM3_HIP_H
#define SM3_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SM3_F32 0
#define SM3_BF16 1
#define SM3_F16 2   /* IEEE half storage + f16 MFMA, fp32 accumulate: the reference's AMP type (backbone_train.py:27,98) */

#define SM3_EINVAL (-1)   /* bad size / null pointer */
#define SM3_EALIGN (-2)   /* channel count not a multiple of the kernel's K chunk */
#define SM3_EDTYPE (-3)

#define SM3_MAX_TAPS 9

/* ABI version, bumped on any signature change. */
int sm3_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Gather-GEMM convolution: forward conv, data-gradient of a conv, and bias-free Linear.
 *   replaces nn.Conv2d forward/backward-data  (src/models/resnet.py:49-67 conv3x3/conv1x1, used at
 *   :144-148,:260; the 7x7 stem :208-210 runs on sm3_stem_conv_fwd straight from the images in all three arithmetic
 *   modes -- sm3_stem_im2col + this GEMM remain as the SM3_DIRECT_STEM=0 A/B path) and nn.Linear(bias=False)
 *   (src/models/simclr.py:17-27).
 *
 *   y[n, oy*osy+ooy, ox*osx+oox, co] = sum_t sum_ci x[n, oy*sy+dy[t], ox*sx+dx[t], ci]
 *                                                 * w[co*w_row_stride + wtap[t]*Ci + ci]   (+ addend)
 *   for (n,oy,ox) in N x Ho x Wo; out-of-range x reads are zero.  Ci must be a multiple of
 *   128 bytes / sizeof(T) (64 bf16 / 32 f32).
 * ------------------------------------------------------------------------------------------ */
typedef struct sm3_conv_desc {
    int32_t dtype;
    int32_t N, Hi, Wi, Ci;          /* x: [N,Hi,Wi,Ci] */
    int32_t Ho, Wo, Co;             /* iteration space and GEMM-N */
    int32_t sy, sx;
    int32_t ntaps;
    int32_t dy[SM3_MAX_TAPS], dx[SM3_MAX_TAPS], wtap[SM3_MAX_TAPS];
    int32_t w_row_stride;           /* elements between rows (co) of w */
    int32_t Hout, Wout;             /* y: [N,Hout,Wout,Co] */
    int32_t osy, osx, ooy, oox;
} sm3_conv_desc;

/* number of [2][Co] fp32 partial-statistics rows sm3_conv_gather_gemm writes for this desc */
int sm3_conv_partial_rows(const sm3_conv_desc* d);

/* stat_partials (nullable): [partial_rows][2][Co] fp32 -- per-row-block sum and sum of squares of
 * the stored (rounded) outputs, for train-mode BatchNorm (resnet.py:145-149,211,261).
 * addend (nullable): same indexing as y, added in the epilogue (may alias y). */
int sm3_conv_gather_gemm(const sm3_conv_desc* d, const void* x, const void* w, void* y,
                         const void* addend, float* stat_partials, void* stream);

/* Data-gradient launch with the FIRST phase of the next BatchNorm backward fused into its epilogue.  The
 * gather-GEMM result (+addend) is dy of a BatchNorm output y = relu?(bn(x)); instead of storing dy this stores
 * dz = dy * (y > 0) (relu_mask NULL: no ReLU) and writes per-row-block partial sums of (dz, dz * xhat) for the
 * channels of y -- exactly what sm3_bn_bwd_reduce would produce from a second pass over dy, y and x.
 * partials: [partial_row_offset + sm3_conv_partial_rows(d)][2][Co]; parity launches of a stride-2 gradient use
 * consecutive row offsets.  x / relu_mask are indexed like dz_out. */
typedef struct sm3_bn_bwd_fuse {
    const uint8_t* relu_mask;
    const void* x;
    const float* mean;           /* [views][Co] */
    const float* invstd;         /* [views][Co] */
    float* partials;
    int32_t partial_row_offset;  /* first partial row of view 0's tiles */
    int32_t views;               /* 0 or 1: one BatchNorm batch.  2: the launch covers two views back to back (rows
                                  * [0, M/2) and [M/2, M), M/2 a multiple of 128), each with its own mean / invstd ... */
    int32_t partial_row_offset_view1; /* ... and view 1's tiles write their partial rows from here */
    int32_t addend_sp_h, addend_sp_w; /* > 0: `addend` is a COMPACT [N, addend_sp_h, addend_sp_w, Co] tensor holding values for
                                  * the even (y, x) output positions only (= (Ho+1)/2 x (Wo+1)/2) and zero elsewhere: the data
                                  * gradient of a Bottleneck's stride-2 1x1 downsample convolution (resnet.py:260), joined with
                                  * conv1's data gradient here instead of by a second pass over the block-input gradient */
} sm3_bn_bwd_fuse;
int sm3_conv_dgrad_bnfuse(const sm3_conv_desc* d, const void* dy_in, const void* w_dgrad, void* dz_out,
                          const void* addend, const sm3_bn_bwd_fuse* fuse, void* stream);
/* fuse->x == NULL (with mean / invstd then unused): the epilogue applies the ReLU mask and sums dz only -- the
 * sum(dz * xhat) slot of every partial row is written as 0.  For a producer BatchNorm whose backward goes through
 * sm3_linbn_stats, which derives that sum from the weight-gradient product instead of a read of x. */

/* The same launch over TWO K segments: dz_out = mask( x0 w0^T + x1 w1^T + col_bias (+ addend) ), for a 1x1 / stride-1
 * descriptor d (x0: [pixels][d->Ci], w0: [d->Co][d->w_row_stride]) and a second operand pair over the same pixels.
 * With x0 = dz of an expanding conv's BatchNorm, w0 = diag(a) W, x1 = that conv's input, w1 = -H, col_bias = const
 * (sm3_linbn_coeffs, sm3_conv_gather_gemm) this is the data gradient of conv -> BatchNorm with the BatchNorm-backward
 * apply pass folded into the GEMM by linearity (see "BatchNorm backward by linearity" below).  16-bit dtypes only.
 * Two views in one launch (fuse->views == 2): view 1's tiles read w0 + w_view_stride, w1 + w1_view_stride (elements)
 * and col_bias + Co. */
typedef struct sm3_conv_seg {
    const void* x1;            /* [pixels][Ci1] */
    const void* w1;            /* [views][Co][Ci1] */
    int32_t Ci1;               /* multiple of 64 */
    int64_t w_view_stride;     /* elements between the views' w0 banks (0: shared) */
    int64_t w1_view_stride;
    const float* col_bias;     /* [views][Co], nullable */
    int32_t views;             /* sm3_conv_gather_gemm_seg only (0 / 1: one view; 2: two views back to back, each a multiple
                                * of 128 rows); sm3_conv_dgrad_seg_bnfuse takes the views from its fuse argument */
} sm3_conv_seg;
int sm3_conv_dgrad_seg_bnfuse(const sm3_conv_desc* d, const void* x0, const void* w0, const sm3_conv_seg* seg,
                              void* dz_out, const void* addend, const sm3_bn_bwd_fuse* fuse, void* stream);
/* The two-segment product alone, y = x0 w0^T + x1 w1^T + col_bias (+ addend): the data gradient of an expanding conv ->
 * BatchNorm pair whose result has no producer BatchNorm to prepare (a Bottleneck's downsample branch: the compact
 * gradient that the block's conv1 data gradient then takes as its sparse addend). */
int sm3_conv_gather_gemm_seg(const sm3_conv_desc* d, const void* x0, const void* w0, const sm3_conv_seg* seg, void* y,
                             const void* addend, void* stream);
/* ... and with a ReLU (+ its bits) on the sum: y = relu?(x0 w0^T + x1 w1^T + col_bias).  With the two BatchNorm scales
 * folded into the banks and the shifts into col_bias (sm3_linbn_scale_banks) this is the WHOLE join of a Bottleneck that has
 * a downsample branch -- conv3, bn3, the 1x1 downsample convolution, its BatchNorm, the add and the ReLU
 * (resnet.py:162-172) -- in one launch, neither pre-BatchNorm tensor stored. */
int sm3_conv_seg_act(const sm3_conv_desc* d, const void* x0, const void* w0, const sm3_conv_seg* seg, int relu, void* y,
                     uint8_t* relu_mask, void* stream);

/* Inference: conv + eval-mode BatchNorm (+ residual) (+ ReLU) in one launch,
 *   y = relu?( conv(x, w) * scale[co] + shift[co] (+ residual) ),
 * scale/shift from sm3_bn_eval_scale_shift.  Replaces the conv->bn->(add)->relu chains of Bottleneck.forward
 * (src/models/resnet.py:154-174) under model.eval(): frozen-encoder feature extraction (simclr.py:393-396,
 * tools/backbone_eval.py --finetune fc, inference.py) never writes or re-reads a pre-BN tensor. */
int sm3_conv_bn_act_eval(const sm3_conv_desc* d, const void* x, const void* w, const float* scale,
                         const float* shift, const void* residual, int relu, void* y, void* stream);
/* The train-mode form: scale / shift are [views][Co] (a launch over two views back to back, each a multiple of 128 rows,
 * uses view v's vectors for view v's rows) and relu_mask (nullable, with relu) receives the ReLU bits of y as sm3_bn_act
 * writes them.  With scale / shift from sm3_linbn_fwd_stats + sm3_bn_finalize this is conv3 -> bn3 -> (+identity) -> ReLU
 * of a Bottleneck (resnet.py:162-172) in ONE launch: the pre-BatchNorm tensor is never written.  Dense outputs only. */
int sm3_conv_bn_act_fused(const sm3_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                          const void* residual, int relu, void* y, uint8_t* relu_mask, int views, void* stream);
/* Same launch with scale/shift derived in the epilogue from the BatchNorm's own tensors,
 *   scale = gamma / sqrt(running_var + eps),  shift = beta - running_mean * scale      (gamma/beta NULL: 1 / 0),
 * bit-identical to sm3_bn_eval_scale_shift followed by sm3_conv_bn_act_eval, without the 53 small launches per encoder
 * pass that computing the vectors first costs (nn.BatchNorm2d.forward in eval mode, src/models/resnet.py:156-169). */
int sm3_conv_bn_eval(const sm3_conv_desc* d, const void* x, const void* w, const float* gamma, const float* beta,
                     const float* running_mean, const float* running_var, float eps, const void* residual, int relu,
                     void* y, void* stream);

/* Weight gradient of the forward conv described by d (autograd of the same call sites):
 *   dw[co*w_row_stride + wtap[t]*Ci + ci] += sum_{n,oy,ox} dy[(n,oy,ox), co] * x[n, oy*sy+dy[t], ox*sx+dx[t], ci]
 * dy is dense [N*Ho*Wo, Co]; dw is fp32 and is accumulated into (float atomics, split over pixels: run-to-run
 * differences of 1 ulp; sm3_conv_wgrad_det below is the fixed-order form).
 * Columns wtap[t]*Ci+ci >= w_row_stride are dropped (the zero-padded K tail of the stem im2col). */
int sm3_conv_wgrad(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, void* stream);
/* The same product over `views` equal pixel ranges that accumulate into dw + v * dw_view_stride (per-view weight-gradient
 * products: P_v = dz_v^T x_v, or with dy = x the Gram matrix G_v = x_v^T x_v), and optionally with dY the channel
 * concatenation [dy (d->Co) | dy1 (Co1)] of two tensors over the same pixels (rows of the second part accumulate into
 * dw1 + v * dw1_view_stride; d->Co then a multiple of 128).  Co1 = 0: dy1 / dw1 unused. */
int sm3_conv_wgrad_cat(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, const void* dy1, int Co1,
                       float* dw1, int views, int64_t dw_view_stride, int64_t dw1_view_stride, void* stream);
/* The same product WITHOUT atomics (plain-store split-K): slice j of view v of the pixel axis stores its whole
 * [Co][taps * Ci] partial product into slabs + (v * *slabs_used + j) * Co * taps * Ci; the caller adds the slabs up in a
 * fixed order (sm3_linbn_moments, sm3_linbn_stats / sm3_linbn_post): deterministic, and at these sizes faster than
 * ~1.3 TB/s of float atomics.  *slabs_used <= slab_capacity slabs per view (host int, written before return); a view is cut
 * the same way whether the launch holds one view or two. */
int sm3_conv_wgrad_slabs(const sm3_conv_desc* d, const void* x, const void* dy, float* slabs, int slab_capacity, int views,
                         int* slabs_used, void* stream);
/* out[e] = (accumulate ? out[e] : 0) + sum_j slabs[j * n + e], j ascending on a fixed tree that depends on (nslabs, e)
 * only; n a multiple of 4, both pointers 16-byte aligned.  The fixed-order sum behind every split-K product of the
 * training step (ABI 8). */
int sm3_slab_reduce(const float* slabs, int nslabs, int64_t n, float* out, int accumulate, void* stream);
/* sm3_conv_wgrad as a FUNCTION OF ITS INPUTS (ABI 8; what the training step uses): the pixel slices store plain slabs
 * into `slabs` (room for slab_capacity matrices [Co][taps * Ci] fp32; d->w_row_stride == taps * Ci) and one
 * sm3_slab_reduce adds them to dw.  The partition depends on the geometry and the device only, so two runs of the same
 * step produce the same bits -- the autograd backward of nn.Conv2d / nn.Linear it replaces (tools/backbone_train.py:125)
 * is a fixed-order reduction too.  A launch that needs a single slice adds its tiles to dw directly (no slab). */
int sm3_conv_wgrad_det(const sm3_conv_desc* d, const void* x, const void* dy, float* dw, float* slabs, int slab_capacity,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (2d and 1d: rows x C), train and eval.  replaces nn.BatchNorm2d/1d (+SyncBatchNorm,
 * tools/backbone_train.py:510) at resnet.py:145-149,211,261 and simclr.py:20-26.
 * ------------------------------------------------------------------------------------------ */
/* sums[0..C) = sum over partial rows of p[r][0][c]; sums[C..2C) likewise of p[r][1][c]  (fp64, deterministic
 * two-stage reduction).  workspace: SM3_BN_REDUCE_GROUPS * 2C doubles. */
#define SM3_BN_REDUCE_GROUPS 64
/* number of fp64 [2C] rows stage A leaves in workspace for `rows` partial rows */
int sm3_bn_reduce_groups(int rows);
/* "views": the two views of a branch go through the convolutions as ONE batch (view 0's rows, then view 1's) but
 * keep separate BatchNorm statistics (simclr.py:58-59).  Every BatchNorm entry point takes `views` >= 1 and treats
 * its tensors as `views` equal row ranges laid out back to back, with per-view parameter vectors [views][C] /
 * [views][2C]; `rows` is always the row count of ONE view.
 * sums == NULL: only stage A runs; pass (workspace, sm3_bn_reduce_groups(rows)) to sm3_bn_finalize.
 * partials: [views][rows][2][C]; sums: [views][2C]; workspace: views * SM3_BN_REDUCE_GROUPS * 2C doubles. */
int sm3_bn_stats_reduce(const float* partials, int rows, int C, double* sums, double* workspace, int views,
                        void* stream);
/* From (possibly all-reduced) sums and the global element count per channel: mean, biased var ->
 * scale = gamma*invstd, shift = beta - mean*scale; running stats momentum update with the unbiased
 * variance; saves mean / invstd for backward.  gamma/beta NULL => affine=False. */
int sm3_bn_finalize(const double* sums, int groups /* sums is [views][groups][2C], groups summed here */, int views,
                    double count /* per view */, int C, const float* gamma, const float* beta,
                    float eps, float momentum, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked /* += views */, float* scale, float* shift, float* save_mean,
                    float* save_invstd /* all four [views][C]; running statistics updated view by view */, void* stream);
/* sm3_bn_stats_reduce (stage A) + sm3_bn_finalize in ONE launch, for a single rank (no exchange between the two): replaces
 * the same nn.BatchNorm2d / BatchNorm1d train-mode statistics (src/models/resnet.py:145-149, src/models/simclr.py:20-26) with one
 * dependent launch fewer per BatchNorm.  The block that draws the last arrival ticket of a 32-channel block finalizes it: same
 * association, same bits as the two-launch form.  workspace: views * SM3_BN_REDUCE_GROUPS * 2C doubles; tickets: (C + 31) / 32
 * uint32, ZERO before the first use and left zero by every launch; both private to the stream (launches on one stream reuse them
 * in order). */
int sm3_bn_stats_finalize(const float* partials, int rows, int C, int views, double* workspace, uint32_t* tickets, double count,
                          const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, float* scale, float* shift, float* save_mean,
                          float* save_invstd, void* stream);
/* eval mode: scale/shift from the running statistics */
int sm3_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, int C, float* scale, float* shift,
                            void* stream);
/* y = [relu]( x*scale + shift [+ residual] ), x,residual,y: [rows, C] of dtype; out_f32 != 0 stores y as fp32.
 * relu_mask (nullable, with relu): one byte per 16-byte vector of y, bit e = (y[e] > 0) -- all that the backward
 * pass needs of y, at 1/16 of its bytes.
 * replaces the bn->relu / bn->add->relu chains of Bottleneck.forward (resnet.py:154-174). */
int sm3_bn_act(int dtype, const void* x, const float* scale, const float* shift, const void* residual,
               int relu, int out_f32, void* y, uint8_t* relu_mask, int64_t rows, int C, int views, void* stream);
/* sm3_bn_act (storage dtype output) that also leaves per-block column sums of the STORED outputs:
 * colsum_partials [views][sm3_bn_act_colsum_rows(rows, C, dtype)][C] fp32 -- the first moment of a convolution input, for
 * the BatchNorm by linearity below (sm3_linbn_moments sums the rows in a fixed order).  A view's rows are cut the same way
 * whether the launch holds one view or two, so its sums do not depend on that. */
int sm3_bn_act_colsum_rows(int64_t rows, int C, int dtype);
int sm3_bn_act_colsum(int dtype, const void* x, const float* scale, const float* shift, const void* residual, int relu,
                      void* y, uint8_t* relu_mask, float* colsum_partials, int64_t rows, int C, int views, void* stream);
/* Column sums of x [N, H, W, C] at the pixels (y % stride == 0, x % stride == 0) as partial rows
 * [views][sm3_subsample_colsum_rows(N / views * Hs * Ws, C, dtype)][C] fp32 (summed by sm3_linbn_moments), and -- y given --
 * those pixels as a compact tensor y [N, Hs, Ws, C], Hs = (H - 1) / stride + 1: the input of a strided 1x1 convolution
 * (a Bottleneck's downsample branch, resnet.py:260) as the dense operand that BatchNorm by linearity needs. */
int sm3_subsample_colsum_rows(int64_t rows, int C, int dtype);
int sm3_subsample_colsum(int dtype, const void* x, void* y, float* colsum_partials, int N, int H, int W, int C, int stride,
                         int views, void* stream);
/* The join of a Bottleneck with a downsample branch in ONE pass (resnet.py:164-172: out = bn3(conv3); identity =
 * downsample(x) [conv + BatchNorm]; out += identity; relu):  y = [relu]( x*scale + shift + x2*scale2 + shift2 ),
 * x2 the pre-BatchNorm output of the downsample convolution, scale2/shift2 [views][C] from its sm3_bn_finalize.
 * The downsample BatchNorm's own apply pass (one read + one write of a block-output-sized tensor) disappears. */
int sm3_bn_add_bn_act(int dtype, const void* x, const float* scale, const float* shift, const void* x2,
                      const float* scale2, const float* shift2, int relu, void* y, uint8_t* relu_mask,
                      int64_t rows, int C, int views, void* stream);
/* Backward, phase 1: dz = dy * (y > 0), the mask taken from relu_mask if given, else from y if given, else all
 * ones; writes dz (may alias dy; NULL to skip) and
 * per-block partial sums [bwd_partial_rows][2][C] of (dz, dz * xhat), xhat = (x-mean)*invstd.  x == NULL (mean / invstd
 * then unused): mask and sum(dz) only, the second slot is written as 0 (BatchNorm by linearity, below). */
int sm3_bn_bwd_partial_rows(int64_t rows, int C);
int sm3_bn_bwd_reduce(int dtype, const void* dy, const void* y, const uint8_t* relu_mask, const void* x,
                      const float* mean, const float* invstd, void* dz, int64_t rows, int C,
                      float* partials /* [views][bwd_partial_rows][2][C] */, int views, void* stream);
/* Backward, phase 2: dx = gamma*invstd*(dz - sum_dz/count - xhat*sum_dz_xhat/count) with the
 * (all-reduced) global sums; dgamma += local sum(dz*xhat), dbeta += local sum(dz) (NULL to skip). */
int sm3_bn_bwd_apply(int dtype, const void* dz, const void* x, const float* mean, const float* invstd,
                     const float* gamma, const double* global_sums, double count,
                     const double* local_sums, float* dgamma, float* dbeta, void* dx, int64_t rows,
                     int C, int views, void* stream);

/* Phase 2 for the TWO BatchNorms of such a join, which receive the same dz: reads dz once, writes both input
 * gradients (arithmetic of each side = sm3_bn_bwd_apply). */
typedef struct sm3_bn_apply_side {
    const void* x;                 /* pre-BatchNorm tensor, [views][rows][C] */
    const float* mean;             /* [views][C] */
    const float* invstd;
    const float* gamma;            /* [C], nullable */
    const double* global_sums;     /* [views][2C] */
    const double* local_sums;      /* [views][2C], nullable with dgamma/dbeta */
    float* dgamma;
    float* dbeta;
    void* dx;
} sm3_bn_apply_side;
int sm3_bn_bwd_apply2(int dtype, const void* dz, double count, const sm3_bn_apply_side* a,
                      const sm3_bn_apply_side* b, int64_t rows, int C, int views, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm BY LINEARITY, for an expanding 1x1 convolution followed by train-mode BatchNorm (Bottleneck conv3 -> bn3:
 * resnet.py:162-163 and their autograd backward; 16-bit dtypes).  With x = y W^T (y: [M, p] conv input, W: [C, p]),
 * s = sum_m y (sm3_bn_act_colsum) and G = y^T y (sm3_conv_wgrad_cat on y alone):
 *   forward   sum_m x = W s,  sum_m x^2 = rowdot(W G, W)  -> sm3_bn_finalize -> sm3_conv_bn_act_fused: x is never stored
 *   backward  dx = a (dz - m1) - b (x - mu)   [a = gamma invstd, b = a invstd mean(dz xhat), m1 = mean(dz)], so with
 *             P = dz^T y (sm3_conv_wgrad_cat):
 *               sum_m dz xhat = invstd (rowdot(W, P) - mu sum_m dz)                                (sm3_linbn_stats)
 *               dy = dz (diag(a) W) - y H + const,  H = W^T diag(b) W   (sm3_linbn_banks, sm3_linbn_post for H,
 *                                                                        sm3_conv_dgrad_seg_bnfuse)
 *               dW += diag(a)(P - m1 s^T) - diag(b)(W G - mu s^T)                                  (sm3_linbn_post)
 * so neither sm3_bn_bwd_apply's pass over dz / x / dx nor any other read of x is made.  All per-channel vectors are
 * [views][C]; coef is [views][C][4] fp32 = (a, b, m1, mu).
 * ------------------------------------------------------------------------------------------ */
/* out [views][n] fp32 = the sum of the nslabs slabs per view sm3_conv_wgrad_slabs left (n = p * p for G = y^T y, C * p for
 * P = dz^T y), and -- colsum_partials given -- s_out [views][p] fp64 = the sum of the colsum_rows partial rows of
 * sm3_bn_act_colsum; both in a fixed order. */
int sm3_linbn_moments(const float* slabs, int nslabs, int64_t n, float* out, const float* colsum_partials, int colsum_rows,
                      double* s_out, int p, int views, void* stream);
/* Tm [views][C][p] fp32 = W G_v (exact-f32 MFMA), and the batch sums of x as sums_ws [views][p/32][2C] fp64 -- pass it to
 * sm3_bn_finalize with groups = p/32 (data parallel: all-reduce it first).  G: [views][p][p] fp32; s: [views][p] fp64;
 * w_dgrad: dtype [p][C]; w_fwd: dtype [C][p].  C, p multiples of 32. */
int sm3_linbn_fwd_stats(int dtype, const float* G, const void* w_dgrad, const void* w_fwd, const double* s, float* Tm,
                        double* sums_ws, int C, int p, int views, void* stream);
/* reduce_ws / groups: what sm3_bn_stats_reduce(partials of (dz, .), sums = NULL) left -- its stage B runs here.
 * lsums: [views][2C] fp64, both halves written (sum dz | sum dz xhat); dgamma += sum dz xhat, dbeta += sum dz (NULL to
 * skip).  count > 0 (single rank: the local sums are the global ones): coef is written too; count <= 0: the caller
 * all-reduces lsums and calls sm3_linbn_coef.  P: [views][C][p] fp32; w_fwd: dtype [C][p]. */
int sm3_linbn_stats(int dtype, const float* P, const void* w_fwd, const float* mean, const float* invstd, const float* gamma,
                    const double* reduce_ws, int groups, double* lsums, float* dgamma, float* dbeta, double count,
                    float* coef, int C, int p, int views, void* stream);
int sm3_linbn_coef(const double* global_sums, double count, const float* gamma, const float* mean, const float* invstd,
                   float* coef, int C, int views, void* stream);
/* wa = diag(a) W and wbn = -diag(b) W in data-gradient order (dtype [views][p][C], from w_dgrad [p][C]);
 * col_const [views][p] = (b mu - a m1) W, summed over the rounded products. */
int sm3_linbn_banks(int dtype, const void* w_dgrad, const float* coef, void* wa, void* wbn, float* col_const, int C, int p,
                    int views, void* stream);
/* out[v][e] = sum over the `groups` partial rows of sums_ws [views][groups][n] (n = 2C), fixed order: what a data-parallel
 * run exchanges between ranks (then sm3_bn_finalize with groups = 1). */
int sm3_linbn_fold(const double* sums_ws, int groups, int n, int views, double* out, void* stream);
/* out3[v][c][:] = scale3[v][c] w3[c][:] ([C][K3] banks), outd[v][c][:] = scaled[v][c] wd[c][:] ([C][Kd]), and
 * bias[v][c] = shift3[v][c] + shiftd[v][c]: what sm3_conv_seg_act needs to run conv3 + bn3 + downsample conv + its BatchNorm
 * + add + ReLU as one two-segment GEMM. */
int sm3_linbn_scale_banks(int dtype, const void* w3, int K3, const float* scale3, const float* shift3, void* out3,
                          const void* wd, int Kd, const float* scaled, const float* shiftd, void* outd, float* bias, int C,
                          int views, void* stream);
/* One launch of 32 x 32 MFMA tiles:  hn [views][p][p] (dtype) = wbn_v w_dgrad^T = -H_v, and
 * dw[C][p] += sum_v diag(a_v)(P_v - m1_v s_v^T) - diag(b_v)(W G_v - mu_v s_v^T), W G_v from Tm (sm3_linbn_fwd_stats) or,
 * Tm NULL, recomputed from G.  C % 128 == 0, p % 32 == 0. */
int sm3_linbn_post(int dtype, const void* wbn, const void* w_dgrad, void* hn, const float* P, const float* G,
                   const float* Tm, const double* s, const float* coef, float* dw, int C, int p, int views, void* stream);
/* sm3_linbn_banks + sm3_linbn_post as ONE launch (the links of the chain between a unit's backward GEMMs are launch
 * latency): wa, col_const, hn and dw as those two produce them, bit for bit; the -diag(b) W bank is formed in registers
 * (same product, same rounding) and never stored. */
int sm3_linbn_banks_post(int dtype, const void* w_dgrad, const float* coef, void* wa, float* col_const, void* hn,
                         const float* P, const float* G, const float* Tm, const double* s, float* dw, int C, int p, int views,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Stem, pooling.  replaces resnet.py:208-213,224,294-305.
 * ------------------------------------------------------------------------------------------ */
/* 7x7/2 pad-3 im2col of an NCHW fp32 image batch into rows [N*Ho*Wo, Kpad] of dtype,
 * k = (kh*7+kw)*3 + c for k < 147, zero for 147 <= k < Kpad. */
int sm3_stem_im2col(int dtype, const float* x_nchw, void* cols, int N, int H, int W, int Kpad, void* stream);
/* Direct stem (bf16 / fp16 on the 16-bit MFMA; SM3_F32 on v_mfma_f32_32x32x2_f32, fp32 patch fragments and filter bank):
 * the 7x7/2 pad-3 convolution straight from the NCHW fp32 images, no im2col matrix in HBM.
 *   w_stem: dtype [64][176] from sm3_stem_weight_prep (K order (kh, c, kw padded to 8); master is [64][kh][kw][c]);
 *   y: [N*Ho*Wo, 64] dtype; stat_partials (nullable): [sm3_stem_partial_rows][2][64], one row per tile of <= 128
 *   output pixels of one output row -- tiles are image-major, so a view's rows are contiguous. */
int sm3_stem_partial_rows(int N, int H, int W);
int sm3_stem_weight_prep(int dtype, const float* w_master, void* w_stem, void* stream);
int sm3_stem_weight_prep_if(int dtype, const float* w_master, void* w_stem, const int* only_if, void* stream);
int sm3_stem_conv_fwd(int dtype, const float* x_nchw, const void* w_stem, void* y, float* stat_partials, int N, int H,
                      int W, void* stream);
/* Stem weight gradient with phase 2 of bn1's backward fused into its operand load:
 *   dxo = gamma*invstd*(dz - sum_dz/count - xhat*sum_dz_xhat/count)   (never written to HBM: the stem has no data
 *   gradient, so this tensor has no other consumer),  dw[64][147] += dxo^T * im2col(x),
 *   dgamma += local sum(dz*xhat), dbeta += local sum(dz).  Arguments as sm3_bn_bwd_apply; dz, xo: [N*Ho*Wo, 64].
 *   dw_slabs (ABI 8; nullable): room for SM3_STEM_WGRAD_SLABS matrices [64][147] fp32 -- every workgroup of the persistent
 *   grid stores its partial product there and one sm3_slab_reduce adds them to dw in a fixed order (what the training step
 *   uses); NULL: float atomics straight into dw. */
#define SM3_STEM_WGRAD_SLABS 768
int sm3_stem_wgrad_bn(int dtype, const float* x_nchw, const void* dz, const void* xo, const float* mean,
                      const float* invstd, const float* gamma, const double* global_sums, double count,
                      const double* local_sums, float* dgamma, float* dbeta, float* dw, float* dw_slabs, int N, int H, int W,
                      int views, void* stream);
/* The same two kernels on images rounded to the 16-bit type ONCE per step (ABI 8; what the training step uses in the
 * 16-bit modes; every output bit equals sm3_stem_conv_fwd / sm3_stem_wgrad_bn on the fp32 images -- the rounding is the
 * same, it only happens once instead of per staged tile, forward and again in the weight gradient):
 *   sm3_stem_image_prep: ximg [views * n_per_view][3][H][Wp] dtype, Wp = sm3_stem_image_cols(W) = round_up(W + 6, 8),
 *     ximg[.., q] = x[.., q - 3] for 3 <= q < W + 3, else 0 (the convolution's zero padding materialised); the views of a
 *     branch come from two NCHW fp32 tensors (x_view1 NULL with views = 1) -- no concatenated copy of the images is made.
 *   sm3_stem_conv_fwd16 / sm3_stem_wgrad_bn16: arguments as the fp32-image forms with ximg in place of x_nchw; the staged
 *     rows are aligned 16-byte chunks of ximg moved by LDS-DMA one tile ahead.  resnet.py:208-213,294-295. */
int sm3_stem_image_cols(int W);
int sm3_stem_image_prep(int dtype, const float* x_view0, const float* x_view1, void* ximg, int n_per_view, int views, int H,
                        int W, void* stream);
int sm3_stem_conv_fwd16(int dtype, const void* ximg, const void* w_stem, void* y, float* stat_partials, int N, int H, int W,
                        void* stream);
int sm3_stem_wgrad_bn16(int dtype, const void* ximg, const void* dz, const void* xo, const float* mean, const float* invstd,
                        const float* gamma, const double* global_sums, double count, const double* local_sums,
                        float* dgamma, float* dbeta, float* dw, float* dw_slabs, int N, int H, int W, int views,
                        void* stream);
/* argmax (nullable): [N,Ho,Wo,C] bytes, window position kh*3+kw of the first maximum in scan order (ATen's tie rule) */
int sm3_maxpool3x3s2_fwd(int dtype, const void* x, void* y, uint8_t* argmax, int N, int H, int W, int C, void* stream);
/* dx[n,iy,ix,c] = sum of dy over the windows whose recorded argmax is (iy,ix); gather form, no atomics */
int sm3_maxpool3x3s2_bwd(int dtype, const uint8_t* argmax, const void* dy, void* dx, int N, int H, int W, int C,
                         void* stream);
/* Stem chain bn1 -> relu -> maxpool (resnet.py:295-297) in ONE pass over the pre-BatchNorm stem output x [N,H,W,C]:
 * y[N,Ho,Wo,C] = maxpool3x3s2(relu(x*scale + shift)), argmax as above; scale/shift [views][C] (images [0,N/views) are
 * view 0).  Bit-identical to sm3_bn_act followed by sm3_maxpool3x3s2_fwd; the post-ReLU map is never stored. */
int sm3_bn_relu_maxpool_fwd(int dtype, const void* x, const float* scale, const float* shift, void* y,
                            uint8_t* argmax, int N, int H, int W, int C, int views, void* stream);
/* ... and its backward up to BatchNorm-backward phase 1: dz[N,H,W,C] = (x*scale+shift > 0) * maxpool_bwd(dy), plus
 * partial sums [views][sm3_maxpool_bn_bwd_partial_rows][2][C] of (dz, dz*xhat) as sm3_bn_bwd_reduce writes them. */
int sm3_maxpool_bn_bwd_partial_rows(int N, int H, int W, int views);
int sm3_maxpool_bn_bwd(int dtype, const uint8_t* argmax, const void* dy, const void* x, const float* scale,
                       const float* shift, const float* mean, const float* invstd, void* dz, float* partials,
                       int N, int H, int W, int C, int views, void* stream);
/* feat[n,c] = mean over HW; feat_f32 and feat_t (dtype copy for the projector GEMM) both optional */
int sm3_avgpool_fwd(int dtype, const void* x, float* feat_f32, void* feat_t, int N, int HW, int C, void* stream);
int sm3_avgpool_bwd(int dtype, const void* dfeat, void* dx, int N, int HW, int C, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight layout preparation (per optimizer step).  w: fp32 master [Co][taps][Ci].
 *   w_fwd   (nullable): dtype [Co][ld_fwd]  (ld_fwd >= taps*Ci, tail zero-filled)
 *   w_dgrad (nullable): dtype [Ci][taps][Co]   (the transposed filter bank used by the data gradient)
 * ------------------------------------------------------------------------------------------ */
int sm3_weight_prep(int dtype, const float* w, int Co, int taps, int Ci, void* w_fwd, int ld_fwd,
                    void* w_dgrad, void* stream);
/* the same for n filter banks in ONE launch; items_device: device array of n descriptors */
typedef struct sm3_wprep_item {
    const float* w;
    void* w_fwd;
    void* w_dgrad;
    int32_t Co, taps, Ci, ld_fwd;
} sm3_wprep_item;
int sm3_weight_prep_batch(int dtype, const sm3_wprep_item* items_device, int n, void* stream);
/* Frozen weights keep their banks (frozen-encoder loops: tools/backbone_eval.py --finetune fc, tools/mlc_train.py,
 * inference.py) without the host ever reading device memory or trusting a framework's dirty bits:
 *   sm3_weights_changed: position-weighted 64-bit hash of the n fp32 words at `flat`; *changed = (hash != state[1]);
 *                        state[1] = hash.  state: 2 x uint64 on the device, zero-initialised by the caller once.
 *   sm3_weight_prep_batch_if / sm3_stem_weight_prep_if: the launches above, whose workgroups return at once when
 *                        *only_if == 0 (only_if NULL = unconditional).  Two launches + early-exit kernels: ~0.03 ms
 *                        instead of the 0.2 ms re-layout of 47 M weights. */
int sm3_weights_changed(const float* flat, int64_t n, uint64_t* state, int* changed, void* stream);
int sm3_weight_prep_batch_if(int dtype, const sm3_wprep_item* items_device, int n, const int* only_if, void* stream);
/* elementwise cast fp32 -> dtype */
int sm3_cast_from_f32(int dtype, const float* src, void* dst, int64_t n, void* stream);
int sm3_cast_to_f32(int dtype, const void* src, float* dst, int64_t n, void* stream);

/* conv2 of a Bottleneck reading conv1's RAW output (resnet.py:144-150: conv1 -> bn1 -> relu -> conv2): ONE launch computes
 *   act = relu(x_raw * in_scale[v] + in_shift[v])  (bn1's train-mode apply + ReLU, the arithmetic of sm3_bn_act),
 *   y = conv3x3(act) + BatchNorm partial sums of y (as sm3_conv_gather_gemm),
 * writing act (same layout as x_raw) and its ReLU bits (1 byte per 16-byte vector, as sm3_bn_act's mask) on the side: the
 * affine runs on the halo-resident A image in LDS, so bn1's separate apply pass (one read of x_raw, one launch) disappears.
 * Bit-identical to sm3_bn_act followed by sm3_conv_gather_gemm.  Only for launches the halo-resident kernel takes --
 * stride-1 full 3 x 3, 16-bit, more than 256 workgroups, the A image within a quarter of a CU's LDS:
 * sm3_conv3x3_bnin_ok(d, views) == 1; anything else returns SM3_EINVAL and the caller keeps the two-pass form.
 * in_scale / in_shift: [views][Ci]; views = 2: two views back to back, each a multiple of 128 rows. */
int sm3_conv3x3_bnin_ok(const sm3_conv_desc* d, int views);
int sm3_conv3x3_bnin(const sm3_conv_desc* d, const void* x_raw, const float* in_scale, const float* in_shift, int views,
                     void* act_out, uint8_t* mask_out, const void* w, void* y, float* stat_partials, void* stream);

/* ------------------------------------------------------------------------------------------
 * NT-Xent.  replaces F.normalize + matmul + mask/select + /T (simclr.py:62-88, 294-320) and
 * nn.CrossEntropyLoss (tools/backbone_train.py:531, applied :101-102,119-120).
 * R = 2B rows, D = proj_dim.
 * ------------------------------------------------------------------------------------------ */
/* zn = z / max(|z|,1e-12); logits[i][0] = S[i][p]/T, logits[i][1..] = S[i][j != i,p]/T ascending j, p=(i+R/2)%R */
int sm3_ntxent_logits(const float* z, int R, int D, float temperature, float* zn, float* inv_norm,
                      float* logits, void* stream);
/* d(loss)/dz from d(loss)/dlogits (reference layout), written as dtype */
int sm3_ntxent_logits_bwd(int dtype, const float* dlogits, const float* zn, const float* inv_norm, int R,
                          int D, float temperature, void* dz, void* stream);
/* Every loss entry point below adds ONE value per call to loss[0]: the per-row terms are written to a buffer and summed
 * in a fixed order (fp64, one workgroup) -- like the reference's single CrossEntropyLoss reduction over a materialised
 * logits tensor (tools/backbone_train.py:531), the loss is a function of its inputs bit for bit (no float atomics).
 * The calls that accumulate into the same loss[0] must be enqueued on one stream. */
/* mean cross-entropy against label 0 of logits [R][Cc]; loss[0] += weight*CE; dlogits = weight*grad.
 * row_terms: R floats of scratch (required when loss is given). */
int sm3_ce_label0(const float* logits, int R, int Cc, float weight, float* row_terms, float* loss, float* dlogits,
                  void* stream);
/* fused: loss[0] += weight * NTXent(z); dz = weight * dNTXent/dz (dtype); never materialises logits.
 * workspace: R*D + 3*R floats (normalised rows, inverse norms, per-row logsumexp, per-row loss terms). */
int sm3_ntxent_fused(int dtype, const float* z, int R, int D, float temperature, float weight,
                     float* workspace, float* loss, void* dz, void* stream);

/* the same with dz additionally multiplied by the device scalar dz_scale[0] (loss scaling of the fp16 mode; the loss itself
 * stays unscaled) */
int sm3_ntxent_fused_scaled(int dtype, const float* z, int R, int D, float temperature, float weight,
                            const float* dz_scale, float* workspace, float* loss, void* dz, void* stream);

/* n <= 4 NT-Xent terms of EQUAL shape in three launches instead of 3 n (round 6; ABI 8 addition): the four loss terms of a
 * step -- derm, clinic and the two cross-modal ones (tools/backbone_train.py:99-102,119-121) -- sit on the main stream between
 * the lanes' forward and backward, where nothing else runs.  z, dz: host arrays of n device pointers; weights: host array of n;
 * dz_scale: NULL or the device scalar of sm3_ntxent_fused_scaled; workspace: n * round_up(R*D + 3*R, 4) floats, 16-byte aligned.  Every
 * dz[t] and the loss are bit-identical to n calls of sm3_ntxent_fused(_scaled) in term order.  D % 4 == 0, D <= 128
 * (SM3_EINVAL otherwise: the caller keeps the per-term calls). */
int sm3_ntxent_fused_batch(int dtype, int nterms, const float* const* z, int R, int D, float temperature, const float* weights,
                           const float* dz_scale, float* workspace, float* loss, void* const* dz, void* stream);

/* Global negatives under data parallelism (BASELINE.json north_star: all-gather of the projection embeddings; NOT the
 * reference's behaviour -- its negatives are the local batch, SURVEY.md section 0 -- hence an opt-in mode of the trainer):
 *   zn = normalise(z) (sm3_normalize_rows), all-gathered over RCCL to zg [Rg = world*Rl][D];
 *   S = zn zg^T (sm3_conv_gather_gemm, exact-f32);  sm3_ntxent_rect: loss += weight * mean_i(-S_ip/T + log sum_{j != self}
 *   exp(S_ij/T)) over the Rl local rows (self = column self_offset + i, positive = self_offset + (i + Rl/2) % Rl) and
 *   S <- d(loss)/dS in place (x dz_scale[0] if given);  d(zn) = dS zg (anchor role, sm3_conv_gather_gemm) + the local
 *   rows of all_reduce(dS^T zn) (candidate role, sm3_conv_wgrad);  sm3_normalize_rows_bwd: dz = inv_norm * (v - zn (zn.v)),
 *   v = dzn_a + dzn_b (dzn_b nullable). */
int sm3_normalize_rows(const float* z, int R, int D, float* zn, float* inv_norm, void* stream);
int sm3_ntxent_rect(float* S, int Rl, int Rg, int self_offset, float temperature, float weight, const float* dz_scale,
                    float* row_terms /* Rl floats of scratch */, float* loss, void* stream);
int sm3_normalize_rows_bwd(int dtype, const float* dzn_a, const float* dzn_b, const float* zn, const float* inv_norm,
                           int R, int D, void* dz, void* stream);

/* ------------------------------------------------------------------------------------------
 * AdamW over a flat fp32 buffer.  replaces torch.optim.AdamW(eps=1e-5, wd) + GradScaler unscale
 * (tools/backbone_train.py:124-127, 525-527).  g is multiplied by grad_scale first.
 * found_inf (nullable): if *found_inf != 0 the step is skipped (GradScaler semantics).
 * ------------------------------------------------------------------------------------------ */
int sm3_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int step, float grad_scale, const int32_t* found_inf,
              void* stream);
/* The fp16 mode's dynamic loss scaling (torch.cuda.amp.GradScaler, backbone_train.py:125-127,480) with the scaler's state
 * on the DEVICE, so that a step needs no host synchronisation: loss_scale[1] (the factor sm3_ntxent_fused_scaled applied to
 * dz), steps_taken[1] (optimizer steps actually taken: a skipped step does not advance Adam's bias correction).
 * sm3_adamw_dynamic: as sm3_adamw with g multiplied by grad_scale / loss_scale[0] and step = steps_taken[0] + 1; skipped when
 * found_inf[0] != 0.  sm3_loss_scale_update: GradScaler.update() -- on overflow scale *= backoff and the growth tracker
 * restarts, else steps_taken += 1 and after growth_interval clean steps scale *= growth; clears found_inf. */
int sm3_adamw_dynamic(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, float grad_scale, const float* loss_scale,
                      const int32_t* steps_taken, const int32_t* found_inf, void* stream);
int sm3_loss_scale_update(float* loss_scale, int32_t* found_inf, int32_t* growth_tracker, int32_t* steps_taken,
                          float growth_factor, float backoff_factor, int growth_interval, void* stream);
/* Momentum ("target") encoder of BASELINE.json's north_star -- an extension: the reference has no target network
 * (SURVEY.md section 0).  target = momentum * target + (1 - momentum) * online over the flat fp32 parameter buffer. */
int sm3_ema_update(float* target, const float* online, int64_t n, float momentum, void* stream);
/* found_inf[0] |= any(!isfinite(g)) */
int sm3_check_finite(const float* g, int64_t n, int32_t* found_inf, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline on the GPU: the SimCLR augmentation chain of tools/backbone_train.py:448-466 (torchvision 0.13
 * transforms on PIL images in DataLoader workers there) over a batch of decoded RGB images resident in HBM.
 * Random parameters are drawn by the host (sm3hip/augment.py, torchvision's sampling rules) and passed per sample.
 * ------------------------------------------------------------------------------------------ */
/* RandomResizedCrop + RandomHorizontalFlip + ToTensor: crop box[b] = (top, left, height, width) of src [B,Hs,Ws,3] uint8,
 * antialiased bilinear resample (PIL's triangle filter, support max(scale,1)) to [H,W], mirrored when flip[b] != 0;
 * out [B,3,H,W] fp32 in [0,1]. */
int sm3_aug_resized_crop(const uint8_t* src, int B, int Hs, int Ws, const int32_t* box, const uint8_t* flip, float* out,
                         int H, int W, void* stream);
/* One position of ColorJitter's randomly ordered chain, in place on img [B,3,H,W]: op[b] = 0 none, 1 brightness, 2 contrast,
 * 3 saturation, 4 hue, with factor[b] (torchvision functional_tensor: _blend / rgb_to_grayscale / _rgb2hsv / _hsv2rgb).
 * gray_mean: [B] scratch (the per-image grayscale mean the contrast blend needs, recomputed by every call). */
int sm3_aug_color_op(float* img, int B, int H, int W, const int32_t* op, const float* factor, float* gray_mean, void* stream);
/* RandomGrayscale (gray[b] != 0: 3-channel grayscale) -> GaussianBlur 3x3 with reflect padding (sigma[b] > 0, else none) ->
 * Normalize: out = (x - mean3[c]) / std3[c].  mean3 / std3 are HOST arrays of 3 floats. */
int sm3_aug_finish(const float* img, int B, int H, int W, const uint8_t* gray, const float* sigma, const float* mean3,
                   const float* std3, float* out, void* stream);

/* ---- multi-label heads of the inference model (reference inference.py:53-96; eval mode) -------------------
 * The Linear layers of that model run as sm3_conv_gather_gemm (1x1, bias-free) + sm3_bn_act(scale=1, shift=bias).
 * qkv: [B*S, 3*D] rows b*S+s, columns [q|k|v] (nn.MultiheadAttention's in_proj layout); out: [B*S, D]:
 * softmax(q k^T / sqrt(D/nhead)) v over the S <= 8 label tokens of each sample, per head. */
int sm3_token_attention(int dtype, const void* qkv, void* out, int B, int S, int D, int nhead, void* stream);
/* out = LayerNorm(a + b) * gamma + beta over the last axis (b nullable); rows x D of dtype, D <= 1024
 * (the post-norm residual joins of nn.TransformerEncoderLayer, inference.py:58-60). */
int sm3_add_layernorm(int dtype, const void* a, const void* b, const float* gamma, const float* beta, float eps,
                      void* out, int64_t rows, int D, void* stream);
/* out[b, t] = <x[b, token_of[t], :], W[t, :]> + bias[t], x: [B, S, D] of dtype (each token L2-normalised first when
 * l2_norm != 0), W: [T, D] fp32 = the prototype Linears concatenated (inference.py:62-71,90-94). */
int sm3_token_heads(int dtype, const void* x, const float* W, const float* bias, const int* token_of, int l2_norm,
                    float* out, int B, int S, int D, int T, void* stream);

/* ---- TRAINING of those heads (reference tools/mlc_train.py:58-90,241-283; tools/mlc_eval.py), all fp32 ------------------
 * The Linear layers (label projectors, attention in/out projections, feed-forward) are sm3_conv_gather_gemm / sm3_conv_wgrad
 * in SM3_F32; these entry points are everything else of one train-mode nn.TransformerEncoderLayer (post-norm, ReLU) over the
 * S <= 8 label tokens, the prototype heads, the pseudo-label cross-entropy and the spherical k-means behind the pseudo-labels.
 * Dropout keeps no mask: element e of stream `seed` is kept iff hash(seed, e) >= p; backward recomputes it. */
/* Token rows: label_major = 0: row(b, s) = b*S + s (the inference path's layout); 1: row(b, s) = s*B + b, the reference's
 * [S, B, D] stacking (mlc_train.py:79).
 * out[rows, D] = softmax(QK^T/sqrt(hd)) (dropout p on the probabilities) V, qkv [rows, 3D]; and its backward */
int sm3_mlc_attention_fwd(const float* qkv, float* out, int B, int S, int D, int nhead, float p, uint32_t seed,
                          int label_major, void* stream);
int sm3_mlc_attention_bwd(const float* qkv, const float* dout, float* dqkv, int B, int S, int D, int nhead, float p,
                          uint32_t seed, int label_major, void* stream);
/* out = LayerNorm(a + dropout_p(b)) * gamma + beta; stats[rows][2] = (mean, rstd).  Backward: da = d(sum), db = mask/(1-p) *
 * d(sum), dgamma / dbeta accumulated (atomics). */
int sm3_mlc_add_ln_fwd(const float* a, const float* b, const float* gamma, const float* beta, float eps, float p,
                       uint32_t seed, float* out, float* stats, int64_t rows, int D, void* stream);
int sm3_mlc_add_ln_bwd(const float* dout, const float* a, const float* b, const float* stats, const float* gamma, float p,
                       uint32_t seed, float* da, float* db, float* dgamma, float* dbeta, int64_t rows, int D, void* stream);
/* h = relu(y + bias) [rows, N], hd = dropout_p(h); backward dh = dhd * mask/(1-p) * (h > 0), dbias += column sums */
int sm3_mlc_bias_relu_drop_fwd(const float* y, const float* bias, float p, uint32_t seed, float* h, float* hd, int64_t rows,
                               int N, void* stream);
int sm3_mlc_relu_drop_bwd(const float* dhd, const float* h, float p, uint32_t seed, float* dh, float* dbias, int64_t rows,
                          int N, void* stream);
/* db[c] += sum_r dy[r][c]: the bias gradient of a Linear */
int sm3_mlc_colsum(const float* dy, float* db, int64_t rows, int N, void* stream);
/* loss[0] += mean_h mean_b CE(logits[b, off[h]:off[h+1]] / T, targets[h][b]); dlogits written (mlc_train.py:252-261).
 * ONE workgroup adds the B * H terms in a fixed fp64 order and does a plain (non-atomic) `loss[0] +=`: the caller zeroes
 * loss[0] before the first term of a step and issues every launch that adds to it on ONE stream (sm3hip/mlc.py does).
 * Launch time is linear in B * H on one CU (4 096 pairs at config 4's size: ~20 us); a caller with far larger B * H
 * should split the batch over calls. */
int sm3_mlc_ce(const float* logits, const int64_t* targets, const int* head_offsets, int H, int B, int Tn, float temperature,
               float* loss, float* dlogits, void* stream);
/* prototype heads in fp32 with either row layout (bias nullable: mlc_train.py's prototypes have none), and the backward:
 * dx written, dW [Tn,D] and dbias [Tn] (nullable) accumulated */
int sm3_mlc_heads_fwd(const float* x, const float* W, const float* bias, const int* token_of, int l2_norm, float* out, int B,
                      int S, int D, int Tn, int label_major, void* stream);
int sm3_mlc_heads_bwd(const float* dlogits, const float* x, const float* W, const int* token_of, int l2_norm, float* dx,
                      float* dW, float* dbias, int B, int S, int D, int Tn, int label_major, void* stream);
/* spherical k-means (mlc_train.py:146-177): E step assign[n] = argmax_k <emb[n], centroids[k]> (+ sums / counts of the M
 * step when given); M step centroid = normalise(sum / count) for non-empty clusters, every centroid L2-normalised */
int sm3_mlc_kmeans_assign(const float* emb, const float* centroids, int64_t* assign, float* sums, int* counts, int N, int D,
                          int K, void* stream);
int sm3_mlc_kmeans_update(float* centroids, const float* sums, const int* counts, int K, int D, void* stream);

/* ---- peer-to-peer SyncBatchNorm statistics exchange on one node (csrc/p2p.hip; opt-in, RCCL is the default) ----------------
 * Replaces the all-reduce torch.nn.SyncBatchNorm performs per BatchNorm and direction (tools/backbone_train.py:510) for the
 * fp64 [views][2C] sums of sm3_bn_stats_reduce / sm3_linbn_fold / sm3_linbn_stats.
 * sm3_p2p_alloc: a zeroed mailbox of sm3_p2p_mailbox_bytes() in device memory + its 64-byte hipIpc handle (to be sent to the
 * peers by any means); sm3_p2p_open / _close: map / unmap a peer's mailbox; sm3_p2p_free: release one's own.
 * sm3_p2p_allreduce_f64: buf[0..n) += the same range of every other rank, in place, ONE launch on `stream`, result
 * bit-identical on all ranks (contributions added in rank order).  mailboxes: host array of `world` device pointers indexed by
 * rank (one's own included); seq: 1, 2, 3, ... -- the same value on every rank for the same exchange, per mailbox set;
 * n <= sm3_p2p_max_elems(); err_flag (device int): set to 1 when a peer's contribution did not arrive within timeout_s --
 * that exchange and EVERY later one with the same err_flag then writes NaN into buf and returns at once (no further waits),
 * so the failure shows in whatever is computed from the sums even if the flag is never read.
 * Mailboxes are fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained), as RCCL's IPC buffers);
 * *kind_out (nullable) = 1, or 0 when that allocation / its IPC export failed and plain hipMalloc memory was used instead
 * (also forced by SM3_P2P_FINEGRAINED=0, for an A/B of the two). */
int sm3_p2p_mailbox_bytes(void);
int sm3_p2p_max_elems(void);
/* largest world size a mailbox has room for (8: one node), and the byte layout of a mailbox -- the data area
 * [data_first, data_first + data_bytes) source rank `src` writes for exchange slot `slot` (0 / 1) and the 8-byte arrival
 * flag of its block `block` (elems_per_block doubles each) -- so that a host-side check can prove, without a GPU, that the
 * areas of max_world ranks x 2 slots x every block are disjoint and inside sm3_p2p_mailbox_bytes(). */
int sm3_p2p_max_world(void);
int sm3_p2p_layout(int slot, int src, int block, int64_t* data_first, int64_t* data_bytes, int64_t* flag_off,
                   int* elems_per_block);
int sm3_p2p_alloc(void** ptr, void* ipc_handle_64, int* kind_out);
int sm3_p2p_open(const void* ipc_handle_64, void** ptr);
int sm3_p2p_close(void* ptr);
int sm3_p2p_free(void* ptr);
int sm3_p2p_allreduce_f64(double* buf, int n, void* const* mailboxes, int rank, int world, uint64_t seq, int* err_flag,
                          double timeout_s, void* stream);

#ifdef __cplusplus
}
#endif
#endif
