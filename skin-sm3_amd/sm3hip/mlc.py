"""Training of the multi-label heads on the HIP kernels (reference tools/mlc_train.py:58-90 `Model`,
:241-283 loop, :116-189 spherical k-means; tools/mlc_eval.py reuses the same model with biased prototypes).

    feats [B, 4096] -> S per-label Linear(4096, D) (src/models/projector.py:65-78) -> stack [S, B, D]
    -> nn.TransformerEncoderLayer(D, nhead, ff, dropout) in TRAIN mode over the S label tokens of each sample
    -> optional L2 norm -> per-label prototype Linear -> logits

Every Linear (forward, data gradient, weight gradient) is the exact-f32 MFMA gather-GEMM / weight-gradient kernel; the
attention over the 8 tokens, the two residual LayerNorms with their dropouts, bias + ReLU + dropout, the prototype heads
and the pseudo-label cross-entropy are csrc/heads_train.hip.  fp32 throughout, as the reference runs this part
(mlc_train.py:294-295).  One torch.autograd.Function spans the heads, so the frozen-extractor default
(mlc_train.py:347-348) and --finetune-backbone (gradient into the HIP encoders through sm3hip.bridge) both work, and the
caller keeps the reference's literal loop (CrossEntropyLoss on the predictions, torch.optim.AdamW)."""
import torch

from . import _lib, ops
from ._lib import SM3_F32, check

_P = ops._ptr


def _st():
    return ops._stream()


class MLCHeads:
    """Kernel sequencing for one `Model` (its projectors / mlc_sa / prototypes own the parameters)."""

    def __init__(self, model):
        self.model = model
        sa = model.mlc_sa
        if sa.norm_first or getattr(sa, "activation_relu_or_gelu", 1) != 1:
            raise NotImplementedError("only the post-norm ReLU TransformerEncoderLayer of mlc_train.py is built")
        self.S = len(model.projectors.projectors)
        self.D = sa.self_attn.embed_dim
        self.nhead = sa.self_attn.num_heads
        self.p = float(sa.dropout.p)
        if float(sa.self_attn.dropout) != self.p or float(sa.dropout1.p) != self.p or float(sa.dropout2.p) != self.p:
            raise NotImplementedError("one dropout rate for the whole layer, as nn.TransformerEncoderLayer builds it")
        self.sizes = [l.weight.shape[0] for l in model.prototypes]
        self.has_pbias = model.prototypes[0].bias is not None

    def params(self):
        m, sa = self.model, self.model.mlc_sa
        ps = []
        for l in m.projectors.projectors:
            ps += [l[0].weight, l[0].bias]
        ps += [sa.self_attn.in_proj_weight, sa.self_attn.in_proj_bias, sa.self_attn.out_proj.weight, sa.self_attn.out_proj.bias,
               sa.linear1.weight, sa.linear1.bias, sa.linear2.weight, sa.linear2.bias, sa.norm1.weight, sa.norm1.bias,
               sa.norm2.weight, sa.norm2.bias]
        for l in m.prototypes:
            ps.append(l.weight)
            if self.has_pbias:
                ps.append(l.bias)
        return ps

    # ---- primitives -----------------------------------------------------------------------------------------------
    @staticmethod
    def _gemm(x, w, out=None, addend=None):
        """out[rows, N] = x[rows, K] @ w[N, K]^T (+ addend), exact-f32 MFMA."""
        rows, K = x.shape
        N = w.shape[0]
        if out is None:
            out = torch.empty(rows, N, dtype=torch.float32, device=x.device)
        ops.conv_gemm(ops.fwd_desc(SM3_F32, rows, 1, 1, K, N, 1, 1, 0), x, w, out, addend, None)
        return out

    @staticmethod
    def _wgrad(x, dy, dw):
        """dw[N, K] += dy[rows, N]^T @ x[rows, K]."""
        rows, K = x.shape
        ops.conv_wgrad(ops.fwd_desc(SM3_F32, rows, 1, 1, K, dy.shape[1], 1, 1, 0), x, dy, dw)

    def _bias(self, y, bias, ones):
        rows, N = y.shape
        ops.bn_act(SM3_F32, y, ones[:N], bias, None, False, y, rows, N)

    @staticmethod
    def _colsum(dy, db):
        check(_lib.load().sm3_mlc_colsum(_P(dy), _P(db), dy.shape[0], dy.shape[1], _st()), "sm3_mlc_colsum")

    # ---- forward / backward ---------------------------------------------------------------------------------------
    def forward(self, feats, seed, train=True):
        m, sa, lib = self.model, self.model.mlc_sa, _lib.load()
        if not feats.is_cuda or feats.dtype != torch.float32:
            raise ValueError("the SM3 HIP path has no CPU fallback; features must be fp32 CUDA")
        feats = feats.contiguous()
        dev, B, S, D = feats.device, feats.shape[0], self.S, self.D
        R, p = S * B, (self.p if train else 0.0)
        seed = int(seed) & 0x7FFFFFFF
        ones = torch.ones(max(3 * D, sa.linear1.out_features), dtype=torch.float32, device=dev)
        x0 = torch.empty(R, D, dtype=torch.float32, device=dev)          # row = s*B + b (the reference's [S, B, D])
        for s, l in enumerate(m.projectors.projectors):
            blk = x0[s * B:(s + 1) * B]
            self._gemm(feats, l[0].weight.detach(), blk)
            self._bias(blk, l[0].bias.detach(), ones)
        att = sa.self_attn
        qkv = self._gemm(x0, att.in_proj_weight.detach())
        self._bias(qkv, att.in_proj_bias.detach(), ones)
        a = torch.empty(R, D, dtype=torch.float32, device=dev)
        check(lib.sm3_mlc_attention_fwd(_P(qkv), _P(a), B, S, D, self.nhead, p, seed + 1, 1, _st()), "sm3_mlc_attention_fwd")
        o = self._gemm(a, att.out_proj.weight.detach())
        self._bias(o, att.out_proj.bias.detach(), ones)
        x1, st1 = torch.empty_like(x0), torch.empty(R, 2, dtype=torch.float32, device=dev)
        check(lib.sm3_mlc_add_ln_fwd(_P(x0), _P(o), _P(sa.norm1.weight.detach()), _P(sa.norm1.bias.detach()), sa.norm1.eps, p,
                                     seed + 2, _P(x1), _P(st1), R, D, _st()), "sm3_mlc_add_ln_fwd")
        y1 = self._gemm(x1, sa.linear1.weight.detach())
        F = y1.shape[1]
        h, hd = torch.empty_like(y1), torch.empty_like(y1)
        check(lib.sm3_mlc_bias_relu_drop_fwd(_P(y1), _P(sa.linear1.bias.detach()), p, seed + 3, _P(h), _P(hd), R, F, _st()),
              "sm3_mlc_bias_relu_drop_fwd")
        fo = self._gemm(hd, sa.linear2.weight.detach())
        self._bias(fo, sa.linear2.bias.detach(), ones)
        x2, st2 = torch.empty_like(x0), torch.empty(R, 2, dtype=torch.float32, device=dev)
        check(lib.sm3_mlc_add_ln_fwd(_P(x1), _P(fo), _P(sa.norm2.weight.detach()), _P(sa.norm2.bias.detach()), sa.norm2.eps, p,
                                     seed + 4, _P(x2), _P(st2), R, D, _st()), "sm3_mlc_add_ln_fwd")
        wp = torch.cat([l.weight.detach() for l in m.prototypes], 0).contiguous()          # [T, D]
        bp = torch.cat([l.bias.detach() for l in m.prototypes], 0).contiguous() if self.has_pbias else None
        T = wp.shape[0]
        tok = torch.tensor([i % S for i, n in enumerate(self.sizes) for _ in range(n)], dtype=torch.int32, device=dev)
        logits = torch.empty(B, T, dtype=torch.float32, device=dev)
        check(lib.sm3_mlc_heads_fwd(_P(x2), _P(wp), _P(bp), _P(tok), int(bool(m.l2_norm)), _P(logits), B, S, D, T, 1, _st()),
              "sm3_mlc_heads_fwd")
        saved = dict(feats=feats, x0=x0, qkv=qkv, a=a, o=o, x1=x1, st1=st1, h=h, hd=hd, fo=fo, x2=x2, st2=st2, wp=wp, tok=tok,
                     B=B, p=p, seed=seed)
        sa_feats = x2.view(S, B, D)
        if m.l2_norm:
            sa_feats = torch.nn.functional.normalize(sa_feats, dim=-1, p=2)
        return sa_feats, logits, saved

    def backward(self, sv, dlogits, need_dfeats):
        m, sa, lib = self.model, self.model.mlc_sa, _lib.load()
        att = sa.self_attn
        B, S, D, p, seed = sv["B"], self.S, self.D, sv["p"], sv["seed"]
        R = S * B
        dev = dlogits.device
        z = lambda t: torch.zeros_like(t, dtype=torch.float32)
        g = {id(q): z(q) for q in self.params()}
        G = lambda q: g[id(q)]
        dlogits = dlogits.contiguous().float()
        T = sv["wp"].shape[0]
        dx2 = torch.empty(R, D, dtype=torch.float32, device=dev)
        dwp = torch.zeros(T, D, dtype=torch.float32, device=dev)
        dbp = torch.zeros(T, dtype=torch.float32, device=dev) if self.has_pbias else None
        check(lib.sm3_mlc_heads_bwd(_P(dlogits), _P(sv["x2"]), _P(sv["wp"]), _P(sv["tok"]), int(bool(m.l2_norm)), _P(dx2),
                                    _P(dwp), _P(dbp), B, S, D, T, 1, _st()), "sm3_mlc_heads_bwd")
        off = 0
        for l, n in zip(m.prototypes, self.sizes):
            G(l.weight).copy_(dwp[off:off + n])
            if self.has_pbias:
                G(l.bias).copy_(dbp[off:off + n])
            off += n
        # LayerNorm 2 (+ dropout2): d(x1 + drop(fo))
        dx1, dfo = torch.empty_like(dx2), torch.empty_like(dx2)
        check(lib.sm3_mlc_add_ln_bwd(_P(dx2), _P(sv["x1"]), _P(sv["fo"]), _P(sv["st2"]), _P(sa.norm2.weight.detach()), p,
                                     seed + 4, _P(dx1), _P(dfo), _P(G(sa.norm2.weight)), _P(G(sa.norm2.bias)), R, D, _st()),
              "sm3_mlc_add_ln_bwd")
        # feed-forward
        self._colsum(dfo, G(sa.linear2.bias))
        self._wgrad(sv["hd"], dfo, G(sa.linear2.weight))
        dhd = self._gemm(dfo, sa.linear2.weight.detach().t().contiguous())
        dh = torch.empty_like(dhd)
        check(lib.sm3_mlc_relu_drop_bwd(_P(dhd), _P(sv["h"]), p, seed + 3, _P(dh), _P(G(sa.linear1.bias)), R, dh.shape[1],
                                        _st()), "sm3_mlc_relu_drop_bwd")
        self._wgrad(sv["x1"], dh, G(sa.linear1.weight))
        self._gemm(dh, sa.linear1.weight.detach().t().contiguous(), out=dx1, addend=dx1)
        # LayerNorm 1 (+ dropout1): d(x0 + drop(o))
        dx0, do = torch.empty_like(dx2), torch.empty_like(dx2)
        check(lib.sm3_mlc_add_ln_bwd(_P(dx1), _P(sv["x0"]), _P(sv["o"]), _P(sv["st1"]), _P(sa.norm1.weight.detach()), p,
                                     seed + 2, _P(dx0), _P(do), _P(G(sa.norm1.weight)), _P(G(sa.norm1.bias)), R, D, _st()),
              "sm3_mlc_add_ln_bwd")
        # attention
        self._colsum(do, G(att.out_proj.bias))
        self._wgrad(sv["a"], do, G(att.out_proj.weight))
        da = self._gemm(do, att.out_proj.weight.detach().t().contiguous())
        dqkv = torch.empty(R, 3 * D, dtype=torch.float32, device=dev)
        check(lib.sm3_mlc_attention_bwd(_P(sv["qkv"]), _P(da), _P(dqkv), B, S, D, self.nhead, p, seed + 1, 1, _st()),
              "sm3_mlc_attention_bwd")
        self._colsum(dqkv, G(att.in_proj_bias))
        self._wgrad(sv["x0"], dqkv, G(att.in_proj_weight))
        self._gemm(dqkv, att.in_proj_weight.detach().t().contiguous(), out=dx0, addend=dx0)
        # label projectors
        dfeats = None
        for s, l in enumerate(m.projectors.projectors):
            blk = dx0[s * B:(s + 1) * B]
            self._colsum(blk, G(l[0].bias))
            self._wgrad(sv["feats"], blk, G(l[0].weight))
            if need_dfeats:
                wt = l[0].weight.detach().t().contiguous()
                if dfeats is None:
                    dfeats = self._gemm(blk, wt)
                else:
                    self._gemm(blk, wt, out=dfeats, addend=dfeats)
        return [g[id(q)] for q in self.params()], dfeats


class _HeadsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, heads, seed, train, feats, *params):
        sa_feats, logits, saved = heads.forward(feats, seed, train)
        ctx.heads, ctx.saved = heads, saved
        ctx.need_dfeats = feats.requires_grad
        ctx.mark_non_differentiable(sa_feats)  # the memory bank takes it detached (mlc_train.py:268-271)
        return sa_feats, logits

    @staticmethod
    def backward(ctx, _dsa, dlogits):
        with ops.stream_scope():
            grads, dfeats = ctx.heads.backward(ctx.saved, dlogits, ctx.need_dfeats)
        ctx.saved = None
        return (None, None, None, dfeats) + tuple(grads)


def heads_forward(model, feats, seed=None):
    """(sa_feats [S, B, D], [logits_i [B, n_i]]) of `model` (projectors / mlc_sa / prototypes) on the HIP kernels, with
    autograd attached to the head parameters (and to `feats` when it requires grad)."""
    heads = model.__dict__.get("_sm3_mlc_heads")
    if heads is None:
        heads = MLCHeads(model)
        model.__dict__["_sm3_mlc_heads"] = heads
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)))  # torch's CPU generator: reproducible under fix_random_seeds
    with ops.stream_scope():
        sa_feats, logits = _HeadsFn.apply(heads, seed, model.mlc_sa.training, feats.float(), *heads.params())
    return sa_feats, list(logits.split(heads.sizes, dim=1))


# ---- pseudo-labels: spherical k-means over the memory bank (mlc_train.py:116-189) -----------------------------------
@torch.no_grad()
def spherical_kmeans(embeddings, K, iters=10, generator=None):
    """embeddings [N, D] fp32 CUDA (one label's memory bank) -> (centroids [K, D], assignments [N] int64): random
    samples as the initial centroids, `iters` rounds of (assign by largest dot product; centroid = L2-normalised mean of
    its members), then the final assignment."""
    if not embeddings.is_cuda or embeddings.dtype != torch.float32:
        raise ValueError("spherical_kmeans: fp32 CUDA embeddings")
    emb = embeddings.contiguous()
    N, D = emb.shape
    if N < K:
        raise ValueError("please reduce the number of centroids")  # mlc_train.py:148
    idx = torch.randperm(N, generator=generator)[:K].to(emb.device)
    cent = emb[idx].contiguous()
    assign = torch.empty(N, dtype=torch.int64, device=emb.device)
    sums = torch.empty(K, D, dtype=torch.float32, device=emb.device)
    counts = torch.empty(K, dtype=torch.int32, device=emb.device)
    lib = _lib.load()
    with ops.stream_scope():
        for _ in range(iters):
            sums.zero_()
            counts.zero_()
            check(lib.sm3_mlc_kmeans_assign(_P(emb), _P(cent), _P(assign), _P(sums), _P(counts), N, D, K, _st()),
                  "sm3_mlc_kmeans_assign")
            check(lib.sm3_mlc_kmeans_update(_P(cent), _P(sums), _P(counts), K, D, _st()), "sm3_mlc_kmeans_update")
        check(lib.sm3_mlc_kmeans_assign(_P(emb), _P(cent), _P(assign), None, None, N, D, K, _st()), "sm3_mlc_kmeans_assign")
    return cent, assign


def pseudo_label_loss(logits, targets, temperature):
    """Fused form of the loop body mlc_train.py:252-261 (forward only, for monitoring): mean over heads of
    CrossEntropyLoss(pred / temperature, target).  logits: list of [B, n_i]; targets: [H, B] int64."""
    cat = torch.cat(logits, 1).contiguous().float()
    B, T = cat.shape
    H = len(logits)
    off = torch.tensor([0] + list(torch.tensor([l.shape[1] for l in logits]).cumsum(0)), dtype=torch.int32, device=cat.device)
    loss = torch.zeros(1, dtype=torch.float32, device=cat.device)
    dl = torch.empty_like(cat)
    with ops.stream_scope():
        check(_lib.load().sm3_mlc_ce(_P(cat), _P(targets.contiguous()), _P(off), H, B, T, float(temperature), _P(loss), _P(dl),
                                     _st()), "sm3_mlc_ce")
    return loss, dl
