"""Eval-mode multi-label heads of the inference model on the HIP kernels (reference inference.py:79-96):

    feats [B, 4096] -> 8 x Linear(4096, 512) label tokens -> one nn.TransformerEncoderLayer over the 8 tokens
    (post-norm, ReLU, dropout inactive) -> optional L2 norm -> 8 prototype Linear heads.

Every Linear is sm3_conv_gather_gemm (the MFMA gather-GEMM as a 1x1 convolution, M = rows) followed by sm3_bn_act
with scale = 1 and shift = bias; the 8 label projectors are ONE GEMM with their weights stacked to [4096, 4096].
Tokens are kept sample-major ([B, 8, 512] = rows b*8+t), so every later GEMM sees one [8B, 512] matrix and the
attention kernel one contiguous block per sample.  No autograd here: training these heads (tools/mlc_train.py,
tools/mlc_eval.py fine-tuning) goes through sm3hip/mlc.py."""
import torch

from . import ops


def _bf16_or_f32(dtype):
    return ops.dtype_code(dtype)


class LabelHeads:
    """Weights of a reference `Model` (inference.py) repacked once per (dtype, parameter version)."""

    def __init__(self, model):
        self.model = model
        self._key = None

    def _versions(self):
        return tuple(p._version for p in self._params()) + tuple(p.data_ptr() for p in self._params())

    def _params(self):
        m = self.model
        ps = [l[0].weight for l in m.projectors.projectors] + [l[0].bias for l in m.projectors.projectors]
        ps += list(m.mlc_sa.parameters()) + [q for l in m.prototypes for q in (l.weight, l.bias)]
        return ps

    def _prepare(self, dt, dev):
        key = (dt, dev, self._versions())
        if key == self._key:
            return
        m, sa = self.model, self.model.mlc_sa
        if sa.norm_first or getattr(sa, "activation_relu_or_gelu", 1) != 1:
            raise NotImplementedError("only the post-norm ReLU TransformerEncoderLayer of inference.py is built")
        f32 = lambda t: t.detach().to(dev, torch.float32).contiguous()
        w = lambda t: t.detach().to(dev, dt).contiguous()
        self.S = len(m.projectors.projectors)
        self.D = m.feat_dim
        self.nhead = sa.self_attn.num_heads
        self.w_proj = w(torch.cat([l[0].weight for l in m.projectors.projectors], 0))      # [S*D, in]
        self.b_proj = f32(torch.cat([l[0].bias for l in m.projectors.projectors], 0))
        self.w_in, self.b_in = w(sa.self_attn.in_proj_weight), f32(sa.self_attn.in_proj_bias)
        self.w_out, self.b_out = w(sa.self_attn.out_proj.weight), f32(sa.self_attn.out_proj.bias)
        self.w1, self.b1 = w(sa.linear1.weight), f32(sa.linear1.bias)
        self.w2, self.b2 = w(sa.linear2.weight), f32(sa.linear2.bias)
        self.g1, self.be1, self.eps1 = f32(sa.norm1.weight), f32(sa.norm1.bias), sa.norm1.eps
        self.g2, self.be2, self.eps2 = f32(sa.norm2.weight), f32(sa.norm2.bias), sa.norm2.eps
        self.w_heads = f32(torch.cat([l.weight for l in m.prototypes], 0))                 # [T, D]
        self.b_heads = f32(torch.cat([l.bias for l in m.prototypes], 0))
        self.sizes = [l.weight.shape[0] for l in m.prototypes]
        tok = [i % self.S for i, n in enumerate(self.sizes) for _ in range(n)]
        self.token_of = torch.tensor(tok, dtype=torch.int32, device=dev)
        self.ones = torch.ones(max(self.S * self.D, 3 * self.D), dtype=torch.float32, device=dev)
        self._key = key

    def _linear(self, code, x, w, bias, rows, relu=False):
        """y[rows, N] = x[rows, K] @ w[N, K]^T + bias  (MFMA gather-GEMM + bias epilogue kernel)."""
        N, K = w.shape
        d = ops.fwd_desc(code, rows, 1, 1, K, N, 1, 1, 0)
        y = torch.empty(rows, N, dtype=x.dtype, device=x.device)
        ops.conv_gemm(d, x, w, y, None, None)
        ops.bn_act(code, y, self.ones[:N], bias, None, relu, y, rows, N)
        return y

    @torch.no_grad()
    def __call__(self, feats, dtype):
        """feats: [B, in] fp32 CUDA (concatenated encoder features) -> list of 8 fp32 logit tensors."""
        if not feats.is_cuda:
            raise ValueError("the SM3 HIP path has no CPU fallback")
        dev, B = feats.device, feats.shape[0]
        code = _bf16_or_f32(dtype)
        self._prepare(dtype, dev)
        S, D = self.S, self.D
        x = feats.contiguous()
        if dtype != torch.float32:
            xt = torch.empty_like(x, dtype=dtype)
            ops.cast_from_f32(code, x, xt)
            x = xt
        tokens = self._linear(code, x, self.w_proj, self.b_proj, B).view(B * S, D)        # rows b*S + t
        qkv = self._linear(code, tokens, self.w_in, self.b_in, B * S)
        att = torch.empty(B * S, D, dtype=dtype, device=dev)
        ops.token_attention(code, qkv, att, B, S, D, self.nhead)
        att = self._linear(code, att, self.w_out, self.b_out, B * S)
        x1 = torch.empty_like(tokens)
        ops.add_layernorm(code, tokens, att, self.g1, self.be1, self.eps1, x1, B * S, D)
        ff = self._linear(code, self._linear(code, x1, self.w1, self.b1, B * S, relu=True), self.w2, self.b2, B * S)
        x2 = torch.empty_like(tokens)
        ops.add_layernorm(code, x1, ff, self.g2, self.be2, self.eps2, x2, B * S, D)
        T = self.w_heads.shape[0]
        out = torch.empty(B, T, dtype=torch.float32, device=dev)
        ops.token_heads(code, x2, self.w_heads, self.b_heads, self.token_of, bool(self.model.l2_norm), out, B, S, D, T)
        return list(out.split(self.sizes, dim=1))
