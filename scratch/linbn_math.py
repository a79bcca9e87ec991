"""fp64 check of the 'BatchNorm backward by linearity' identities for an expanding 1x1 convolution x3 = y2 W^T
followed by train-mode BN (scratch; mirrors what sm3_linbn_* compute)."""
import torch
torch.manual_seed(0)
M, p, Co = 640, 16, 64
y2 = torch.relu(torch.randn(M, p, dtype=torch.float64) + 0.3).requires_grad_()
W = (torch.randn(Co, p, dtype=torch.float64) * 0.2).requires_grad_()
gamma = (torch.rand(Co, dtype=torch.float64) + 0.5).requires_grad_()
beta = torch.randn(Co, dtype=torch.float64).requires_grad_()
x3 = y2 @ W.t()
mu, var = x3.mean(0), x3.var(0, unbiased=False)
istd = (var + 1e-5).rsqrt()
z = (x3 - mu) * istd * gamma + beta
dz = torch.randn(M, Co, dtype=torch.float64) * (torch.rand(M, Co, dtype=torch.float64) > 0.4)
z.backward(dz)
# ---- linear form
with torch.no_grad():
    S1 = dz.sum(0)
    P = dz.t() @ y2                      # wgrad product
    G = y2.t() @ y2
    s = y2.sum(0)
    mu_s = mu                            # saved mean
    S2 = istd * ((P * W).sum(1) - mu_s * S1)
    assert torch.allclose(S2, (dz * (x3 - mu) * istd).sum(0))
    m1, m2 = S1 / M, S2 / M
    a = gamma * istd
    b = a * istd * m2
    Wa = a[:, None] * W                  # [Co, p]
    H = W.t() @ (b[:, None] * W)         # [p, p]
    const = ((b * mu_s - a * m1)[None, :] @ W)[0]
    dy2 = dz @ Wa - y2 @ H + const
    T = W @ G                            # = x3^T y2
    dW = a[:, None] * (P - m1[:, None] * s[None, :]) - b[:, None] * (T - mu_s[:, None] * s[None, :])
    print("dy2   ", (dy2 - y2.grad).abs().max().item(), y2.grad.abs().max().item())
    print("dW    ", (dW - W.grad).abs().max().item(), W.grad.abs().max().item())
    print("dgamma", (S2 - gamma.grad).abs().max().item(), "dbeta", (S1 - beta.grad).abs().max().item())
