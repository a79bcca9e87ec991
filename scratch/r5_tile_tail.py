"""VERDICT r4 item 4 (second half): what is the half-empty last round of tiles of the MFMA-bound 3x3 launches worth -- alone, and
beside the other lane?  The 3x3 256 -> 256 forward launch of the step has 1 568 tiles of 128 x 128 on 1 024 workgroup slots
(4 per CU): one full round and one half-empty one.  A stream-K split of the last round could at best make the launch cost
1 568 / 2 048 of two full rounds.  Measured here WITHOUT building it: the same kernel on 668 images (2 046 tiles: two full
rounds) against 512 images (1 568 tiles), per tile, (a) alone and (b) while a second stream runs what the other lane would
be running -- an HBM-bound BatchNorm pass, or the same convolution.  If (b) shows the same time per tile for both tile counts,
the tail is already filled by the other lane and a stream-K split has nothing to recover in the two-lane step."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "skin-sm3_amd")]
from sm3hip import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16; code = ops.dtype_code(dt)


def conv_job(N, H, Ci, Co, k):
    d = ops.fwd_desc(code, N, H, H, Ci, Co, k, 1, k // 2); M = N * H * H
    x = torch.randn(M, Ci, device=dev).to(dt); w = (torch.randn(Co, k * k * Ci, device=dev) * 0.05).to(dt)
    y = torch.empty(M, Co, dtype=dt, device=dev); part = torch.empty(ops.conv_partial_rows(d) * 2 * Co, device=dev)
    return (lambda: ops.conv_gemm(d, x, w, y, None, part)), ((M + 127) // 128) * ((Co + 127) // 128)


def bn_apply_job(rows, C):
    dz = torch.randn(rows, C, device=dev).to(dt); x = torch.randn(rows, C, device=dev).to(dt); dx = torch.empty_like(dz)
    mean, istd, gamma = torch.randn(2 * C, device=dev), torch.rand(2 * C, device=dev) + 0.5, torch.rand(C, device=dev) + 0.5
    gs = torch.randn(2 * 2 * C, dtype=torch.float64, device=dev); ls = gs.clone()
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    return lambda: ops.bn_bwd_apply(code, dz, x, mean, istd, gamma, gs, float(rows), ls, dg, db, dx, rows // 2, C, views=2)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(pairs):
    torch.cuda.synchronize()
    evs = []
    for st in (s1, s2): st.wait_stream(torch.cuda.current_stream())
    for fn, n, st in pairs:
        with torch.cuda.stream(st), ops.stream_scope():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): fn()
            e1.record()
            evs.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in evs]


print(__doc__)
bn = bn_apply_job(1605632, 64)
for name, H, C in (("3x3 256 -> 256, 14 x 14", 14, 256), ("3x3 128 -> 128, 28 x 28", 28, 128)):
    per = H * H
    n_full = (1023 * 128) // per if H == 14 else (4095 * 128 // 1) // per  # images for just under a whole number of rounds
    jobs = {}
    for N in (512, n_full):
        fn, tiles = conv_job(N, H, C, C, 3)
        jobs[N] = (fn, tiles)
    print(f"\n{name}")
    reps = 24
    for N, (fn, tiles) in jobs.items():
        timed([(fn, 3, s1)])
        alone = min(timed([(fn, reps, s1)])[0] for _ in range(3)) / reps * 1e3
        other = jobs[N][0]
        with_bn = min(timed([(fn, reps, s1), (bn, 4 * reps, s2)])[0] for _ in range(3)) / reps * 1e3
        with_conv = min(timed([(fn, reps, s1), (other, 2 * reps, s2)])[0] for _ in range(3)) / reps * 1e3
        print(f"  {N:4d} images, {tiles:5d} tiles ({tiles / 1024:.2f} rounds of 1 024 slots): alone {alone:6.1f} us = {alone / tiles * 1e3:6.1f} ns per tile; "
              f"beside an HBM-bound pass {with_bn:6.1f} us = {with_bn / tiles * 1e3:6.1f} ns per tile; beside the same convolution "
              f"{with_conv:6.1f} us = {with_conv / tiles * 1e3:6.1f} ns per tile")
