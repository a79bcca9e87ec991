"""Test helper: run the host-side control flow of sm3hip (engine, trainer, bridge) on CPU tensors with the
C-ABI calls replaced by a recorder that launches nothing.  Used by the `not gpu` tests to check host logic:
shape/dtype validation in sm3hip.ops, kernel sequencing, gradient-bucket schedule, collective call pattern."""
import contextlib

import torch


class _FakeFn:
    def __init__(self, name, log):
        self.name, self.log = name, log
        self.argtypes, self.restype = None, None

    def __call__(self, *args):
        self.log.append(self.name)
        if self.name == "sm3_conv_partial_rows":
            d = args[0]._obj
            return (d.N * d.Ho * d.Wo + 127) // 128
        if self.name == "sm3_bn_bwd_partial_rows":
            return max(1, min(1024, (int(args[0]) + 63) // 64))
        if self.name == "sm3_maxpool_bn_bwd_partial_rows":
            n, h, w, v = (int(a) for a in args)
            return max(1, min(1024, (n // v * h * w + 63) // 64))
        if self.name == "sm3_stem_partial_rows":
            n, h, w = (int(a) for a in args)
            return n * ((h - 1) // 2 + 1) * (((w - 1) // 2 + 1 + 127) // 128)
        if self.name == "sm3_bn_act_colsum_rows":
            return 1
        if self.name == "sm3_conv_wgrad_slabs":
            args[6]._obj.value = 1  # slabs used per view
            return 0
        if self.name == "sm3_abi_version":
            return 8
        return 0


class FakeLib:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if name.startswith("sm3_"):
            return _FakeFn(name, self.calls)
        raise AttributeError(name)


@contextlib.contextmanager
def installed():
    from sm3hip import _lib, ops
    fake = FakeLib()
    saved = (_lib._lib, ops._chk, ops._stream)

    def chk(t, dtype=None, name="tensor"):
        if t is None:
            return
        if not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous")
        if dtype is not None and t.dtype != dtype:
            raise ValueError(f"{name}: expected {dtype}, got {t.dtype}")

    _lib._lib, ops._chk, ops._stream = fake, chk, (lambda: None)
    try:
        yield fake
    finally:
        _lib._lib, ops._chk, ops._stream = saved
