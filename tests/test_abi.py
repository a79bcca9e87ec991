"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/sm3_hip.h declares
(no compute calls here -- there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sm3_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(sm3_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from sm3hip import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_library_loads_and_exports_every_symbol():
    from sm3hip import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.sm3_abi_version() == 8


def test_arg_rejection_launches_nothing():
    """Argument validation happens on the host before any launch, so it is checkable without a GPU."""
    import ctypes as C
    from sm3hip import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    assert lib.sm3_conv_gather_gemm(C.byref(d), None, None, None, None, None, None) == -1
    assert lib.sm3_bn_act(7, None, None, None, None, 0, 0, None, None, 0, 0, 1, None) == -1
    assert lib.sm3_adamw(None, None, None, None, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 1, 1.0, None, None) == -1


def test_product_path_fails_loudly_without_library(monkeypatch, tmp_path):
    from sm3hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.SM3LibraryError):
        _lib.load()


def test_dgrad_descriptors_cover_every_input_pixel_once():
    from sm3hip import ops
    for (H, W, k, s, p) in [(14, 14, 3, 1, 1), (13, 11, 3, 2, 1), (56, 56, 3, 2, 1), (10, 10, 1, 2, 0), (7, 9, 1, 1, 0)]:
        descs, full = ops.dgrad_descs(0, 2, H, W, 64, 128, k, s, p)
        seen = set()
        for d in descs:
            for i in range(d.Ho):
                for j in range(d.Wo):
                    pos = (i * d.osy + d.ooy, j * d.osx + d.oox)
                    assert pos not in seen and pos[0] < H and pos[1] < W
                    seen.add(pos)
        if full:
            assert len(seen) == H * W
        else:
            assert k == 1 and s == 2 and seen == {(y, x) for y in range(0, H, 2) for x in range(0, W, 2)}
        # every (input row, output row, kh) triple of the forward conv is used by exactly one dgrad tap
        Ho = (H + 2 * p - k) // s + 1
        pairs = set()
        for d in descs:
            for i in range(d.Ho):
                for t in range(d.ntaps):
                    oy = i + d.dy[t]
                    if 0 <= oy < Ho:
                        pairs.add((i * d.osy + d.ooy, oy, d.wtap[t] // k))
        want = {(oy * s - p + kh, oy, kh) for oy in range(Ho) for kh in range(k) if 0 <= oy * s - p + kh < H}
        assert pairs == want
