"""Driver entry points: build() compiles the HIP library for gfx950; smoke() runs one tiny SM3 pretrain
step on cuda:0 and checks it against the CPU oracle."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "skin-sm3_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def build():
    """hipcc --offload-arch=gfx950 -> skin-sm3_amd/sm3hip/libsm3hip.so (in-tree), then import the package.
    The oracle is Python (torch-CPU restatement): nothing to compile there, and the reference is Python,
    so there is no oracle/_ref build either (DESIGN.md, section Oracle)."""
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j4"])
    from sm3hip import _lib
    _lib.load()
    import sm3hip  # noqa: F401


def smoke():
    from sm3hip import selftest
    selftest.smoke()


if __name__ == "__main__":
    build()
    print("build ok")
