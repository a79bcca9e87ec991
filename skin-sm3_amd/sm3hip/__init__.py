"""sm3hip -- MI355X-native kernels and engine for the SM3 pre-training hot path.

`import sm3hip` never touches the GPU; the HIP library is loaded on first use and its absence is a
hard error (sm3hip._lib.SM3LibraryError): there is no CPU fallback in the product path.
"""
import os

# Kernel arguments in device memory instead of host memory (a ROCm runtime switch; the HIP runtime reads it when it
# initialises, so it is set before `import torch`): a kernel that starts no longer fetches its argument block over PCIe.
# With ~1 160 dependent launches per step that is +2.2 % of the two-lane step and +3.5 % single-lane
# (profiles/r06_dev_kernarg_ab.txt).  Effective when sm3hip is imported before the
# process makes its first HIP call (bench.py and the tools set it themselves, first thing).  An explicit HIP_FORCE_DEV_KERNARG in the environment wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

_DEFAULT_DTYPE = {"bf16": torch.bfloat16, "f32": torch.float32, "fp32": torch.float32, "f16": torch.float16,
                  "fp16": torch.float16}[
    os.environ.get("SM3_DTYPE", "bf16").lower()
]


def set_default_dtype(dtype):
    """Activation/MFMA dtype of engines created afterwards: torch.bfloat16 (throughput), torch.float16 (the reference's
    AMP storage type; the fused trainer then runs dynamic loss scaling) or torch.float32 (exact-f32 MFMA parity mode)."""
    global _DEFAULT_DTYPE
    if dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise ValueError("sm3hip supports torch.bfloat16, torch.float16 and torch.float32")
    _DEFAULT_DTYPE = dtype


def default_dtype():
    return _DEFAULT_DTYPE
