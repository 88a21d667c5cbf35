"""sfron -- MI355X-native SFR-on unlearning hot path (package dir: unified-unlearning-w-remain-geometry_amd).

Host side (Python, PyTorch-ROCm for device memory / streams / torch.distributed) over the C-ABI
library ``libsfron.so`` (hand-written HIP for gfx950, sources in ``csrc/``).  There is NO CPU or
PyTorch fallback: every op raises if the library is missing or the tensors are not on the GPU.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
