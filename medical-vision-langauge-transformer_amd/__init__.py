"""MI355X-native (gfx950) implementation of the MVLT vision-language hot path.

Drop-in ``nn.Module`` classes with the reference's constructor / ``forward``
signatures and state-dict keys, backed by hand-written HIP kernels behind the
C-ABI in ``include/mvlt_hip.h`` (``libmvlt_hip.so``).  See DESIGN.md.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
