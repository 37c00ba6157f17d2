"""MI355X-native (gfx950) implementation of the MVLT vision-language hot path.

Drop-in ``nn.Module`` classes with the reference's constructor / ``forward``
signatures and state-dict keys, backed by hand-written HIP kernels behind the
C-ABI in ``include/mvlt_hip.h`` (``libmvlt_hip.so``).  See DESIGN.md.
"""
from . import _lib  # noqa: F401
from .runtime import manual_seed, set_compute_dtype  # noqa: F401
from .swin import SwinTransformer  # noqa: F401
from .bert import MVLBert  # noqa: F401
from .model import (Conv_layer, MVLBertConfig, MVLBertConfigforVQA, MVLBertConfigForImageCaption,  # noqa: F401
                    MVLBertForImageCaption, MVLBertForPretraining, MVLBertForRetrieval, MVLBertForVQA,
                    MVLBertPretrainConfig, MVLBertRetrieval)
