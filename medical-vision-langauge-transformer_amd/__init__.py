"""MI355X-native (gfx950) implementation of the MVLT vision-language hot path.

Drop-in ``nn.Module`` classes with the reference's constructor / ``forward``
signatures and state-dict keys, backed by hand-written HIP kernels behind the
C-ABI in ``include/mvlt_hip.h`` (``libmvlt_hip.so``).  See DESIGN.md.
"""
import os as _os

# Hardware queues.  The HIP runtime maps a process' streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The training
# step runs two streams of its own (dgrad chain / weight gradients); a process group over RCCL brings RCCL's streams: with more
# streams than queues two of them share a queue and serialise -- measured on one rank: 12.65 ms per step against 11.63 ms
# with nothing but the communicator created, 11.68 ms with 8 queues (profiles/r5_ddp_one_rank.md).  The runtime reads the
# variable when it initialises, i.e. at the first HIP call of the process: set here, on import, unless the user chose a value.
HWQ_SET_LATE = False          # True: the variable was unset AND the HIP runtime had already started when this package was imported
if "GPU_MAX_HW_QUEUES" not in _os.environ:
    import torch as _torch
    HWQ_SET_LATE = bool(_torch.cuda.is_initialized())
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"

from . import _lib  # noqa: F401
from .runtime import manual_seed, set_compute_dtype  # noqa: F401
from .swin import SwinTransformer  # noqa: F401
from .bert import MVLBert  # noqa: F401
from .model import (Conv_layer, MVLBertConfig, MVLBertConfigforVQA, MVLBertConfigForImageCaption,  # noqa: F401
                    MVLBertForImageCaption, MVLBertForPretraining, MVLBertForRetrieval, MVLBertForVQA,
                    MVLBertPretrainConfig, MVLBertRetrieval)
