"""mmt_amd -- MI355X-native hot path of googleinterns/multimodal-long-transformer-2021.

Host-side mirror (Python, like the reference) of the relative-attention path of
`MmtEncoder`, over a C ABI (include/mmt_attn.h) into hand-written HIP kernels for gfx950.
"""
from . import _lib
from .ops import (AttentionPattern, relative_attention, relative_attention_backward, relative_attention_qkv,
                  relative_attention_forward, side_inputs)

from .encoder import MmtEncoder
from .models import MmtClassificationModel, MmtPretrainingModel
from .benchmarks import make_train_step_bench
from . import configs, distribute, fused, input_utils, layers, optimization, registry_imports, tasks

__all__ = ['MmtEncoder', 'MmtPretrainingModel', 'MmtClassificationModel', 'make_train_step_bench',
           'configs', 'tasks', 'distribute', 'optimization', 'layers', 'input_utils',
           'AttentionPattern', 'relative_attention', 'relative_attention_forward',
           'relative_attention_backward', 'side_inputs', '_lib']
