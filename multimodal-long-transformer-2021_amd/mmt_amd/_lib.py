"""ctypes binding of the C ABI declared in include/mmt_attn.h.

The shared library is built in-tree (csrc/Makefile -> mmt_amd/libmmt_attn.so).  There is
no CPU fallback: if the library is missing, loading fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), 'csrc')
LIB_PATH = os.path.join(_HERE, 'libmmt_attn.so')

MMT_ABI_VERSION = 4
MMT_F32, MMT_BF16 = 0, 1
MMT_IDS_NONE, MMT_IDS_1D, MMT_IDS_2D = 0, 1, 2
MMT_FLAG_SCALE_BEFORE_ADD = 1
MMT_FLAG_ACCUM_REL_GRADS = 2
# mmt_attn_desc.tuning (include/mmt_attn.h): kernel-selection switches; 0 = the library's defaults
MMT_TUNE_FWD_WALK = 0x01
MMT_TUNE_FWD_PWIN = 0x100
MMT_TUNE_FWD_ROWS_ONE_WG = 0x200
MMT_TUNE_FWD_NO_WIN = 0x02
MMT_TUNE_FWD_FORCE_WIN = 0x04
MMT_TUNE_BWD_NO_HANDOVER = 0x08
MMT_TUNE_BWD_HO_PER_WAVE = 0x10
MMT_TUNE_BWD_NO_PEEL_DQ = 0x20
MMT_TUNE_BWD_NO_PEEL_DKV = 0x40
MMT_TUNE_BWD_DQ_PLANE_MAJOR = 0x80

EXPORTS = ('mmt_abi_version', 'mmt_last_error', 'mmt_write_step_scalars', 'mmt_workspace_bytes', 'mmt_attn_fwd',
           'mmt_attn_bwd', 'mmt_side_inputs',
           # include/mmt_layer.h
           'mmt_layer_workspace_bytes', 'mmt_ln_fwd', 'mmt_ln_bwd', 'mmt_residual_block_fwd',
           'mmt_residual_block_bwd', 'mmt_bias_gelu_fwd', 'mmt_bias_gelu_bwd', 'mmt_colsum_reduce', 'mmt_colsum_reduce_batch', 'mmt_accumulate_grad', 'mmt_grad_clip_scale', 'mmt_adamw_step', 'mmt_wgrad_accumulate',
           'mmt_wgrad_bias_accumulate', 'mmt_wgrad_grouped', 'mmt_wgrad_group_workspace_bytes', 'mmt_wgrad_workspace_bytes', 'mmt_wgrad_set_cu_budget', 'mmt_embed_fwd', 'mmt_embed_bwd',
           'mmt_embed_workspace_bytes', 'mmt_xent_fwd', 'mmt_xent_fwd_argmax', 'mmt_xent_bwd', 'mmt_xent_bwd_scaled', 'mmt_weighted_loss', 'mmt_colsum', 'mmt_colsum_workspace_bytes', 'mmt_ln_bwd_add', 'mmt_ffn_gelu_gemm', 'mmt_ffn_dgelu_gemm', 'mmt_ffn_set_cu_budget')


class EmbedDesc(ctypes.Structure):
  _fields_ = [('rows', ctypes.c_int64), ('S', ctypes.c_int32), ('H', ctypes.c_int32), ('dtype', ctypes.c_int32),
              ('vocab', ctypes.c_int32), ('seg_vocab', ctypes.c_int32), ('patch_start', ctypes.c_int32),
              ('n_patch', ctypes.c_int32), ('eps', ctypes.c_float), ('dropout_p', ctypes.c_float),
              ('accumulate', ctypes.c_int32), ('dropout_seed', ctypes.c_uint64), ('dropout_epoch', ctypes.c_void_p)]


class RowsDesc(ctypes.Structure):
  _fields_ = [('rows', ctypes.c_int64), ('H', ctypes.c_int32), ('dtype', ctypes.c_int32),
              ('eps', ctypes.c_float), ('dropout_p', ctypes.c_float),
              ('dropout_seed', ctypes.c_uint64), ('accumulate', ctypes.c_int32),
              ('defer_reduce', ctypes.c_int32), ('dropout_epoch', ctypes.c_void_p)]


class MaskDesc(ctypes.Structure):
  _fields_ = [('valid_len', ctypes.c_void_p), ('local_radius', ctypes.c_int32),
              ('global_start', ctypes.c_int32), ('n_global', ctypes.c_int32),
              ('id_mode', ctypes.c_int32), ('max_dist', ctypes.c_int32),
              ('patches_per_row', ctypes.c_int32), ('core_layers', ctypes.c_int32),
              ('global_index', ctypes.c_void_p)]


class AttnDesc(ctypes.Structure):
  _fields_ = [('B', ctypes.c_int32), ('S', ctypes.c_int32), ('N', ctypes.c_int32),
              ('D', ctypes.c_int32), ('R', ctypes.c_int32), ('dtype', ctypes.c_int32),
              ('q_stride', ctypes.c_int64 * 3), ('k_stride', ctypes.c_int64 * 3),
              ('v_stride', ctypes.c_int64 * 3), ('o_stride', ctypes.c_int64 * 3),
              ('scale', ctypes.c_float), ('mask_value', ctypes.c_float),
              ('flags', ctypes.c_uint32), ('dropout_p', ctypes.c_float),
              ('dropout_seed', ctypes.c_uint64), ('mask', MaskDesc),
              ('dropout_epoch', ctypes.c_void_p), ('tuning', ctypes.c_uint32), ('sync_words', ctypes.c_uint32),
              ('sync', ctypes.c_void_p)]


class AdamwDesc(ctypes.Structure):
  _fields_ = [('n', ctypes.c_int64), ('lr', ctypes.c_float), ('beta1', ctypes.c_float),
              ('beta2', ctypes.c_float), ('eps', ctypes.c_float), ('bias_correction1', ctypes.c_float),
              ('bias_correction2', ctypes.c_float), ('zero_grad', ctypes.c_int32),
              ('reserved', ctypes.c_int32), ('hyper', ctypes.c_void_p)]


class WgradProblem(ctypes.Structure):
  _fields_ = [('dw', ctypes.c_void_p), ('ldw', ctypes.c_int64), ('dbias', ctypes.c_void_p), ('dy', ctypes.c_void_p),
              ('ldy', ctypes.c_int64), ('x', ctypes.c_void_p), ('ldx', ctypes.c_int64), ('M', ctypes.c_int32),
              ('N', ctypes.c_int32)]


class ColsumItem(ctypes.Structure):
  _fields_ = [('workspace', ctypes.c_void_p), ('o0', ctypes.c_void_p), ('o1', ctypes.c_void_p), ('o2', ctypes.c_void_p),
              ('rows', ctypes.c_int64), ('H', ctypes.c_int32), ('kind', ctypes.c_int32), ('accumulate', ctypes.c_int32),
              ('reserved', ctypes.c_int32)]


class MmtError(RuntimeError):
  """Raised when an entry point of libmmt_attn returns a negative code."""

  def __init__(self, code: int, message: str):
    super().__init__(f'libmmt_attn error {code}: {message}')
    self.code = code


def build(force: bool = False) -> str:
  """Compiles the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
  srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(('.hip', '.h'))]
  srcs.append(os.path.join(os.path.dirname(os.path.dirname(_HERE)), 'include', 'mmt_attn.h'))
  stale = (not os.path.exists(LIB_PATH) or
           any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs))
  if force or stale:
    subprocess.check_call(['make', '-s', '-j4', '-C', _CSRC])
  return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} is missing: the HIP extension has not been built '
        '(run `python -c "import __graft_entry__ as g; g.build()"` or `make -C csrc`). '
        'There is no CPU fallback for the hot path.')
  L = ctypes.CDLL(LIB_PATH)
  vp, i32p = ctypes.c_void_p, ctypes.c_void_p
  L.mmt_abi_version.restype = ctypes.c_int
  L.mmt_abi_version.argtypes = []
  L.mmt_last_error.restype = ctypes.c_char_p
  L.mmt_last_error.argtypes = []
  L.mmt_workspace_bytes.restype = ctypes.c_size_t
  L.mmt_workspace_bytes.argtypes = [ctypes.POINTER(AttnDesc)]
  L.mmt_attn_fwd.restype = ctypes.c_int
  L.mmt_attn_fwd.argtypes = [ctypes.POINTER(AttnDesc), vp, vp, vp, vp, vp, i32p, i32p, vp, vp,
                             vp, ctypes.c_size_t, vp]
  L.mmt_attn_bwd.restype = ctypes.c_int
  L.mmt_attn_bwd.argtypes = [ctypes.POINTER(AttnDesc), vp, vp, vp, vp, vp, i32p, i32p, vp, vp, vp,
                             vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
  L.mmt_side_inputs.restype = ctypes.c_int
  L.mmt_side_inputs.argtypes = [ctypes.POINTER(MaskDesc), ctypes.c_int32, ctypes.c_int32, i32p,
                                i32p, ctypes.c_int32, i32p, i32p, i32p, vp]
  rd = ctypes.POINTER(RowsDesc)
  L.mmt_layer_workspace_bytes.restype = ctypes.c_size_t
  L.mmt_layer_workspace_bytes.argtypes = [rd]
  for name, nargs in (('mmt_ln_fwd', 7), ('mmt_ln_bwd', 11), ('mmt_residual_block_fwd', 10),
                      ('mmt_residual_block_bwd', 14), ('mmt_bias_gelu_fwd', 4), ('mmt_bias_gelu_bwd', 8)):
    fn = getattr(L, name)
    fn.restype = ctypes.c_int
    fn.argtypes = [rd] + [vp] * nargs
  L.mmt_ln_bwd.argtypes = [rd] + [vp] * 9 + [ctypes.c_size_t, vp]
  L.mmt_ln_bwd_add.restype = ctypes.c_int
  L.mmt_ln_bwd_add.argtypes = [rd] + [vp] * 10 + [ctypes.c_size_t, vp]
  L.mmt_residual_block_bwd.argtypes = [rd] + [vp] * 12 + [ctypes.c_size_t, vp]
  L.mmt_bias_gelu_bwd.argtypes = [rd] + [vp] * 6 + [ctypes.c_size_t, vp]
  L.mmt_colsum_reduce.restype = ctypes.c_int
  L.mmt_colsum_reduce.argtypes = [rd, ctypes.c_int32, vp, vp, vp, vp, vp]
  L.mmt_colsum_reduce_batch.restype = ctypes.c_int
  L.mmt_colsum_reduce_batch.argtypes = [ctypes.c_int32, ctypes.POINTER(ColsumItem), vp]
  L.mmt_wgrad_accumulate.restype = ctypes.c_int
  L.mmt_wgrad_accumulate.argtypes = [vp, ctypes.c_int64, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_int64, vp, ctypes.c_size_t, vp]
  L.mmt_wgrad_bias_accumulate.restype = ctypes.c_int
  L.mmt_wgrad_bias_accumulate.argtypes = [vp, ctypes.c_int64, vp, vp, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_int32, ctypes.c_int64, vp, ctypes.c_size_t, vp]
  wp = ctypes.POINTER(WgradProblem)
  L.mmt_wgrad_group_workspace_bytes.restype = ctypes.c_size_t
  L.mmt_wgrad_group_workspace_bytes.argtypes = [ctypes.c_int32, wp, ctypes.c_int64]
  L.mmt_wgrad_grouped.restype = ctypes.c_int
  L.mmt_wgrad_grouped.argtypes = [ctypes.c_int32, wp, ctypes.c_int64, vp, ctypes.c_size_t, vp]
  ed = ctypes.POINTER(EmbedDesc)
  L.mmt_embed_fwd.restype = ctypes.c_int
  L.mmt_embed_fwd.argtypes = [ed] + [vp] * 13
  L.mmt_embed_bwd.restype = ctypes.c_int
  L.mmt_embed_bwd.argtypes = [ed] + [vp] * 12 + [ctypes.c_size_t, vp]
  L.mmt_embed_workspace_bytes.restype = ctypes.c_size_t
  L.mmt_embed_workspace_bytes.argtypes = [ed]
  L.mmt_xent_fwd_argmax.restype = ctypes.c_int
  L.mmt_xent_fwd_argmax.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int64, vp, vp, vp, vp, vp]
  L.mmt_xent_fwd.restype = ctypes.c_int
  L.mmt_xent_fwd.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int64, vp, vp, vp, vp]
  L.mmt_colsum_workspace_bytes.restype = ctypes.c_size_t
  L.mmt_colsum_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
  L.mmt_colsum.restype = ctypes.c_int
  L.mmt_colsum.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int64, vp, ctypes.c_int32, vp,
                           ctypes.c_size_t, vp]
  L.mmt_xent_bwd_scaled.restype = ctypes.c_int
  L.mmt_xent_bwd_scaled.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int64, vp, vp, vp, vp,
                                    vp, ctypes.c_int64, vp]
  L.mmt_weighted_loss.restype = ctypes.c_int
  L.mmt_weighted_loss.argtypes = [ctypes.c_int64, vp, vp, vp, vp, ctypes.c_int64, vp, vp, vp]
  L.mmt_xent_bwd.restype = ctypes.c_int
  L.mmt_xent_bwd.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int64, vp, vp, vp, vp,
                             ctypes.c_int64, vp]
  i64 = ctypes.c_int64
  L.mmt_ffn_gelu_gemm.restype = ctypes.c_int
  L.mmt_ffn_gelu_gemm.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, vp]
  L.mmt_ffn_dgelu_gemm.restype = ctypes.c_int
  L.mmt_ffn_dgelu_gemm.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, i64, i64, i64, i64, vp]
  L.mmt_ffn_set_cu_budget.restype = None
  L.mmt_ffn_set_cu_budget.argtypes = [ctypes.c_int32]
  L.mmt_wgrad_set_cu_budget.restype = None
  L.mmt_wgrad_set_cu_budget.argtypes = [ctypes.c_int32]
  L.mmt_wgrad_workspace_bytes.restype = ctypes.c_size_t
  L.mmt_wgrad_workspace_bytes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64]
  L.mmt_adamw_step.restype = ctypes.c_int
  L.mmt_adamw_step.argtypes = [ctypes.POINTER(AdamwDesc)] + [vp] * 8
  L.mmt_grad_clip_scale.restype = ctypes.c_int
  L.mmt_grad_clip_scale.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_float,
                                    ctypes.c_float, vp, vp, vp, ctypes.c_size_t, vp]
  L.mmt_accumulate_grad.restype = ctypes.c_int
  L.mmt_accumulate_grad.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_int64, vp]
  L.mmt_write_step_scalars.restype = ctypes.c_int
  L.mmt_write_step_scalars.argtypes = [vp, vp, ctypes.c_uint64, ctypes.c_float, ctypes.c_float, ctypes.c_float, vp]
  if L.mmt_abi_version() != MMT_ABI_VERSION:
    raise ImportError('libmmt_attn ABI version mismatch')
  _lib = L
  return L


def check(rc: int) -> None:
  if rc != 0:
    raise MmtError(rc, lib().mmt_last_error().decode())
