"""ORACLE (test infrastructure) -- ctypes loader for the plain-C restatement
`oracle/mmt_oracle.c` (built by `oracle/Makefile` / `__graft_entry__.build()`)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
  so = os.path.join(_HERE, 'libmmt_oracle.so')
  src = os.path.join(_HERE, 'mmt_oracle.c')
  if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(['make', '-s', '-C', _HERE, 'libmmt_oracle.so'])
  return so


def lib() -> ctypes.CDLL:
  global _LIB
  if _LIB is None:
    _LIB = ctypes.CDLL(build())
  return _LIB


def _i32(a):
  return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _f32(a):
  return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) if a is not None else None


def relative_ids(S, id_mode, max_dist, P=0, r=0):
  out = np.empty((S, S), np.int32)
  lib().oracle_relative_ids(_i32(out), S, id_mode, max_dist, P, r)
  return out


def att_mask(S, valid_len, radius=-1, global_start=0, n_global=0):
  out = np.empty((S, S), np.int32)
  lib().oracle_att_mask(_i32(out), S, valid_len, radius, global_start, n_global)
  return out


def segment_ids(S, img_wp, txt_wp):
  out = np.empty((S,), np.int32)
  lib().oracle_segment_ids(_i32(out), S, img_wp, txt_wp)
  return out


def rel_attention_fwd(q, k, v, rel_emb, rel_bias, att_mask_, rel_ids_, scale=None,
                      mask_value=-10000.0, scale_after_add=True, acc64=False):
  q = np.ascontiguousarray(q, np.float32); k = np.ascontiguousarray(k, np.float32)
  v = np.ascontiguousarray(v, np.float32)
  B, S, N, D = q.shape
  R = 0 if rel_emb is None else rel_emb.shape[0]
  rel_emb = None if rel_emb is None else np.ascontiguousarray(rel_emb, np.float32)
  rel_bias = None if rel_bias is None else np.ascontiguousarray(rel_bias, np.float32)
  def prep(a):
    if a is None:
      return None, 0
    a = np.ascontiguousarray(a, np.int32)
    return a, (S * S if a.ndim == 3 and a.shape[0] > 1 else 0)
  am, ams = prep(att_mask_)
  ri, ris = prep(rel_ids_)
  out = np.empty_like(q); lse = np.empty((B, N, S), np.float32)
  if scale is None:
    scale = 1.0 / np.sqrt(D)
  f = lib().oracle_rel_attention_fwd
  f.restype = ctypes.c_int
  f.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_long, ctypes.c_long, ctypes.c_void_p,
                ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_float,
                ctypes.c_int, ctypes.c_int]
  ptr = lambda a: None if a is None else a.ctypes.data
  rc = f(ptr(q), ptr(k), ptr(v), ptr(rel_emb), ptr(rel_bias), ptr(am), ptr(ri), ams, ris,
         ptr(out), ptr(lse), B, S, N, D, R, float(scale), float(mask_value),
         int(scale_after_add), int(acc64))
  if rc != 0:
    raise MemoryError('oracle_rel_attention_fwd failed')
  return out, lse
