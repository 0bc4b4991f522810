"""Library-GEMM solution selection for the forward / dgrad products that stay on hipBLASLt / rocBLAS.

PyTorch's TunableOp picks, per GEMM shape, the fastest solution it measured instead of the heuristic's
first answer.  `tuned/gemm_gfx950_rocm72.csv` holds the selections measured on an MI355X with this image's
libraries for the BASELINE shapes (B*S = 16384 rows, BERT-base widths): FFN1 forward 68 -> 59 us, FFN2 forward
72 -> 53 us, FFN1 dgrad 84 -> 62 us; 16.7 -> 16.4 ms per train step.  The file's validator lines (PyTorch, HIP,
hipBLASLt, rocBLAS versions and the GCN arch string) must match the running stack, otherwise TunableOp ignores
it and every GEMM takes the default path; shapes that are not listed also take the default path (tuning itself
stays off: nothing is measured at run time).  Re-measure with
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=out.csv python bench.py
Opt out with MMT_GEMM_TUNING=0; if the PYTORCH_TUNABLEOP_* variables are set the user's settings win.
"""
from __future__ import annotations

import os
import tempfile

import torch

TUNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned', 'gemm_gfx950_rocm72.csv')
_state = {'done': False, 'loaded': False}


def ensure(path: str = TUNED_FILE) -> bool:
  """Idempotent; returns True when the tuned selections were loaded.  MMT_GEMM_TUNING_FILE names another selections
  file (A/B of a re-measurement against the shipped one)."""
  if _state['done']:
    return _state['loaded']
  _state['done'] = True
  path = os.environ.get('MMT_GEMM_TUNING_FILE', path)
  if (os.environ.get('MMT_GEMM_TUNING', '1') == '0' or os.environ.get('PYTORCH_TUNABLEOP_ENABLED') is not None
      or not torch.cuda.is_available() or not os.path.exists(path)):
    return False
  try:
    import torch.cuda.tunable as tun
    tun.enable(True)
    tun.tuning_enable(False)
    # results are written to get_filename() when the process ends: keep that away from the shipped file
    tun.set_filename(os.path.join(tempfile.gettempdir(), f'mmt_tunableop_{os.getpid()}.csv'))
    _state['loaded'] = bool(tun.read_file(path))
  except Exception:        # an unexpected TunableOp build: the default GEMM path is always valid
    _state['loaded'] = False
  return _state['loaded']
