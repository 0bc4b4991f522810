"""The shipped TunableOp selections: well-formed on CPU, loaded and harmless on the GPU."""
import csv
import os

import pytest
import torch


def test_tuned_file_is_wellformed():
  from mmt_amd import gemm_tuning
  rows = list(csv.reader(open(gemm_tuning.TUNED_FILE)))
  validators = {r[1]: r[2] for r in rows if r[0] == 'Validator'}
  assert {'PT_VERSION', 'HIPBLASLT_VERSION', 'ROCBLAS_VERSION', 'GCN_ARCH_NAME'} <= set(validators)
  assert validators['GCN_ARCH_NAME'].startswith('gfx950')
  ops = [r for r in rows if r[0] != 'Validator']
  assert ops and all(len(r) == 4 and float(r[3]) > 0 for r in ops)
  assert not torch.cuda.is_available() or True
  if not torch.cuda.is_available():
    assert gemm_tuning.ensure() is False          # nothing to do (and nothing enabled) without a GPU


@pytest.mark.gpu
def test_tuned_selections_load_and_match_default_gemm():
  from mmt_amd import gemm_tuning
  torch.manual_seed(0)
  x = torch.randn(16384, 768, device='cuda', dtype=torch.bfloat16)
  w = torch.randn(3072, 768, device='cuda', dtype=torch.bfloat16) * 0.05
  ref = (x[:64].float() @ w.float().t())
  if os.environ.get('MMT_GEMM_TUNING', '1') != '0' and os.environ.get('PYTORCH_TUNABLEOP_ENABLED') is None:
    assert gemm_tuning.ensure()
    import torch.cuda.tunable as tun
    assert tun.is_enabled() and not tun.tuning_is_enabled()
  y = torch.nn.functional.linear(x, w)
  assert float((y[:64].float() - ref).abs().max()) < 0.25          # bf16 rounding of O(10) outputs
