"""Dev tool: correctness + speed of mmt_wgrad_accumulate vs torch.mm (not part of the product)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import fused
torch.manual_seed(0)
def t(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
K = 16384
for M, N in ((128, 256), (3072, 768), (768, 3072), (2304, 768), (768, 768)):
  dY = torch.randn(K, M, device='cuda', dtype=torch.bfloat16)
  X = torch.randn(K, N, device='cuda', dtype=torch.bfloat16)
  dw = torch.zeros(M, N, device='cuda')
  assert fused.wgrad_accumulate_(dw, dY, X)
  ref = dY.float().t() @ X.float()
  err = float((dw - ref).abs().max()) / float(ref.abs().max())
  ours = t(lambda: fused.wgrad_accumulate_(dw, dY, X))
  lib = t(lambda: torch.mm(dY.t(), X))
  fl = 2 * K * M * N
  print(f'M{M} N{N}: rel err {err:.2e}  ours {ours:.1f}us ({fl/ours/1e6:.0f} TF)  torch.mm {lib:.1f}us ({fl/lib/1e6:.0f} TF)', flush=True)
