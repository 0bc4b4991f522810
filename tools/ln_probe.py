"""Dev tool: residual-block forward / backward kernels (K6) at the config-3 shape, us per launch and achieved HBM rate:
  MMT_ROOT=_ab/A python tools/ln_probe.py ; python tools/ln_probe.py"""
import os, sys
root = os.environ.get('MMT_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import fused, _lib
rows, H = 16384, 768
torch.manual_seed(0)
o, x, dxn, dh = (torch.randn(rows, H, device='cuda').to(torch.bfloat16) for _ in range(4))
bias, gamma, beta = (torch.randn(H, device='cuda') for _ in range(3))
x_new, h, d_o, dx = (torch.empty_like(x) for _ in range(4))
mean, rstd = torch.empty(rows, device='cuda'), torch.empty(rows, device='cuda')
dbias, dg, db = (torch.zeros(H, device='cuda') for _ in range(3))
L = _lib.lib()
def t(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
p_ = lambda t_: t_.data_ptr()
st = torch.cuda.current_stream().cuda_stream
for p in (0.0, 0.1):
  d = fused._desc(x, 1e-12, p, 77)
  ws = fused._ws(d, x)
  fwd = lambda: _lib.check(L.mmt_residual_block_fwd(d, p_(o), p_(bias), p_(x), p_(gamma), p_(beta), p_(x_new), p_(h), p_(mean), p_(rstd), st))
  bwd = lambda: _lib.check(L.mmt_residual_block_bwd(d, p_(dxn), p_(dh), p_(x_new), p_(gamma), p_(mean), p_(rstd), p_(d_o), p_(dx), p_(dbias), p_(dg), p_(db), p_(ws), ws.numel(), st))
  tf, tb = t(fwd), t(bwd)
  mb = rows * H * 2 / 1e6
  print(f'{root[-6:]} p={p}: fwd {tf:.1f} us ({4 * mb / tf:.2f} TB/s over 4 arrays)  bwd+reduce {tb:.1f} us ({5 * mb / tb:.2f} TB/s over 5 arrays)')
