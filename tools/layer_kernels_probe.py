"""Dev tool: the row-wise layer kernels (residual + dropout + LayerNorm forward / backward, bias + GELU, AdamW) alone at the
config-3 shapes: HIP-event time per call and the HBM rate their algorithmic bytes imply.
  python tools/layer_kernels_probe.py [rows] [H] [dropout_p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
from mmt_amd import fused
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
pd = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
dt = torch.bfloat16
torch.manual_seed(0)
o = torch.randn(rows, H, device='cuda', dtype=dt, requires_grad=True)
x = torch.randn(rows, H, device='cuda', dtype=dt, requires_grad=True)
bias = torch.nn.Parameter(torch.randn(H, device='cuda') * 0.1)
gamma = torch.nn.Parameter(torch.ones(H, device='cuda')); beta = torch.nn.Parameter(torch.zeros(H, device='cuda'))
for prm in (bias, gamma, beta):
  prm.grad = torch.zeros_like(prm)
flush = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device='cuda')     # larger than L2 + MALL


def timeit(fn, n=30, cold=True):
  for _ in range(3): fn()
  ts = []
  for _ in range(n):
    if cold: flush.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
  ts.sort()
  return ts[len(ts) // 2], ts[0]


def report(name, fn, nbytes):
  for cold in (True, False):
    med, mn = timeit(fn, cold=cold)
    print(f'{name:34s} {"cold" if cold else "warm"}: median {med:7.1f} us  min {mn:7.1f} us   {nbytes / med / 1e6:6.2f} TB/s (algorithmic {nbytes / 1e6:.0f} MB)', flush=True)

E = rows * H * 2
with torch.no_grad():
  pass
res = {}
def fwd():
  res['y'] = fused.residual_block(o, bias, x, gamma, beta, 1e-12, pd, 1234)
report('residual + dropout + LN forward', fwd, 4 * E)
xn, h = res['y']
gh, gx = torch.randn_like(h), torch.randn_like(xn)
def bwd():
  torch.autograd.grad((h, xn), (o, x), (gh, gx), retain_graph=True)
report('... backward (+ column sums)', bwd, 5 * E)
u = torch.randn(rows, 4 * H, device='cuda', dtype=dt)
b1 = torch.randn(4 * H, device='cuda')
report('bias + GELU forward', lambda: fused.bias_gelu_forward_(u, b1), 2 * 4 * E)
def ln():
  res['l'] = fused.layer_norm(x, gamma, beta)
report('LayerNorm forward', ln, 2 * E)
