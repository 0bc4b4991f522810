"""Dev tool: attention forward (config 3 shape) timing split per kernel with HIP events, for A/B of work decompositions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
B, S, N = 4, 4096, 12
dt = torch.bfloat16
q, k, v = (torch.randn(B, S, N, 64, device='cuda', dtype=dt) for _ in range(3))
emb = (torch.randn(32, N, 64, device='cuda') * 0.02).to(dt); bias = (torch.randn(32, N, device='cuda') * 0.02).to(dt)
pat = mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=1, max_dist=12)
def t(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for drop in (0.0, 0.1):
  kw = dict(pattern=pat, dropout_p=drop, dropout_seed=1234)
  print(os.environ.get('MMT_LEAN_BPW', 'auto'), 'drop', drop, 'fwd us', round(t(lambda: mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)), 1))
