"""Dev tool: what the 8 global tokens cost -- attention forward / backward at the config-3 shape with and without them."""
import os, sys
root = os.environ.get('MMT_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
B, S, N, R = 4, 4096, 12, 32
dt = torch.bfloat16
q, k, v = (torch.randn(B, S, N, 64, device='cuda', dtype=dt) for _ in range(3))
emb = (torch.randn(R, N, 64, device='cuda') * 0.02).to(dt); bias = (torch.randn(R, N, device='cuda') * 0.02).to(dt)
def t(fn, n=30):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for ng, g0 in ((0, 0), (8, 3971), (8, 0), (32, 3971)):
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=g0, n_global=ng, id_mode=1, max_dist=12)
  kw = dict(pattern=pat, dropout_p=0.1, dropout_seed=1234)
  out, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
  dout = torch.randn_like(out)
  dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
  de = torch.zeros(R, N, 64, device='cuda'); db = torch.zeros(R, N, device='cuda')
  tf = t(lambda: mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw))
  tb = t(lambda: mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, grads_out=(dq, dk, dv), rel_grads_accum=(de, db), **kw))
  print(f'n_global {ng:3d} at {g0:4d}: fwd {tf:6.1f} us  bwd {tb:6.1f} us')
