"""Dev tool: attention forward / backward timing with 2-D relative ids at the config-3 shape (lean kernels; r = 1 is
the reference's *_2d*.yaml setting and runs at table width 32, r = 2 needs width 64), with and without dropout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
B, S, N, R = 4, 4096, 12, 49
dt = torch.bfloat16
q, k, v = (torch.randn(B, S, N, 64, device='cuda', dtype=dt) for _ in range(3))
emb = (torch.randn(R, N, 64, device='cuda') * 0.02).to(dt); bias = (torch.randn(R, N, device='cuda') * 0.02).to(dt)
def t(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
for name, pat in (('1-D ids', mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=1, max_dist=12)),
                  ('2-D ids r=1', mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=2, max_dist=12, patches_per_row=63, core_layers=1)),
                  ('2-D ids r=2', mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=2, max_dist=12, patches_per_row=63, core_layers=2))):
  e, bb = (emb[:32].contiguous(), bias[:32].contiguous()) if name == '1-D ids' else (emb, bias)
  out, lse = mmt_amd.relative_attention_forward(q, k, v, e, bb, pattern=pat)
  dout = torch.randn_like(out)
  for dp in (0.0, 0.1):
    kw = dict(pattern=pat, dropout_p=dp, dropout_seed=5)
    out, lse = mmt_amd.relative_attention_forward(q, k, v, e, bb, **kw)
    print(name, 'dropout', dp, 'fwd us', round(t(lambda: mmt_amd.relative_attention_forward(q, k, v, e, bb, **kw)), 1),
          'bwd us', round(t(lambda: mmt_amd.relative_attention_backward(dout, q, k, v, e, bb, out, lse, **kw)), 1))
