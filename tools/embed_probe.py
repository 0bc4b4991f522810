"""Dev tool: embedding assembly forward / backward timing at the config-3 shape (3969 of 4096 positions hold the
[PATCH] id, as in the reference's batches), for same-box A/B runs:  MMT_ROOT=_ab/A python tools/embed_probe.py"""
import os, sys
root = os.environ.get('MMT_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import fused
torch.manual_seed(0)
B, S, H, V = 4, 4096, 768, 30522
ids = torch.randint(1000, V, (B, S), device='cuda'); ids[:, 2:2 + 3969] = 5
seg = torch.zeros(B, S, dtype=torch.long, device='cuda'); seg[:, 3971:] = 1
wt = torch.nn.Parameter(torch.randn(V, H, device='cuda') * 0.02); st = torch.nn.Parameter(torch.randn(2, H, device='cuda') * 0.02)
g = torch.nn.Parameter(torch.ones(H, device='cuda')); b = torch.nn.Parameter(torch.zeros(H, device='cuda'))
for p_ in (wt, st, g, b): p_.grad = torch.zeros_like(p_)
patch = torch.randn(B, 3969, H, device='cuda').to(torch.bfloat16).requires_grad_(True)
dout = torch.randn(B, S, H, device='cuda').to(torch.bfloat16)
def step():
  out = fused.embed_assemble(ids, seg, wt, st, g, b, patch_proj=patch, p=0.1, seed=3)
  out.backward(dout)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): step()
e1.record(); torch.cuda.synchronize()
print(root[-6:], 'embed fwd+bwd us', round(e0.elapsed_time(e1) / 20 * 1e3, 1))
