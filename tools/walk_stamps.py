"""Dev tool (diagnostic -DMMT_STAMP build only): in-kernel s_memtime stamps of two workgroups of the plane-walk forward
kernel (attn_fwd_walk.hip): per wave and super-step, cycles from the step's start to (a) the end of its tile work /
rows step, (b) its arrival at the barrier, (c) its release from the barrier.
  python tools/walk_stamps.py [globals] [dropout_p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
B, S, N = 4, 4096, 12
ng = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pdrop = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
dt = torch.bfloat16
qkv = torch.randn(B, S, 3, N, 64, device='cuda', dtype=dt)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
emb = (torch.randn(32, N, 64, device='cuda') * 0.02).to(dt); bias = (torch.randn(32, N, device='cuda') * 0.02).to(dt)
pat = mmt_amd.AttentionPattern(local_radius=64, global_start=S - 125, n_global=ng, id_mode=1, max_dist=12)
kw = dict(pattern=pat, dropout_p=pdrop, dropout_seed=1234)
dbg = torch.zeros(2 * 8 * 64, dtype=torch.int64, device='cuda')
os.environ['MMT_DBG_PTR'] = hex(dbg.data_ptr())
for _ in range(30): mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
torch.cuda.synchronize()
d = dbg.cpu().view(2, 8, 64)
for wg in range(2):
  t00 = int(d[wg, :, 0].min())
  print(f'workgroup {wg}: set-up {int(d[wg, 0, 1]) - t00} cycles; life {int(d[wg].max()) - t00} cycles')
  for s in range(15):
    if int(d[wg, 0, 2 + 4 * s]) == 0: break
    parts = []
    for w in range(8):
      a, b, c, e = (int(d[wg, w, 2 + 4 * s + i]) for i in range(4))
      role = 'S' if ((s - (w >> 1) + 1 + 2) & 3) == 3 else 'a'      # approximate: phases are relative to T0 (even jb/2)
      parts.append(f'w{w}:{(b - a) if b else 0:5d}/{c - a:5d}/{e - a:5d}')
    print(f' step {s:2d} (+{int(d[wg, 0, 2 + 4 * s]) - t00:6d})  ' + '  '.join(parts))
print('per wave: work-end / barrier-arrival / barrier-release, cycles since the step start')
