"""Dev tool (diagnostic -DMMT_STAMP build only): in-kernel s_memtime stamps of one band workgroup of the lean backward
kernels (MMT_DBG_MODE=0: dQ pass, 1: dK/dV pass): per wave, cycles between consecutive stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
B, S, N = 4, 4096, 12
dt = torch.bfloat16
qkv = torch.randn(B, S, 3, N, 64, device='cuda', dtype=dt)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
emb = (torch.randn(32, N, 64, device='cuda') * 0.02).to(dt); bias = (torch.randn(32, N, device='cuda') * 0.02).to(dt)
pat = mmt_amd.AttentionPattern(local_radius=64, global_start=S - 125, n_global=8, id_mode=1, max_dist=12)
kw = dict(pattern=pat, dropout_p=0.1, dropout_seed=1234)
out, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
dout = torch.randn_like(out)
dbg = torch.zeros(4 * 32, dtype=torch.int64, device='cuda')
os.environ['MMT_DBG_PTR'] = hex(dbg.data_ptr())
mode = int(os.environ.get('MMT_DBG_MODE', 0))
for _ in range(20): mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, **kw)
torch.cuda.synchronize()
d = dbg.cpu().view(4, 32)
names_q = ['start', 'loads issued', 'E image+barrier', 'delta/table', 't0', 't1', 't2', 't3', 't4', 't5', 't6', 't7', 'loop end', 'dQ stored', 'dE contracted', 'barrier', 'dE partial stored']
names_k = ['start', 'loads issued', 'E image+barrier', '-', 't0', 't1', 't2', 't3', 't4', 't5', 't6', 't7', 'loop end', 'stored']
nm = names_k if (mode & 1) else names_q
t00 = int(d[:, 0].min())
for w in range(4):
  row = d[w]; out_s = [f'w{w} start+{int(row[0]) - t00:6d}']; prev = int(row[0])
  for i in range(1, len(nm)):
    x = int(row[i])
    if x == 0: continue
    out_s.append(f'{nm[i]} {x - prev}'); prev = x
  out_s.append(f'| life {prev - int(row[0])}')
  print('  '.join(out_s))
