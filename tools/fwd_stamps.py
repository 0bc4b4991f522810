"""Dev tool (diagnostic -DMMT_STAMP build only): prints the in-kernel s_memtime stamps of three workgroups of the
window forward kernel (attn_fwd_win.hip): per wave, cycles between consecutive stamps."""
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
dbg = torch.zeros(4 * 4 * 128, dtype=torch.int64, device='cuda')
os.environ['MMT_DBG_PTR'] = hex(dbg.data_ptr())
for _ in range(30): mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
torch.cuda.synchronize()
d = dbg.cpu().view(4, 4, 128)
rnames = ['start', 'prologue'] + [f'r{i}' for i in range(12)] + ['loop end']
names = ['start', 'dma issued', 'Q/E/table', 'vmcnt0', 'barrier', 'peeled', 't0', 't1', 't2', 't3', 't4', 't5', 't6', 't7', 'end']
g00 = min(int(d[wg, :, 0].min()) for wg in range(4) if int(d[wg, :, 0].min()) > 0)
for wg in range(4):
  t00 = int(d[wg, :, 0].min())
  print(f'wg{wg}: first stamp at +{t00 - g00} cycles after the earliest stamped workgroup; last stamp at +{int(d[wg].max()) - g00}')
  for w in range(4):
    row = d[wg, w]
    nm = rnames if wg == 3 else names
    out = [f'wg{wg} w{w} start+{int(row[0]) - t00:6d}']
    prev = int(row[0])
    for i in range(1, 15):
      x = int(row[i])
      if x == 0: continue
      out.append(f'{nm[i]} {x - prev}')
      prev = x
    out.append(f'| life {prev - int(row[0])}')
    print('  '.join(out))

print('sub-tile stamps of wg1 (cycles since the previous stamp): S issued | s2 done | max done | exp+sum done | dropout done | PV issued')
for w in range(4):
  row = d[1, w]
  for t in range(6):
    base = int(row[5 + t])
    sub = [int(row[16 + t * 8 + j]) for j in range(5)] + [int(row[6 + t])]
    if sub[0] == 0: continue
    prev = base; out = []
    for x in sub:
      out.append(x - prev); prev = x
    print(f'  w{w} tile{t}:', out)
