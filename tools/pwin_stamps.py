"""Dev tool (diagnostic -DMMT_STAMP build only): in-kernel s_memtime stamps of two workgroups of the sliding-window
forward kernel (attn_fwd_pwin.hip): per wave and block, cycles from the block's start to each of its landmarks.
  python tools/pwin_stamps.py [globals] [dropout_p]"""
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
names = ['issued', 'landed', 'bar1', 'rows2', 'tiles', 'bar2', 'bar3', 'stored', 'end']
for wg in range(2):
  t00 = int(d[wg, :, 0].min())
  print(f'workgroup {wg}: life {int(d[wg].max()) - t00} cycles')
  for blk in range(6):
    if int(d[wg, 0, 10 * blk]) == 0 and blk > 0: break
    print(f' block {blk} (+{int(d[wg, 0, 10 * blk]) - t00}):   ' + ' '.join(f'{n:>7s}' for n in names))
    for w in range(8):
      a = int(d[wg, w, 10 * blk])
      print(f'   wave {w} ({"AB"[w >> 2]}{w & 3})        ' + ' '.join(f'{int(d[wg, w, 10 * blk + i]) - a:7d}' for i in range(1, 10)))
