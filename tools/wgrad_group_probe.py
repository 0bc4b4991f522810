"""Dev tool: the grouped weight-gradient launch alone at config 3's shapes (K = 16384) for groups of 1, 2, 3, 5, 7 encoder
blocks: HIP-event time per launch, per 256-tile round, and PFLOP/s."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import _lib
L = _lib.lib()
K = 16384
shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
torch.manual_seed(0)
ops = {}
for M, N in shapes:
  ops[(M, N)] = (torch.randn(K, M, device='cuda', dtype=torch.bfloat16), torch.randn(K, N, device='cuda', dtype=torch.bfloat16))
for blocks in (1, 2, 3, 5, 7):
  n = 4 * blocks
  probs, keep = (_lib.WgradProblem * n)(), []
  for q, (M, N) in zip(probs, shapes * blocks):
    dy, x = ops[(M, N)]
    dw = torch.zeros(M, N, device='cuda')
    q.dw, q.ldw, q.dbias = dw.data_ptr(), N, None
    q.dy, q.ldy, q.x, q.ldx, q.M, q.N = dy.data_ptr(), M, x.data_ptr(), N, M, N
    keep.append(dw)
  need = L.mmt_wgrad_group_workspace_bytes(n, probs, K)
  ws = torch.empty(max(need, 16), dtype=torch.uint8, device='cuda')
  st = torch.cuda.current_stream().cuda_stream
  f = lambda: _lib.check(L.mmt_wgrad_grouped(n, probs, K, ws.data_ptr(), ws.numel(), st))
  for _ in range(3): f()
  ts = []
  for rnd in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5 * 1e3)
  us = sorted(ts)[2]
  tiles = 108 * blocks
  flops = 2.0 * K * sum(M * N for M, N in shapes) * blocks
  print(f'{blocks} block(s): {tiles:4d} tiles  {us:8.1f} us per launch  {us / (tiles / 256):7.1f} us per 256 tiles  {flops / (us * 1e-6) / 1e15:.2f} PFLOP/s   slabs {need / 1e6:.0f} MB', flush=True)
