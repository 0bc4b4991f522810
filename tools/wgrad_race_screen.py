"""Dev tool: race screen of the K7 ring schedule -- the grouped weight-gradient launch repeated many times on the
same operands (several K lengths) must give bit-identical results every time and match the fp32 reference."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import _lib
L = _lib.lib()
torch.manual_seed(0)
shapes = [(768, 3072, False), (3072, 768, True), (768, 768, False), (2304, 768, True)]
for K in (16384, 2048, 4160, 192):
  keep = []
  probs = (_lib.WgradProblem * 4)()
  for q, (M, N, wb) in zip(probs, shapes):
    dy = torch.randn(K, M, device='cuda', dtype=torch.bfloat16); x = torch.randn(K, N, device='cuda', dtype=torch.bfloat16)
    dw = torch.zeros(M, N, device='cuda'); db = torch.zeros(M, device='cuda') if wb else None
    q.dw, q.ldw, q.dbias = dw.data_ptr(), N, (None if db is None else db.data_ptr())
    q.dy, q.ldy, q.x, q.ldx, q.M, q.N = dy.data_ptr(), M, x.data_ptr(), N, M, N
    keep.append((dy, x, dw, db))
  ws = torch.empty(max(L.mmt_wgrad_group_workspace_bytes(4, probs, K), 16), dtype=torch.uint8, device='cuda')
  first = None
  for it in range(150):
    for _, _, dw, db in keep:
      dw.zero_()
      if db is not None: db.zero_()
    _lib.check(L.mmt_wgrad_grouped(4, probs, K, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    cur = [t[2].clone() for t in keep] + [t[3].clone() for t in keep if t[3] is not None]
    if first is None:
      first = cur
      for (dy, x, dw, db) in keep:
        ref = dy.float().t() @ x.float()
        assert float((dw - ref).abs().max()) / float(ref.abs().max()) < 1e-5
    else:
      assert all(torch.equal(a, b) for a, b in zip(first, cur)), (K, it)
  print('K', K, 'ok: 150 identical runs')
