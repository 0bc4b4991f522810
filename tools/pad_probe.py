import torch, time
torch.manual_seed(0)
M, V, H = 1024, 30522, 768
Vp = (V + 63) // 64 * 64
x = torch.randn(M, H, device='cuda').bfloat16(); w = torch.randn(V, H, device='cuda').bfloat16(); b = torch.randn(V, device='cuda').bfloat16()
def t(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize(); return round(e0.elapsed_time(e1) / n * 1e3, 1)
ref = torch.addmm(b, x, w.t())
buf = torch.zeros(M, Vp, device='cuda', dtype=torch.bfloat16); out = buf[:, :V]
try:
  torch.addmm(b, x, w.t(), out=out)
  print('addmm strided out ok', out.data_ptr() == buf.data_ptr(), float((out.float() - ref.float()).abs().max()), 'pad untouched', float(buf[:, V:].abs().max()))
  print('fwd contiguous us', t(lambda: torch.addmm(b, x, w.t())), 'strided-out us', t(lambda: torch.addmm(b, x, w.t(), out=out)))
except Exception as e:
  print('addmm strided out failed', e)
dy = torch.randn(M, V, device='cuda').bfloat16()
dbuf = torch.zeros(M, Vp, device='cuda', dtype=torch.bfloat16); dys = dbuf[:, :V]; dys.copy_(dy)
print('dgrad contiguous us', t(lambda: torch.mm(dy, w)), 'strided us', t(lambda: torch.mm(dys, w)), 'max diff', float((torch.mm(dy, w).float() - torch.mm(dys, w).float()).abs().max()))
print('wgrad contiguous us', t(lambda: torch.mm(dy.t(), x)), 'strided us', t(lambda: torch.mm(dys.t(), x)))
print('sum0 contiguous us', t(lambda: dy.sum(0, dtype=torch.float32)), 'strided us', t(lambda: dys.sum(0, dtype=torch.float32)))
ones = torch.ones(1, M, device='cuda', dtype=torch.bfloat16)
print('ones-gemm colsum us', t(lambda: torch.mm(ones, dys)))
