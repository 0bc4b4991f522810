"""Dev tool: the tied-logits dgrad dx[1024,768] = dy[1024,30522] . W[30522,768] (K = 30522: 12-64 output tiles for 256
CUs) as one library GEMM against split-K forms built from torch ops."""
import torch, time
torch.manual_seed(0)
M, V, H = 1024, 30522, 768
dy = (torch.randn(M, V, device='cuda') * 0.01).bfloat16()
W = (torch.randn(V, H, device='cuda') * 0.02).bfloat16()
def timeit(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(n): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
ref = torch.mm(dy.float(), W.float())
print('mm           %.1f us' % timeit(lambda: torch.mm(dy, W)))
for c in (2, 3, 6):
  k = V // c
  a = dy.as_strided((c, M, k), (k, V, 1))
  b = W.view(c, k, H)
  f = lambda: torch.bmm(a, b).sum(0, dtype=torch.float32)
  err = (f() - ref).abs().max().item() / ref.abs().max().item()
  print('bmm x%d + sum %.1f us   rel err %.2e' % (c, timeit(f), err))
  # fp32 accumulate through baddbmm into an fp32 buffer is not available for bf16 inputs; try addmm chain
err = (torch.mm(dy, W).float() - ref).abs().max().item() / ref.abs().max().item()
print('mm rel err %.2e' % err)
# transposed-A form: dy^T contiguous [V, M]
dyT = dy.t().contiguous()
print('mm(dyT.t(), W) %.1f us' % timeit(lambda: torch.mm(dyT.t(), W)))
