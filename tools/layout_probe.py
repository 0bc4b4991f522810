"""Dev tool: the attention calls on token-major q/k/v/dout ([B,S,N,D] contiguous: a head's row is 128 bytes every 1.5-4.6 KB,
what the QKV GEMM writes) against head-major ones ([B,N,S,D] contiguous, viewed as [B,S,N,D]: a head's rows are contiguous)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
B, S, N = 4, 4096, 12
dt = torch.bfloat16
emb = (torch.randn(32, N, 64, device='cuda') * 0.02).to(dt); bias = (torch.randn(32, N, device='cuda') * 0.02).to(dt)
pat = mmt_amd.AttentionPattern(local_radius=64, global_start=S - 125, n_global=8, id_mode=1, max_dist=12)
kw = dict(pattern=pat, dropout_p=0.1, dropout_seed=1234)
def timeit(fn, n=100):
  for _ in range(20): fn()
  torch.cuda.synchronize()
  res = []
  for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / n * 1e3)
  return min(res)
for name in ('token-major (fused qkv buffer)', 'token-major (separate)', 'head-major'):
  if name.startswith('token-major (fused'):
    qkv = torch.randn(B, S, 3, N, 64, device='cuda', dtype=dt)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    dout = torch.randn(B, S, N, 64, device='cuda', dtype=dt)
  elif name.startswith('token-major'):
    q, k, v, dout = (torch.randn(B, S, N, 64, device='cuda', dtype=dt) for _ in range(4))
  else:
    q, k, v, dout = (torch.randn(B, N, S, 64, device='cuda', dtype=dt).permute(0, 2, 1, 3) for _ in range(4))
  out, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
  go = (torch.empty_like(q), torch.empty_like(k), torch.empty_like(v))
  f = timeit(lambda: mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw))
  dout = dout if dout.stride() == out.stride() else dout.contiguous().as_strided(out.shape, out.stride())
  b = timeit(lambda: mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, grads_out=go, **kw), 50)
  print(f'{name:32s} fwd {f:6.1f} us   bwd {b:6.1f} us   out strides {tuple(out.stride())}')
