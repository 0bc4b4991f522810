"""Dev tool: the Dense-layer GEMMs of csrc/dense_gemm.hip against the library (torch -> hipBLASLt) on the encoder's
shapes at config 3 (M = B S = 16384): results (max error against an fp32 product of the same bf16 operands) and HIP-event
times in interleaved rounds, both as PFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'multimodal-long-transformer-2021_amd'))
import torch, torch.nn.functional as F
from mmt_amd import _lib, gemm_tuning
try:
  gemm_tuning.ensure()
except Exception as e:
  print('gemm tuning not enabled:', e)
import ctypes
L = ctypes.CDLL(os.environ.get('DENSE_SO', '/tmp/libdense.so'))      # built from tools/experiments/dense_gemm.hip (README.md)
vp, i64 = ctypes.c_void_p, ctypes.c_int64
L.mmt_dense_fwd.argtypes = [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, ctypes.c_int32, vp]
L.mmt_dense_dgrad.argtypes = [vp, i64, vp, i64, vp, i64, i64, i64, i64, ctypes.c_int32, vp]
dev = 'cuda'
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
st = lambda: torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)

def fwd(x, w, b, gelu):
  Mx, K = x.shape; N = w.shape[0]
  y = torch.empty(Mx, N, device=dev, dtype=torch.bfloat16)
  gg = torch.empty_like(y) if gelu else None
  assert 0 == (L.mmt_dense_fwd(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), b.data_ptr(), y.data_ptr(), y.stride(0),
                             gg.data_ptr() if gelu else None, gg.stride(0) if gelu else 0, Mx, N, K, 0, st()))
  return (y, gg)

def dgrad(dy, w):
  Mx, K = dy.shape; N = w.shape[1]
  dx = torch.empty(Mx, N, device=dev, dtype=torch.bfloat16)
  assert 0 == (L.mmt_dense_dgrad(dy.data_ptr(), dy.stride(0), w.data_ptr(), w.stride(0), dx.data_ptr(), dx.stride(0), Mx, N, K, 0, st()))
  return dx

def timeit(fns, flops, rounds=5, iters=20):
  res = {k: [] for k in fns}
  for _ in range(rounds):
    for k, f in fns.items():
      for _ in range(3): f()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(iters): f()
      e1.record(); torch.cuda.synchronize()
      res[k].append(e0.elapsed_time(e1) / iters * 1e3)
  return {k: (sorted(v)[len(v) // 2], flops / (sorted(v)[len(v) // 2] * 1e-6) / 1e15) for k, v in res.items()}

for name, N, K, gelu in (('QKV fwd', 2304, 768, False), ('attn-out fwd', 768, 768, False), ('FFN1 fwd + GELU', 3072, 768, True),
                         ('FFN1 fwd', 3072, 768, False), ('FFN2 fwd', 768, 3072, False)):
  x, w, b = rnd(M, K), rnd(N, K) * 0.05, torch.randn(N, device=dev, generator=g)
  y, gg = fwd(x, w, b, gelu)
  ref = x.float() @ w.float().t() + b
  err = float((y.float() - ref).abs().max() / ref.abs().max())
  if gelu:
    gref = F.gelu(y.float(), approximate='tanh')
    err = max(err, float((gg.float() - gref).abs().max() / gref.abs().max()))
  bb = b.to(torch.bfloat16)
  lib = (lambda: F.gelu(F.linear(x, w, bb), approximate='tanh')) if gelu else (lambda: F.linear(x, w, bb))
  t = timeit({'hand': lambda: fwd(x, w, b, gelu), 'library': lib}, 2.0 * M * N * K)
  print(f'{name:18s} N={N:5d} K={K:5d} rel err {err:.2e}   hand {t["hand"][0]:6.1f} us = {t["hand"][1]:.2f} PF   library {t["library"][0]:6.1f} us = {t["library"][1]:.2f} PF', flush=True)

for name, N, K in (('QKV dgrad', 768, 2304), ('attn-out dgrad', 768, 768), ('FFN1 dgrad', 768, 3072), ('FFN2 dgrad', 3072, 768)):
  dy, w = rnd(M, K), rnd(K, N) * 0.05
  dx = dgrad(dy, w)
  ref = dy.float() @ w.float()
  err = float((dx.float() - ref).abs().max() / ref.abs().max())
  t = timeit({'hand': lambda: dgrad(dy, w), 'library': lambda: torch.mm(dy, w)}, 2.0 * M * N * K)
  print(f'{name:18s} N={N:5d} K={K:5d} rel err {err:.2e}   hand {t["hand"][0]:6.1f} us = {t["hand"][1]:.2f} PF   library {t["library"][0]:6.1f} us = {t["library"][1]:.2f} PF', flush=True)
