"""Dev tool: attention forward call at BASELINE config 3 (B=4), dropout 0.1: HIP-event time per call, min / median of
several repeats.  Run under rocprofv3 --kernel-trace --stats for per-kernel durations."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd
torch.manual_seed(0)
B, S, N = 4, int(os.environ.get('PROBE_S', 4096)), 12
ng = int(os.environ.get('PROBE_NG', 8))
dt = torch.bfloat16
qkv = torch.randn(B, S, 3, N, 64, device='cuda', dtype=dt)
q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
emb = (torch.randn(32, N, 64, device='cuda') * 0.02).to(dt); bias = (torch.randn(32, N, device='cuda') * 0.02).to(dt)
pat = mmt_amd.AttentionPattern(local_radius=64, global_start=S - 125, n_global=ng, id_mode=1, max_dist=12)
kw = dict(pattern=pat, dropout_p=float(os.environ.get('PROBE_DROP', 0.1)), dropout_seed=1234)
fn = lambda: mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
for _ in range(50): fn()
torch.cuda.synchronize()
res = []
for rep in range(5):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(100): fn()
  e1.record(); torch.cuda.synchronize()
  res.append(e0.elapsed_time(e1) / 100 * 1e3)
print(f"fwd us  min {min(res):.1f}  median {statistics.median(res):.1f}   env WIN={os.environ.get('MMT_FWD_WIN')} TSTRIDE={os.environ.get('MMT_WIN_TSTRIDE')} ng={ng} S={S}")
if os.environ.get('PROBE_BWD'):
  out, lse = fn()
  dout = torch.randn_like(out)
  fb = lambda: mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, **kw)
  for _ in range(20): fb()
  torch.cuda.synchronize()
  res = []
  for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fb()
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50 * 1e3)
  print(f"bwd us  min {min(res):.1f}  median {statistics.median(res):.1f}")
