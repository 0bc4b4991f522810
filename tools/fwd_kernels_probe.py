"""Dev tool: one attention-forward call of a BASELINE configuration through each forward kernel (plane walk, window,
per-wave; `tuning` switches of the descriptor), HIP-event time per call in interleaved rounds + agreement of the outputs.
  python tools/fwd_kernels_probe.py [config 2|3|5] [globals] [dropout_p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd, bench
from mmt_amd import _lib
cfg = bench.get_config(int(sys.argv[1]) if len(sys.argv) > 1 else 3, sys.argv[2] if len(sys.argv) > 2 else None)
pdrop = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
B, S, N, D, R = cfg['B'], cfg['S'], cfg['N'], cfg['D'], cfg['R']
g = torch.Generator(device='cuda').manual_seed(1234)
q, k, v = (torch.randn(B, S, N, D, device='cuda', generator=g).to(torch.bfloat16) for _ in range(3))
emb = (torch.randn(R, N, D, device='cuda', generator=g) * 0.02).to(torch.bfloat16)
bias = (torch.randn(R, N, device='cuda', generator=g) * 0.02).to(torch.bfloat16)
pat = mmt_amd.AttentionPattern(local_radius=cfg['radius'], global_start=cfg['g0'], n_global=cfg['ng'], id_mode=1, max_dist=cfg['m'])
kinds = {'default': 0, 'sliding window': _lib.MMT_TUNE_FWD_PWIN, 'window': _lib.MMT_TUNE_FWD_FORCE_WIN, 'window, one rows workgroup': _lib.MMT_TUNE_FWD_FORCE_WIN | _lib.MMT_TUNE_FWD_ROWS_ONE_WG, 'walk': _lib.MMT_TUNE_FWD_WALK, 'per-wave': _lib.MMT_TUNE_FWD_NO_WIN}
call = lambda t: mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat, dropout_p=pdrop, dropout_seed=12345, tuning=t)
outs = {n: call(t) for n, t in kinds.items()}
torch.cuda.synchronize()
ref = outs['per-wave']
for n, (o, l) in outs.items():
  print(f'{n:26s} max|out - per-wave| = {float((o.float() - ref[0].float()).abs().max()):.3e}   max|lse - per-wave| = {float((l - ref[1]).abs().max()):.3e}'
        f'   finite: {bool(torch.isfinite(o.float()).all())}', flush=True)
times = {n: [] for n in kinds}
for rnd in range(5):
  for n, t in kinds.items():
    for _ in range(3):
      call(t)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
      call(t)
    e1.record()
    torch.cuda.synchronize()
    times[n].append(e0.elapsed_time(e1) / 20 * 1e3)
for n, ts in times.items():
  print(f'{n:26s} us per call: median {sorted(ts)[len(ts) // 2]:.1f}  min {min(ts):.1f}  all {[round(x, 1) for x in ts]}')
