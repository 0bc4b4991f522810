"""Benchmark of the hot path on MI355X (contract: see the task statement / DESIGN.md).

  python bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE config 3 -- BERT-base dims (12 heads x 64), S=4096
(2 + 63^2 patches + 125 text), local radius 64 + 8 global text tokens [3971,3979), bf16 I/O with
fp32 softmax/accumulation, per-GPU batch 4, synthetic N(0,1) data, random-init weights.
A "step" is one pass of the hot path over one batch.  One process per GPU; for N>1 the batch
is sharded (weak scaling, per-GPU batch fixed) and, in train_step mode, gradients are
all-reduced with RCCL.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'multimodal-long-transformer-2021_amd')):
  if _p not in sys.path:
    sys.path.insert(0, _p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA


def config3():
  return dict(S=4096, N=12, D=64, R=32, B=4, radius=64, g0=3971, ng=8, m=12, H=768, L=12, I=3072)


def pattern_pairs(S, radius, g0, ng):
  """Exact count of unmasked (q,k) pairs of the band+global pattern (valid_len = S)."""
  import numpy as np
  q = np.arange(S)
  lo = np.maximum(0, q - radius); hi = np.minimum(S - 1, q + radius)
  band = hi - lo + 1
  is_g = (q >= g0) & (q < g0 + ng)
  g_in_band = np.clip(np.minimum(hi, g0 + ng - 1) - np.maximum(lo, g0) + 1, 0, None)
  pairs = np.where(is_g, S, band + ng - g_in_band)
  return int(pairs.sum())


def attn_algorithmic(cfg, elt):
  """per attention layer per sample, forward."""
  S, N, D, R = cfg['S'], cfg['N'], cfg['D'], cfg['R']
  pairs = pattern_pairs(S, cfg['radius'], cfg['g0'], cfg['ng'])
  flops = 4 * D * N * pairs + 2 * S * R * N * D
  byts = 4 * S * N * D * elt + S * N * 4 + R * N * (D + 1) * elt
  return flops, byts


def cpu_baseline_attention(cfg, seed=1234):
  """Times the dense CPU restatement of the reference operator (oracle/attention.py; TF is not
  available offline) on ONE sample x ONE attention layer forward (all heads, dense [S,S]
  mask + ids materialised exactly as the reference feeds them)."""
  import numpy as np
  from oracle import attention as oa
  from oracle import side_inputs as si
  S, N, D, R = cfg['S'], cfg['N'], cfg['D'], cfg['R']
  rng = np.random.default_rng(seed)
  q, k, v = (rng.standard_normal((1, S, N, D)).astype(np.float32) for _ in range(3))
  emb = (rng.standard_normal((R, N, D)) * 0.02).astype(np.float32)
  bias = (rng.standard_normal((R, N)) * 0.02).astype(np.float32)
  mask = si.sparse_pattern_mask(S, S, cfg['radius'], cfg['g0'], cfg['ng'])[None]
  ids = si.relative_ids_from_desc(S, 1, cfg['m'])[None]
  from threadpoolctl import threadpool_limits
  cores = min(16, os.cpu_count() or 1)      # the box's CPU share for one GPU
  n, t0 = 0, time.perf_counter()
  with threadpool_limits(limits=cores):
    while True:                             # bounded sample: >= 12 s of CPU work
      oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
      n += 1
      if time.perf_counter() - t0 > 12.0:
        break
  dt = time.perf_counter() - t0
  return {'value': n / dt, 'unit': 'attention-layer-fwd samples/s', 'cores': cores, 'kind': 'port',
          'sample': f'{n} x (1 sequence x 1 attention layer forward, S={S}, {N} heads, dense int32 '
                    f'mask+ids as the reference feeds them), numpy/BLAS restatement of the TF2 CPU '
                    f'path (TF unavailable offline), {dt:.1f} s on {cores} threads'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--mode', default='auto', choices=['auto', 'attn_fwd', 'train_step'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # RCCL ("nccl" on ROCm).  MMT_DIST_BACKEND=gloo + MMT_ONE_GPU=1 is a rehearsal mode that runs
    # several ranks on ONE GPU (the development box has a single device).
    backend = os.environ.get('MMT_DIST_BACKEND', 'nccl')
    if os.environ.get('MMT_ONE_GPU'):
      local_rank = 0
    kw = {'device_id': torch.device('cuda', local_rank)} if backend == 'nccl' else {}
    dist.init_process_group(backend, **kw)
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)

  import mmt_amd
  cfg = config3()
  mode = args.mode
  if mode == 'auto':
    mode = 'train_step' if hasattr(mmt_amd, 'make_train_step_bench') else 'attn_fwd'

  g = torch.Generator(device=dev).manual_seed(1234 + rank)
  B, S, N, D, R = cfg['B'], cfg['S'], cfg['N'], cfg['D'], cfg['R']
  pat = mmt_amd.AttentionPattern(local_radius=cfg['radius'], global_start=cfg['g0'],
                                 n_global=cfg['ng'], id_mode=1, max_dist=cfg['m'])
  q, k, v = (torch.randn(B, S, N, D, device=dev, generator=g).to(torch.bfloat16) for _ in range(3))
  emb = (torch.randn(R, N, D, device=dev, generator=g) * 0.02).to(torch.bfloat16)
  bias = (torch.randn(R, N, device=dev, generator=g) * 0.02).to(torch.bfloat16)

  def attn_fwd():
    return mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat)

  if mode == 'train_step':
    step_fn, step_info = mmt_amd.make_train_step_bench(cfg, dev, rank, world)
  else:
    step_fn, step_info = attn_fwd, {}

  def barrier():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    step_fn()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step_fn()
  barrier()
  dt = time.perf_counter() - t0
  if world > 1:
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
  ms_per_step = dt / args.steps * 1e3
  value = B * world / (dt / args.steps)

  # ---- roofline of the dominant kernel: attention forward, HIP events on the launch stream ----
  for _ in range(5):
    attn_fwd()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  iters = 50
  torch.cuda.synchronize()
  e0.record()
  for _ in range(iters):
    attn_fwd()
  e1.record()
  torch.cuda.synchronize()
  attn_ms = e0.elapsed_time(e1) / iters
  flops, byts = attn_algorithmic(cfg, 2)
  gbs = byts * B / (attn_ms * 1e-3) / 1e9
  tfs = flops * B / (attn_ms * 1e-3) / 1e12
  traffic = None   # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE)
  try:
    with open(os.path.join(ROOT, 'profiles', 'attn_fwd_traffic.json')) as f:
      traffic = json.load(f)['hbm_bytes_per_launch']
  except (OSError, KeyError, ValueError):
    pass
  roofline = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
              'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': traffic,
              'kernel': 'one attention-forward call, B=4: attn_fwd_band_bf16_kernel (band + global-key tiles + '
                        'global-row chunks) followed by attn_rows_combine_kernel',
              'launch_us': round(attn_ms * 1e3, 2),
              'algorithmic_bytes_per_launch': byts * B,
              'mfma_tflops': round(tfs, 2), 'mfma_frac': round(tfs / MFMA_BF16_PEAK_TF, 5)}

  if rank == 0:
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
      cpu = cpu_baseline_attention(cfg)
    line = {
        'metric': 'train-step samples/sec + attention TFLOPS, 4096-tok seq',
        'value': round(value, 3),
        'unit': 'samples/s' if mode == 'train_step' else 'attention-layer-fwd samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': 'BASELINE config 3: BERT-base dims, S=4096 (2+63^2 patches+125 text), '
                               'radius 64 + 8 global tokens, bf16, per-GPU batch 4',
                   'step': mode, 'per_gpu_batch': B, 'global_batch': B * world, 'seq_len': S,
                   'parallelism': f'dp{world}', **step_info},
        'attention_fwd': {'us_per_layer_call': round(attn_ms * 1e3, 2), 'tflops': round(tfs, 2),
                          'samples_per_s': round(B / (attn_ms * 1e-3), 1)},
        'roofline': roofline,
        'cpu_baseline': cpu,
    }
    print(json.dumps(line), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
