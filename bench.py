"""Benchmark of the hot path on MI355X (contract: see the task statement / DESIGN.md section 5).

  python bench.py --gpus N --steps K --warmup W

Workload: BASELINE config 3 -- BERT-base dims (12 heads x 64), S=4096 (2 + 63^2 patches + 125 text),
local radius 64 + 8 global text tokens [3971,3979), bf16 I/O with fp32 softmax/accumulation, per-GPU
batch 4, synthetic N(0,1) data, random-init weights.  A "step" is one full optimisation step
(forward, MLM+MPP+ITM losses, backward, gradient all-reduce, clip, AdamW).

One process per GPU.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment
starts the N ranks itself (child processes through `python -m torch.distributed.run`, before this
process has touched the GPU) and relays rank 0's JSON line; launched by torchrun it is one of the
ranks.  The batch is sharded (weak scaling, per-GPU batch fixed) and gradients are all-reduced over
RCCL (backend "nccl" on ROCm) -- `src/distribute_utils.py:97-188`, `src/tasks/pretraining.py:273`.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'multimodal-long-transformer-2021_amd')):
  if _p not in sys.path:
    sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0   # dense bf16 MFMA
XGMI_LINK_GBS = 153.0        # per xGMI link, 7 links per GPU


F32_MFMA_PEAK_TF = 157.3     # exact-f32 MFMA (32x32x2) = the f32 vector rate


def traffic_file(cfg):
  """profiles/ file that holds the PMC traffic + rocprofv3 kernel durations of this configuration's attention calls."""
  if cfg['config'] == 3 and cfg['ng'] == 8:
    return 'attn_traffic.json'
  return f"r04_cfg{cfg['config']}_g{cfg['ng']}_attn_traffic.json"


def config3():
  return get_config(3)


def get_config(n=3, ng=None):
  """The single-GPU BASELINE configurations.  S = 2 + P^2 patches + text; the global tokens
  are the first `ng` text positions; per-GPU batch as the survey fixes it."""
  base = dict(N=12, D=64, R=32, radius=64, m=12, H=768, L=12, I=3072)
  if n == 2:      # BERT-base dims, S=1024 (image 448), fp32 (exact-f32 MFMA), B=8
    base.update(S=1024, P=28, B=8, dtype='f32', name='BASELINE config 2: BERT-base dims, S=1024 (2+28^2 patches+238 text), '
                                                     'radius 64 + {ng} global tokens, fp32, per-GPU batch 8')
  elif n == 3:    # the headline: S=4096 (image 1008), bf16, B=4
    base.update(S=4096, P=63, B=4, dtype='bf16', name='BASELINE config 3: BERT-base dims, S=4096 (2+63^2 patches+125 text), '
                                                      'radius 64 + {ng} global tokens, bf16, per-GPU batch 4')
  elif n == 5:    # Fashion-Gen-shaped S=8192 (image 1408), bf16, B=2, global tokens swept 8 / 32 / 128
    base.update(S=8192, P=88, B=2, dtype='bf16', name='BASELINE config 5 (single-GPU shape): BERT-base dims, S=8192 (2+88^2 patches+446 text), '
                                                      'radius 64 + {ng} global tokens, bf16, per-GPU batch 2')
  else:
    raise ValueError(f'no single-GPU BASELINE configuration {n}')
  base['ng'] = 8 if ng is None else int(ng)
  base['g0'] = 2 + base['P'] ** 2          # the [ATT] marker's position and the tokens after it
  base['config'] = n
  base['name'] = base['name'].format(ng=base['ng'])
  return base


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
  """per attention layer per sample.  Returns (forward flops, forward bytes,
  backward bytes); backward flops = 2.5 x forward (five products against two)."""
  S, N, D, R = cfg['S'], cfg['N'], cfg['D'], cfg['R']
  pairs = pattern_pairs(S, cfg['radius'], cfg['g0'], cfg['ng'])
  flops = 4 * D * N * pairs + 2 * S * R * N * D
  byts = 4 * S * N * D * elt + S * N * 4 + R * N * (D + 1) * elt
  # backward: read Q, K, V, O, dO, write dQ, dK, dV (8 arrays), read LSE, read tables, write fp32 table grads
  bwd = 8 * S * N * D * elt + S * N * 4 + R * N * (D + 1) * (elt + 4)
  return flops, byts, bwd


# ---------------------------------------------------------------------------------------------
# CPU baselines (oracle = the checker; timed here on the host cores only, never shipped)
# ---------------------------------------------------------------------------------------------

def _host_threads():
  return min(16, os.cpu_count() or 1)      # the box's CPU share for one GPU


def cpu_baseline_attention(cfg, seed=1234, budget_s=10.0, backward=False):
  """Times the dense CPU restatement of the reference operator (oracle/attention.py; TF is not
  available offline) on ONE sample x ONE attention layer (all heads, dense [S,S] mask + ids
  materialised exactly as the reference feeds them); with `backward`, forward + analytic backward."""
  import numpy as np
  from oracle import attention as oa
  from oracle import side_inputs as si
  S, N, D, R = cfg['S'], cfg['N'], cfg['D'], cfg['R']
  rng = np.random.default_rng(seed)
  q, k, v = (rng.standard_normal((1, S, N, D)).astype(np.float32) for _ in range(3))
  emb = (rng.standard_normal((R, N, D)) * 0.02).astype(np.float32)
  bias = (rng.standard_normal((R, N)) * 0.02).astype(np.float32)
  dout = rng.standard_normal((1, S, N, D)).astype(np.float32)
  mask = si.sparse_pattern_mask(S, S, cfg['radius'], cfg['g0'], cfg['ng'])[None]
  ids = si.relative_ids_from_desc(S, 1, cfg['m'])[None]
  from threadpoolctl import threadpool_limits
  cores = _host_threads()
  n, t0 = 0, time.perf_counter()
  with threadpool_limits(limits=cores):
    while True:                             # bounded sample
      oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
      if backward:
        oa.relative_attention_bwd(dout, q, k, v, emb, bias, mask, ids)
      n += 1
      if time.perf_counter() - t0 > budget_s:
        break
  dt = time.perf_counter() - t0
  what = 'forward + backward' if backward else 'forward'
  return {'value': n / dt, 'unit': f'attention-layer {"fwd+bwd" if backward else "fwd"} samples/s', 'cores': cores,
          'host_cores_visible': os.cpu_count(),    # the box's logical CPUs; `cores` = threads used (capped at 16, the one-GPU share)
          'kind': 'port',
          'sample': f'{n} x (1 sequence x 1 attention layer {what}, S={S}, {N} heads, dense int32 '
                    f'mask+ids as the reference feeds them), numpy/BLAS restatement of the TF2 CPU '
                    f'path (TF unavailable offline), {dt:.1f} s on {cores} threads'}


def cpu_baseline_train_step(budget_s=10.0):
  """BASELINE config 1 (the reference's CPU-runnable case: 2 layers, hidden 128, 2 heads, S=256 =
  2 + 14^2 + 58 text, R=32, dense segmented mask + 1-D ids, fp32): forward + MLM/MPP/ITM loss +
  backward of the dense torch-CPU restatement (oracle/encoder.py), samples/s on the host threads.
  Per-replica batch 64 of the reference (pretraining.py:39) is run as micro-batches of 8."""
  import numpy as np
  import torch
  from oracle import encoder as oe
  from oracle import side_inputs as si
  cores = _host_threads()
  torch.set_num_threads(cores)
  H, N, I, L, S, P, R, V, m = 128, 2, 512, 2, 256, 14, 32, 30522, 12
  g = torch.Generator().manual_seed(0)
  tn = lambda *s: (torch.randn(*s, generator=g) * 0.02).requires_grad_(True)
  one = lambda *s: torch.ones(*s).requires_grad_(True)
  zero = lambda *s: torch.zeros(*s).requires_grad_(True)
  sd = {'encoder._word_embedding_layer.embedding_table': tn(V, H),
        'encoder._segment_embedding_layer.embedding_table': tn(16, H),
        'encoder._embedding_norm_layer.weight': one(H), 'encoder._embedding_norm_layer.bias': zero(H),
        'encoder._patch_projection_weight': tn(H, 768), 'encoder._patch_projection_bias': zero(H),
        'masked_lm.dense_weight': tn(H, H), 'masked_lm.dense_bias': zero(H),
        'masked_lm.layer_norm.weight': one(H), 'masked_lm.layer_norm.bias': zero(H),
        'masked_lm.output_bias': zero(V),
        'masked_pp.layer_norm.weight': one(H), 'masked_pp.layer_norm.bias': zero(H),
        'masked_pp.dense_weight': tn(512, H), 'masked_pp.dense_bias': zero(512), 'masked_pp.bias': zero(512),
        'classification_heads.0.dense_weight': tn(H, H), 'classification_heads.0.dense_bias': zero(H),
        'classification_heads.0.out_proj_weight': tn(2, H), 'classification_heads.0.out_proj_bias': zero(2)}
  for l in range(L):
    lp = f'encoder._transformer_layers.layers.{l}.'
    sd.update({lp + 'attention.qkv_weight': tn(3 * H, H), lp + 'attention.qkv_bias': zero(3 * H),
               lp + 'attention.relative_emb_table': tn(R, N, H // N), lp + 'attention.relative_bias_table': tn(R, N),
               lp + 'attention.output_weight': tn(H, H), lp + 'attention.output_bias': zero(H),
               lp + 'attention_layer_norm.weight': one(H), lp + 'attention_layer_norm.bias': zero(H),
               lp + 'ffn_layer_norm.weight': one(H), lp + 'ffn_layer_norm.bias': zero(H),
               lp + 'intermediate_weight': tn(I, H), lp + 'intermediate_bias': zero(I),
               lp + 'ffn_output_weight': tn(H, I), lp + 'ffn_output_bias': zero(H)})
  cfg = {'hidden_size': H, 'num_attention_heads': N, 'num_hidden_layers': L, 'use_pre_activation_order': True}
  Bm = 8
  n_img = 2 + P * P
  side = si.add_side_input_features(n_img, S - n_img, S, m)      # data_utils.py:285-380, one example
  rep = lambda a: torch.from_numpy(np.stack([a] * Bm))
  inputs = {'word_ids': torch.randint(1000, V, (Bm, S), generator=g, dtype=torch.int32),
            'segment_ids': rep(side['segment_ids']), 'att_mask': rep(side['att_mask']),
            'relative_att_ids': rep(side['relative_att_ids']),
            'patch_embeddings': torch.randn(Bm, P * P, 768, generator=g),
            'mlm_positions': torch.randint(n_img + 1, S, (Bm, 32), generator=g, dtype=torch.int32),
            'mpp_positions': torch.randint(2, n_img, (Bm, 16), generator=g, dtype=torch.int32)}
  labels = {'mlm_label_ids': torch.randint(1000, V, (Bm, 32), generator=g), 'mlm_label_weights': torch.ones(Bm, 32),
            'mpp_label_ids': torch.randint(0, 512, (Bm, 16), generator=g), 'mpp_label_weights': torch.ones(Bm, 16),
            'itm_label_ids': torch.ones(Bm, dtype=torch.int64), 'itm_label_weights': torch.ones(Bm)}
  params = [p for p in sd.values() if p.requires_grad]
  n, t0 = 0, time.perf_counter()
  while True:
    loss = oe.pretraining_loss(sd, cfg, inputs, labels, dtype=torch.float32)
    grads = torch.autograd.grad(loss, params, allow_unused=True)
    del grads
    n += 1
    if time.perf_counter() - t0 > budget_s:
      break
  dt = time.perf_counter() - t0
  return {'value': n * Bm / dt, 'unit': 'train-step (fwd+loss+bwd) samples/s', 'cores': cores, 'host_cores_visible': os.cpu_count(), 'kind': 'port',
          'sample': f'{n} micro-batches of {Bm} sequences, BASELINE config 1 (2 layers, hidden 128, 2 heads, S=256, '
                    f'dense segmented mask + 1-D ids, fp32), dense torch-CPU restatement of MmtPretrainingModel + '
                    f'build_losses with autograd (TF unavailable offline), {dt:.1f} s on {cores} threads'}


# ---------------------------------------------------------------------------------------------
# launcher: N ranks as children of this process (which never touches the GPU itself)
# ---------------------------------------------------------------------------------------------

def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def launch_ranks(n, argv):
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this pool (RCCL needs it)
  env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
  return subprocess.call(cmd, env=env)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=10)
  ap.add_argument('--mode', default='auto', choices=['auto', 'attn_fwd', 'train_step', 'allreduce'])
  ap.add_argument('--allreduce-mb', type=float, default=None,
                  help='allreduce mode: gradient bytes per rank in MB (default: the model\'s 444 MB)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--config', type=int, default=3, choices=[2, 3, 5],
                  help='single-GPU BASELINE configuration (SURVEY.md 8d): 3 = the headline (default); 2 = S=1024 fp32; '
                       '5 = S=8192 bf16 (use --globals 8|32|128)')
  ap.add_argument('--globals', type=int, default=None, dest='n_globals', help='number of global tokens (default 8)')
  ap.add_argument('--ids2d', action='store_true',
                  help="the same workload with the 2-D relative ids of the reference's *_2d*.yaml (one core layer, "
                       'relative_vocab_size 49) instead of the 1-D ids of BASELINE config 3 -- a side measurement, not the headline')
  args = ap.parse_args()
  if args.gpus < 1:
    ap.error('--gpus must be >= 1')

  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus:
    sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch exactly --gpus ranks')

  import torch
  import torch.distributed as dist

  # MMT_DIST_BACKEND=gloo (+ MMT_ONE_GPU=1: every rank on cuda:0) is the rehearsal mode for boxes with
  # fewer devices than ranks; without a GPU only `--mode allreduce` runs (gloo, CPU tensors).
  on_gpu = torch.cuda.device_count() > 0
  backend = os.environ.get('MMT_DIST_BACKEND', 'nccl' if on_gpu else 'gloo')
  if os.environ.get('MMT_ONE_GPU'):
    local_rank = 0
  if not on_gpu and args.mode != 'allreduce':
    sys.exit('bench.py: no GPU visible -- the hot path has no CPU fallback (only --mode allreduce runs on gloo/CPU)')
  dev = torch.device('cuda', local_rank) if on_gpu else torch.device('cpu')
  # MMT_FORCE_DIST=1 (rehearsal): one rank, but through the collective backend and the multi-rank reducer -- on a
  # one-GPU box this is the only way to execute the RCCL path (init, async bucket all-reduces under backward, waits)
  force_dist = world == 1 and os.environ.get('MMT_FORCE_DIST') == '1'
  if world > 1 or force_dist:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if force_dist:
      os.environ.setdefault('MASTER_PORT', str(_free_port()))
      os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    kw = {'device_id': dev} if backend == 'nccl' else {}
    dist.init_process_group(backend, **kw)
  if on_gpu:
    torch.cuda.set_device(local_rank)

  def sync():
    if on_gpu:
      torch.cuda.synchronize()

  def barrier():
    if world > 1 or force_dist:
      dist.barrier()
    sync()

  ranks_seen = 1
  if world > 1 or force_dist:        # every rank must be reachable through the collective backend before anything is timed
    t = torch.ones(1, device=dev)
    dist.all_reduce(t)
    sync()
    ranks_seen = int(t.item())
    assert ranks_seen == world, (ranks_seen, world)

  cfg = get_config(args.config, args.n_globals)
  if args.ids2d:
    cfg.update(R=49, core=1)
  B, S, N, D, R = cfg['B'], cfg['S'], cfg['N'], cfg['D'], cfg['R']
  f32 = cfg['dtype'] == 'f32'
  tdtype = torch.float32 if f32 else torch.bfloat16
  elt = 4 if f32 else 2
  mode = args.mode
  if mode == 'auto':
    mode = 'train_step'

  import mmt_amd
  step_info = {}
  if mode == 'allreduce':
    from mmt_amd import benchmarks
    step_fn, step_info = benchmarks.make_allreduce_bench(dev, world, args.allreduce_mb)
  else:
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    pat = mmt_amd.AttentionPattern(local_radius=cfg['radius'], global_start=cfg['g0'],
                                   n_global=cfg['ng'], id_mode=2 if args.ids2d else 1, max_dist=cfg['m'],
                                   patches_per_row=cfg.get('P', 0) if args.ids2d else 0, core_layers=cfg.get('core', 0))
    q, k, v = (torch.randn(B, S, N, D, device=dev, generator=g).to(tdtype) for _ in range(3))
    emb = (torch.randn(R, N, D, device=dev, generator=g) * 0.02).to(tdtype)
    bias = (torch.randn(R, N, device=dev, generator=g) * 0.02).to(tdtype)
    drop = dict(dropout_p=0.1, dropout_seed=12345)       # the train step's attention_probs_dropout_prob

    def attn_fwd():
      return mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat, **drop)

    if mode == 'train_step':
      step_fn, step_info = mmt_amd.make_train_step_bench(cfg, dev, rank, world, dtype=tdtype)
    else:
      step_fn = attn_fwd

  if mode == 'train_step' and step_info.get('step_launch', '').startswith('HIP graph'):
    for _ in range(4):       # set-up, not warm-up: three eager steps and the one that records the graph (mmt_amd/graphed.py)
      step_fn()
  for _ in range(args.warmup):
    step_fn()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step_fn()
  barrier()
  dt = time.perf_counter() - t0
  getattr(step_fn, 'close', lambda: None)()      # back to host-side step scalars for the per-call measurements below
  if world > 1:
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
  ms_per_step = dt / args.steps * 1e3

  if mode == 'allreduce':
    assert step_fn.check(), 'all-reduce result is not the sum over the ranks'
    if rank == 0:
      nbytes = step_info['bytes_per_rank']
      busbw = 2 * (world - 1) / world * nbytes / (dt / args.steps) / 1e9 if world > 1 else 0.0
      print(json.dumps({
          'metric': 'gradient all-reduce of one optimisation step', 'value': round(nbytes / (dt / args.steps) / 1e9, 3),
          'unit': 'GB/s (gradient bytes per rank / time)', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
          'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
          'dtype': 'f32', 'data': 'synthetic',
          'config': {'workload': f'bucketed SUM all-reduce of {nbytes / 1e6:.0f} MB fp32 gradients per rank '
                                 f'(GradientBucketReducer, 48 MB buckets)', 'backend': backend,
                     'ranks_seen': ranks_seen, 'parallelism': f'dp{world}', **step_info},
          'bus_bandwidth_GBps': round(busbw, 2),
          'xgmi_ring_frac': round(busbw / XGMI_LINK_GBS, 4) if backend == 'nccl' else None}), flush=True)
    if world > 1 or force_dist:
      dist.destroy_process_group()
    return

  value = B * world / (dt / args.steps)

  # ---- rooflines of the attention kernels: HIP events on the launch stream (torch's current stream
  # IS the stream the C-ABI launches on: ops._stream_ptr) ----
  def timed(fn, iters=50):
    for _ in range(5):
      fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

  flops, byts, bwd_byts = attn_algorithmic(cfg, elt)
  attn_ms = timed(attn_fwd)
  out, lse = attn_fwd()
  dout = torch.randn_like(out)
  dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
  de = torch.zeros(R, N, D, device=dev); db = torch.zeros(R, N, device=dev)

  def attn_bwd():
    return mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, pattern=pat,
                                               grads_out=(dq, dk, dv), rel_grads_accum=(de, db), **drop)
  bwd_ms = timed(attn_bwd, iters=30)

  traffic = traffic_bwd = traffic_commit = None   # HBM bytes per launch from the committed PMC passes
  us_prof = us_prof_bwd = None                    # rocprofv3 kernel durations of the same workload (profiles/, same commit)
  try:
    with open(os.path.join(ROOT, 'profiles', traffic_file(cfg))) as f:
      tj = json.load(f)
    traffic, traffic_bwd, traffic_commit = tj['fwd_hbm_bytes_per_launch'], tj['bwd_hbm_bytes_per_launch'], tj.get('commit')
    us_prof, us_prof_bwd = tj.get('fwd_kernel_us_rocprof'), tj.get('bwd_kernel_us_rocprof')
  except (OSError, KeyError, ValueError):
    pass

  mfma_peak = F32_MFMA_PEAK_TF if f32 else MFMA_BF16_PEAK_TF

  def roof(ms, nbytes, nflops, traffic, kernel, us_rocprof=None):
    """Both roofline terms ("report both fractions for every run and name the binding one"): the HBM
    fraction of the algorithmic bytes and the MFMA fraction of the algorithmic flops -- against the exact-f32 MFMA
    rate for fp32 I/O (config 2), the dense bf16 rate otherwise; `bound` is the larger of the two."""
    gbs = nbytes * B / (ms * 1e-3) / 1e9
    tfs = nflops * B / (ms * 1e-3) / 1e12
    hbm_frac, mfma_frac = gbs / HBM_PEAK_GBS, tfs / mfma_peak
    by_hbm = hbm_frac >= mfma_frac
    us_sum = sum(us_rocprof.values()) * 1e-6 if us_rocprof else None
    return {'bound': 'hbm' if by_hbm else ('mfma_f32' if f32 else 'mfma'),
            'achieved': round(gbs, 1) if by_hbm else round(tfs, 2), 'peak': HBM_PEAK_GBS if by_hbm else mfma_peak,
            'unit': 'GB/s' if by_hbm else 'TFLOP/s', 'frac': round(max(hbm_frac, mfma_frac), 4),
            'traffic': traffic, 'traffic_measured_at': traffic_commit,
            'kernel': kernel, 'launch_us': round(ms * 1e3, 2), 'algorithmic_bytes_per_launch': nbytes * B,
            'algorithmic_flops_per_launch': int(nflops * B),
            'hbm_gbs': round(gbs, 1), 'hbm_frac': round(hbm_frac, 4),
            'mfma_tflops': round(tfs, 2), 'mfma_peak_tflops': mfma_peak, 'mfma_frac': round(mfma_frac, 5),
            'dropout_p': drop['dropout_p'],
            # per-kernel average durations under rocprofv3 --kernel-trace --stats (profiles/*_attn_kernel_stats.csv):
            # frac can be recomputed from them as algorithmic bytes (or flops) per launch / sum(us) / peak
            'kernel_us_rocprof': us_rocprof,
            'frac_rocprof': (round((nbytes * B / us_sum / 1e9 / HBM_PEAK_GBS) if by_hbm else (nflops * B / us_sum / 1e12 / mfma_peak), 4)
                             if us_rocprof else None)}

  roofline = roof(attn_ms, byts, flops, traffic,
                  f'one attention-forward call, B={B} (mmt_attn_fwd)', us_prof)
  roofline_bwd = roof(bwd_ms, bwd_byts, 2.5 * flops, traffic_bwd,
                      f'one attention-backward call, B={B} (mmt_attn_bwd: dQ pass, dK/dV pass, table-gradient reduce)', us_prof_bwd)

  if rank == 0:
    cpu = cpu_fb = cpu_step = None
    if world == 1 and not args.no_cpu_baseline:
      if S <= 4096:      # the dense restatement materialises [N,S,S] fp32 tensors: 3.2 GB each at S=8192 -- not run there
        cpu = cpu_baseline_attention(cfg)
        cpu_fb = cpu_baseline_attention(cfg, backward=True)
      cpu_step = cpu_baseline_train_step()
    line = {
        'metric': f'train-step samples/sec + attention TFLOPS, {S}-tok seq',
        'value': round(value, 3),
        'unit': 'samples/s' if mode == 'train_step' else 'attention-layer-fwd samples/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32' if f32 else 'bf16', 'data': 'synthetic',
        'config': {'workload': cfg['name']
                               + (' -- SIDE MEASUREMENT with the 2-D relative ids of *_2d*.yaml (1 core layer, R=49)' if args.ids2d else ''),
                   'step': mode, 'per_gpu_batch': B, 'global_batch': B * world, 'seq_len': S,
                   'parallelism': f'dp{world}', 'backend': backend if (world > 1 or force_dist) else None,
                   'ranks_seen': ranks_seen, **step_info},
        'attention_fwd': {'us_per_layer_call': round(attn_ms * 1e3, 2), 'tflops': roofline['mfma_tflops'],
                          'samples_per_s': round(B / (attn_ms * 1e-3), 1)},
        'attention_bwd': {'us_per_layer_call': round(bwd_ms * 1e3, 2), 'tflops': roofline_bwd['mfma_tflops'],
                          'samples_per_s': round(B / (bwd_ms * 1e-3), 1)},
        'roofline': roofline,
        'roofline_bwd': roofline_bwd,
        'cpu_baseline': cpu,
        'cpu_baseline_attention_fwd_bwd': cpu_fb,
        'cpu_baseline_train_step': cpu_step,
    }
    print(json.dumps(line), flush=True)
  if world > 1 or force_dist:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
