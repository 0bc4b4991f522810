"""Builds the closures `bench.py` times: the train step of BASELINE config 3 (BERT-base dims, S=4096,
radius 64 + 8 global text tokens, bf16, per-GPU batch 4; synthetic data, full optimisation step:
forward + losses + backward + gradient all-reduce + clip + AdamW) and the gradient exchange alone."""
from __future__ import annotations

import torch

import os

from . import configs, distribute, graphed, optimization, tasks


def _exchanging(world: int) -> bool:
  """More than one rank, or the one-rank rehearsal of the exchange machinery (MMT_FORCE_DIST=1, bench.py)."""
  return world > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized())


def _exchange_label(world: int, bucket_mb: float) -> str:
  if not _exchanging(world):
    return 'none (1 GPU)'
  backend = torch.distributed.get_backend()
  name = 'RCCL (nccl backend)' if backend == 'nccl' else backend
  return f'{name}, bucketed ({bucket_mb:.0f} MB) SUM all-reduce overlapped with backward'


def make_train_step_bench(cfg: dict, device, rank: int, world: int, dtype=torch.bfloat16, graph=None):
  """`graph`: replay the step as a HIP graph after three eager steps (`graphed.GraphedTrainStep`); None = the
  MMT_STEP_GRAPH switch: unset = on for a single-process run, off when gradients are exchanged (the capture of the RCCL
  all-reduces is exercised with one rank only -- MMT_FORCE_DIST=1 MMT_STEP_GRAPH=1 -- not between devices); bf16 only."""
  exp = configs.get_exp_config('mmt/pretraining')
  P = cfg.get('P', 63)          # patches per image row (image side 16 P)
  exp.override({
      'task': {
          'micro_batch_size': cfg['B'],
          'model': {'encoder': {'mmt': {'relative_pos_max_distance': cfg['m'],
                                       'relative_att_num_core_layers': cfg.get('core', 0),
                                       'relative_vocab_size': cfg['R']}},
                    'cls_heads': [{'inner_dim': 768, 'num_classes': 2, 'name': 'itm'}]},
          # mlm_max_selections_per_seq / mpp_max_selections_per_seq stay at the reference defaults
          # (256 / 98, pretrain_dataloader.py:37-38); mpp_fraction_to_mask 0.0 as in every shipped YAML
          'train_data': {'max_seq_len': cfg['S'], 'image_size': 16 * P, 'patch_size': 16,
                         'global_batch_size': cfg['B'] * world, 'tasks': 'mlm,itm',
                         'mpp_fraction_to_mask': 0.0,
                         'relative_pos_max_distance': cfg['m'], 'local_radius': cfg['radius'],
                         'relative_att_num_core_layers': cfg.get('core', 0),     # > 0: 2-D ids (*_2d*.yaml)
                         'num_global_tokens': cfg['ng']},
      }})
  strategy = distribute.DataParallelStrategy(torch.distributed.get_backend() if _exchanging(world) else None)
  task = tasks.PretrainingTask(exp.task, compute_dtype=dtype, num_replicas=world)
  torch.manual_seed(0)
  model = task.build_model().to(device)
  opt_cfg = exp.trainer.optimizer_config
  reduce = tasks.gradient_reduce_mode(exp.task)
  reducer = strategy.make_reducer(list(model.parameters()), reduce=reduce)
  optimizer = optimization.create_optimizer(model, opt_cfg, reducer=reducer)
  data = task.build_inputs(exp.task.train_data, device=device, rank=rank, batch_size=cfg['B'])
  batch = next(data)       # inputs resident in HBM before the timed region
  state = {'step': 0}
  if graph is None:
    env = os.environ.get('MMT_STEP_GRAPH')
    graph = (not _exchanging(world)) if env is None else env != '0'
  graph = bool(graph) and dtype == torch.bfloat16 and hasattr(optimizer, 'slabs')

  if graph:
    graphed_step = graphed.GraphedTrainStep(task, model, optimizer, reducer, opt_cfg, clip_norm=opt_cfg.gradient_clip_norm,
                                            static_inputs=True)     # the batch is resident and the same every step

    def step():
      state['step'] += 1
      return graphed_step(batch, state['step'])
    step.close = graphed_step.close
  else:
    def step():
      optimization.set_learning_rate(optimizer, optimization.learning_rate_at(opt_cfg, state['step']))
      state['step'] += 1
      return task.train_step(batch, model, optimizer, reducer=reducer, clip_norm=opt_cfg.gradient_clip_norm,
                             step=state['step'])
    step.close = lambda: None

  step.optimizer, step.model = optimizer, model
  n_params = sum(p.numel() for p in model.parameters())
  d = exp.task.train_data
  info = {'model': 'MmtPretrainingModel (12 layers, hidden 768, 12 heads, intermediate 3072), mlm+itm heads',
          'params': n_params, 'optimizer': 'AdamW (fused flat step, bf16 shadow weights) + polynomial lr, clip 1.0',
          'dropout': 0.1, 'mlm_max_selections_per_seq': d.mlm_max_selections_per_seq,
          'mpp_max_selections_per_seq': d.mpp_max_selections_per_seq,
          'grad_allreduce': _exchange_label(world, strategy.bucket_bytes / (1 << 20)), 'grad_reduce': reduce,
          'step_launch': 'HIP graph replay (recorded after 3 eager steps)' if graph else 'eager (one launch per kernel)'}
  return step, info


def make_allreduce_bench(device, world: int, mbytes=None):
  """The exchange step alone (`optimizer.apply_gradients`' implicit all-reduce, `pretraining.py:273`):
  fp32 gradient buckets of the BERT-base model (111.0 M parameters = 444 MB) pushed through
  `GradientBucketReducer` exactly as a backward pass does -- ready hooks in reverse parameter order,
  one async all-reduce per 48 MB bucket, `finish()`."""
  n = int((mbytes if mbytes is not None else 444.0) * 1e6 / 4)
  chunk = 2_359_296                      # one 768 x 3072 weight
  sizes = [chunk] * (n // chunk) + ([n % chunk] if n % chunk else [])
  params = [torch.nn.Parameter(torch.zeros(s, device=device)) for s in sizes]
  strategy = distribute.DataParallelStrategy(torch.distributed.get_backend() if _exchanging(world) else None)
  reducer = strategy.make_reducer(params, reduce='sum')

  def step():
    reducer.zero_grad()
    for p in reversed(params):           # "backward": each gradient is written, then reported ready
      p.grad.fill_(1.0)
      reducer._on_grad_ready(p)
    reducer.finish()

  step.check = lambda: float(reducer.buckets[0][0]) == float(world)   # SUM of ones over the ranks
  return step, {'bytes_per_rank': 4 * sum(sizes), 'buckets': len(reducer.buckets),
                'bucket_mb': strategy.bucket_bytes / (1 << 20)}
