"""Builds the train-step closure `bench.py` times: BASELINE config 3 (BERT-base dims, S=4096,
radius 64 + 8 global text tokens, bf16, per-GPU batch 4), synthetic data, full optimisation step
(forward + losses + backward + gradient all-reduce + clip + AdamW)."""
from __future__ import annotations

import torch

from . import configs, distribute, optimization, tasks


def make_train_step_bench(cfg: dict, device, rank: int, world: int, dtype=torch.bfloat16):
  exp = configs.get_exp_config('mmt/pretraining')
  P = cfg.get('P', 63)          # patches per image row (image side 16 P)
  exp.override({
      'task': {
          'micro_batch_size': cfg['B'],
          'model': {'encoder': {'mmt': {'relative_pos_max_distance': cfg['m'],
                                       'relative_vocab_size': cfg['R']}},
                    'cls_heads': [{'inner_dim': 768, 'num_classes': 2, 'name': 'itm'}]},
          'train_data': {'max_seq_len': cfg['S'], 'image_size': 16 * P, 'patch_size': 16,
                         'global_batch_size': cfg['B'] * world, 'tasks': 'mlm,itm',
                         'mpp_fraction_to_mask': 0.0, 'mlm_max_selections_per_seq': 32,
                         'relative_pos_max_distance': cfg['m'], 'local_radius': cfg['radius'],
                         'num_global_tokens': cfg['ng']},
      }})
  strategy = distribute.DataParallelStrategy(torch.distributed.get_backend() if world > 1 else None)
  task = tasks.PretrainingTask(exp.task, compute_dtype=dtype, num_replicas=world)
  torch.manual_seed(0)
  model = task.build_model().to(device)
  opt_cfg = exp.trainer.optimizer_config
  reducer = strategy.make_reducer(list(model.parameters()), reduce='mean')
  optimizer = optimization.create_optimizer(model, opt_cfg, reducer=reducer)
  data = task.build_inputs(exp.task.train_data, device=device, rank=rank, batch_size=cfg['B'])
  batch = next(data)       # inputs resident in HBM before the timed region
  state = {'step': 0}

  def step():
    optimization.set_learning_rate(optimizer, optimization.learning_rate_at(opt_cfg, state['step']))
    state['step'] += 1
    return task.train_step(batch, model, optimizer, reducer=reducer, clip_norm=opt_cfg.gradient_clip_norm)

  n_params = sum(p.numel() for p in model.parameters())
  info = {'model': 'MmtPretrainingModel (12 layers, hidden 768, 12 heads, intermediate 3072), mlm+itm heads',
          'params': n_params, 'optimizer': 'AdamW (fused flat step, bf16 shadow weights) + polynomial lr, clip 1.0', 'dropout': 0.1,
          'grad_allreduce': 'RCCL bucketed (48 MB) overlapped with backward' if world > 1 else 'none (1 GPU)'}
  return step, info
