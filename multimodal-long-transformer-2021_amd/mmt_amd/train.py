"""Training driver with the reference's flag surface (`src/train.py:37-91`):
  python -m mmt_amd.train --experiment mmt/pretraining --mode train --model_dir /tmp/m \\
      --config_file a.yaml --params_override task.train_data.max_seq_len=1024
Launch one process per GPU (torchrun / torch.distributed.run); data are synthetic
(`input_utils.synthetic_batch`) -- real-data ingestion is out of scope.
"""
from __future__ import annotations

import argparse
import json
import os
import time

import torch

from . import checkpoint, configs, distribute, graphed, optimization, tasks


def run_experiment(params: configs.ExperimentConfig, mode: str, model_dir: str, device=None,
                   log_every: int = 10, max_steps=None):
  """Minimal `train_lib.run_experiment`: build task/model/optimizer under the strategy, loop."""
  rt = params.runtime
  strategy = distribute.get_distribution_strategy(
      distribution_strategy=rt.distribution_strategy, all_reduce_alg=rt.all_reduce_alg,
      num_gpus=rt.num_gpus, tpu_address=rt.tpu)
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  device = device or torch.device('cuda', local_rank)
  torch.cuda.set_device(device)
  task = tasks.get_task(params.task, logging_dir=model_dir,
                        compute_dtype=tasks._compute_dtype(rt.mixed_precision_dtype),
                        num_replicas=strategy.num_replicas_in_sync)
  torch.manual_seed(0)     # identical initial weights on every replica
  model = task.build_model().to(device)
  task.initialize(model)              # warm start from task.init_checkpoint (no-op when empty)
  opt_cfg = params.trainer.optimizer_config
  reducer = strategy.make_reducer(list(model.parameters()), reduce=tasks.gradient_reduce_mode(params.task))
  optimizer = optimization.create_optimizer(model, opt_cfg, reducer=reducer)
  data = task.build_inputs(params.task.train_data, device=device, rank=strategy.rank)
  steps = max_steps or params.trainer.train_steps
  logs = []
  start = 0
  resume_from = checkpoint.latest_checkpoint(model_dir) if model_dir and os.path.isdir(model_dir) else None
  if resume_from:                       # restart: model, optimizer moments and step from the latest checkpoint
    start = checkpoint.restore(resume_from, model, optimizer)
    if hasattr(optimizer, 'refresh_shadow'):
      optimizer.refresh_shadow()
    if strategy.rank == 0:
      print(json.dumps({'resumed_from': resume_from, 'step': start}), flush=True)
  ckpt_every = params.trainer.checkpoint_interval

  def save(step):
    if strategy.rank == 0 and model_dir:
      checkpoint.save(model_dir, step, model, optimizer, max_to_keep=params.trainer.max_to_keep)

  from . import metrics as metrics_lib
  if 'train' in mode:
    t0 = time.perf_counter()
    # `task.build_metrics()` objects, updated on the device inside every step (process_metrics, pretraining.py:297);
    # read -- one all-reduce + one host copy each -- only on logging steps, then reset like orbit's summary loop
    train_metrics = task.build_metrics(training=True)
    # the step as a HIP graph (recorded after three eager steps, graphed.py) when everything it touches has a fixed
    # address: flat optimizer on the GPU, bf16 compute.  MMT_STEP_GRAPH: unset = on with one replica, 0 / 1 = off / on
    graphed_step = None
    env = os.environ.get('MMT_STEP_GRAPH')
    if ((strategy.num_replicas_in_sync == 1 if env is None else env != '0') and hasattr(optimizer, 'slabs')
        and task.compute_dtype == torch.bfloat16):
      graphed_step = graphed.GraphedTrainStep(task, model, optimizer, reducer, opt_cfg, metrics=train_metrics,
                                              clip_norm=opt_cfg.gradient_clip_norm)
    run_experiment.last_step_launch = 'graph' if graphed_step is not None else 'eager'
    try:
      for step in range(start, steps):
        if graphed_step is not None:
          out = graphed_step(next(data), step + 1)
        else:
          optimization.set_learning_rate(optimizer, optimization.learning_rate_at(opt_cfg, step))
          out = task.train_step(next(data), model, optimizer, metrics=train_metrics, reducer=reducer,
                                clip_norm=opt_cfg.gradient_clip_norm, step=step + 1)
        if step % log_every == 0 or step == steps - 1:
          loss = float(out[task.loss])
          logs.append({'step': step, 'loss': loss, 'elapsed_s': time.perf_counter() - t0,
                       **{k: round(v, 6) for k, v in metrics_lib.results(train_metrics).items()}})
          metrics_lib.reset(train_metrics)
          if strategy.rank == 0:
            print(json.dumps(logs[-1]), flush=True)
        if ckpt_every and (step + 1) % ckpt_every == 0 and step + 1 < steps:
          save(step + 1)
      if steps > start:
        save(steps)
    finally:
      if graphed_step is not None:       # also when the loop raised: the device-resident step scalars must not outlive it
        graphed_step.close()
  if 'eval' in mode:
    vdata = task.build_inputs(params.task.validation_data, device=device, rank=strategy.rank)
    eval_metrics = task.build_metrics(training=False)
    out = task.validation_step(next(vdata), model, metrics=eval_metrics)
    logs.append({'validation_loss': float(out[task.loss]),
                 **{f'validation_{k}': round(v, 6) for k, v in metrics_lib.results(eval_metrics).items()}})
    if strategy.rank == 0:
      print(json.dumps(logs[-1]), flush=True)
  return model, logs


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__)
  ap.add_argument('--experiment', required=True)
  ap.add_argument('--mode', required=True,
                  choices=['train', 'eval', 'train_and_eval', 'continuous_train_and_eval'])
  ap.add_argument('--model_dir', required=True)
  ap.add_argument('--config_file', action='append', default=[])
  ap.add_argument('--params_override', default=None)
  ap.add_argument('--tpu', default=None)
  ap.add_argument('--tpu_zone', default=None)
  ap.add_argument('--pretrain_steps', type=int, default=None)
  ap.add_argument('--max_steps', type=int, default=None)
  args = ap.parse_args(argv)
  params = configs.parse_configuration(args.experiment, args.config_file, args.params_override)
  if 'train' in args.mode and args.model_dir:
    os.makedirs(args.model_dir, exist_ok=True)
    with open(os.path.join(args.model_dir, 'params.yaml'), 'w') as f:
      import yaml
      yaml.safe_dump(params.as_dict(), f)
  run_experiment(params, args.mode, args.model_dir, max_steps=args.max_steps)


if __name__ == '__main__':
  main()
