"""A train step recorded once as a HIP graph and replayed.

One eager step of BASELINE config 3 launches ~350 kernels from Python; with a GPU-side sleep queued ahead of it (so that
the host runs ahead) the same step takes 1.8 ms less than its 15.1 ms wall time -- the GPU spends an eighth of the step
waiting for launches (tools/host_bound_probe.py).  Everything the step touches already has a fixed address: gradient
buckets and AdamW slabs are flat buffers (`distribute.GradientBucketReducer`, `optimization.FusedAdamW`), the metrics
accumulate on the device, and the only values that change from step to step -- the step's share of the dropout seeds,
the learning rate, AdamW's bias corrections -- are read from device memory by the kernels (`step_scalars`,
`mmt_set_step_scalars`).  So the step (forward, losses, metrics, backward, gradient exchange, clip, AdamW) is captured
with `torch.cuda.graph` after a few eager steps and then replayed: one launch per step.

Same numbers as the eager step: the kernels are deterministic and the seeds are the same sums (tests/test_gpu_graph.py).
The reference's counterpart is the `tf.function` around its train step (src/tasks/pretraining.py:224-298)."""
from __future__ import annotations

import warnings
from typing import Optional

import torch

from . import optimization, step_scalars


def _tensors(tree):
  for v in tree.values():
    if torch.is_tensor(v):
      yield v


class GraphedTrainStep:
  """`step(batch, step_index) -> {loss: tensor}`; the first `eager_steps` calls run `task.train_step` as it is, the next
  one records the graph (inputs are copied into buffers of its own from then on).  The returned loss tensor is the
  graph's output buffer: read or clone it before the next call."""

  def __init__(self, task, model, optimizer, reducer, opt_cfg, *, metrics=None, clip_norm: Optional[float] = None,
               eager_steps: int = 3, static_inputs: bool = False):
    if not hasattr(optimizer, 'slabs'):
      raise ValueError('GraphedTrainStep needs the flat optimizer (optimization.FusedAdamW): its state has fixed addresses')
    self.task, self.model, self.optimizer, self.reducer = task, model, optimizer, reducer
    self.opt_cfg, self.metrics, self.clip_norm = opt_cfg, metrics, clip_norm
    self.eager_left = max(1, int(eager_steps))
    # static_inputs: the caller passes the SAME batch tensors every step (a benchmark's resident batch): they are the
    # graph's inputs as they are; otherwise the graph gets buffers of its own and every call copies its batch into them
    self.static_inputs = bool(static_inputs)
    self.graph = None
    self.scalars = None
    self.static_batch = None
    self.out = None

  def _eager(self, batch, step: int):
    optimization.set_learning_rate(self.optimizer, optimization.learning_rate_at(self.opt_cfg, step - 1))
    return self.task.train_step(batch, self.model, self.optimizer, metrics=self.metrics, reducer=self.reducer,
                                clip_norm=self.clip_norm, step=step)

  def _write_scalars(self, step: int):
    c = self.opt_cfg
    self.scalars.write(step, optimization.learning_rate_at(c, step - 1), self.optimizer.t + 1, c.beta_1, c.beta_2)

  def _copy_in(self, batch):
    for src_tree, dst_tree in zip(batch, self.static_batch):
      for k, v in src_tree.items():
        if torch.is_tensor(v) and v.data_ptr() != dst_tree[k].data_ptr():
          dst_tree[k].copy_(v, non_blocking=True)

  def _record(self, batch, step: int):
    inputs, labels = batch
    dev = next(_tensors(inputs)).device
    own = (lambda v: v) if self.static_inputs else (lambda v: v.clone())
    self.static_batch = ({k: (own(v) if torch.is_tensor(v) else v) for k, v in inputs.items()},
                         {k: (own(v) if torch.is_tensor(v) else v) for k, v in labels.items()})
    self.scalars = step_scalars.DeviceStepScalars(dev)
    self.scalars.enable()
    self._write_scalars(step)
    t_before = self.optimizer.t
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    try:
      with torch.cuda.graph(graph):
        self.out = self.task.train_step(self.static_batch, self.model, self.optimizer, metrics=self.metrics,
                                        reducer=self.reducer, clip_norm=self.clip_norm, step=step)
    finally:
      self.optimizer.t = t_before          # recording runs nothing: the step itself is the first replay
    self.graph = graph

  def __call__(self, batch, step: int):
    if self.graph is None:
      if self.eager_left > 0:
        self.eager_left -= 1
        return self._eager(batch, step)
      try:
        self._record(batch, step)
      except Exception as e:               # something in this configuration cannot be captured: stay eager, say so once
        warnings.warn(f'train step not recorded as a HIP graph ({type(e).__name__}: {e}); continuing with eager steps')
        self.close()
        self.eager_left = 1 << 62
        torch.cuda.synchronize()
        return self._eager(batch, step)
    self._copy_in(batch)
    self._write_scalars(step)
    self.graph.replay()
    self.optimizer.t += 1
    return self.out

  def close(self):
    """Back to host-side step scalars (eager steps, other models in the process).  Also runs when the object is
    collected: the library must not keep pointers into scalars that are gone."""
    if self.scalars is not None:
      self.scalars.disable()
      self.scalars = None
    self.graph = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
