"""A train step recorded once as a HIP graph and replayed.

One eager step of BASELINE config 3 launches ~350 kernels from Python; with a GPU-side sleep queued ahead of it (so that
the host runs ahead) the same step takes 1.8 ms less than its 15.1 ms wall time -- the GPU spends an eighth of the step
waiting for launches (tools/host_bound_probe.py).  Everything the step touches already has a fixed address: gradient
buckets and AdamW slabs are flat buffers (`distribute.GradientBucketReducer`, `optimization.FusedAdamW`), the metrics
accumulate on the device, and the only values that change from step to step -- the step's share of the dropout seeds,
the learning rate, AdamW's bias corrections -- are read from device memory by the kernels (`step_scalars`; the
descriptors name the words: `dropout_epoch`, `mmt_adamw_desc.hyper`).  So the step (forward, losses, metrics, backward, gradient exchange, clip, AdamW) is captured
with `torch.cuda.graph` after a few eager steps and then replayed: one launch per step.

Same numbers as the eager step: the kernels are deterministic and the seeds are the same sums (tests/test_gpu_graph.py).
The reference's counterpart is the `tf.function` around its train step (src/tasks/pretraining.py:224-298)."""
from __future__ import annotations

import warnings
from typing import Optional

import torch

from . import fused, optimization, step_scalars


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
    self._warned_shape = False

  def _eager(self, batch, step: int):
    optimization.set_learning_rate(self.optimizer, optimization.learning_rate_at(self.opt_cfg, step - 1))
    return self.task.train_step(batch, self.model, self.optimizer, metrics=self.metrics, reducer=self.reducer,
                                clip_norm=self.clip_norm, step=step)

  def _write_scalars(self, step: int):
    c = self.opt_cfg
    self.scalars.write(step, optimization.learning_rate_at(c, step - 1), self.optimizer.t + 1, c.beta_1, c.beta_2)

  def _same_signature(self, batch) -> bool:
    """Does `batch` fit the recorded graph's input buffers?  Same keys, same tensor shapes / dtypes / devices, equal
    non-tensor values.  (`copy_` would broadcast a short last batch over the recorded one and train on duplicated rows;
    a key missing from the new batch would silently keep the recorded step's tensor.)"""
    if len(batch) != len(self.static_batch):
      return False
    for src, dst in zip(batch, self.static_batch):
      if src.keys() != dst.keys():
        return False
      for k, v in src.items():
        d = dst[k]
        if torch.is_tensor(v) != torch.is_tensor(d):
          return False
        if torch.is_tensor(v):
          if v.shape != d.shape or v.dtype != d.dtype or v.device != d.device:
            return False
        elif v != d:
          return False
    return True

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
    self._write_scalars(step)
    t_before = self.optimizer.t
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    # the device-resident scalars are "active" (named by every descriptor that is filled in) ONLY while the step is
    # being recorded: the recorded launches keep the pointers, replays fill in no descriptor at all, and eager steps
    # of this or any other model in the process go on taking their epoch and learning rate from the host
    self.scalars.enable()
    try:
      with torch.cuda.graph(graph):
        self.out = self.task.train_step(self.static_batch, self.model, self.optimizer, metrics=self.metrics,
                                        reducer=self.reducer, clip_norm=self.clip_norm, step=step)
    finally:
      self.scalars.disable()
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
        self._abandon_recording()
        return self._eager(batch, step)
    if not self._same_signature(batch):
      if not self._warned_shape:
        self._warned_shape = True
        warnings.warn('batch does not match the recorded HIP graph (keys, shapes or constants differ -- e.g. a short last '
                      'batch): this step runs eagerly')
      return self._eager(batch, step)
    self._copy_in(batch)
    self._write_scalars(step)
    # the host's copy of the learning rate follows the schedule too: a checkpoint written from a graphed run carries the
    # same optimizer state as one written from the eager loop (optimization.FusedAdamW.state_dict)
    optimization.set_learning_rate(self.optimizer, optimization.learning_rate_at(self.opt_cfg, step - 1))
    self.graph.replay()
    self.optimizer.t += 1
    return self.out

  def _abandon_recording(self):
    """A capture that raised has run the HOST side of `task.train_step` up to the failure (and none of its kernels):
    weight-gradient products queued by the part of backward that was traced, the reducer's ready counts and its "buckets
    are already zero" note (set by the traced optimizer step, which cleared nothing), device scalars enabled.  Put all
    of that back before the step is retried eagerly."""
    self.close()
    self.eager_left = 1 << 62
    fused.reset_host_queues()
    if self.reducer is not None:
      self.reducer.buckets_are_zero = False      # the traced optimizer step zeroed nothing
      self.reducer.zero_grad()
    step_scalars.set_step(0)
    torch.cuda.synchronize()

  def close(self):
    """Drops the recorded graph and the device-resident scalars its launches read (the graph goes first: it holds
    their addresses)."""
    self.graph = None
    if self.scalars is not None:
      self.scalars.disable()
      self.scalars = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass
