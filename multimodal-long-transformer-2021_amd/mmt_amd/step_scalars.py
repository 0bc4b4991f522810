"""Per-step scalars of a train step -- the step's share of every dropout seed, the learning rate and AdamW's bias
corrections -- either folded into the descriptors on the host (eager steps) or kept in device memory and read by the
kernels when they start (`mmt_set_step_scalars`, include/mmt_attn.h), so that a step recorded once as a HIP graph
(`graphed.GraphedTrainStep`) replays with fresh values.  Both ways give the same numbers: a dropout seed is
descriptor seed + epoch (mod 2^64) wherever the addition happens.

The reference keeps the same quantities in tf.Variables of its tf.function train step (`optimizer.iterations`, the
learning-rate schedule evaluated on it; src/tasks/pretraining.py:224-298 runs under `tf.function`)."""
from __future__ import annotations

import torch

from . import _lib

_MASK64 = (1 << 64) - 1
_EPOCH_MUL = 0x9E3779B97F4A7C15
_host = {'epoch': 0}
_device = None             # the active DeviceStepScalars, if any


def epoch_of(step: int) -> int:
  return (int(step) * _EPOCH_MUL) & _MASK64


def set_step(step: int) -> None:
  """Host side: the train step whose masks the next launches draw (`fused.set_seed_stream`)."""
  _host['epoch'] = epoch_of(step)


def host_epoch() -> int:
  """What a descriptor adds to its dropout seed on the host: the step's epoch, or 0 while the kernels add it."""
  return 0 if _device is not None else _host['epoch']


def device_active() -> bool:
  return _device is not None


class DeviceStepScalars:
  """The device-resident copy: one uint64 epoch and {lr, 1 - beta1^t, 1 - beta2^t}.  `write` queues their new values on
  the current stream (one small launch: the values travel as kernel arguments, so the host may run any number of steps
  ahead of the device)."""

  def __init__(self, device):
    self.epoch = torch.zeros(1, dtype=torch.int64, device=device)
    self.hyper = torch.ones(3, dtype=torch.float32, device=device)

  def enable(self) -> None:
    global _device
    _lib.check(_lib.lib().mmt_set_step_scalars(self.epoch.data_ptr(), self.hyper.data_ptr()))
    _device = self

  def disable(self) -> None:
    global _device
    if _device is self:
      _lib.check(_lib.lib().mmt_set_step_scalars(None, None))
      _device = None

  def __del__(self):
    # the library holds raw pointers into these tensors: never let them outlive the registration
    try:
      self.disable()
    except Exception:
      pass

  def write(self, step: int, lr: float, t: int, beta1: float, beta2: float) -> None:
    dev = self.epoch.device
    with torch.cuda.device(dev):
      _lib.check(_lib.lib().mmt_write_step_scalars(self.epoch.data_ptr(), self.hyper.data_ptr(), epoch_of(step), float(lr),
                                                  1.0 - beta1 ** t, 1.0 - beta2 ** t, torch.cuda.current_stream(dev).cuda_stream))
