"""Per-step scalars of a train step -- the step's share of every dropout seed, the learning rate and AdamW's bias
corrections -- either folded into the descriptors on the host (eager steps) or kept in device memory and read by the
kernels when they start, so that a step recorded once as a HIP graph (`graphed.GraphedTrainStep`) replays with fresh
values.  Both ways give the same numbers: a dropout seed is descriptor seed + epoch (mod 2^64) wherever the addition
happens.  The device words are named by the descriptors themselves (ABI 4: `dropout_epoch` of mmt_attn_desc /
mmt_rows_desc / mmt_embed_desc, `hyper` of mmt_adamw_desc, include/mmt_attn.h); the library keeps no registration, so
what is "active" is this module's business: one `DeviceStepScalars` per device at most.

The reference keeps the same quantities in tf.Variables of its tf.function train step (`optimizer.iterations`, the
learning-rate schedule evaluated on it; src/tasks/pretraining.py:224-298 runs under `tf.function`)."""
from __future__ import annotations

import torch

from . import _lib

_MASK64 = (1 << 64) - 1
_EPOCH_MUL = 0x9E3779B97F4A7C15
_host = {'epoch': 0}
_active = {}               # device index -> the active DeviceStepScalars of that device


def _index(device) -> int:
  device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
  return device.index if device.index is not None else torch.cuda.current_device()


def epoch_of(step: int) -> int:
  return (int(step) * _EPOCH_MUL) & _MASK64


def set_step(step: int) -> None:
  """Host side: the train step whose masks the next launches draw (`fused.set_seed_stream`)."""
  _host['epoch'] = epoch_of(step)


def host_epoch(device=None) -> int:
  """What a descriptor for `device` adds to its dropout seed on the host: the step's epoch, or 0 while the kernels of
  that device add it themselves."""
  return 0 if (_active and _index(device) in _active) else _host['epoch']


def epoch_ptr(device=None):
  """Address of the device-resident epoch word for descriptors on `device` (their `dropout_epoch` field), or None."""
  a = _active.get(_index(device)) if _active else None
  return None if a is None else a.epoch.data_ptr()


def hyper_ptr(device=None):
  """Address of the device-resident {lr, bias corrections} for mmt_adamw_desc.hyper on `device`, or None."""
  a = _active.get(_index(device)) if _active else None
  return None if a is None else a.hyper.data_ptr()


def device_active(device=None) -> bool:
  return bool(_active) and _index(device) in _active


class DeviceStepScalars:
  """The device-resident copy: one uint64 epoch and {lr, 1 - beta1^t, 1 - beta2^t}.  `write` queues their new values on
  the current stream (one small launch: the values travel as kernel arguments, so the host may run any number of steps
  ahead of the device)."""

  def __init__(self, device):
    self.epoch = torch.zeros(1, dtype=torch.int64, device=device)
    self.hyper = torch.ones(3, dtype=torch.float32, device=device)

  def enable(self) -> None:
    _active[_index(self.epoch.device)] = self

  def disable(self) -> None:
    if _active.get(_index(self.epoch.device)) is self:
      del _active[_index(self.epoch.device)]

  def __del__(self):
    try:
      self.disable()
    except Exception:
      pass

  def write(self, step: int, lr: float, t: int, beta1: float, beta2: float) -> None:
    dev = self.epoch.device
    with torch.cuda.device(dev):
      _lib.check(_lib.lib().mmt_write_step_scalars(self.epoch.data_ptr(), self.hyper.data_ptr(), epoch_of(step), float(lr),
                                                  1.0 - beta1 ** t, 1.0 - beta2 ** t, torch.cuda.current_stream(dev).cuda_stream))
