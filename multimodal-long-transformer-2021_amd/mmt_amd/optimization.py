"""AdamW with the reference's decay-exclusion list + polynomial decay with polynomial warm-up
(`src/configs/pretraining_experiments.py:24-47`; TFM `optimization.OptimizerFactory`)."""
from __future__ import annotations

import re

import torch

from . import step_scalars
from .configs import OptimizerConfig


def learning_rate_at(cfg: OptimizerConfig, step: int) -> float:
  """TFM PolynomialDecay wrapped in PolynomialWarmUp (both power 1 by default)."""
  s = min(step, cfg.decay_steps)
  decayed = ((cfg.initial_learning_rate - cfg.end_learning_rate) *
             (1 - s / max(cfg.decay_steps, 1)) ** cfg.power + cfg.end_learning_rate)
  if cfg.warmup_steps and step < cfg.warmup_steps:
    return decayed * (step / cfg.warmup_steps) ** cfg.warmup_power
  return decayed


def split_decay_groups(named_params, exclude_patterns):
  decay, no_decay = [], []
  for name, p in named_params:
    if not p.requires_grad:
      continue
    (no_decay if any(re.search(pat, name) for pat in exclude_patterns) else decay).append(p)
  return decay, no_decay


class FusedAdamW:
  """AdamW over the reducer's flat layout: one `mmt_adamw_step` launch per gradient bucket updates
  the fp32 master parameters and moments, applies the global-norm clip factor while reading the
  gradients, writes the bf16 shadow weights the forward pass uses (`param._mmt_shadow`) and
  clears the gradient bucket.  Same update rule as torch.optim.AdamW."""

  def __init__(self, named_params, cfg: OptimizerConfig, reducer, shadow_dtype=torch.bfloat16):
    from . import _lib
    self._lib = _lib
    self.cfg, self.reducer = cfg, reducer
    self.param_groups = [{'lr': cfg.initial_learning_rate}]
    self.t = 0
    names = {p: n for n, p in named_params}
    self.slabs = []
    for gi, grad in enumerate(reducer.buckets):
      n = grad.numel()
      dev = grad.device
      slab = dict(grad=grad, param=torch.zeros(n, device=dev), m=torch.zeros(n, device=dev),
                  v=torch.zeros(n, device=dev), shadow=torch.zeros(n, device=dev, dtype=shadow_dtype),
                  wd=torch.zeros(n >> 10, device=dev))
      self.slabs.append(slab)
    for p, gi, off in reducer.layout:
      slab = self.slabs[gi]
      k = p.numel()
      slab['param'][off:off + k].copy_(p.data.reshape(-1))
      p.data = slab['param'][off:off + k].view_as(p)              # re-home the parameter
      excluded = any(re.search(pat, names.get(p, '')) for pat in cfg.exclude_from_weight_decay)
      slab['wd'][off >> 10:(off + k + 1023) >> 10] = 0.0 if excluded else cfg.weight_decay_rate
      p._mmt_shadow = slab['shadow'][off:off + k].view_as(p)
    for slab in self.slabs:
      slab['shadow'].copy_(slab['param'])
    for p, _, _ in reducer.layout:
      p._mmt_shadow_version = p._version     # layers._current_shadow re-syncs after any later torch-side write

  def zero_grad(self, set_to_none: bool = False):
    self.reducer.zero_grad()

  def state_dict(self):
    """Step count, learning rate and the flat moment slabs (the parameters themselves are model state)."""
    return {'t': self.t, 'lr': self.param_groups[0]['lr'],
            'slabs': [{'m': s['m'].detach().cpu(), 'v': s['v'].detach().cpu()} for s in self.slabs]}

  def load_state_dict(self, state):
    if len(state['slabs']) != len(self.slabs) or any(a['m'].numel() != b['m'].numel() for a, b in zip(state['slabs'], self.slabs)):
      raise ValueError('optimizer state does not match this model / bucket layout')
    self.t = int(state['t'])
    self.param_groups[0]['lr'] = float(state['lr'])
    for src, dst in zip(state['slabs'], self.slabs):
      dst['m'].copy_(src['m']); dst['v'].copy_(src['v'])
    self.refresh_shadow()

  def refresh_shadow(self):
    """Re-derive the bf16 shadow weights after the master parameters were written from outside
    (checkpoint restore, manual initialisation)."""
    for slab in self.slabs:
      slab['shadow'].copy_(slab['param'])
    for p, _, _ in self.reducer.layout:
      p._mmt_shadow_version = p._version

  @torch.no_grad()
  def step(self, grad_scale=None):
    self.t += 1
    c = self.cfg
    d = self._lib.AdamwDesc()
    d.lr, d.beta1, d.beta2, d.eps = self.param_groups[0]['lr'], c.beta_1, c.beta_2, c.epsilon
    d.bias_correction1, d.bias_correction2 = 1 - c.beta_1 ** self.t, 1 - c.beta_2 ** self.t
    d.zero_grad = 1
    L = self._lib.lib()
    for slab in self.slabs:
      d.n = slab['grad'].numel()
      dev = slab['grad'].device
      d.hyper = step_scalars.hyper_ptr(dev)       # device-resident {lr, bias corrections} of a replayed step, or None
      with torch.cuda.device(dev):
        self._lib.check(L.mmt_adamw_step(
            d, slab['param'].data_ptr(), slab['grad'].data_ptr(), slab['m'].data_ptr(), slab['v'].data_ptr(),
            slab['shadow'].data_ptr(), slab['wd'].data_ptr(),
            None if grad_scale is None else grad_scale.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream))
    self.reducer.buckets_are_zero = True


def create_optimizer(model: torch.nn.Module, cfg: OptimizerConfig, reducer=None):
  """torch.optim.AdamW (fused) by default; with a gradient reducer on the GPU, the flat
  `FusedAdamW` that shares the reducer's bucket layout."""
  if reducer is not None and all(p.is_cuda for p in reducer.params):
    return FusedAdamW(list(model.named_parameters()), cfg, reducer)
  decay, no_decay = split_decay_groups(model.named_parameters(), cfg.exclude_from_weight_decay)
  groups = [{'params': decay, 'weight_decay': cfg.weight_decay_rate},
            {'params': no_decay, 'weight_decay': 0.0}]
  fused = all(p.is_cuda for p in decay + no_decay)
  return torch.optim.AdamW(groups, lr=cfg.initial_learning_rate, betas=(cfg.beta_1, cfg.beta_2),
                           eps=cfg.epsilon, fused=fused)


def set_learning_rate(optimizer: torch.optim.Optimizer, lr: float) -> None:
  for g in optimizer.param_groups:
    g['lr'] = lr
