"""AdamW with the reference's decay-exclusion list + polynomial decay with polynomial warm-up
(`src/configs/pretraining_experiments.py:24-47`; TFM `optimization.OptimizerFactory`)."""
from __future__ import annotations

import re

import torch

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


def create_optimizer(model: torch.nn.Module, cfg: OptimizerConfig) -> torch.optim.Optimizer:
  decay, no_decay = split_decay_groups(model.named_parameters(), cfg.exclude_from_weight_decay)
  groups = [{'params': decay, 'weight_decay': cfg.weight_decay_rate},
            {'params': no_decay, 'weight_decay': 0.0}]
  fused = all(p.is_cuda for p in decay + no_decay)
  return torch.optim.AdamW(groups, lr=cfg.initial_learning_rate, betas=(cfg.beta_1, cfg.beta_2),
                           eps=cfg.epsilon, fused=fused)


def set_learning_rate(optimizer: torch.optim.Optimizer, lr: float) -> None:
  for g in optimizer.param_groups:
    g['lr'] = lr
