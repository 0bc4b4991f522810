"""The metrics of the train / validation step, on the device.

`build_metrics` of the reference returns `tf.keras.metrics` objects -- `Mean` for every loss term,
`SparseCategoricalAccuracy` for every head, `AUC(curve='PR')` for binary classification
(src/tasks/pretraining.py:183-196, src/tasks/classification.py:132-148) -- which `process_metrics` and the loss
function update inside every (micro) step (`pretraining.py:198-222, 297`, `classification.py:150-170`,
`weighted_sparse_categorical_crossentropy_loss.py:42`).  Under a distribution strategy their variables are
SUM-aggregated and synchronised when read.  Here each metric keeps its running sums as a small fp32
tensor on the training device: `update_state` only enqueues device work (no host read inside the step), `result()`
all-reduces a copy of the sums over the data-parallel group (SUM) when one is initialised, and `reset_state()`
zeroes them.  The update rules restate Keras 2.5 (`tf.keras.metrics`, a pinned dependency of the reference,
src/requirements.txt: tensorflow==2.5.0) -- not present in this image: restated from its published source;
oracle/metrics.py is the numpy twin the tests compare with.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch


def _dist_sum_(t: torch.Tensor) -> torch.Tensor:
  if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
  return t


def _divide_no_nan(num: torch.Tensor, den: torch.Tensor) -> torch.Tensor:
  safe = torch.where(den != 0, den, torch.ones_like(den))
  return torch.where(den != 0, num / safe, torch.zeros_like(num))


class Metric:
  """Running sums in `self.state` (fp32, created on the device of the first update)."""
  n_state = 2

  def __init__(self, name: str):
    self.name = name
    self.state: Optional[torch.Tensor] = None

  def _ensure(self, device):
    if self.state is None or self.state.device != torch.device(device):
      self.state = torch.zeros(self.n_state, dtype=torch.float32, device=device)
    return self.state

  def reset_state(self):
    if self.state is not None:
      self.state.zero_()

  reset_states = reset_state                    # Keras 2.5 spelling

  def synced_state(self) -> torch.Tensor:
    """The sums over every replica (a copy: the local sums keep accumulating)."""
    if self.state is None:
      return torch.zeros(self.n_state, dtype=torch.float32)
    return _dist_sum_(self.state.clone())


class Mean(Metric):
  """`tf.keras.metrics.Mean`: weighted running mean, total / count with divide_no_nan."""

  def update_state(self, values, sample_weight=None):
    v = torch.as_tensor(values).detach().to(torch.float32)
    st = self._ensure(v.device)
    if sample_weight is None:
      st[0] += v.sum()
      st[1] += float(v.numel())
    else:
      w = torch.broadcast_to(torch.as_tensor(sample_weight, device=v.device).to(torch.float32), v.shape)
      st[0] += (v * w).sum()
      st[1] += w.sum()

  def result(self) -> torch.Tensor:
    st = self.synced_state()
    return _divide_no_nan(st[0], st[1])


class SparseCategoricalAccuracy(Mean):
  """`tf.keras.metrics.SparseCategoricalAccuracy`: the weighted mean of [argmax(y_pred, -1) == y_true]; the arg-max
  is the FIRST index of the largest value (tf.argmax).  `argmax` may carry a precomputed one (the loss kernel's pass
  over the logits reports it: `mmt_xent_fwd_argmax`), in which case `y_pred` is not read."""

  def update_state(self, y_true, y_pred, sample_weight=None, argmax=None):
    y_true = torch.as_tensor(y_true)
    if argmax is None:
      argmax = torch.argmax(y_pred.detach().to(torch.float32), dim=-1)
    match = (argmax.reshape(y_true.shape).to(torch.int64) == y_true.to(torch.int64)).to(torch.float32)
    if sample_weight is not None:
      sample_weight = torch.as_tensor(sample_weight, device=match.device).reshape(match.shape)
    super().update_state(match, sample_weight)


class AUC(Metric):
  """`tf.keras.metrics.AUC(curve='PR')` with its defaults (200 thresholds, summation_method 'interpolation'):
  weighted confusion counts at every threshold, area by the precision-interpolation of Davis & Goadrich
  (Keras `interpolate_pr_auc`).  `curve='ROC'` sums trapezoids of (FPR, TPR) as Keras does."""
  EPS = 1e-7

  def __init__(self, name: str = 'auc', curve: str = 'PR', num_thresholds: int = 200):
    super().__init__(name)
    if curve not in ('PR', 'ROC'):
      raise ValueError(f'curve must be PR or ROC, got {curve!r}')
    if num_thresholds <= 1:
      raise ValueError('`num_thresholds` must be > 1.')
    self.curve, self.num_thresholds = curve, num_thresholds
    self.n_state = 4 * num_thresholds                       # tp | fp | tn | fn
    inner = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
    self._thresholds = [0.0 - self.EPS] + inner + [1.0 + self.EPS]
    self._thr: Optional[torch.Tensor] = None

  def update_state(self, y_true, y_pred, sample_weight=None):
    p = torch.as_tensor(y_pred).detach().to(torch.float32).reshape(-1)
    st = self._ensure(p.device).view(4, self.num_thresholds)
    if self._thr is None or self._thr.device != p.device:
      self._thr = torch.tensor(self._thresholds, dtype=torch.float32, device=p.device)
    pos = torch.as_tensor(y_true, device=p.device).reshape(-1) != 0
    w = (torch.ones_like(p) if sample_weight is None
         else torch.as_tensor(sample_weight, device=p.device).to(torch.float32).reshape(-1))
    pred_pos = p[None, :] > self._thr[:, None]                      # [T, n]
    wp, wn = (w * pos)[None, :], (w * ~pos)[None, :]
    st[0] += (pred_pos * wp).sum(1)
    st[1] += (pred_pos * wn).sum(1)
    st[2] += (~pred_pos * wn).sum(1)
    st[3] += (~pred_pos * wp).sum(1)

  def result(self) -> torch.Tensor:
    st = self.synced_state().view(4, self.num_thresholds)
    tp, fp, tn, fn = st[0], st[1], st[2], st[3]
    n = self.num_thresholds
    if self.curve == 'PR':
      dtp = tp[:n - 1] - tp[1:]
      p = tp + fp
      dp = p[:n - 1] - p[1:]
      slope = _divide_no_nan(dtp, dp.clamp(min=0))
      intercept = tp[1:] - slope * p[1:]
      ratio = torch.where((p[:n - 1] > 0) & (p[1:] > 0), _divide_no_nan(p[:n - 1], p[1:].clamp(min=0)),
                          torch.ones_like(p[1:]))
      inc = _divide_no_nan(slope * (dtp + intercept * torch.log(ratio)), (tp[1:] + fn[1:]).clamp(min=0))
      return inc.sum()
    recall = _divide_no_nan(tp, tp + fn)
    fpr = _divide_no_nan(fp, fp + tn)
    return ((fpr[:n - 1] - fpr[1:]) * (recall[:n - 1] + recall[1:]) / 2.0).sum()


def by_name(metrics) -> Dict[str, Metric]:
  """The `dict([(metric.name, metric) for metric in metrics])` every reference consumer starts with."""
  if metrics is None:
    return {}
  if isinstance(metrics, dict):
    return {k: v for k, v in metrics.items() if isinstance(v, Metric)}
  return {m.name: m for m in metrics}


def results(metrics: Iterable[Metric]) -> Dict[str, float]:
  """name -> python float of every metric (one host read each; for the logging step, not for the train step)."""
  return {m.name: float(m.result()) for m in by_name(metrics).values()}


def reset(metrics) -> None:
  for m in by_name(metrics).values():
    m.reset_state()
