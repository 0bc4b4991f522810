"""TEST INFRASTRUCTURE (oracle): numpy restatement of the Keras 2.5 metrics the reference's tasks build
(src/tasks/pretraining.py:183-222, src/tasks/classification.py:132-170): Mean, SparseCategoricalAccuracy and
AUC(curve='PR', 200 thresholds, 'interpolation').  tensorflow==2.5.0 is a pinned dependency of the reference
(src/requirements.txt) and absent from /root/reference and this image: restated from its published source
(keras/metrics.py `Mean.update_state`, `sparse_categorical_accuracy`, `AUC.interpolate_pr_auc`) -- PARITY UNPINNED.
Only tests/ import this file."""
import numpy as np


def divide_no_nan(num, den):
  num, den = np.asarray(num, np.float32), np.asarray(den, np.float32)
  return np.where(den != 0, num / np.where(den != 0, den, 1), 0).astype(np.float32)


def mean_update(state, values, sample_weight=None):
  """state = [total, count] (float32)."""
  v = np.asarray(values, np.float32)
  if sample_weight is None:
    return np.array([state[0] + v.sum(dtype=np.float32), state[1] + np.float32(v.size)], np.float32)
  w = np.broadcast_to(np.asarray(sample_weight, np.float32), v.shape)
  return np.array([state[0] + (v * w).sum(dtype=np.float32), state[1] + w.sum(dtype=np.float32)], np.float32)


def sparse_categorical_accuracy_update(state, y_true, y_pred, sample_weight=None):
  match = (np.argmax(np.asarray(y_pred, np.float32), axis=-1).reshape(np.shape(y_true)) == np.asarray(y_true)).astype(np.float32)
  return mean_update(state, match, None if sample_weight is None else np.reshape(sample_weight, match.shape))


def auc_thresholds(n=200, eps=1e-7):
  return np.array([0.0 - eps] + [(i + 1) * 1.0 / (n - 1) for i in range(n - 2)] + [1.0 + eps], np.float32)


def auc_update(state, y_true, y_pred, sample_weight=None, n=200):
  """state = [4, n] float32: tp, fp, tn, fn per threshold."""
  thr = auc_thresholds(n)
  p = np.asarray(y_pred, np.float32).reshape(-1)
  pos = np.asarray(y_true).reshape(-1) != 0
  w = np.ones_like(p) if sample_weight is None else np.asarray(sample_weight, np.float32).reshape(-1)
  out = np.array(state, np.float32).copy()
  for t in range(n):
    pp = p > thr[t]
    out[0, t] += (w * (pp & pos)).sum(dtype=np.float32)
    out[1, t] += (w * (pp & ~pos)).sum(dtype=np.float32)
    out[2, t] += (w * (~pp & ~pos)).sum(dtype=np.float32)
    out[3, t] += (w * (~pp & pos)).sum(dtype=np.float32)
  return out


def auc_pr_result(state):
  tp, fp, tn, fn = (np.asarray(x, np.float32) for x in state)
  n = tp.shape[0]
  dtp = tp[:n - 1] - tp[1:]
  p = tp + fp
  dp = p[:n - 1] - p[1:]
  slope = divide_no_nan(dtp, np.maximum(dp, 0))
  intercept = tp[1:] - slope * p[1:]
  ratio = np.where((p[:n - 1] > 0) & (p[1:] > 0), divide_no_nan(p[:n - 1], np.maximum(p[1:], 0)), 1.0).astype(np.float32)
  inc = divide_no_nan(slope * (dtp + intercept * np.log(ratio)), np.maximum(tp[1:] + fn[1:], 0))
  return np.float32(inc.sum(dtype=np.float32))
