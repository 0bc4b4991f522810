"""ORACLE (test infrastructure, not product code) -- the element-wise chain of one residual
block, restated with numpy: bias add, dropout (explicit keep mask), residual add, LayerNorm
(eps 1e-12), tanh-GELU, and their gradients.

Follows SURVEY.md App. A.3 (etcmodel ResidualBlock / DenseLayers in pre-activation order; the
package is un-vendored, so this part of the float path is PARITY UNPINNED like
oracle/attention.py) and `src/modeling/models/mmt_encoder.py:53-54` (approximate GELU).
The dropout keep mask restates the counter hash of the product kernels bit for bit
(`csrc/layer_common.h: layer_drop_bits16`, `csrc/mmt_common.h: mix32, drop_row_base, drop_pair_finish`) so masks can be compared
exactly; TF's own RNG stream is not reproducible and is not part of parity.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
  x = x.astype(np.uint64) & _M32
  x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & _M32
  x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & _M32
  x ^= x >> np.uint64(16)
  return x & _M32


def dropout_keep_mask(rows: int, H: int, p: float, seed: int) -> np.ndarray:
  """keep[row, col] of the fused row-wise kernels (csrc/layer_common.h: layer_drop_row / layer_drop_bits16): the
  attention kernels' hash (csrc/mmt_common.h: drop_row_base, drop_pair_finish) on (row, column) -- 16 bits per
  element, one mix per row and one finisher (xor, fold, 24-bit multiply, fold) per column pair."""
  thresh = min(max(int(p * 65536.0 + 0.5), 1), 65535)
  seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
  row = np.arange(rows, dtype=np.uint64).reshape(rows, 1)
  col = np.arange(H, dtype=np.uint64).reshape(1, H)
  row_base = (_mix32(seed_lo ^ (((row >> np.uint64(32)) * np.uint64(0x9E3779B9)) & _M32)) + seed_hi
              + (row & _M32) * np.uint64(0x85EBCA6B)) & _M32
  x = row_base ^ (((col >> np.uint64(1)) * np.uint64(0xC2B2AE35)) & _M32)
  x ^= x >> np.uint64(16); x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0xEB352D)) & _M32; x ^= x >> np.uint64(15)
  bits = np.where((col & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
  keep = bits >= np.uint64(thresh)
  return keep, 65536.0 / (65536.0 - thresh)


def layer_norm(x, gamma, beta, eps=1e-12):
  mu = x.mean(-1, keepdims=True)
  var = ((x - mu) ** 2).mean(-1, keepdims=True)
  rstd = 1.0 / np.sqrt(var + eps)
  return (x - mu) * rstd * gamma + beta, mu[..., 0], rstd[..., 0]


def layer_norm_bwd(dy, x, gamma, eps=1e-12):
  mu = x.mean(-1, keepdims=True)
  rstd = 1.0 / np.sqrt(((x - mu) ** 2).mean(-1, keepdims=True) + eps)
  xh = (x - mu) * rstd
  dyh = dy * gamma
  dx = rstd * (dyh - dyh.mean(-1, keepdims=True) - xh * (dyh * xh).mean(-1, keepdims=True))
  return dx, (dy * xh).sum(0), dy.sum(0)


def residual_block_fwd(o, bias, x, gamma=None, beta=None, keep=None, inv_keep=1.0, eps=1e-12):
  t = o + bias
  if keep is not None:
    t = np.where(keep, t * inv_keep, 0.0)
  x_new = x + t
  h = None if gamma is None else layer_norm(x_new, gamma, beta, eps)[0]
  return x_new, h


def residual_block_bwd(dx_new, dh, x_new, gamma=None, keep=None, inv_keep=1.0, eps=1e-12):
  g = np.zeros_like(x_new) if dx_new is None else dx_new.copy()
  dgamma = dbeta = None
  if gamma is not None:
    dxl, dgamma, dbeta = layer_norm_bwd(dh, x_new, gamma, eps)
    g = g + dxl
  d_o = g if keep is None else np.where(keep, g * inv_keep, 0.0)
  return d_o, g, d_o.sum(0), dgamma, dbeta


def gelu_tanh(z):
  return 0.5 * z * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (z + 0.044715 * z ** 3)))


def gelu_tanh_grad(z):
  k = np.sqrt(2.0 / np.pi)
  t = np.tanh(k * (z + 0.044715 * z ** 3))
  return 0.5 * (1 + t) + 0.5 * z * (1 - t * t) * k * (1 + 3 * 0.044715 * z * z)


# ---- embedding assembly (src/modeling/models/mmt_encoder.py:189-218) ----------
def embed_assemble_fwd(word_ids, seg_ids, word_table, seg_table, gamma, beta, pos_table=None,
                       patch_proj=None, patch_start=2, keep=None, inv_keep=1.0, eps=1e-12):
  """out[b,s] = Dropout(LN(WordEmb[word_ids])) + SegEmb[seg_ids] (+ PosEmb[s]) (+ patches at
  [patch_start, patch_start + n_patch)).  Ids outside the table contribute a zero row (one-hot
  lookup, mmt_encoder.py:110).  `patch_proj` [B, n_patch, H] already includes the projection bias.
  `keep` [B*S, H] is the explicit dropout mask on the LayerNorm output."""
  B, S = word_ids.shape
  V, H = word_table.shape
  ok = (word_ids >= 0) & (word_ids < V)
  we = np.where(ok[..., None], word_table[np.clip(word_ids, 0, V - 1)], 0.0)
  we, mu, rstd = layer_norm(we, gamma, beta, eps)
  if keep is not None:
    we = np.where(keep.reshape(B, S, H), we * inv_keep, 0.0)
  sok = (seg_ids >= 0) & (seg_ids < seg_table.shape[0])
  out = we + np.where(sok[..., None], seg_table[np.clip(seg_ids, 0, seg_table.shape[0] - 1)], 0.0)
  if pos_table is not None:
    out = out + pos_table[:S][None]
  if patch_proj is not None:
    n = patch_proj.shape[1]
    out[:, patch_start:patch_start + n] += patch_proj
  return out


def embed_assemble_bwd(dout, word_ids, seg_ids, word_table, seg_table, gamma, n_patch=0, patch_start=2,
                       keep=None, inv_keep=1.0, eps=1e-12, has_pos=False):
  """Gradients of embed_assemble_fwd: dict(word_table, seg_table, gamma, beta, patch_proj, pos)."""
  B, S = word_ids.shape
  V, H = word_table.shape
  d = dout.reshape(B * S, H)
  ids = word_ids.reshape(-1)
  ok = (ids >= 0) & (ids < V)
  x = np.where(ok[:, None], word_table[np.clip(ids, 0, V - 1)], 0.0)
  t = d if keep is None else np.where(keep, d * inv_keep, 0.0)
  dx, dgamma, dbeta = layer_norm_bwd(t, x, gamma, eps)
  dword = np.zeros_like(word_table)
  np.add.at(dword, ids[ok], dx[ok])
  dseg = np.zeros_like(seg_table)
  sg = seg_ids.reshape(-1)
  sok = (sg >= 0) & (sg < seg_table.shape[0])
  np.add.at(dseg, sg[sok], d[sok])
  res = dict(word_table=dword, seg_table=dseg, gamma=dgamma, beta=dbeta)
  if n_patch:
    res['patch_proj'] = dout[:, patch_start:patch_start + n_patch].copy()
  if has_pos:
    res['pos'] = dout.sum(0)
  return res


# ---- per-row softmax cross-entropy (weighted_sparse_categorical_crossentropy_loss.py:17-43) --------
def softmax_xent(logits, labels):
  """loss[row] = logsumexp(logits[row]) - logits[row, label]; labels outside [0, C) -> 0 (no target).
  Returns (loss, dlogits_per_unit_coef) with dlogits = softmax - onehot (zero rows for no-target)."""
  x = logits.astype(np.float64)
  m = x.max(-1, keepdims=True)
  lse = m[:, 0] + np.log(np.exp(x - m).sum(-1))
  C = x.shape[1]
  ok = (labels >= 0) & (labels < C)
  lab = np.clip(labels, 0, C - 1)
  loss = np.where(ok, lse - x[np.arange(x.shape[0]), lab], 0.0)
  d = np.exp(x - lse[:, None])
  d[np.arange(x.shape[0]), lab] -= 1.0
  d[~ok] = 0.0
  return loss, d


def weighted_loss(row_loss, weight, lmul=None, mask=None, mask_div=1):
  """`weighted_sparse_categorical_crossentropy_loss.py:36-43` with the loss bookkeeping of `pretraining.py:101-109`
  folded in: w_i = weight_i * mask[i // mask_div], l_i = row_loss_i * lmul_i; returns
  (divide_no_nan(sum w l, sum w), d loss / d row_loss)."""
  w = weight.astype(np.float64).reshape(-1)
  if mask is not None:
    w = w * np.repeat(mask.astype(np.float64).reshape(-1), mask_div)
  lm = np.ones_like(w) if lmul is None else lmul.astype(np.float64).reshape(-1)
  num, den = float((w * lm * row_loss.astype(np.float64)).sum()), float(w.sum())
  if den == 0.0:
    return 0.0, np.zeros_like(w)
  return num / den, w * lm / den
