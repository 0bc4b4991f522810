"""ORACLE (test infrastructure, not product code) -- the element-wise chain of one residual
block, restated with numpy: bias add, dropout (explicit keep mask), residual add, LayerNorm
(eps 1e-12), tanh-GELU, and their gradients.

Follows SURVEY.md App. A.3 (etcmodel ResidualBlock / DenseLayers in pre-activation order; the
package is un-vendored, so this part of the float path is PARITY UNPINNED like
oracle/attention.py) and `src/modeling/models/mmt_encoder.py:53-54` (approximate GELU).
The dropout keep mask restates the counter hash of the product kernels bit for bit
(`csrc/fused_layer.hip: drop_bits16`, `csrc/mmt_common.h: mix32`) so masks can be compared
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
  """keep[row, col] of the fused kernels: 16 bits per element, one mix per element pair."""
  thresh = min(max(int(p * 65536.0 + 0.5), 1), 65535)
  idx = np.arange(rows * H, dtype=np.uint64)
  pair = idx >> np.uint64(1)
  seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
  inner = _mix32((pair >> np.uint64(32)) ^ seed_hi)
  h = _mix32(((pair & _M32) * np.uint64(0x9E3779B9) + inner + seed_lo) & _M32)
  bits = np.where(idx & np.uint64(1), h >> np.uint64(16), h & np.uint64(0xFFFF))
  keep = (bits >= np.uint64(thresh)).reshape(rows, H)
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
