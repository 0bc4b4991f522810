"""ORACLE (test infrastructure, not product code) -- dense relative attention.

CPU/numpy restatement of the floating-point hot op on the reference's path:
`etcmodel.layers.attention.QkvRelativeAttention` as instantiated by
`etc_layers.RelativeTransformerLayers` from `src/modeling/models/mmt_encoder.py:124-135`
and called with a dense `att_mask[B,S,S]` / `relative_att_ids[B,S,S]` at
`mmt_encoder.py:220-224`.

PARITY UNPINNED (float path): `etcmodel` is an un-vendored, version-less
third-party dependency (`src/README.md:10-11`), absent from /root/reference and
from this image, and the reference holds no test or fixture that exercises the
encoder or the attention output.  TensorFlow / etcmodel are not installed
(ordinary `ModuleNotFoundError`), so no reference output could be generated.
This module follows the published algorithm as specified in SURVEY.md App. A.3;
the uncertain points (App. A.4) are explicit keyword flags so flipping one is a
one-line change.  Self-generated vectors under tests/golden/ are
"self-consistency vectors", not reference vectors.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module.

Op order mirrored (App. A.3):
  content = einsum(Q, K); relall = einsum(Q, E) (+ bias); rel = gather(relall, ids)
  (0 where id >= R: one-hot lookup semantics, App. B q1); s = (content + rel)/sqrt(D);
  s += (1 - att_mask) * -10000; p = softmax_k(s); p = dropout(p); o = einsum(p, V).
"""
from __future__ import annotations

import numpy as np


def _gather_rel(relall: np.ndarray, ids: np.ndarray) -> np.ndarray:
  """relall [N,S,R], ids [S,S] -> rel [N,S,S]; ids outside [0,R) select nothing (0)."""
  R = relall.shape[-1]
  ok = (ids >= 0) & (ids < R)
  safe = np.where(ok, ids, 0)
  rel = np.take_along_axis(relall, np.broadcast_to(safe[None], (relall.shape[0],) + ids.shape),
                           axis=-1)
  return np.where(ok[None], rel, relall.dtype.type(0))


def relative_attention_fwd(q, k, v, rel_emb=None, rel_bias=None, att_mask=None,
                           rel_ids=None, *, scale=None, mask_value=-10000.0,
                           scale_after_add=True, keep_mask=None, keep_prob=1.0,
                           dtype=np.float32, return_probs=False):
  """Dense QkvRelativeAttention forward.

  q, k, v: [B,S,N,D]; rel_emb: [R,N,D]; rel_bias: [R,N] or None;
  att_mask, rel_ids: int [B,S,S] (or [S,S], broadcast) or None.
  keep_mask: optional bool [B,N,S,S] dropout keep mask (p * keep / keep_prob).
  Returns out [B,S,N,D], lse [B,N,S] (natural log of the softmax denominator of s).
  """
  q = np.asarray(q, dtype); k = np.asarray(k, dtype); v = np.asarray(v, dtype)
  B, S, N, D = q.shape
  if scale is None:
    scale = 1.0 / np.sqrt(D)
  scale = dtype(scale)
  out = np.empty((B, S, N, D), dtype)
  lse = np.empty((B, N, S), dtype)
  probs = np.empty((B, N, S, S), dtype) if return_probs else None
  if rel_emb is not None:
    rel_emb = np.asarray(rel_emb, dtype)
  if rel_bias is not None:
    rel_bias = np.asarray(rel_bias, dtype)
  for b in range(B):
    qb = np.transpose(q[b], (1, 0, 2))            # [N,S,D]
    kb = np.transpose(k[b], (1, 0, 2))
    vb = np.transpose(v[b], (1, 0, 2))
    content = np.matmul(qb, np.transpose(kb, (0, 2, 1)))          # [N,S,S]
    if not scale_after_add:
      content = content * scale
    if rel_emb is not None and rel_ids is not None:
      relall = np.einsum('nqd,rnd->nqr', qb, rel_emb).astype(dtype)
      if rel_bias is not None:
        relall = relall + np.transpose(rel_bias, (1, 0))[:, None, :]
      ids = np.asarray(rel_ids)
      ids = ids[b] if ids.ndim == 3 else ids
      content = content + _gather_rel(relall, ids)
    s = content * scale if scale_after_add else content
    if att_mask is not None:
      m = np.asarray(att_mask)
      m = m[b] if m.ndim == 3 else m
      s = s + ((1 - m).astype(dtype) * dtype(mask_value))[None]
    smax = s.max(axis=-1, keepdims=True)
    e = np.exp(s - smax)
    den = e.sum(axis=-1, keepdims=True)
    p = e / den
    lse[b] = (smax + np.log(den))[..., 0]
    if keep_mask is not None:
      p = p * np.asarray(keep_mask[b], dtype) / dtype(keep_prob)
    if return_probs:
      probs[b] = p
    out[b] = np.transpose(np.matmul(p, vb), (1, 0, 2))
  if return_probs:
    return out, lse, probs
  return out, lse


def relative_attention_bwd(dout, q, k, v, rel_emb=None, rel_bias=None, att_mask=None,
                           rel_ids=None, *, scale=None, mask_value=-10000.0,
                           scale_after_add=True, keep_mask=None, keep_prob=1.0,
                           dtype=np.float64):
  """Analytic gradients of `relative_attention_fwd` (hand-derived; checked against
  torch autograd and finite differences in tests/test_oracle_attention.py).

  Returns dict(dq, dk, dv, drel_emb, drel_bias).
  """
  q = np.asarray(q, dtype); k = np.asarray(k, dtype); v = np.asarray(v, dtype)
  dout = np.asarray(dout, dtype)
  B, S, N, D = q.shape
  if scale is None:
    scale = 1.0 / np.sqrt(D)
  scale = dtype(scale)
  dq = np.zeros_like(q); dk = np.zeros_like(k); dv = np.zeros_like(v)
  have_rel = rel_emb is not None and rel_ids is not None
  if have_rel:
    rel_emb = np.asarray(rel_emb, dtype)
    R = rel_emb.shape[0]
    drel_emb = np.zeros_like(rel_emb)
    drel_bias = np.zeros((R, N), dtype) if rel_bias is not None else None
  else:
    drel_emb = drel_bias = None
  _, _, probs = relative_attention_fwd(
      q, k, v, rel_emb, rel_bias, att_mask, rel_ids, scale=scale, mask_value=mask_value,
      scale_after_add=scale_after_add, keep_mask=None, dtype=dtype, return_probs=True)
  for b in range(B):
    qb = np.transpose(q[b], (1, 0, 2)); kb = np.transpose(k[b], (1, 0, 2))
    vb = np.transpose(v[b], (1, 0, 2)); dob = np.transpose(dout[b], (1, 0, 2))
    p = probs[b]                                    # pre-dropout softmax [N,S,S]
    if keep_mask is not None:
      drop = np.asarray(keep_mask[b], dtype) / dtype(keep_prob)
      pd = p * drop
    else:
      drop = None
      pd = p
    dv[b] = np.transpose(np.matmul(np.transpose(pd, (0, 2, 1)), dob), (1, 0, 2))
    dpd = np.matmul(dob, np.transpose(vb, (0, 2, 1)))          # [N,S,S]
    dp = dpd * drop if drop is not None else dpd
    ds = p * (dp - (dp * p).sum(axis=-1, keepdims=True))       # grad wrt masked, scaled score
    g_content = ds * scale                                      # grad wrt QK^T
    g_rel = ds * scale if scale_after_add else ds               # grad wrt gathered rel score
    dqb = np.matmul(g_content, kb)
    dkb = np.matmul(np.transpose(g_content, (0, 2, 1)), qb)
    if have_rel:
      ids = np.asarray(rel_ids)
      ids = ids[b] if ids.ndim == 3 else ids
      drelall = np.zeros((N, S, R), dtype)
      ok = (ids >= 0) & (ids < R)
      for r in range(R):
        sel = (ids == r) & ok
        if sel.any():
          drelall[:, :, r] = (g_rel * sel[None]).sum(axis=-1)
      dqb = dqb + np.einsum('nqr,rnd->nqd', drelall, rel_emb)
      drel_emb += np.einsum('nqr,nqd->rnd', drelall, qb)
      if drel_bias is not None:
        drel_bias += np.transpose(drelall.sum(axis=1), (1, 0))
    dq[b] = np.transpose(dqb, (1, 0, 2)); dk[b] = np.transpose(dkb, (1, 0, 2))
  return dict(dq=dq, dk=dk, dv=dv, drel_emb=drel_emb, drel_bias=drel_bias)


# ---- attention-probability dropout: the keep mask the HIP kernels generate in-kernel ---------------------------
# Restates csrc/mmt_common.h (mix32, drop_row_base, drop_pair_finish, drop_bits16) and the parameter derivation of
# csrc/mmt_api.hip (16-bit threshold, exact keep probability).  The reference draws its mask from TF's stateless
# RNG (tf.nn.dropout inside QkvRelativeAttention, etc_layers/attention.py); no two frameworks share that stream, so
# the contract here is: this mask, exactly, in the forward and in both backward kernels.
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
  x = x & _M32
  x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & _M32
  x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & _M32
  x ^= x >> np.uint64(16)
  return x


def dropout_keep_mask(B: int, N: int, S: int, p: float, seed: int):
  """(keep [B,N,S,S] bool, keep_prob) for attention dropout probability p and the 64-bit seed."""
  t = int(p * 65536.0 + 0.5)
  t = 1 if t < 1 else (65535 if t > 65535 else t)
  seed_lo, seed_hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
  bn = np.arange(B * N, dtype=np.uint64).reshape(B * N, 1, 1)
  q = np.arange(S, dtype=np.uint64).reshape(1, S, 1)
  k = np.arange(S, dtype=np.uint64).reshape(1, 1, S)
  row_base = (_mix32(seed_lo ^ ((bn * np.uint64(0x9E3779B9)) & _M32)) + seed_hi + q * np.uint64(0x85EBCA6B)) & _M32
  x = row_base ^ (((k >> np.uint64(1)) * np.uint64(0xC2B2AE35)) & _M32)
  x ^= x >> np.uint64(16); x = ((x & np.uint64(0xFFFFFF)) * np.uint64(0xEB352D)) & _M32; x ^= x >> np.uint64(15)      # (24-bit multiply: csrc/mmt_common.h)
  bits = np.where((k & np.uint64(1)) == 1, x >> np.uint64(16), x & np.uint64(0xFFFF))
  keep = (bits >= np.uint64(t)).reshape(B, N, S, S)
  return keep, (65536.0 - t) / 65536.0
