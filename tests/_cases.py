"""Shared seeded input builders for the parity tests (oracle vs HIP path)."""
import numpy as np

from oracle import side_inputs as si


def attention_inputs(B, S, N, R, seed=0, D=64, scale_q=1.0):
  rng = np.random.default_rng(seed)
  q = (rng.standard_normal((B, S, N, D)) * scale_q).astype(np.float32)
  k = rng.standard_normal((B, S, N, D)).astype(np.float32)
  v = rng.standard_normal((B, S, N, D)).astype(np.float32)
  emb = (rng.standard_normal((R, N, D)) * 0.5).astype(np.float32) if R else None
  bias = (rng.standard_normal((R, N)) * 0.5).astype(np.float32) if R else None
  return q, k, v, emb, bias


def dense_side_inputs(B, S, valid, radius, g0, ng, id_mode, m, P=0, r=0, gidx=None):
  """Materialised [B,S,S] mask + ids for a pattern (what the reference would be fed)."""
  valid = valid if valid is not None else [S] * B
  mask = np.stack([si.sparse_pattern_mask(S, vl, radius, g0, ng, gidx) for vl in valid]).astype(np.int32)
  if id_mode:
    ids = si.relative_ids_from_desc(S, id_mode, m, P, r)
    ids = np.broadcast_to(ids, (B, S, S)).astype(np.int32).copy()
  else:
    ids = None
  return mask, ids


def bf16_round(x):
  import torch
  return torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
