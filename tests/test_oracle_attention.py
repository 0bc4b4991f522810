"""Oracle (float path) self-checks -- CPU only.

The float oracle is PARITY UNPINNED (no reference fixtures exist, see
oracle/attention.py).  What can be checked here: the numpy restatement and the
independent plain-C restatement agree; the hand-derived backward agrees with
torch autograd of an independently written torch expression and with finite
differences; documented edge semantics (ids >= R contribute 0, additive -10000
mask, never-all-masked rows).
"""
import numpy as np
import pytest
import torch

from oracle import attention as oa
from oracle import c_port
from oracle import side_inputs as si


def make_case(B=2, S=24, N=2, D=8, R=9, seed=0, id_mode=1, m=3, P=3, r=1, valid=None,
              radius=None, g0=0, ng=0):
  rng = np.random.default_rng(seed)
  q, k, v = (rng.standard_normal((B, S, N, D)).astype(np.float32) for _ in range(3))
  emb = (rng.standard_normal((R, N, D)) * 0.5).astype(np.float32)
  bias = (rng.standard_normal((R, N)) * 0.5).astype(np.float32)
  ids = si.relative_ids_from_desc(S, id_mode, m, P, r)
  valid = valid or [S] * B
  if radius is None:
    mask = np.stack([si.add_side_input_features(vl, 0, S, 0)['att_mask'] for vl in valid])
  else:
    mask = np.stack([si.sparse_pattern_mask(S, vl, radius, g0, ng) for vl in valid])
  return q, k, v, emb, bias, mask, np.broadcast_to(ids, (B, S, S)).copy()


@pytest.mark.parametrize('kw', [dict(), dict(valid=[20, 7]), dict(id_mode=2, R=20, m=3, S=30),
                                dict(radius=4, g0=10, ng=2, valid=[24, 18])])
def test_numpy_vs_c_port(kw):
  q, k, v, emb, bias, mask, ids = make_case(**kw)
  o1, l1 = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  o2, l2 = c_port.rel_attention_fwd(q, k, v, emb, bias, mask, ids)
  np.testing.assert_allclose(o1, o2, rtol=2e-5, atol=2e-6)
  np.testing.assert_allclose(l1, l2, rtol=2e-5, atol=2e-6)
  o3, _ = c_port.rel_attention_fwd(q, k, v, emb, bias, mask, ids, acc64=True)
  np.testing.assert_allclose(o1, o3, rtol=2e-5, atol=2e-6)


def test_flags_and_edge_semantics():
  q, k, v, emb, bias, mask, ids = make_case(R=5, m=3)       # ids reach 6 >= R=5
  assert ids.max() >= 5
  o, _ = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  # ids >= R select nothing: same as zero-extending the tables (App. B q1)
  emb2 = np.concatenate([emb, np.zeros((2,) + emb.shape[1:], np.float32)])
  bias2 = np.concatenate([bias, np.zeros((2, bias.shape[1]), np.float32)])
  o2, _ = oa.relative_attention_fwd(q, k, v, emb2, bias2, mask, ids)
  np.testing.assert_allclose(o, o2, rtol=1e-6, atol=1e-6)
  # no ids / no table = plain softmax attention
  o3, _ = oa.relative_attention_fwd(q, k, v, None, None, mask, None)
  qt, kt, vt = (torch.from_numpy(x).permute(0, 2, 1, 3) for x in (q, k, v))
  ref = torch.nn.functional.scaled_dot_product_attention(qt, kt, vt).permute(0, 2, 1, 3)
  np.testing.assert_allclose(o3, ref.numpy(), rtol=1e-4, atol=1e-5)
  # scale-before-add flag changes only the rel term's scaling
  o4, _ = oa.relative_attention_fwd(q, k, v, emb / 8 ** 0.5, bias / 8 ** 0.5, mask, ids,
                                    scale_after_add=False)
  np.testing.assert_allclose(o, o4, rtol=1e-4, atol=1e-5)


def test_additive_mask_is_not_minus_inf():
  # A row whose keys are all masked keeps a (uniform-shifted) softmax over all keys.
  q, k, v, emb, bias, _, ids = make_case(B=1, S=8)
  mask = np.zeros((1, 8, 8), np.int32)
  o, _ = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  o_ref, _ = oa.relative_attention_fwd(q, k, v, emb, bias, np.ones_like(mask), ids)
  np.testing.assert_allclose(o, o_ref, rtol=1e-3, atol=1e-3)


def _torch_ref(q, k, v, emb, bias, mask, ids, keep=None, keep_prob=1.0):
  D = q.shape[-1]
  R = emb.shape[0]
  content = torch.einsum('bqnd,bknd->bnqk', q, k)
  relall = torch.einsum('bqnd,rnd->bnqr', q, emb) + bias.t()[None, :, None, :]
  onehot = torch.nn.functional.one_hot(ids.clamp(0, R), R + 1)[..., :R].to(q.dtype)  # [B,S,S,R]
  rel = torch.einsum('bnqr,bqkr->bnqk', relall, onehot)
  s = (content + rel) / D ** 0.5 + (1 - mask)[:, None].to(q.dtype) * -10000.0
  p = torch.softmax(s, -1)
  if keep is not None:
    p = p * keep / keep_prob
  return torch.einsum('bnqk,bknd->bqnd', p, v)


@pytest.mark.parametrize('kw', [dict(), dict(valid=[16, 9], S=16), dict(radius=3, g0=6, ng=2, S=16),
                                dict(id_mode=2, R=12, S=14)])
@pytest.mark.parametrize('dropout', [False, True])
def test_backward_vs_torch_autograd(kw, dropout):
  q, k, v, emb, bias, mask, ids = make_case(**kw)
  rng = np.random.default_rng(5)
  dout = rng.standard_normal(q.shape)
  B, S, N, _ = q.shape
  keep = (rng.random((B, N, S, S)) > 0.25) if dropout else None
  tq, tk, tv, te, tb = (torch.tensor(x, dtype=torch.float64, requires_grad=True)
                        for x in (q, k, v, emb, bias))
  out = _torch_ref(tq, tk, tv, te, tb, torch.tensor(mask), torch.tensor(ids).long(),
                   None if keep is None else torch.tensor(keep, dtype=torch.float64), 0.75)
  o_np, _ = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids, keep_mask=keep,
                                      keep_prob=0.75, dtype=np.float64)
  np.testing.assert_allclose(o_np, out.detach().numpy(), rtol=1e-9, atol=1e-10)
  out.backward(torch.tensor(dout))
  g = oa.relative_attention_bwd(dout, q, k, v, emb, bias, mask, ids, keep_mask=keep,
                                keep_prob=0.75)
  for name, t in (('dq', tq), ('dk', tk), ('dv', tv), ('drel_emb', te), ('drel_bias', tb)):
    np.testing.assert_allclose(g[name], t.grad.numpy(), rtol=1e-8, atol=1e-10, err_msg=name)


def test_backward_finite_differences():
  q, k, v, emb, bias, mask, ids = make_case(B=1, S=6, N=1, D=4, R=7)
  rng = np.random.default_rng(9)
  dout = rng.standard_normal(q.shape)
  g = oa.relative_attention_bwd(dout, q, k, v, emb, bias, mask, ids)
  def loss(**over):
    a = dict(q=q, k=k, v=v, rel_emb=emb, rel_bias=bias); a.update(over)
    o, _ = oa.relative_attention_fwd(a['q'], a['k'], a['v'], a['rel_emb'], a['rel_bias'], mask, ids,
                                     dtype=np.float64)
    return float((o * dout).sum())
  eps = 1e-6
  for name, arr, key in (('dq', q, 'q'), ('dk', k, 'k'), ('dv', v, 'v'),
                         ('drel_emb', emb, 'rel_emb'), ('drel_bias', bias, 'rel_bias')):
    a = arr.astype(np.float64)
    for idx in list(np.ndindex(a.shape))[::3]:
      p = a.copy(); p[idx] += eps
      n = a.copy(); n[idx] -= eps
      fd = (loss(**{key: p}) - loss(**{key: n})) / (2 * eps)
      assert abs(fd - g[name][idx]) < 1e-6 * max(1.0, abs(fd)), (name, idx, fd, g[name][idx])


def test_dropout_keep_mask_statistics():
  """The attention-dropout keep mask (counter hash of (seed, plane, q, k >> 1), 16 bits per element, 24-bit multiply in
  the per-pair finisher: csrc/mmt_common.h, restated in oracle.attention.dropout_keep_mask): keep rate within 3e-4 of
  1 - p, row / column keep-rate spread binomial, neighbour correlations (along k, along q, diagonals, across planes,
  across a tile) below 2.5e-3 -- for several seeds, including one with a step epoch added."""
  from oracle import attention as oa
  for p in (0.1, 0.25):
    for seed in (12345, 0xDEADBEEF12345678, (77 * 0x9E3779B97F4A7C15 + 5) & ((1 << 64) - 1)):
      keep, kp = oa.dropout_keep_mask(3, 2, 768, p, seed)
      m = keep.reshape(6, 768, 768).astype(np.float64)
      assert abs(m.mean() - kp) < 6e-4 and abs(kp - (1 - p)) < 1e-4
      c = lambda a, b: abs(float(np.corrcoef(a.ravel(), b.ravel())[0, 1]))
      assert c(m[:, :, :-1], m[:, :, 1:]) < 2.5e-3          # the two elements of a pair, and neighbouring pairs
      assert c(m[:, :, :-2], m[:, :, 2:]) < 2.5e-3
      assert c(m[:, :-1], m[:, 1:]) < 2.5e-3                # neighbouring rows
      assert c(m[:-1], m[1:]) < 2.5e-3                      # neighbouring planes
      assert c(m[:, :-1, :-1], m[:, 1:, 1:]) < 2.5e-3 and c(m[:, :-1, 1:], m[:, 1:, :-1]) < 2.5e-3
      assert c(m[:, :, :-64], m[:, :, 64:]) < 2.5e-3 and c(m[:, :-32], m[:, 32:]) < 2.5e-3
      sd = np.sqrt(p * (1 - p) / 768)
      assert abs(m.mean(2).std() - sd) < 0.15 * sd and abs(m.mean(1).std() - sd) < 0.15 * sd


def test_layer_dropout_keep_mask_statistics():
  """The row-wise kernels' dropout mask (round 4: the same hash on (row, column): csrc/layer_common.h, restated in
  oracle.layer_ops.dropout_keep_mask) at the residual block's shape: keep rate, binomial row / column spread, neighbour
  correlations along the row, down the column and on the diagonals."""
  from oracle import layer_ops as lo
  for p in (0.1, 0.5):
    for seed in (1234, 0x9ABCDEF012345678, (3 * 0x9E3779B97F4A7C15 + 11) & ((1 << 64) - 1)):
      keep, inv_keep = lo.dropout_keep_mask(4096, 768, p, seed)
      m = keep.astype(np.float64)
      assert abs(m.mean() - 1.0 / inv_keep) < 6e-4 and abs(1.0 / inv_keep - (1 - p)) < 1e-4
      c = lambda a, b: abs(float(np.corrcoef(a.ravel(), b.ravel())[0, 1]))
      for a, b in ((m[:, :-1], m[:, 1:]), (m[:, :-2], m[:, 2:]), (m[:-1], m[1:]), (m[:-2], m[2:]), (m[:-4], m[4:]),
                   (m[:-1, :-1], m[1:, 1:]), (m[:-1, 1:], m[1:, :-1]), (m[:, :-128], m[:, 128:])):
        assert c(a, b) < 2.5e-3
      assert abs(m.mean(1).std() - np.sqrt(p * (1 - p) / 768)) < 0.1 * np.sqrt(p * (1 - p) / 768)
      assert abs(m.mean(0).std() - np.sqrt(p * (1 - p) / 4096)) < 0.1 * np.sqrt(p * (1 - p) / 4096)
