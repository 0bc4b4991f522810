"""Size-independent properties of the attention path at the FULL BASELINE config-3 size (B=4, S=4096, N=12, D=64,
radius 64 + 8 global tokens, R=32, bf16) -- where the dense CPU oracle is too slow to run every head.  Each
property follows from the operator's definition and involves every tile class, the
global-row items and their combine kernels:

  * softmax rows sum to one          -> V = 1 gives O = 1 (any mask, any relative term)
  * the output is linear in V        -> O(V1 + V2) = O(V1) + O(V2)
  * (batch, head) planes are independent and the kernels are deterministic -> permuting planes permutes the
    outputs BIT-EXACTLY (band rows; the 8 global tokens' rows to output rounding: the plane-walk kernel merges them
    from as many partial sums as the plane has runs, 10 or 11 by the plane's place in the grid); two runs are
    bit-identical (forward and backward)
  * dO = 1 gives dV[k,:] = sum_q P[q,k], so sum_k dV[k,d] = S; and sum_k dS[q,k] = 0 for every row, so the
    relative-bias gradient sums to zero over the ids (all 1-D ids are inside the table).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, S, N, D, R, M = 4, 4096, 12, 64, 32, 12


def _pattern(id_mode=1):
  import mmt_amd
  if id_mode == 2:
    return mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=2, max_dist=M,
                                    patches_per_row=63, core_layers=1)
  return mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=1, max_dist=M)


def _inputs(seed=0):
  g = torch.Generator(device='cuda').manual_seed(seed)
  q, k, v = (torch.randn(B, S, N, D, device='cuda', generator=g).to(torch.bfloat16) for _ in range(3))
  emb = (torch.randn(R, N, D, device='cuda', generator=g) * 0.1).to(torch.bfloat16)
  bias = (torch.randn(R, N, device='cuda', generator=g) * 0.1).to(torch.bfloat16)
  return q, k, v, emb, bias


@pytest.mark.parametrize('id_mode', [1, 2], ids=['ids1d', 'ids2d'])
def test_rows_sum_to_one_and_output_is_linear_in_v(id_mode):
  import mmt_amd
  q, k, v, emb, bias = _inputs(1)
  pat = _pattern(id_mode)
  valid = torch.tensor([S, 4000, S, 3000], dtype=torch.int32, device='cuda')      # ragged batch
  ones = torch.ones_like(v)
  o1, _ = mmt_amd.relative_attention_forward(q, k, ones, emb, bias, pattern=pat, valid_len=valid)
  # P is rounded to bf16 before the P.V product (8 bits): the row sum of up to 145 rounded terms
  assert float((o1.float() - 1).abs().max()) < 4e-3
  v2 = torch.randn_like(v)
  oa, _ = mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat, valid_len=valid)
  ob, _ = mmt_amd.relative_attention_forward(q, k, v2, emb, bias, pattern=pat, valid_len=valid)
  oab, _ = mmt_amd.relative_attention_forward(q, k, (v.float() + v2.float()).to(torch.bfloat16), emb, bias, pattern=pat,
                                              valid_len=valid)
  err = (oab.float() - (oa.float() + ob.float())).abs().max()
  assert float(err) < 6e-2                 # three bf16-rounded outputs of magnitude <= ~4 and the rounded V sum


def test_planes_are_independent_and_runs_are_bit_identical():
  import mmt_amd
  q, k, v, emb, bias = _inputs(2)
  pat = _pattern()
  kw = dict(pattern=pat, dropout_p=0.0)
  o, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
  o2, lse2 = mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
  assert torch.equal(o, o2) and torch.equal(lse, lse2)
  perm_b = torch.tensor([2, 0, 3, 1], device='cuda')
  perm_n = torch.randperm(N, device='cuda', generator=torch.Generator(device='cuda').manual_seed(3))
  qp, kp, vp = (t[perm_b][:, :, perm_n].contiguous() for t in (q, k, v))
  op, lsep = mmt_amd.relative_attention_forward(qp, kp, vp, emb[:, perm_n].contiguous(), bias[:, perm_n].contiguous(), **kw)
  want, want_lse = o[perm_b][:, :, perm_n], lse[perm_b][:, perm_n]
  band = torch.ones(S, dtype=torch.bool, device='cuda')
  band[3971:3979] = False                                   # the global tokens' rows
  assert torch.equal(op[:, band], want[:, band])
  assert torch.equal(lsep[:, :, band], want_lse[:, :, band])
  # rows of the global tokens: the same sums in a partition that depends on the plane's place in the grid
  assert float((op[:, ~band].float() - want[:, ~band].float()).abs().max()) < 2e-3
  assert float((lsep[:, :, ~band] - want_lse[:, :, ~band]).abs().max()) < 1e-5
  dout = torch.randn_like(o)
  g1 = mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, o, lse, **kw)
  g2 = mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, o, lse, **kw)
  for a, b in zip(g1, g2):
    assert torch.equal(a, b)


@pytest.mark.parametrize('id_mode', [1, 2], ids=['ids1d', 'ids2d'])
def test_backward_column_and_row_sum_identities(id_mode):
  import mmt_amd
  q, k, v, emb, bias = _inputs(4)
  pat = _pattern(id_mode)
  o, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat)
  dout = torch.ones_like(o)
  dq, dk, dv, demb, dbias = mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, o, lse, pattern=pat)
  # dV[k,:] = sum_q P[q,k]: every head dim equal, and summed over the keys it counts the query rows
  col = dv.float().sum(1)                                        # [B, N, D]
  assert float((col - S).abs().max()) < 0.02 * S ** 0.5 + 8      # 4096 bf16-rounded dV rows (each P sum ~1, 2^-9 relative)
  assert float((dv.float() - dv.float()[..., :1]).abs().max()) < 2e-2 * float(dv.float().abs().max())
  # rows of dS sum to zero -> so do the relative-bias gradients over the ids, per head
  # (1-D ids only: with 2-D ids the image x text pairs carry the part ids >= R, which have no table row, App. B q1)
  if id_mode == 1:
    scale = float(dbias.abs().max()) + 1e-6
    assert float(dbias.sum(0).abs().max()) < 2e-2 * scale * R
  for t in (dq, dk, dv, demb, dbias):
    assert torch.isfinite(t.float()).all()
