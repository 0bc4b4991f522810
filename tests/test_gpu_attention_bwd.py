"""Backward parity: HIP kernels (through the C ABI + autograd) vs the analytic CPU oracle.

Tolerances: fp32 path 2e-3 absolute (gradients are O(1..10); observed ~1e-5); bf16 path is
compared against the fp64 oracle run on bf16-rounded inputs, relative to each gradient's
max magnitude (3e-2: bf16 outputs + bf16 P/dS operands)."""
import numpy as np
import pytest
import torch

from oracle import attention as oa
from tests._cases import attention_inputs, bf16_round, dense_side_inputs

pytestmark = pytest.mark.gpu


def run_bwd(B, S, N, R, dtype, *, dense, valid=None, radius=1 << 30, g0=0, ng=0, id_mode=1, m=3,
            P=0, r=0, seed=0, scale_before_add=False, use_bias=True, tuning=0):
  import mmt_amd
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed)
  rng = np.random.default_rng(seed + 100)
  dout = rng.standard_normal(q.shape).astype(np.float32)
  if not use_bias:
    bias = None
  if dtype == torch.bfloat16:
    q, k, v, dout = (bf16_round(x) for x in (q, k, v, dout))
    emb = None if emb is None else bf16_round(emb)
    bias = None if bias is None else bf16_round(bias)
  if R == 0:
    id_mode = 0
  mask, ids = dense_side_inputs(B, S, valid, radius, g0, ng, id_mode, m, P, r)
  ref = oa.relative_attention_bwd(dout, q, k, v, emb, bias, mask, ids,
                                  scale_after_add=not scale_before_add)
  dev = lambda x, dt=dtype: None if x is None else torch.from_numpy(x).cuda().to(dt).contiguous()
  tq, tk, tv, te, tb = (None if x is None else dev(x).requires_grad_(True) for x in (q, k, v, emb, bias))
  kw = dict(scale_before_add=scale_before_add)
  if dense:
    out = mmt_amd.relative_attention(tq, tk, tv, te, tb, att_mask=dev(mask, torch.int32),
                                     relative_att_ids=dev(ids, torch.int32), **kw)
  else:
    pat = mmt_amd.AttentionPattern(local_radius=radius, global_start=g0, n_global=ng, id_mode=id_mode,
                                   max_dist=m, patches_per_row=P, core_layers=r)
    vl = None if valid is None else torch.tensor(valid, dtype=torch.int32, device='cuda:0')
    out = mmt_amd.relative_attention(tq, tk, tv, te, tb, pattern=pat, valid_len=vl, tuning=tuning, **kw)
  out.backward(dev(dout))
  torch.cuda.synchronize()
  worst = 0.0
  for name, t in (('dq', tq), ('dk', tk), ('dv', tv), ('drel_emb', te), ('drel_bias', tb)):
    if t is None:
      continue
    got = t.grad.float().cpu().numpy()
    want = ref[name]
    assert np.isfinite(got).all(), name
    if dtype == torch.float32:
      err = np.abs(got - want).max()
      assert err < 2e-3, f'{name}: max abs err {err}'
    else:
      err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
      assert err < 3e-2, f'{name}: max err relative to max |grad| = {err}'
    worst = max(worst, err)
  return worst


DTYPES = [torch.float32, torch.bfloat16]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [
    dict(B=2, S=96, N=2, R=9),
    dict(B=1, S=100, N=2, R=32, m=12, valid=[61]),
    dict(B=1, S=64, N=1, R=0),
    dict(B=1, S=96, N=2, R=5, m=3),                              # ids >= R contribute 0
    dict(B=1, S=96, N=2, R=49, id_mode=2, m=12, P=6, r=2),
    dict(B=1, S=96, N=2, R=9, scale_before_add=True),
    dict(B=1, S=96, N=2, R=9, use_bias=False),
    dict(B=1, S=128, N=2, R=100, m=40),                          # relative vocabulary above 64: the 128-wide table
    dict(B=1, S=112, N=2, R=121, id_mode=2, m=12, P=8, r=4),     # 2-D ids with a 9 x 9 core window
], ids=lambda c: '-'.join(f'{k}{v}' for k, v in c.items() if k in ('S', 'R', 'id_mode')))
def test_dense_operator_backward(cfg, dtype):
  run_bwd(dtype=dtype, dense=True, **cfg)


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [
    dict(B=2, S=256, N=2, R=32, radius=1 << 30, m=12),
    dict(B=2, S=256, N=2, R=32, radius=16, m=12),
    dict(B=2, S=256, N=2, R=32, radius=16, g0=200, ng=8, m=12),
    dict(B=2, S=300, N=2, R=32, radius=64, g0=250, ng=8, m=12, valid=[300, 211]),
    dict(B=1, S=256, N=1, R=32, radius=0, g0=3, ng=40, m=12),
    dict(B=1, S=200, N=2, R=49, radius=20, g0=150, ng=5, id_mode=2, m=12, P=10, r=2),
    # a small image: the cross-modal part ids P^2 + 8 + 2m + 1 (+ 1) = 31 / 32 are BELOW R and index real table rows
    dict(B=1, S=64, N=2, R=49, radius=8, g0=40, ng=2, id_mode=2, m=3, P=4, r=1),
    # 2-D ids with a patch row of >= 32 positions (look-up-table tiles of the lean kernels, bf16): table width 32
    # (r = 1, the reference's *_2d*.yaml) and 64 (r = 2); ragged batch; R below the text id range
    dict(B=2, S=1200, N=2, R=49, radius=64, g0=1100, ng=8, id_mode=2, m=12, P=33, r=1, valid=[1200, 1111]),
    dict(B=1, S=1152, N=2, R=49, radius=40, g0=0, ng=8, id_mode=2, m=12, P=32, r=2),
    dict(B=1, S=1500, N=1, R=25, radius=100, g0=1400, ng=40, id_mode=2, m=12, P=37, r=1),
    dict(B=1, S=96, N=1, R=0, radius=8, g0=0, ng=1),
    dict(B=1, S=512, N=2, R=32, radius=64, g0=400, ng=8, m=12),
    dict(B=1, S=512, N=2, R=41, radius=64, g0=400, ng=8, m=20),            # 1-D ids, table width 64 on the lean path
    dict(B=2, S=64, N=1, R=9, radius=8, g0=10, ng=2, valid=[0, 64]),      # an all-padding example next to a full one
    dict(B=1, S=20, N=2, R=9, radius=4, g0=0, ng=1),                       # shorter than one 32-row tile
    dict(B=1, S=96, N=1, R=1, radius=16, m=0),                             # a single relative id (m = 0)
    # BASELINE config 2 shape: S=1024 = 2 + 28^2 + 238 text, radius 64, 8 globals [786,794) (fp32 there)
    dict(B=1, S=1024, N=2, R=32, radius=64, g0=786, ng=8, m=12),
    dict(B=1, S=320, N=2, R=100, radius=64, g0=250, ng=8, m=40),           # relative vocabulary above 64 (general kernels, 128-wide table)
    dict(B=1, S=210, N=2, R=128, radius=24, g0=150, ng=5, id_mode=2, m=12, P=10, r=4),
], ids=lambda c: '-'.join(f'{k}{v}' for k, v in c.items() if k in ('S', 'radius', 'ng', 'id_mode')))
def test_structured_pattern_backward(cfg, dtype):
  run_bwd(dtype=dtype, dense=False, **cfg)


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_rel_grads_accumulate_into_master_buffers(dtype):
  """MMT_FLAG_ACCUM_REL_GRADS: drel_emb / drel_bias are added to caller buffers (fp32 master
  gradients) -- bit-identical to the overwrite mode's result plus the buffer's old contents."""
  import mmt_amd
  B, S, N, R = 2, 160, 2, 9
  q, k, v, emb, bias = (torch.from_numpy(x).cuda().to(dtype) for x in attention_inputs(B, S, N, R, 5))
  pat = mmt_amd.AttentionPattern(local_radius=24, global_start=150, n_global=4, id_mode=1, max_dist=4)
  out, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat)
  dout = torch.randn_like(out)
  dq, dk, dv, de, db = mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, pattern=pat)
  de0, db0 = torch.randn_like(de), torch.randn_like(db)
  de1, db1 = de0.clone(), db0.clone()
  dq2, dk2, dv2, de2, db2 = mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, pattern=pat,
                                                                rel_grads_accum=(de1, db1))
  assert de2 is de1 and db2 is db1
  assert torch.equal(de1, de0 + de) and torch.equal(db1, db0 + db)
  assert torch.equal(dq2, dq) and torch.equal(dk2, dk) and torch.equal(dv2, dv)
  with pytest.raises(ValueError):
    mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, pattern=pat,
                                        rel_grads_accum=(de1.to(torch.bfloat16), db1))


def test_qkv_fn_adds_fp32_table_gradients_to_master_params():
  """relative_attention_qkv(rel_grad_sinks=...): the fp32 table gradients land in the master
  parameters' .grad (accumulating over calls) and match the autograd path's bf16-rounded ones."""
  import mmt_amd
  torch.manual_seed(3)
  B, S, N, R = 1, 192, 2, 9
  qkv = torch.randn(B, S, 3, N, 64, device='cuda', dtype=torch.bfloat16, requires_grad=True)
  emb_p = torch.nn.Parameter(torch.randn(R, N, 64, device='cuda') * 0.1)
  bias_p = torch.nn.Parameter(torch.randn(R, N, device='cuda') * 0.1)
  pat = mmt_amd.AttentionPattern(local_radius=32, global_start=180, n_global=4, id_mode=1, max_dist=4)
  g = torch.randn(B, S, N, 64, device='cuda', dtype=torch.bfloat16)
  fired = []
  emb_p._mmt_grad_ready_hooks = [lambda p: fired.append('emb')]
  for _ in range(2):      # two micro-steps accumulate
    mmt_amd.relative_attention_qkv(qkv, emb_p.detach().bfloat16(), bias_p.detach().bfloat16(),
                                   rel_grad_sinks=(emb_p, bias_p), pattern=pat).backward(g)
  assert fired == ['emb', 'emb'] and emb_p.grad.dtype == torch.float32
  e2 = emb_p.detach().bfloat16().requires_grad_(True)
  b2 = bias_p.detach().bfloat16().requires_grad_(True)
  mmt_amd.relative_attention_qkv(qkv, e2, b2, pattern=pat).backward(g)
  for got, want in ((emb_p.grad, e2.grad), (bias_p.grad, b2.grad)):
    err = float((got / 2 - want.float()).abs().max()) / float(want.float().abs().max())
    assert err < 1e-2, err


def test_config3_shape_backward_against_oracle():
  """BASELINE config 3 shape (S=4096, radius 64, 8 globals, 1-D ids with m=12, bf16), one head: every gradient
  of the lean backward -- band items, the global-row / global-key chunk partials and their combines riding in
  the next launches, the per-workgroup dE partials -- against the dense fp64 oracle."""
  worst = run_bwd(1, 4096, 1, 32, torch.bfloat16, dense=False, radius=64, g0=3971, ng=8, m=12, seed=5)
  assert worst < 3e-2


# ---- the three forms of the dK/dV pass on the lean bf16 path ----
DKV_FORMS = {'recompute': 0x08, 'handover-wave': 0x10, 'handover-window': 0}      # _lib.MMT_TUNE_BWD_NO_HANDOVER / _BWD_HO_PER_WAVE


@pytest.mark.parametrize('form', list(DKV_FORMS))
@pytest.mark.parametrize('cfg', [
    dict(B=2, S=256, N=2, R=32, radius=16, g0=200, ng=8, m=12),
    dict(B=2, S=300, N=2, R=32, radius=64, g0=251, ng=8, m=12, valid=[300, 211]),     # ragged, odd global start (two key tiles)
    dict(B=1, S=700, N=2, R=32, radius=40, g0=100, ng=5, m=7),                        # radius not a tile multiple
    dict(B=1, S=640, N=2, R=25, radius=64, g0=630, ng=3, m=12),                       # globals in the last tile
    dict(B=1, S=520, N=2, R=0, radius=64, g0=0, ng=8),                                # no relative term
    dict(B=1, S=700, N=2, R=32, radius=96, g0=333, ng=8, m=12),                       # 7 band tiles per block: per-wave form only
    dict(B=1, S=600, N=2, R=32, radius=64, m=12),                                     # no global tokens
    dict(B=2, S=64, N=1, R=9, radius=8, g0=10, ng=2, valid=[0, 64]),
    dict(B=1, S=20, N=2, R=9, radius=4, g0=0, ng=1),
    dict(B=1, S=96, N=1, R=32, radius=200, g0=40, ng=8, m=12),                        # radius beyond the sequence: no split items
    dict(B=1, S=1024, N=2, R=32, radius=64, g0=786, ng=8, m=12),
    # 2-D ids (look-up-table tiles, table width 32): the dQ pass's peeled global keys look their columns up
    dict(B=2, S=1200, N=2, R=49, radius=64, g0=1100, ng=8, id_mode=2, m=12, P=33, r=1, valid=[1200, 1111]),
    dict(B=1, S=200, N=2, R=49, radius=20, g0=150, ng=5, id_mode=2, m=12, P=10, r=2),
], ids=lambda c: '-'.join(f'{k}{v}' for k, v in c.items() if k in ('S', 'radius', 'ng', 'm', 'id_mode')))
def test_dkv_pass_forms(cfg, form):
  """The dK/dV pass that recomputes S / dP / P (attn_bwd_dkv_band_bf16_kernel) and the two forms that read the dQ
  pass's probabilities (attn_bwd_dkv_ho_kernel: per-wave tiles, workgroup window), each against the oracle."""
  run_bwd(dtype=torch.bfloat16, dense=False, tuning=DKV_FORMS[form], **cfg)


def test_dkv_pass_forms_agree_under_dropout():
  """Same seed: the hand-over carries the dQ pass's keep decisions (sign bit of the stored probability), the
  recomputing pass regenerates them -- dK / dV agree to bf16 rounding."""
  import mmt_amd
  B, S, N, R = 2, 1024, 2, 32
  q, k, v, emb, bias = (torch.from_numpy(bf16_round(x)).cuda().bfloat16() for x in attention_inputs(B, S, N, R, seed=21))
  dout = torch.from_numpy(bf16_round(np.random.default_rng(5).standard_normal(q.shape).astype(np.float32))).cuda().bfloat16()
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=771, n_global=8, id_mode=1, max_dist=12)
  kw = dict(pattern=pat, dropout_p=0.25, dropout_seed=77)
  out, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
  res = {}
  for form, tuning in DKV_FORMS.items():
    res[form] = [g.float().cpu().numpy() for g in mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, tuning=tuning, **kw)]
  for form in ('handover-wave', 'handover-window'):
    for a, b in zip(res['recompute'], res[form]):
      assert np.abs(a - b).max() <= 3e-2 * max(1.0, np.abs(a).max()), form
  for a, b in zip(res['handover-wave'], res['handover-window']):      # same arithmetic, different staging
    assert np.abs(a - b).max() <= 1e-6 * max(1.0, np.abs(a).max())


# ---- attention-probability dropout: the in-kernel keep mask against its restatement (oracle.dropout_keep_mask) ----
@pytest.mark.parametrize('path', ['lean_bf16', 'general_f32', 'dense_f32'])
def test_dropout_mask_matches_oracle_forward_and_backward(path):
  """With the restated keep mask handed to the oracle, outputs and every gradient agree as without dropout: the
  forward and the two backward kernels of each path regenerate exactly that mask."""
  import mmt_amd
  B, S, N, R, m = 2, 320, 2, 32, 12
  pdrop, seed = 0.25, 0x1234_5678_9ABC_DEF1
  dtype = torch.bfloat16 if path == 'lean_bf16' else torch.float32
  q, k, v, emb, bias = attention_inputs(B, S, N, R, 3)
  dout = np.random.default_rng(9).standard_normal(q.shape).astype(np.float32)
  if dtype == torch.bfloat16:
    q, k, v, emb, bias, dout = (bf16_round(x) for x in (q, k, v, emb, bias, dout))
  g0, ng, radius = 300, 8, 64
  mask, ids = dense_side_inputs(B, S, None, radius, g0, ng, 1, m)
  keep, keep_prob = oa.dropout_keep_mask(B, N, S, pdrop, seed)
  assert abs(keep.mean() - 0.75) < 0.01
  ref_o, _ = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids, keep_mask=keep, keep_prob=keep_prob, dtype=np.float64)
  ref = oa.relative_attention_bwd(dout, q, k, v, emb, bias, mask, ids, keep_mask=keep, keep_prob=keep_prob)
  dev = lambda x, dt=dtype: torch.from_numpy(x).cuda().to(dt).contiguous()
  tq, tk, tv, te, tb = (dev(x).requires_grad_(True) for x in (q, k, v, emb, bias))
  kw = dict(dropout_p=pdrop, dropout_seed=seed)
  if path == 'dense_f32':
    out = mmt_amd.relative_attention(tq, tk, tv, te, tb, att_mask=dev(mask, torch.int32), relative_att_ids=dev(ids, torch.int32), **kw)
  else:
    pat = mmt_amd.AttentionPattern(local_radius=radius, global_start=g0, n_global=ng, id_mode=1, max_dist=m)
    out = mmt_amd.relative_attention(tq, tk, tv, te, tb, pattern=pat, **kw)
  out.backward(dev(dout))
  tol = 3e-2 if dtype == torch.bfloat16 else 2e-3
  err = np.abs(out.detach().float().cpu().numpy() - ref_o).max()
  assert err < tol, f'out: {err}'
  for name, t in (('dq', tq), ('dk', tk), ('dv', tv), ('drel_emb', te), ('drel_bias', tb)):
    got, want = t.grad.float().cpu().numpy(), ref[name]
    err = np.abs(got - want).max() / (max(1.0, np.abs(want).max()) if dtype == torch.bfloat16 else 1.0)
    assert err < tol, f'{name}: {err}'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_2d_ids_long_sequence_forward_backward_against_oracle(dtype):
  """The `*_2d*.yaml` experiments at the long-sequence shape: S=4096 = 2 + 63^2 + 125, 2-D relative ids
  (MmtRelativePositionGenerator, feature_utils.py:114-184: P=63, 2 core layers, R=49, m=12), radius 64 + 8 global
  tokens, one head: output, LSE and every gradient against the dense oracle (ids >= R contribute 0, q1)."""
  import mmt_amd
  B, S, N, R, P, r, m = 1, 4096, 1, 49, 63, 2, 12
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=21)
  dout = np.random.default_rng(22).standard_normal(q.shape).astype(np.float32)
  if dtype == torch.bfloat16:
    q, k, v, emb, bias, dout = (bf16_round(x) for x in (q, k, v, emb, bias, dout))
  mask, ids = dense_side_inputs(B, S, None, 64, 3971, 8, 2, m, P, r)
  assert ids.max() > R                                     # the cross-modal part ids lie outside the vocabulary
  ref_o, ref_lse = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  ref = oa.relative_attention_bwd(dout, q, k, v, emb, bias, mask, ids)
  dev = lambda x: torch.from_numpy(x).cuda().to(dtype).contiguous()
  tq, tk, tv, te, tb = (dev(x).requires_grad_(True) for x in (q, k, v, emb, bias))
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=2, max_dist=m,
                                 patches_per_row=P, core_layers=r)
  out = mmt_amd.relative_attention(tq, tk, tv, te, tb, pattern=pat)
  out.backward(dev(dout))
  tol_o, tol_g = (2e-2, 3e-2) if dtype == torch.bfloat16 else (1e-3, 2e-3)
  assert np.abs(out.detach().float().cpu().numpy() - ref_o).max() < tol_o
  for name, t in (('dq', tq), ('dk', tk), ('dv', tv), ('drel_emb', te), ('drel_bias', tb)):
    got, want = t.grad.float().cpu().numpy(), ref[name]
    err = np.abs(got - want).max() / (max(1.0, np.abs(want).max()) if dtype == torch.bfloat16 else 1.0)
    assert err < tol_g, f'{name}: {err}'


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_listed_global_tokens_forward_backward_and_materialised_mask(dtype):
  """`mmt_mask_desc.global_index` (ABI 2): a scattered set of global tokens.  The materialised mask
  (`mmt_side_inputs`, materialize_pattern) is bit-exact against the numpy restatement; output and every gradient,
  through the dense operator fed that mask, match the dense oracle; a listed set that is a contiguous run gives
  bit for bit what the range form gives (it takes the structured kernels); the structured entry points refuse a
  listed set themselves."""
  import mmt_amd
  from mmt_amd import _lib
  B, S, N, R, m = 2, 200, 2, 32, 12
  gidx = (3, 4, 77, 150, 151, 199)
  valid = [200, 161]
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=31)
  dout = np.random.default_rng(32).standard_normal(q.shape).astype(np.float32)
  if dtype == torch.bfloat16:
    q, k, v, emb, bias, dout = (bf16_round(x) for x in (q, k, v, emb, bias, dout))
  mask, ids = dense_side_inputs(B, S, valid, 16, 0, 0, 1, m, gidx=gidx)
  pat = mmt_amd.AttentionPattern(local_radius=16, id_mode=1, max_dist=m, global_index=(151, 3, 150, 4, 199, 77, 77))
  vl = torch.tensor(valid, dtype=torch.int32, device='cuda')
  got = mmt_amd.side_inputs(pat, vl, torch.zeros_like(vl), S, materialize_pattern=True)
  assert np.array_equal(got['att_mask'].cpu().numpy(), mask)
  assert np.array_equal(got['relative_att_ids'].cpu().numpy(), ids)
  ref_o, ref_lse = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  ref = oa.relative_attention_bwd(dout, q, k, v, emb, bias, mask, ids)
  dev = lambda x: torch.from_numpy(x).cuda().to(dtype).contiguous()
  tq, tk, tv, te, tb = (dev(x).requires_grad_(True) for x in (q, k, v, emb, bias))
  out = mmt_amd.relative_attention(tq, tk, tv, te, tb, pattern=pat, valid_len=vl)
  out.backward(dev(dout))
  tol_o, tol_g = (2e-2, 3e-2) if dtype == torch.bfloat16 else (1e-3, 2e-3)
  assert np.abs(out.detach().float().cpu().numpy() - ref_o).max() < tol_o
  for name, t in (('dq', tq), ('dk', tk), ('dv', tv), ('drel_emb', te), ('drel_bias', tb)):
    got_g, want = t.grad.float().cpu().numpy(), ref[name]
    err = np.abs(got_g - want).max() / (max(1.0, np.abs(want).max()) if dtype == torch.bfloat16 else 1.0)
    assert err < tol_g, f'{name}: {err}'
  # a contiguous run, listed: the range form's kernels, bit for bit
  run = mmt_amd.AttentionPattern(local_radius=16, id_mode=1, max_dist=m, global_index=(152, 150, 151, 153))
  rng_ = mmt_amd.AttentionPattern(local_radius=16, id_mode=1, max_dist=m, global_start=150, n_global=4)
  o1, l1 = mmt_amd.relative_attention_forward(tq.detach(), tk.detach(), tv.detach(), te.detach(), tb.detach(), pattern=run, valid_len=vl)
  o2, l2 = mmt_amd.relative_attention_forward(tq.detach(), tk.detach(), tv.detach(), te.detach(), tb.detach(), pattern=rng_, valid_len=vl)
  assert torch.equal(o1, o2) and torch.equal(l1, l2)
  # the C entry point itself refuses a listed set without a dense mask
  from mmt_amd import ops
  d = ops._make_desc(tq.detach(), tk.detach(), tv.detach(), o1, R, pat.normalized(), vl, None, -10000.0, False, 0.0, 0)
  ws = torch.empty(1 << 20, dtype=torch.uint8, device='cuda')
  rc = _lib.lib().mmt_attn_fwd(d, tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), te.data_ptr(), tb.data_ptr(), None, None,
                               o1.data_ptr(), None, ws.data_ptr(), ws.numel(), None)
  assert rc == -2 and b'listed global-token set' in _lib.lib().mmt_last_error()      # MMT_E_UNSUPPORTED


def test_listed_global_mask_cache_is_not_keyed_on_the_address():
  """Two batches whose valid_len tensors land at the SAME device address (the caching allocator hands a freed block
  back) with different lengths: the second call must build its own dense mask, not reuse the first one's (the cache
  entry is tied to the tensor object, and dies with it)."""
  import gc
  import mmt_amd
  from mmt_amd import ops
  B, S, N, R, m = 2, 160, 1, 32, 12
  gidx = (5, 9, 120)
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=41)
  dev = lambda x: torch.from_numpy(x).cuda().contiguous()
  tq, tk, tv, te, tb = (dev(x) for x in (q, k, v, emb, bias))
  pat = mmt_amd.AttentionPattern(local_radius=16, id_mode=1, max_dist=m, global_index=gidx)
  ops.clear_pattern_cache()
  outs, ptrs = [], []
  for valid in ([160, 100], [90, 160]):
    vl = torch.tensor(valid, dtype=torch.int32, device='cuda')
    ptrs.append(vl.data_ptr())
    o, _ = mmt_amd.relative_attention_forward(tq, tk, tv, te, tb, pattern=pat, valid_len=vl)
    mask, ids = dense_side_inputs(B, S, valid, 16, 0, 0, 1, m, gidx=gidx)
    ref, _ = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
    assert np.abs(o.cpu().numpy() - ref).max() < 1e-3, valid
    del vl, o
    gc.collect()
    assert 'last' not in ops._DENSE_CACHE          # the entry went with its valid_len tensor
  if ptrs[0] != ptrs[1]:
    pytest.skip('results correct, but the allocator did not hand the block back: address reuse not exercised')
