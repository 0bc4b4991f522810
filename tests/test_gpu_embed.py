"""Embedding assembly (mmt_embed_fwd / mmt_embed_bwd) vs the numpy oracle (oracle/layer_ops.py,
restating mmt_encoder.py:189-218).  fp32 outputs: 1e-5 absolute on O(1) values; bf16 outputs are
compared after rounding the oracle's result to bf16 (one ulp = 2^-8 relative).  The word-table
scatter must be bitwise reproducible (no atomics): two runs are compared with torch.equal."""
import numpy as np
import pytest
import torch

from oracle import layer_ops as lo

pytestmark = pytest.mark.gpu


def make_case(B, S, H, V, Vs, n_patch, seed, repeat_id=None, repeat_n=0, bad_ids=False, pos=False):
  rng = np.random.default_rng(seed)
  word_ids = rng.integers(0, V, size=(B, S)).astype(np.int64)
  if repeat_id is not None:       # a heavily repeated id (padding / [MASK]): runs longer than the 32-cut
    flat = word_ids.reshape(-1)
    flat[rng.choice(B * S, size=repeat_n, replace=False)] = repeat_id
  seg_ids = rng.integers(0, Vs, size=(B, S)).astype(np.int64)
  if bad_ids:
    word_ids[0, 3], word_ids[-1, -1], seg_ids[0, 5] = V + 7, -2, Vs
  c = dict(word_ids=word_ids, seg_ids=seg_ids,
           word_table=rng.standard_normal((V, H)).astype(np.float32) * 0.5,
           seg_table=rng.standard_normal((Vs, H)).astype(np.float32) * 0.5,
           gamma=(1 + 0.1 * rng.standard_normal(H)).astype(np.float32),
           beta=(0.1 * rng.standard_normal(H)).astype(np.float32),
           pos_table=(rng.standard_normal((S + 3, H)).astype(np.float32) * 0.5) if pos else None,
           patch=(rng.standard_normal((B, n_patch, H)).astype(np.float32)) if n_patch else None,
           dout=rng.standard_normal((B, S, H)).astype(np.float32))
  return c


def run_gpu(c, dtype, p=0.0, seed=0):
  from mmt_amd import fused
  dev = lambda x, dt=None: None if x is None else torch.from_numpy(x).cuda().to(dt or torch.from_numpy(x).dtype)
  t = {k: dev(c[k]) for k in ('word_ids', 'seg_ids')}
  prm = {k: (None if c[k] is None else dev(c[k]).requires_grad_(True)) for k in ('word_table', 'seg_table', 'gamma', 'beta', 'pos_table')}
  patch = None if c['patch'] is None else dev(c['patch'], dtype).requires_grad_(True)
  out = fused.embed_assemble(t['word_ids'], t['seg_ids'], prm['word_table'], prm['seg_table'], prm['gamma'], prm['beta'],
                             pos_table=prm['pos_table'], patch_proj=patch, p=p, seed=seed, out_dtype=dtype)
  out.backward(dev(c['dout'], dtype))
  torch.cuda.synchronize()
  grads = {k: (None if v is None else v.grad) for k, v in prm.items()}
  grads['patch'] = None if patch is None else patch.grad
  return out, grads


def bf16r(x):
  return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).float().numpy()


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('kw', [
    dict(B=2, S=40, H=64, V=50, Vs=4, n_patch=9),
    dict(B=3, S=70, H=768, V=300, Vs=16, n_patch=25, repeat_id=0, repeat_n=100, pos=True),     # runs cross the 32-cut
    dict(B=2, S=33, H=136, V=20, Vs=3, n_patch=0, bad_ids=True),                               # out-of-range ids, no patches
    dict(B=1, S=130, H=1024, V=7, Vs=2, n_patch=100, repeat_id=3, repeat_n=70),                 # tiny vocab: every run is long
], ids=['small', 'bert-width+pos+repeats', 'bad-ids', 'tiny-vocab'])
def test_embed_assemble_matches_oracle(kw, dtype):
  c = make_case(seed=11, **kw)
  p, seed = 0.25, 0x1234567890ABCDEF
  B, S = c['word_ids'].shape
  H = c['word_table'].shape[1]
  keep, inv_keep = lo.dropout_keep_mask(B * S, H, p, seed)
  f64 = lambda x: None if x is None else x.astype(np.float64)
  patch = c['patch'] if dtype == torch.float32 or c['patch'] is None else bf16r(c['patch'])
  dout = c['dout'] if dtype == torch.float32 else bf16r(c['dout'])
  want = lo.embed_assemble_fwd(c['word_ids'], c['seg_ids'], f64(c['word_table']), f64(c['seg_table']), f64(c['gamma']),
                               f64(c['beta']), pos_table=f64(c['pos_table']), patch_proj=f64(patch), keep=keep, inv_keep=inv_keep)
  ref = lo.embed_assemble_bwd(f64(dout), c['word_ids'], c['seg_ids'], f64(c['word_table']), f64(c['seg_table']),
                              f64(c['gamma']), n_patch=kw['n_patch'], keep=keep, inv_keep=inv_keep, has_pos=kw.get('pos', False))
  out, g = run_gpu(c, dtype, p, seed)
  got = out.detach().float().cpu().numpy()
  if dtype == torch.float32:
    assert np.abs(got - want).max() < 1e-5
  else:
    assert np.abs(got - want).max() <= 2.0 ** -7 * max(1.0, np.abs(want).max())      # bf16 output rounding
  tol = 2e-5 if dtype == torch.float32 else 1e-3      # bf16: only dout / dpatch are low precision; sums stay fp32
  for name, key in (('word_table', 'word_table'), ('seg_table', 'seg_table'), ('gamma', 'gamma'), ('beta', 'beta')):
    gw = ref[key]
    err = np.abs(g[name].double().cpu().numpy() - gw).max() / max(1.0, np.abs(gw).max())
    assert err < tol, (name, err)
  if kw['n_patch']:
    assert np.array_equal(g['patch'].float().cpu().numpy(), ref['patch_proj'].astype(np.float32))     # a plain copy
  if kw.get('pos'):
    gp = g['pos_table'].double().cpu().numpy()
    assert np.abs(gp[:S] - ref['pos']).max() / max(1.0, np.abs(ref['pos']).max()) < tol and not gp[S:].any()


def test_embed_scatter_is_bitwise_reproducible_and_accumulates():
  c = make_case(B=4, S=96, H=256, V=40, Vs=5, n_patch=30, seed=3, repeat_id=1, repeat_n=150)
  _, g1 = run_gpu(c, torch.bfloat16, 0.1, 7)
  _, g2 = run_gpu(c, torch.bfloat16, 0.1, 7)
  for k in ('word_table', 'gamma', 'beta', 'seg_table'):
    assert torch.equal(g1[k], g2[k]), k
  # accumulation into an existing master gradient (nn.Parameter with a preset fp32 .grad)
  from mmt_amd import fused
  wt = torch.nn.Parameter(torch.from_numpy(c['word_table']).cuda())
  wt.grad = torch.ones_like(wt)
  args = [torch.from_numpy(c[k]).cuda() for k in ('word_ids', 'seg_ids')]
  st, ga, be = (torch.from_numpy(c[k]).cuda().requires_grad_(True) for k in ('seg_table', 'gamma', 'beta'))
  out = fused.embed_assemble(args[0], args[1], wt, st, ga, be, p=0.1, seed=7, out_dtype=torch.bfloat16)
  out.backward(torch.from_numpy(c['dout']).cuda().to(torch.bfloat16))
  assert torch.allclose(wt.grad - 1.0, g1['word_table'], atol=1e-5)


def test_wgrad_ragged_k():
  """K not a multiple of 32 (patch rows: B * 63^2): the tail rows of the last slab stage as zeros."""
  from mmt_amd import fused
  torch.manual_seed(0)
  K, M, N = 3 * 49 + 5, 256, 256
  dy = torch.randn(K, M, device='cuda').to(torch.bfloat16)
  x = torch.randn(K, N, device='cuda').to(torch.bfloat16)
  dw, db = torch.zeros(M, N, device='cuda'), torch.zeros(M, device='cuda')
  assert fused.wgrad_accumulate_(dw, dy, x, db)
  want = dy.double().t() @ x.double()
  assert float((dw.double() - want).abs().max()) / float(want.abs().max()) < 2e-5
  assert float((db.double() - dy.double().sum(0)).abs().max()) < 1e-3
