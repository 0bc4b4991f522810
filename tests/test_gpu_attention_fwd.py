"""Forward parity: HIP kernels (through the C ABI) vs the dense CPU oracle.

Tolerances: fp32 path 1e-3 absolute on O(1) outputs (north_star; observed ~1e-5);
bf16 path is compared against the oracle run on the bf16-rounded inputs with 2e-2
(bf16 output rounding 2^-9 relative plus bf16 P in the PV product).
"""
import numpy as np
import pytest
import torch

from oracle import attention as oa
from tests._cases import attention_inputs, bf16_round, dense_side_inputs

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3
BF16_TOL = 2e-2


def _to_dev(x, dtype):
  return None if x is None else torch.from_numpy(x).to('cuda:0').to(dtype).contiguous()


def run_case(B, S, N, R, dtype, *, dense, valid=None, radius=1 << 30, g0=0, ng=0, id_mode=1, m=3,
             P=0, r=0, seed=0, scale_before_add=False, use_bias=True, tuning=0):
  import mmt_amd
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed)
  if not use_bias:
    bias = None
  if dtype == torch.bfloat16:
    q, k, v = bf16_round(q), bf16_round(k), bf16_round(v)
    emb = None if emb is None else bf16_round(emb)
    bias = None if bias is None else bf16_round(bias)
  if R == 0:
    id_mode = 0
  mask, ids = dense_side_inputs(B, S, valid, radius, g0, ng, id_mode, m, P, r)
  ref, ref_lse = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids,
                                           scale_after_add=not scale_before_add)
  tq, tk, tv, te, tb = (_to_dev(x, dtype) for x in (q, k, v, emb, bias))
  kw = dict(scale_before_add=scale_before_add)
  if dense:
    out, lse = mmt_amd.relative_attention_forward(
        tq, tk, tv, te, tb, att_mask=_to_dev(mask, torch.int32),
        relative_att_ids=_to_dev(ids, torch.int32), **kw)
  else:
    pat = mmt_amd.AttentionPattern(local_radius=radius, global_start=g0, n_global=ng, id_mode=id_mode,
                                   max_dist=m, patches_per_row=P, core_layers=r)
    vl = None if valid is None else torch.tensor(valid, dtype=torch.int32, device='cuda:0')
    out, lse = mmt_amd.relative_attention_forward(tq, tk, tv, te, tb, pattern=pat, valid_len=vl, tuning=tuning, **kw)
  torch.cuda.synchronize()
  tol = F32_TOL if dtype == torch.float32 else BF16_TOL
  got = out.float().cpu().numpy()
  err = np.abs(got - ref).max()
  lerr = np.abs(lse.cpu().numpy() - ref_lse).max()
  assert np.isfinite(got).all()
  assert err < tol, f'max |out - oracle| = {err}'
  assert lerr < tol, f'max |lse - oracle| = {lerr}'
  return err


DTYPES = [torch.float32, torch.bfloat16]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [
    dict(B=2, S=96, N=2, R=9),
    dict(B=1, S=128, N=3, R=32, m=12),
    dict(B=2, S=100, N=2, R=9, valid=[100, 37]),          # ragged valid lengths, S % 32 != 0
    dict(B=1, S=40, N=1, R=0),                               # no relative term
    dict(B=1, S=160, N=2, R=5, m=3),                         # ids >= R contribute 0 (App. B q1)
    dict(B=1, S=96, N=2, R=49, id_mode=2, m=12, P=6, r=2),   # 2-D ids, part ids >= R
    dict(B=1, S=64, N=2, R=20, id_mode=2, m=3, P=3, r=1, valid=[50]),
    dict(B=1, S=96, N=2, R=9, scale_before_add=True),
    dict(B=1, S=96, N=2, R=9, use_bias=False),
    dict(B=1, S=128, N=2, R=100, m=40),                          # relative vocabulary above 64: the 128-wide table
    dict(B=1, S=112, N=2, R=121, id_mode=2, m=12, P=8, r=4),     # 2-D ids with a 9 x 9 core window
], ids=lambda c: '-'.join(f'{k}{v}' for k, v in c.items() if k in ('S', 'R', 'id_mode')))
def test_dense_operator(cfg, dtype):
  """K3: the literal reference operator (dense int32 att_mask / relative_att_ids)."""
  run_case(dtype=dtype, dense=True, **cfg)


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('cfg', [
    dict(B=2, S=256, N=2, R=32, radius=1 << 30, m=12),                    # segmented only = reference mask
    dict(B=2, S=256, N=2, R=32, radius=16, m=12),                          # band only
    dict(B=2, S=256, N=2, R=32, radius=16, g0=200, ng=8, m=12),            # band + globals
    dict(B=2, S=300, N=2, R=32, radius=64, g0=250, ng=8, m=12, valid=[300, 211]),
    dict(B=1, S=256, N=1, R=32, radius=0, g0=3, ng=40, m=12),              # radius 0, 2 global row blocks
    dict(B=1, S=200, N=2, R=49, radius=20, g0=150, ng=5, id_mode=2, m=12, P=10, r=2),
    # a small image: the cross-modal part ids P^2 + 8 + 2m + 1 (+ 1) = 31 / 32 are BELOW R and index real table rows
    dict(B=1, S=64, N=2, R=49, radius=8, g0=40, ng=2, id_mode=2, m=3, P=4, r=1),
    # 2-D ids with a patch row of >= 32 positions: image x image tiles go through the (dx, dy) look-up table of
    # the lean kernels (bf16); the reference's *_2d*.yaml use r = 1, m = 12, R = 49 (table width 32), r = 2 needs 64
    dict(B=2, S=1200, N=2, R=49, radius=64, g0=1100, ng=8, id_mode=2, m=12, P=33, r=1, valid=[1200, 1111]),
    dict(B=1, S=1152, N=2, R=49, radius=40, g0=0, ng=8, id_mode=2, m=12, P=32, r=2),
    dict(B=1, S=1500, N=1, R=25, radius=100, g0=1400, ng=40, id_mode=2, m=12, P=37, r=1),   # R cuts the text ids
    dict(B=1, S=96, N=1, R=0, radius=8, g0=0, ng=1),
    dict(B=1, S=512, N=2, R=41, radius=64, g0=400, ng=8, m=20),            # 1-D ids, table width 64 on the lean path
    dict(B=2, S=64, N=1, R=9, radius=8, g0=10, ng=2, valid=[0, 64]),      # an all-padding example next to a full one
    dict(B=1, S=20, N=2, R=9, radius=4, g0=0, ng=1),                       # shorter than one 32-row tile
    dict(B=1, S=96, N=1, R=1, radius=16, m=0),                             # a single relative id (m = 0)
    dict(B=1, S=1024, N=2, R=32, radius=64, g0=786, ng=8, m=12),           # BASELINE config 2 shape (N cut)
    dict(B=1, S=320, N=2, R=100, radius=64, g0=250, ng=8, m=40),           # relative vocabulary above 64 (general kernels, 128-wide table)
    dict(B=1, S=210, N=2, R=128, radius=24, g0=150, ng=5, id_mode=2, m=12, P=10, r=4),
], ids=lambda c: '-'.join(f'{k}{v}' for k, v in c.items() if k in ('S', 'radius', 'ng', 'id_mode')))
def test_structured_pattern(cfg, dtype):
  """K1+K2: in-kernel mask/id generation vs the dense oracle fed the materialised pattern."""
  run_case(dtype=dtype, dense=False, **cfg)


def _fwd_kernels():
  from mmt_amd import _lib
  return {'per-wave': _lib.MMT_TUNE_FWD_NO_WIN, 'window': _lib.MMT_TUNE_FWD_FORCE_WIN, 'walk': _lib.MMT_TUNE_FWD_WALK,
          'window-unsplit-rows': _lib.MMT_TUNE_FWD_FORCE_WIN | _lib.MMT_TUNE_FWD_ROWS_ONE_WG,
          'sliding-window': _lib.MMT_TUNE_FWD_PWIN}


@pytest.mark.parametrize('kernel', ['per-wave', 'window', 'window-unsplit-rows', 'walk', 'sliding-window'])
@pytest.mark.parametrize('cfg', [
    dict(B=2, S=256, N=2, R=32, radius=16, g0=200, ng=8, m=12),
    dict(B=2, S=300, N=2, R=32, radius=64, g0=251, ng=8, m=12, valid=[300, 211]),     # ragged, odd global start
    dict(B=1, S=700, N=3, R=32, radius=64, g0=333, ng=16, m=12),                      # two groups of global keys / rows
    dict(B=1, S=700, N=2, R=32, radius=40, g0=100, ng=5, m=7),                        # radius not a tile multiple
    dict(B=1, S=640, N=2, R=25, radius=64, g0=630, ng=3, m=12),                       # globals in the last tile
    dict(B=1, S=520, N=2, R=0, radius=64, g0=0, ng=8),                                # no relative term
    dict(B=1, S=512, N=2, R=41, radius=64, g0=400, ng=8, m=20),                       # wider table (window kernel only when forced)
    dict(B=2, S=64, N=1, R=9, radius=8, g0=10, ng=2, valid=[0, 64]),
    dict(B=1, S=20, N=2, R=9, radius=4, g0=0, ng=1),
    dict(B=1, S=1024, N=2, R=32, radius=64, g0=786, ng=8, m=12),
], ids=lambda c: '-'.join(f'{k}{v}' for k, v in c.items() if k in ('S', 'radius', 'ng', 'm')))
def test_forward_kernels(cfg, kernel):
  """The per-wave kernel (attn_fwd_band.hip), the window kernel (attn_fwd_win.hip: shared K / V window, peeled
  global keys, flipped global rows) and the plane-walk kernel (attn_fwd_walk.hip: persistent workgroups, sliding K / V
  ring, global rows over the walked keys; opt-in) on the same cases, each against the
  oracle; the descriptor's `tuning` switches pick the kernel."""
  run_case(dtype=torch.bfloat16, dense=False, tuning=_fwd_kernels()[kernel], **cfg)


def test_forward_kernels_share_the_dropout_mask():
  """Same seed, same keep decisions: the three bf16 forward kernels agree to output rounding with dropout on (the keep
  mask itself is checked against its restatement in test_gpu_attention_bwd.py)."""
  import mmt_amd
  B, S, N, R = 1, 1024, 2, 32
  q, k, v, emb, bias = (torch.from_numpy(bf16_round(x)).cuda().bfloat16() for x in attention_inputs(B, S, N, R, seed=11))
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=771, n_global=8, id_mode=1, max_dist=12)
  outs = []
  for tuning in _fwd_kernels().values():
    o, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat, dropout_p=0.25, dropout_seed=77, tuning=tuning)
    outs.append((o.float().cpu().numpy(), lse.cpu().numpy()))
  for other in outs[1:]:
    assert np.abs(outs[0][0] - other[0]).max() < BF16_TOL
    assert np.abs(outs[0][1] - other[1]).max() < 1e-3


def test_online_softmax_rescale_is_exercised():
  """A late key with a much larger score forces the running-max rescale (guide rule 26)."""
  import mmt_amd
  B, S, N, R = 1, 192, 1, 9
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=3)
  k[0, 170, 0] = q[0, 5, 0] * 3.0          # spike far from row 5's first tiles
  mask, ids = dense_side_inputs(B, S, None, 1 << 30, 0, 0, 1, 3)
  ref, _ = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  dev = lambda x: torch.from_numpy(x).cuda()
  out, _ = mmt_amd.relative_attention_forward(dev(q), dev(k), dev(v), dev(emb), dev(bias),
                                              pattern=mmt_amd.AttentionPattern(id_mode=1, max_dist=3))
  assert np.abs(out.cpu().numpy() - ref).max() < F32_TOL


def test_strided_views_and_errors():
  import mmt_amd
  from mmt_amd._lib import MmtError
  B, S, N, R = 2, 64, 2, 9
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=4)
  qkv = torch.from_numpy(np.concatenate([q, k, v], axis=-2).reshape(B, S, 3, N, 64)).cuda()
  mask, ids = dense_side_inputs(B, S, None, 1 << 30, 0, 0, 1, 3)
  ref, _ = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  out, _ = mmt_amd.relative_attention_forward(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2],
                                              torch.from_numpy(emb).cuda(), torch.from_numpy(bias).cuda(),
                                              pattern=mmt_amd.AttentionPattern(id_mode=1, max_dist=3))
  assert np.abs(out.cpu().numpy() - ref).max() < F32_TOL
  with pytest.raises(MmtError):          # head size other than 64 is not built
    x = torch.zeros(1, 32, 1, 32, device='cuda')
    mmt_amd.relative_attention_forward(x, x, x)
  with pytest.raises(RuntimeError):      # no CPU fallback
    x = torch.zeros(1, 32, 1, 64)
    mmt_amd.relative_attention_forward(x, x, x)


def test_config3_shape_against_oracle_sample():
  """BASELINE config 3 shape (S=4096, radius 64, 8 globals, bf16): two heads checked against
  the dense oracle, all heads checked for finiteness and softmax normalisation via V=1."""
  import mmt_amd
  B, S, N, R = 1, 4096, 12, 32
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=7)
  q, k, v, emb, bias = (bf16_round(x) for x in (q, k, v, emb, bias))
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=8, id_mode=1, max_dist=12)
  dev = lambda x: torch.from_numpy(x).cuda().to(torch.bfloat16)
  out, lse = mmt_amd.relative_attention_forward(dev(q), dev(k), dev(v), dev(emb), dev(bias), pattern=pat)
  got = out.float().cpu().numpy()
  assert np.isfinite(got).all()
  mask, ids = dense_side_inputs(1, S, None, 64, 3971, 8, 1, 12)
  for n in (0, 11):
    sl = slice(n, n + 1)
    ref, ref_lse = oa.relative_attention_fwd(q[:, :, sl], k[:, :, sl], v[:, :, sl], emb[:, sl], bias[:, sl],
                                             mask, ids)
    assert np.abs(got[:, :, sl] - ref).max() < BF16_TOL
    assert np.abs(lse[:, sl].cpu().numpy() - ref_lse).max() < BF16_TOL
  ones = torch.ones_like(dev(v))
  out1, _ = mmt_amd.relative_attention_forward(dev(q), dev(k), ones, dev(emb), dev(bias), pattern=pat)
  assert float((out1.float() - 1).abs().max()) < 1e-2     # rows of P sum to 1


@pytest.mark.parametrize('ng', [8, 32, 128])
def test_config5_shape_forward_backward(ng):
  """BASELINE config 5 shape (S=8192 = 2+88^2+446, radius 64, g globals, bf16): one head against the
  dense oracle (forward), all heads finite + normalised; backward finite and dV rows sum to dO."""
  import mmt_amd
  B, S, N, R = 1, 8192, 2, 32
  g0 = 2 + 88 * 88
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=11)
  q, k, v, emb, bias = (bf16_round(x) for x in (q, k, v, emb, bias))
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=g0, n_global=ng, id_mode=1, max_dist=12)
  dev = lambda x: torch.from_numpy(x).cuda().to(torch.bfloat16)
  tq, tk, tv = (dev(x).requires_grad_(True) for x in (q, k, v))
  te, tb = dev(emb).requires_grad_(True), dev(bias).requires_grad_(True)
  out = mmt_amd.relative_attention(tq, tk, tv, te, tb, pattern=pat)
  got = out.detach().float().cpu().numpy()
  assert np.isfinite(got).all()
  mask, ids = dense_side_inputs(1, S, None, 64, g0, ng, 1, 12)
  ref, _ = oa.relative_attention_fwd(q[:, :, :1], k[:, :, :1], v[:, :, :1], emb[:, :1], bias[:, :1], mask, ids)
  assert np.abs(got[:, :, :1] - ref).max() < BF16_TOL
  dout = torch.ones_like(out)
  out.backward(dout)
  for t in (tq, tk, tv, te, tb):
    assert torch.isfinite(t.grad.float()).all()
  # every row of P sums to 1, so sum_k dV[k] = sum_q dO[q] per head
  dv_sum = tv.grad.float().sum(dim=1)
  assert float((dv_sum - float(S)).abs().max()) < 0.02 * S


def test_split_row_groups_long_sequence():
  """The window kernel's row groups walked by several workgroups (S >= 4096 with the host's arrival counters): two
  groups of global rows (12 tokens), a ragged batch; against the oracle without dropout, and with dropout against the
  one-workgroup form and the per-wave kernel (same mask: tests above) -- only summation order may differ."""
  import mmt_amd
  from mmt_amd import _lib
  B, S, N, R = 2, 4096, 1, 32
  q, k, v, emb, bias = attention_inputs(B, S, N, R, seed=21)
  q, k, v, emb, bias = (bf16_round(x) for x in (q, k, v, emb, bias))
  valid = [4096, 3990]
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=3971, n_global=12, id_mode=1, max_dist=12)
  dev = lambda x: torch.from_numpy(x).cuda().to(torch.bfloat16)
  vl = torch.tensor(valid, dtype=torch.int32, device='cuda:0')
  args = (dev(q), dev(k), dev(v), dev(emb), dev(bias))
  out, lse = mmt_amd.relative_attention_forward(*args, pattern=pat, valid_len=vl)
  mask, ids = dense_side_inputs(B, S, valid, 64, 3971, 12, 1, 12)
  ref, ref_lse = oa.relative_attention_fwd(q, k, v, emb, bias, mask, ids)
  got = out.float().cpu().numpy()
  assert np.isfinite(got).all()
  assert np.abs(got - ref).max() < BF16_TOL and np.abs(lse.cpu().numpy() - ref_lse).max() < BF16_TOL
  outs = {}
  for name, tuning in (('split', 0), ('one', _lib.MMT_TUNE_FWD_FORCE_WIN | _lib.MMT_TUNE_FWD_ROWS_ONE_WG), ('per-wave', _lib.MMT_TUNE_FWD_NO_WIN)):
    outs[name] = mmt_amd.relative_attention_forward(*args, pattern=pat, valid_len=vl, dropout_p=0.25, dropout_seed=99, tuning=tuning)
  for name in ('one', 'per-wave'):
    assert float((outs['split'][0].float() - outs[name][0].float()).abs().max()) < BF16_TOL, name
    assert float((outs['split'][1] - outs[name][1]).abs().max()) < 1e-3, name
  # the counters are left zero: a second call gives the same bits
  again, _ = mmt_amd.relative_attention_forward(*args, pattern=pat, valid_len=vl)
  assert torch.equal(again, out)
