"""Fused residual-block kernels (include/mmt_layer.h) vs the numpy oracle.

fp32: 1e-4 abs on O(1) values (north_star bar is 1e-3); bf16: compared on bf16-rounded inputs
with 3e-2 (one bf16 rounding of O(1..4) outputs).  The dropout keep mask must match the oracle's
restatement of the counter hash exactly."""
import numpy as np
import pytest
import torch

from oracle import layer_ops as lo
from tests._cases import bf16_round

pytestmark = pytest.mark.gpu

DT = [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)]


def _mk(rows, H, seed, dtype):
  rng = np.random.default_rng(seed)
  a = lambda *s: rng.standard_normal(s).astype(np.float32)
  o, x, dxn, dh = a(rows, H), a(rows, H), a(rows, H), a(rows, H)
  bias, gamma, beta = a(H) * 0.1, 1 + 0.1 * a(H), 0.1 * a(H)
  if dtype == torch.bfloat16:
    o, x, dxn, dh = (bf16_round(t) for t in (o, x, dxn, dh))
  return o, x, dxn, dh, bias, gamma, beta


def _dev(t, dtype=None):
  out = torch.from_numpy(np.ascontiguousarray(t)).cuda()
  return out if dtype is None else out.to(dtype)


@pytest.mark.parametrize('dtype,tol', DT, ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,H', [(37, 128), (64, 768), (5, 1024), (300, 2048), (16, 8)])
@pytest.mark.parametrize('p', [0.0, 0.1])
@pytest.mark.parametrize('has_ln', [True, False])
def test_residual_block(rows, H, p, has_ln, dtype, tol):
  from mmt_amd import fused
  o, x, dxn, dh, bias, gamma, beta = _mk(rows, H, rows + H, dtype)
  seed = 0x1234_5678_9ABC + rows
  keep, inv_keep = lo.dropout_keep_mask(rows, H, p, seed) if p else (None, 1.0)
  to, tx = _dev(o, dtype).requires_grad_(True), _dev(x, dtype).requires_grad_(True)
  tb = _dev(bias).requires_grad_(True)
  tg = _dev(gamma).requires_grad_(True) if has_ln else None
  tbt = _dev(beta).requires_grad_(True) if has_ln else None
  x_new, h = fused.residual_block(to, tb, tx, tg, tbt, 1e-12, p, seed)
  want_x, want_h = lo.residual_block_fwd(o, bias, x, gamma if has_ln else None, beta, keep, inv_keep)
  if p:   # exact mask check: dropped positions are exactly x
    got_keep = (x_new.detach().float().cpu().numpy() != _dev(x, dtype).float().cpu().numpy())
    assert not (got_keep & ~keep).any()         # a dropped position is never modified
    assert (got_keep == keep).mean() > 0.99     # kept positions may round back to x in bf16
    assert abs(keep.mean() - (1 - p)) < 0.02 + 2.0 / np.sqrt(rows * H)
  assert np.abs(x_new.detach().float().cpu().numpy() - want_x).max() < tol
  loss_terms = [(x_new, _dev(dxn, dtype))]
  if has_ln:
    xn_for_ln = x_new.detach().float().cpu().numpy().astype(np.float64)    # LN sees the stored (rounded) x_new
    want_h = lo.layer_norm(xn_for_ln, gamma, beta)[0]
    assert np.abs(h.detach().float().cpu().numpy() - want_h).max() < tol
    loss_terms.append((h, _dev(dh, dtype)))
  sum((a.float() * b.float()).sum() for a, b in loss_terms).backward()
  w_do, w_dx, w_db, w_dg, w_dbt = lo.residual_block_bwd(
      dxn.astype(np.float64), dh.astype(np.float64) if has_ln else None,
      x_new.detach().float().cpu().numpy().astype(np.float64), gamma if has_ln else None, keep, inv_keep)
  scale = lambda w: max(1.0, np.abs(w).max())
  assert np.abs(to.grad.float().cpu().numpy() - w_do).max() / scale(w_do) < tol
  assert np.abs(tx.grad.float().cpu().numpy() - w_dx).max() / scale(w_dx) < tol
  assert np.abs(tb.grad.cpu().numpy() - w_db).max() / scale(w_db) < tol
  if has_ln:
    assert np.abs(tg.grad.cpu().numpy() - w_dg).max() / scale(w_dg) < tol
    assert np.abs(tbt.grad.cpu().numpy() - w_dbt).max() / scale(w_dbt) < tol


@pytest.mark.parametrize('dtype,tol', DT, ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,H', [(33, 128), (7, 768), (1024, 768)])
def test_layer_norm(rows, H, dtype, tol):
  from mmt_amd import fused
  _, x, _, dy, _, gamma, beta = _mk(rows, H, 3 * rows + H, dtype)
  tx = _dev(x, dtype).requires_grad_(True)
  tg, tb = _dev(gamma).requires_grad_(True), _dev(beta).requires_grad_(True)
  y = fused.layer_norm(tx, tg, tb, 1e-12)
  assert np.abs(y.detach().float().cpu().numpy() - lo.layer_norm(x.astype(np.float64), gamma, beta)[0]).max() < tol
  y.backward(_dev(dy, dtype))
  dx, dg, db = lo.layer_norm_bwd(dy.astype(np.float64), x.astype(np.float64), gamma)
  assert np.abs(tx.grad.float().cpu().numpy() - dx).max() < tol * max(1, np.abs(dx).max())
  assert np.abs(tg.grad.cpu().numpy() - dg).max() < tol * max(1, np.abs(dg).max())
  assert np.abs(tb.grad.cpu().numpy() - db).max() < tol * max(1, np.abs(db).max())


@pytest.mark.parametrize('dtype,tol', DT, ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,H', [(33, 512), (600, 3072), (3, 8)])
def test_bias_gelu(rows, H, dtype, tol):
  from mmt_amd import fused
  rng = np.random.default_rng(rows)
  u = (rng.standard_normal((rows, H)) * 2).astype(np.float32); dy = rng.standard_normal((rows, H)).astype(np.float32)
  bias = (rng.standard_normal(H) * 0.5).astype(np.float32)
  if dtype == torch.bfloat16:
    u, dy = bf16_round(u), bf16_round(dy)
  tu, tb = _dev(u, dtype).requires_grad_(True), _dev(bias).requires_grad_(True)
  y = fused.bias_gelu(tu, tb)
  z = u.astype(np.float64) + bias
  ref = lo.gelu_tanh(z)
  assert (np.abs(y.detach().float().cpu().numpy() - ref) / np.maximum(1.0, np.abs(ref))).max() < tol
  y.backward(_dev(dy, dtype))
  du = dy * lo.gelu_tanh_grad(z)
  assert (np.abs(tu.grad.float().cpu().numpy() - du) / np.maximum(1.0, np.abs(du))).max() < tol
  assert np.abs(tb.grad.cpu().numpy() - du.sum(0)).max() < tol * max(1.0, np.abs(du.sum(0)).max())


def test_errors():
  from mmt_amd import fused
  from mmt_amd._lib import MmtError
  x = torch.zeros(4, 12, device='cuda')            # H not a multiple of 8
  with pytest.raises(MmtError):
    fused.layer_norm(x, torch.ones(12, device='cuda'), torch.zeros(12, device='cuda'))
  with pytest.raises(RuntimeError, match='GPU only'):
    fused.layer_norm(torch.zeros(4, 16), torch.ones(16), torch.zeros(16))


@pytest.mark.parametrize('n', [8, 1000, 768 * 3072 + 5])
@pytest.mark.parametrize('gdt', [torch.float32, torch.bfloat16])
def test_accumulate_grad(n, gdt):
  from mmt_amd import fused
  torch.manual_seed(n)
  acc = torch.randn(n, device='cuda')
  g = torch.randn(n, device='cuda').to(gdt)
  want = acc + g.float()
  fused.accumulate_grad_(acc, g)
  assert torch.equal(acc, want)        # one fp32 add per element: bit-exact


def test_fused_adamw_matches_torch_adamw():
  """mmt_adamw_step (flat, clip factor folded in, bf16 shadow, grad cleared) vs torch.optim.AdamW."""
  from mmt_amd import configs, distribute, optimization
  def make():
    torch.manual_seed(0)
    m = torch.nn.Module()
    m.dense_weight = torch.nn.Parameter(torch.randn(130, 70, device='cuda'))
    m.dense_bias = torch.nn.Parameter(torch.randn(130, device='cuda'))
    m.layer_norm = torch.nn.LayerNorm(70).cuda()
    return m
  cfg = configs.OptimizerConfig(initial_learning_rate=1e-2)
  ref_m, our_m = make(), make()
  ref_opt = optimization.create_optimizer(ref_m, cfg)
  reducer = distribute.DataParallelStrategy(None).make_reducer(list(our_m.parameters()))
  our_opt = optimization.create_optimizer(our_m, cfg, reducer=reducer)
  assert isinstance(our_opt, optimization.FusedAdamW)
  for step in range(3):
    reducer.zero_grad()
    for (_, pr), (_, po) in zip(ref_m.named_parameters(), our_m.named_parameters()):
      g = torch.randn_like(pr) * (3.0 if step == 1 else 0.1)
      pr.grad = g.clone()
      po.grad.copy_(g)
    torch.nn.utils.clip_grad_norm_(ref_m.parameters(), 1.0)
    scale = reducer.clip_by_global_norm(1.0, apply=False)
    ref_opt.step()
    our_opt.step(grad_scale=scale)
    for (n, pr), (_, po) in zip(ref_m.named_parameters(), our_m.named_parameters()):
      assert float((pr - po).abs().max()) < 2e-6, (step, n)
      assert torch.equal(po._mmt_shadow, po.detach().to(torch.bfloat16)), n
      assert float(po.grad.abs().max()) == 0.0


@pytest.mark.parametrize('K,M,N', [(64, 128, 256), (4096, 768, 768), (2048, 2304, 768), (1024, 768, 3072), (96, 256, 512)])
@pytest.mark.parametrize('use_ws', [True, False])
def test_wgrad_accumulate(K, M, N, use_ws):
  """dW += dY^T X (split-K MFMA kernel, fp32 accumulate) vs an fp64 torch product."""
  from mmt_amd import _lib, fused
  torch.manual_seed(K + M)
  dy = torch.randn(K, M, device='cuda').to(torch.bfloat16)
  x = torch.randn(K, N, device='cuda').to(torch.bfloat16)
  dw0 = torch.randn(M, N, device='cuda')
  db0 = torch.randn(M, device='cuda')
  dw, db = dw0.clone(), db0.clone()
  if use_ws:
    assert fused.wgrad_accumulate_(dw, dy, x, db)
  else:   # no workspace: float-atomic epilogue
    _lib.check(_lib.lib().mmt_wgrad_bias_accumulate(dw.data_ptr(), N, db.data_ptr(), dy.data_ptr(), M, x.data_ptr(), N,
                                                    M, N, K, None, 0, torch.cuda.current_stream().cuda_stream))
  want = dw0.double() + dy.double().t() @ x.double()
  err = float((dw.double() - want).abs().max()) / float(want.abs().max())
  assert err < 2e-5, err
  want_b = db0.double() + dy.double().sum(0)            # bias gradient from the same pass over dy
  err_b = float((db.double() - want_b).abs().max()) / float(want_b.abs().max())
  assert err_b < 2e-5, err_b
  dw2 = dw0.clone()                                     # without dbias: the plain entry point, same dW
  assert fused.wgrad_accumulate_(dw2, dy, x)
  if use_ws:
    assert torch.equal(dw2, dw)                         # slab mode is bitwise reproducible
  assert not fused.wgrad_accumulate_(dw, dy[:, :100].contiguous(), x)       # unsupported shape -> caller falls back


def test_linear_fn_matches_autograd():
  from mmt_amd import layers
  torch.manual_seed(0)
  w = torch.nn.Parameter(torch.randn(256, 512, device='cuda') * 0.05)
  b = torch.nn.Parameter(torch.randn(256, device='cuda') * 0.05)
  x = torch.randn(4, 64, 512, device='cuda', dtype=torch.bfloat16, requires_grad=True)
  g = torch.randn(4, 64, 256, device='cuda', dtype=torch.bfloat16)
  layers._linear(x, w, b).backward(g)
  x2 = x.detach().clone().requires_grad_(True)
  w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
  torch.nn.functional.linear(x2, w2.to(torch.bfloat16), b2.to(torch.bfloat16)).backward(g)
  assert float((x.grad.float() - x2.grad.float()).abs().max()) < 1e-2
  assert float((w.grad - w2.grad).abs().max()) / float(w2.grad.abs().max()) < 1e-2     # autograd's dW is bf16-rounded
  assert float((b.grad - b2.grad).abs().max()) / float(b2.grad.abs().max()) < 1e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,C', [(128, 30522), (392, 512), (7, 64), (3, 1001)])
def test_softmax_cross_entropy_matches_oracle(rows, C, dtype):
  """mmt_xent_fwd / mmt_xent_bwd vs the fp64 numpy restatement on the stored (rounded) logits.
  fp32: loss 1e-5, gradient 1e-6 absolute (softmax values <= 1); bf16 gradient: one bf16 ulp."""
  from mmt_amd import fused
  rng = np.random.default_rng(rows + C)
  logits = torch.from_numpy((rng.standard_normal((rows, C)) * 3).astype(np.float32)).cuda().to(dtype).requires_grad_(True)
  labels = rng.integers(0, C, size=rows)
  labels[0] = -1                                            # no target
  if rows > 2:
    labels[2] = C + 5
  lab_t = torch.from_numpy(labels).cuda()
  coef = rng.standard_normal(rows).astype(np.float32)
  loss = fused.softmax_cross_entropy(logits, lab_t)
  (loss * torch.from_numpy(coef).cuda()).sum().backward()
  want_loss, want_d = lo.softmax_xent(logits.detach().float().cpu().numpy(), labels)
  assert np.abs(loss.detach().cpu().numpy() - want_loss).max() < (1e-5 if dtype == torch.float32 else 2e-5) * max(1.0, np.abs(want_loss).max())
  got_d = logits.grad.float().cpu().numpy()
  want_d = want_d * coef[:, None]
  tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8 * max(1e-3, np.abs(want_d).max())
  assert np.abs(got_d - want_d).max() <= tol + 1e-7
  assert not got_d[0].any() and loss[0].item() == 0.0       # no-target row


def test_weighted_loss_uses_fused_xent_and_matches_torch():
  from mmt_amd import layers
  torch.manual_seed(1)
  logits = (torch.randn(4, 32, 1000, device='cuda') * 2).to(torch.bfloat16).requires_grad_(True)
  labels = torch.randint(0, 1000, (4, 32), device='cuda')
  w = (torch.rand(4, 32, device='cuda') > 0.3).float()
  loss = layers.weighted_sparse_categorical_crossentropy_loss(logits, labels, w)
  loss.backward()
  ref_logits = logits.detach().float().requires_grad_(True)
  un = torch.nn.functional.cross_entropy(ref_logits.reshape(-1, 1000), labels.reshape(-1), reduction='none').view(4, 32)
  ref = (un * w).sum() / w.sum()
  ref.backward()
  assert abs(float(loss) - float(ref)) < 1e-4
  assert float((logits.grad.float() - ref_logits.grad).abs().max()) < 2.0 ** -8 * float(ref_logits.grad.abs().max()) + 1e-7
  zero = layers.weighted_sparse_categorical_crossentropy_loss(logits, labels, torch.zeros_like(w))
  assert float(zero) == 0.0                                  # divide_no_nan


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('B,n,C,use_mask,use_lmul', [(4, 256, 30522, True, False), (4, 98, 512, True, True),
                                                      (4, 1, 2, False, False), (3, 5, 7, False, True)])
def test_weighted_loss_one_launch_matches_oracle(B, n, C, use_mask, use_lmul, dtype):
  """mmt_xent_fwd + mmt_weighted_loss + mmt_xent_bwd_scaled (the whole loss term of pretraining.py:95-140: row
  losses, weights masked by the example's ITM label, divide_no_nan, and the gradient under an upstream factor)
  against the fp64 numpy restatement: loss 2e-5 relative, gradient 1e-6 (fp32) / one bf16 ulp."""
  from mmt_amd import layers
  rng = np.random.default_rng(B * 1000 + n + C)
  logits = torch.from_numpy((rng.standard_normal((B, n, C)) * 3).astype(np.float32)).cuda().to(dtype).requires_grad_(True)
  labels = rng.integers(0, C, size=(B, n))
  w = (rng.random((B, n)) > 0.4).astype(np.float32) * rng.random((B, n)).astype(np.float32)
  mask = (np.arange(B) % 2 == 0).astype(np.float32) if use_mask else None
  lmul = (rng.random((B, n)) + 0.5).astype(np.float32) if use_lmul else None
  dev = lambda x: None if x is None else torch.from_numpy(x).cuda()
  loss = layers.weighted_sparse_categorical_crossentropy_loss(logits, dev(labels), dev(w), pos_weights=dev(lmul),
                                                              example_mask=dev(mask))
  (loss * 0.25).backward()                                  # an upstream factor, as loss / num_small_steps
  rows, d_unit = lo.softmax_xent(logits.detach().float().cpu().numpy().reshape(B * n, C), labels.reshape(-1))
  want, coef = lo.weighted_loss(rows, w, lmul, mask, n)
  assert abs(float(loss) - want) <= 2e-5 * max(1.0, abs(want))
  want_d = d_unit * (0.25 * coef)[:, None]
  got_d = logits.grad.float().cpu().numpy().reshape(B * n, C)
  tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8 * max(1e-6, np.abs(want_d).max())
  assert np.abs(got_d - want_d).max() <= tol + 1e-9
  zero = layers.weighted_sparse_categorical_crossentropy_loss(logits, dev(labels), torch.zeros(B, n, device='cuda'))
  assert float(zero) == 0.0                                  # divide_no_nan


def test_wgrad_cu_budget_changes_the_split_not_the_result():
  """mmt_wgrad_set_cu_budget (data-parallel runs leave CUs to the collectives): same dW, other grid."""
  from mmt_amd import _lib, fused
  L = _lib.lib()
  torch.manual_seed(0)
  K, M, N = 8192, 768, 768        # split-K not capped by the slice length: 28 / 24 / 7 slabs
  dy = torch.randn(K, M, device='cuda').to(torch.bfloat16)
  x = torch.randn(K, N, device='cuda').to(torch.bfloat16)
  want = dy.double().t() @ x.double()
  sizes = []
  try:
    for cus in (256, 224, 64):
      L.mmt_wgrad_set_cu_budget(cus)
      sizes.append(L.mmt_wgrad_workspace_bytes(M, N, K))
      dw = torch.zeros(M, N, device='cuda')
      assert fused.wgrad_accumulate_(dw, dy, x)
      assert float((dw.double() - want).abs().max()) / float(want.abs().max()) < 2e-5
  finally:
    L.mmt_wgrad_set_cu_budget(256)
  assert sizes[0] > sizes[1] > sizes[2]          # fewer split-K slabs with fewer compute units


# ---- K11: feed-forward GEMMs with the GELU in the epilogue (mmt_ffn_gelu_gemm / mmt_ffn_dgelu_gemm) ---------
# bf16 operands, fp32 accumulate: compared with the numpy oracle on the bf16-rounded inputs; tolerance = one
# bf16 rounding of the output (2^-8 relative) plus 2e-3 for accumulation order / the hardware exp and rcp.
@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (512, 768, 192), (1024, 256, 768)])
@pytest.mark.parametrize('with_bias', [True, False])
def test_ffn_gelu_gemm_matches_oracle(M, N, K, with_bias):
  from mmt_amd import fused
  rng = np.random.default_rng(M + N + K)
  x = bf16_round(rng.standard_normal((M, K)).astype(np.float32))
  w = bf16_round((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32) * 2)
  b = rng.standard_normal(N).astype(np.float32) if with_bias else None
  out = fused.ffn_gelu_gemm(_dev(x, torch.bfloat16), _dev(w, torch.bfloat16), None if b is None else _dev(b))
  assert out is not None
  u, g = (t.float().cpu().numpy() for t in out)
  u_ref = x.astype(np.float64) @ w.astype(np.float64).T + (0 if b is None else b)
  close = lambda got, ref: bool((np.abs(got - ref) <= 2.0 ** -8 * np.abs(ref) + 2e-3).all())     # one bf16 rounding
  assert close(u, u_ref)
  assert close(g, lo.gelu_tanh(u.astype(np.float64)))          # gelu of the STORED (rounded) u
  assert fused.ffn_gelu_gemm(_dev(x[:100], torch.bfloat16), _dev(w, torch.bfloat16), None) is None   # M % 256 != 0


@pytest.mark.parametrize('M,N,K', [(256, 256, 64), (512, 768, 192), (768, 1024, 256)])
@pytest.mark.parametrize('with_bias', [True, False])
def test_ffn_dgelu_gemm_matches_oracle(M, N, K, with_bias):
  from mmt_amd import fused
  rng = np.random.default_rng(M + 3 * N + K)
  dy = bf16_round(rng.standard_normal((M, K)).astype(np.float32))
  w = bf16_round((rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32) * 2)
  u = bf16_round(rng.standard_normal((M, N)).astype(np.float32) * 2)
  b = rng.standard_normal(N).astype(np.float32) if with_bias else None
  du = fused.ffn_dgelu_gemm(_dev(dy, torch.bfloat16), _dev(w, torch.bfloat16), _dev(u, torch.bfloat16),
                            None if b is None else _dev(b))
  assert du is not None
  ref = (dy.astype(np.float64) @ w.astype(np.float64)) * lo.gelu_tanh_grad(u.astype(np.float64) + (0 if b is None else b))
  assert (np.abs(du.float().cpu().numpy() - ref) <= 2.0 ** -8 * np.abs(ref) + 2e-3).all()       # one bf16 rounding
  # a strided (column-sliced) dy is accepted through its row stride
  wide = torch.zeros(M, K + 64, device='cuda', dtype=torch.bfloat16)
  wide[:, :K] = _dev(dy, torch.bfloat16)
  du2 = fused.ffn_dgelu_gemm(wide[:, :K], _dev(w, torch.bfloat16), _dev(u, torch.bfloat16), None if b is None else _dev(b))
  assert torch.equal(du2, du)


def test_ffn_fn_matches_unfused_pair():
  """`_FfnFn` (GELU' inside the backward GEMM's epilogue) against the library-GEMM + bias_gelu chain it replaces."""
  from mmt_amd import fused, layers
  torch.manual_seed(1)
  H, Fd = 128, 512
  mk = lambda *s: torch.nn.Parameter(torch.randn(*s, device='cuda') * 0.08)
  w1, b1, w2 = mk(Fd, H), mk(Fd), mk(H, Fd)
  x = torch.randn(2, 128, H, device='cuda', dtype=torch.bfloat16, requires_grad=True)
  go = torch.randn(2, 128, H, device='cuda', dtype=torch.bfloat16)
  assert layers._ffn_ok(x, w1, b1, w2)
  f = layers._FfnFn.apply(x, w1, b1, w2)
  f.backward(go)
  got = [f.detach().float(), x.grad.float(), w1.grad.clone(), b1.grad.clone(), w2.grad.clone()]
  x.grad = None
  for t in (w1, b1, w2):
    t.grad = None
  f2 = layers._linear(fused.bias_gelu(layers._linear(x, w1, None), b1), w2, None)
  f2.backward(go)
  want = [f2.detach().float(), x.grad.float(), w1.grad, b1.grad, w2.grad]
  for a, b, name in zip(got, want, ('f', 'dx', 'dw1', 'db1', 'dw2')):
    assert float((a - b).abs().max()) <= 2e-2 * max(1.0, float(b.abs().max())), name


def test_ffn_gemm_cu_budget_changes_the_schedule_not_the_result():
  """The persistent kernels size their grid for the CU budget; every budget must give the same bits."""
  from mmt_amd import fused, _lib
  torch.manual_seed(3)
  x = torch.randn(2048, 192, device='cuda').bfloat16()
  w1 = (torch.randn(768, 192, device='cuda') * 0.1).bfloat16()
  w2 = (torch.randn(192, 768, device='cuda') * 0.1).bfloat16()
  b1 = torch.randn(768, device='cuda')
  try:
    outs = []
    for cus in (256, 40, 32):                        # 24 tiles: one per workgroup whatever the budget
      _lib.lib().mmt_ffn_set_cu_budget(cus)
      u, g = fused.ffn_gelu_gemm(x, w1, b1)
      du = fused.ffn_dgelu_gemm(x, w2, u)
      outs.append((u, g, du))
    big = torch.randn(8192, 192, device='cuda').bfloat16()        # 96 tiles on 32 CUs: 3 per workgroup
    _lib.lib().mmt_ffn_set_cu_budget(32)
    a = fused.ffn_gelu_gemm(big, w1, b1)
    _lib.lib().mmt_ffn_set_cu_budget(256)
    b = fused.ffn_gelu_gemm(big, w1, b1)
  finally:
    _lib.lib().mmt_ffn_set_cu_budget(256)
  for o in outs[1:]:
    assert all(torch.equal(p, q) for p, q in zip(o, outs[0]))
  assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize('blocks', [1, 2, 3, 5, 7], ids=['one-block-split-k', 'two-blocks-unsplit', 'three-blocks-round+tail-of-68x3', 'five-blocks-2-rounds+tail-of-28x8', 'seven-blocks-2-rounds+unsplit-tail'])
def test_grouped_weight_gradients_match_fp32_reference(blocks):
  """mmt_wgrad_grouped: the four weight gradients of one / two encoder blocks (BERT-base feature sizes) in one
  launch, half of them with the bias gradient, accumulated into non-zero fp32 buffers -- against fp32 matmuls, and
  bit for bit against a second run (fixed-order split-K sums; with two blocks K is not split and every tile is
  added by its one workgroup with plain stores; from three blocks on there are more tiles than compute units: whole
  rounds of unsplit tiles plus a tail that is split into compact per-tile slabs -- or not at all when it nearly fills
  the chip)."""
  from mmt_amd import _lib, fused
  L = _lib.lib()
  torch.manual_seed(0)
  K = 2048
  shapes = [(768, 3072, False), (3072, 768, True), (768, 768, False), (2304, 768, True)] * blocks   # (M, N, bias)
  n = len(shapes)
  outs = []
  for rep in range(2):
    torch.manual_seed(1)
    probs, keep, refs = (_lib.WgradProblem * n)(), [], []
    for q, (M, N, with_b) in zip(probs, shapes):
      dy = torch.randn(K, M, device='cuda', dtype=torch.bfloat16)
      x = torch.randn(K, N, device='cuda', dtype=torch.bfloat16)
      dw = torch.randn(M, N, device='cuda')
      db = torch.randn(M, device='cuda') if with_b else None
      refs.append((dw + dy.float().t() @ x.float(), None if db is None else db + dy.float().sum(0)))
      q.dw, q.ldw, q.dbias = dw.data_ptr(), N, (None if db is None else db.data_ptr())
      q.dy, q.ldy, q.x, q.ldx, q.M, q.N = dy.data_ptr(), M, x.data_ptr(), N, M, N
      keep.append((dy, x, dw, db))
    need = L.mmt_wgrad_group_workspace_bytes(n, probs, K)
    assert (need > 0) == (blocks in (1, 3, 5))    # two blocks: 216 tiles fill the chip without splitting K; seven: the tail of 244 neither
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device='cuda')
    _lib.check(L.mmt_wgrad_grouped(n, probs, K, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for (dy, x, dw, db), (rw, rb) in zip(keep, refs):
      assert float((dw - rw).abs().max()) / float(rw.abs().max()) < 1e-5
      if db is not None:
        assert float((db - rb).abs().max()) / float(rb.abs().max()) < 1e-5
    outs.append([t[2].clone() for t in keep] + [t[3].clone() for t in keep if t[3] is not None])
  for a, b in zip(*outs):
    assert torch.equal(a, b)
  # refusals: a shape the 256 x 256 kernel cannot tile, a missing workspace
  bad = (_lib.WgradProblem * 1)()
  bad[0].M, bad[0].N = 128, 256
  assert L.mmt_wgrad_group_workspace_bytes(1, bad, K) == 0
  if blocks == 1:
    assert L.mmt_wgrad_grouped(n, probs, K, None, 0, None) == -3          # MMT_E_WORKSPACE


@pytest.mark.parametrize('pending', [1.0, 0.25])
def test_grad_clip_scale_matches_global_norm(pending):
  """mmt_grad_clip_scale over several slabs: norm = pending * ||all slabs||, scale = min(1, max / (norm + 1e-6)) *
  pending, against float64 torch; clipped and unclipped cases; bitwise repeatable."""
  import ctypes
  from mmt_amd import _lib
  L = _lib.lib()
  torch.manual_seed(3)
  slabs = [torch.randn(n, device='cuda') * s for n, s in ((1024 * 37, 1.0), (1024 * 5, 3.0), (4096, 0.1))]
  ptrs = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in slabs])
  sizes = (ctypes.c_int64 * 3)(*[t.numel() for t in slabs])
  ws = torch.empty(2048, device='cuda')
  out = torch.zeros(2, device='cuda')
  true = pending * float(torch.sqrt(sum((t.double() ** 2).sum() for t in slabs)))
  for max_norm in (1.0, 1e6):
    vals = []
    for _ in range(2):
      _lib.check(L.mmt_grad_clip_scale(3, ptrs, sizes, max_norm, pending, out[0:1].data_ptr(), out[1:2].data_ptr(),
                                       ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream))
      torch.cuda.synchronize()
      vals.append(out.clone())
    assert torch.equal(vals[0], vals[1])
    scale, norm = float(out[0]), float(out[1])
    assert abs(norm - true) / true < 1e-5
    assert abs(scale - min(1.0, max_norm / (true + 1e-6)) * pending) / pending < 1e-5
  assert L.mmt_grad_clip_scale(3, ptrs, sizes, 1.0, 1.0, out.data_ptr(), None, None, 0, None) == -3


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['f32', 'bf16'])
@pytest.mark.parametrize('rows,C,pad', [(1024, 30522, 0), (392, 512, 0), (7, 33, 5), (1, 2, 0), (100, 1, 3)])
def test_colsum_matches_fp64_sum(rows, C, pad, dtype):
  """mmt_colsum (bias gradients dy.sum(0), any row stride / alignment) against an fp64 sum of the stored values; the
  accumulate form adds to an existing fp32 buffer; two runs are bit-identical (fixed-order sums)."""
  from mmt_amd import fused
  g = torch.Generator(device='cuda').manual_seed(rows * 7 + C)
  buf = torch.randn(rows, C + pad, device='cuda', generator=g).to(dtype)
  x = buf[:, :C]
  want = x.double().sum(0)
  got = fused.colsum(x)
  tol = 1e-5 * (1 + float(x.double().abs().sum(0).max()))
  assert float((got.double() - want).abs().max()) < tol
  acc = torch.full((C,), 3.0, device='cuda')
  fused.colsum(x, out=acc)
  assert float((acc.double() - 3.0 - want).abs().max()) < tol
  assert torch.equal(got, fused.colsum(x))


def test_deferred_column_sum_reduces_equal_the_immediate_ones(monkeypatch):
  """Parameter gradients that go straight into fp32 masters with nobody waiting for them: the column-sum reduces of a
  backward pass are queued and launched once at its end (`mmt_colsum_reduce_batch`) -- same partials, same fixed-order
  sums as the per-call launches, so the same bits; more items than one batch holds are split."""
  from mmt_amd import fused
  torch.manual_seed(5)
  rows, H = 512, 256

  def run(defer, n_blocks):
    torch.manual_seed(6)
    monkeypatch.setattr(fused, '_defer_colsum_ok', (lambda direct, *p: direct) if defer else (lambda direct, *p: False))
    params = []
    x = torch.randn(2, rows // 2, H, device='cuda', dtype=torch.bfloat16, requires_grad=True)
    h = x
    for _ in range(n_blocks):
      bias = torch.nn.Parameter(torch.randn(H, device='cuda') * 0.1)
      gamma = torch.nn.Parameter(torch.rand(H, device='cuda') + 0.5)
      beta = torch.nn.Parameter(torch.randn(H, device='cuda') * 0.1)
      b2 = torch.nn.Parameter(torch.randn(H, device='cuda') * 0.1)
      for prm in (bias, gamma, beta, b2):
        prm.grad = torch.full_like(prm, 0.25)                   # fp32 masters with something in them already
      params += [bias, gamma, beta, b2]
      xn, hh = fused.residual_block(h * 0.5, bias, h, gamma, beta, 1e-12, 0.1, 77)
      h = fused.bias_gelu(hh, b2) + xn
    h.float().square().mean().backward()
    torch.cuda.synchronize()
    assert not fused._cs_deferred
    return [prm.grad.clone() for prm in params] + [x.grad.clone()]

  for n_blocks in (3, 30):                                      # 30 blocks: 60 reduces, more than one batch of 48
    a, b = run(False, n_blocks), run(True, n_blocks)
    for u, v in zip(a, b):
      assert torch.equal(u, v)
