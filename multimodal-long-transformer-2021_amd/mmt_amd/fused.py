"""Fused residual-block ops (C ABI include/mmt_layer.h) as autograd functions.

They replace the BiasAdd / Dropout / Add / LayerNormalization / Gelu chain of etcmodel's
`ResidualBlock` + `DenseLayers` in the pre-activation order the reference config uses
(`src/configs/encoders.py:95`; math SURVEY.md App. A.3).  Parameters stay fp32 (master weights)
and their gradients come back fp32 straight from the kernels.
"""
from __future__ import annotations


from typing import Optional

import torch

from . import _lib, step_scalars

# Dropout seeds are a pure function of (train step, micro step, data-parallel rank, index of the
# dropout site within the forward pass): replicas draw different masks for their shards and a run
# resumed from a checkpoint continues with the masks the uninterrupted run would have drawn.  Code that
# never calls `set_seed_stream` (plain `model(...)` calls) just keeps counting from where it is.
_seed_state = {'base': 0, 'n': 0}


def set_seed_stream(step: int, micro_step: int = 0, rank: int = 0) -> None:
  """Called by the tasks' train_step before every micro-step's forward pass.  The step's share of the seeds (its
  "epoch") is added where the descriptors are filled in -- or by the kernels themselves when it lives in device
  memory (`step_scalars`) -- so the seeds handed around on the host do not depend on the step."""
  _seed_state['base'] = ((int(micro_step) * 0xC2B2AE3D27D4EB4F) ^ ((int(rank) + 1) * 0x165667B19E3779F9)) & ((1 << 63) - 1)
  _seed_state['n'] = 0
  step_scalars.set_step(step)


def next_seed(base: int = 0) -> int:
  # 63 bits: the seed travels through autograd.Function.apply as a Python int, which torch's
  # shape-recording profiler converts to int64
  _seed_state['n'] += 1
  return (_seed_state['base'] + int(base) * 0x9E3779B97F4A7C15
          + _seed_state['n'] * 0xD1B54A32D192ED03) & ((1 << 63) - 1)


def _desc(x2d: torch.Tensor, eps=1e-12, p=0.0, seed=0) -> _lib.RowsDesc:
  d = _lib.RowsDesc()
  d.rows, d.H = x2d.shape[0], x2d.shape[1]
  if x2d.dtype == torch.float32:
    d.dtype = _lib.MMT_F32
  elif x2d.dtype == torch.bfloat16:
    d.dtype = _lib.MMT_BF16
  else:
    raise TypeError(f'fused layer ops support float32 and bfloat16, got {x2d.dtype}')
  d.eps, d.dropout_p, d.dropout_seed = float(eps), float(p), (int(seed) + step_scalars.host_epoch(x2d.device)) & ((1 << 64) - 1)
  d.dropout_epoch = step_scalars.epoch_ptr(x2d.device) if p else None
  return d


def _stream(t):
  return torch.cuda.current_stream(t.device).cuda_stream


def _ws(desc, like):
  n = _lib.lib().mmt_layer_workspace_bytes(desc)
  return torch.empty((max(n, 16),), dtype=torch.uint8, device=like.device)


def _check(*ts):
  for t in ts:
    if t is not None and not t.is_cuda:
      raise RuntimeError('fused layer ops run on the GPU only (no CPU fallback)')


def _p(t):
  return None if t is None else t.data_ptr()


def _f32(t):
  return t if t.dtype == torch.float32 and t.is_contiguous() else t.float().contiguous()


def _grad_target(param, like):
  """Where a parameter gradient goes: straight into an existing fp32 `.grad` (accumulate, no
  AccumulateGrad add kernel) when possible, else a fresh buffer handed back to autograd."""
  g = getattr(param, 'grad', None)
  direct = (isinstance(param, torch.nn.Parameter) and g is not None and g.dtype == torch.float32
            and g.is_contiguous() and g.shape == like.shape and g.data_ptr() % 16 == 0)
  return (g, True) if direct else (torch.empty_like(like), False)


def _finish(param, buf, direct):
  """Return value for autograd + gradient-ready notification for directly written grads."""
  if not direct:
    return buf
  for hook in getattr(param, '_mmt_grad_ready_hooks', ()):
    hook(param)
  return None


# ---- column-sum reduces of a backward pass in one launch -------------------------------------------------------------
# Every *_bwd kernel below leaves per-workgroup partial sums of its parameter gradients (dbias / dgamma / dbeta) in its
# workspace and a 5-us reduce launch folds them -- ~32 of those per step.  When the gradients go straight into fp32
# masters and nothing waits for them before the backward pass ends (no parameter carries a reducer's hooks), the
# reduces are queued (desc.defer_reduce; the workspaces stay alive in the queue) and launched ONCE by an engine callback
# at the end of backward (`mmt_colsum_reduce_batch`).
_CS_MAX = 48
_cs_deferred = {}          # device -> (graph-task id, [(workspace, rows, H, kind, outs)])


def _defer_colsum_ok(direct: bool, *params) -> bool:
  return (direct and grouping_is_safe()
          and not any(getattr(p, '_mmt_grad_ready_hooks', ()) for p in params if p is not None))


def _flush_colsum(device) -> None:
  entry = _cs_deferred.pop(device, None)
  if not entry or not entry[1]:
    return
  items = entry[1]
  arr = (_lib.ColsumItem * len(items))()
  for q, (ws, rows, H, kind, outs) in zip(arr, items):
    o = list(outs) + [None] * (3 - len(outs))
    q.workspace, q.o0, q.o1, q.o2 = ws.data_ptr(), _p(o[0]), _p(o[1]), _p(o[2])
    q.rows, q.H, q.kind, q.accumulate = rows, H, kind, 1
  with torch.cuda.device(device):
    _lib.check(_lib.lib().mmt_colsum_reduce_batch(len(items), arr, torch.cuda.current_stream(device).cuda_stream))


def _flush_all_colsum():
  for device in list(_cs_deferred):
    _flush_colsum(device)


def _defer_colsum(ws, rows: int, H: int, kind: int, outs) -> None:
  device = ws.device
  gid = _graph_task_id()
  entry = _cs_deferred.get(device)
  if entry is not None and entry[0] != gid:              # left behind by a backward pass that raised: not ours to launch
    _cs_deferred.pop(device)
    entry = None
  if entry is None:
    entry = _cs_deferred[device] = (gid, [])
    torch.autograd.Variable._execution_engine.queue_callback(_flush_all_colsum)    # end of this backward pass
  entry[1].append((ws, rows, H, kind, tuple(outs)))
  if len(entry[1]) >= _CS_MAX:
    _flush_colsum(device)          # (a later item opens a new entry and queues another callback: harmless)


class _LayerNormFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, x, gamma, beta, eps):
    _check(x, gamma, beta)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).contiguous()
    gamma_p, beta_p = gamma, beta
    gamma, beta = _f32(gamma), _f32(beta)
    y = torch.empty_like(x2)
    mean = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    d = _desc(x2, eps)
    with torch.cuda.device(x.device):
      _lib.check(_lib.lib().mmt_ln_fwd(d, _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _stream(x)))
    ctx.save_for_backward(x2, gamma, mean, rstd)
    ctx.eps, ctx.shape = eps, shape
    ctx.params = (gamma_p, beta_p)
    return y.view(shape)

  @staticmethod
  def backward(ctx, dy):
    x2, gamma, mean, rstd = ctx.saved_tensors
    gamma_p, beta_p = ctx.params
    dy2 = dy.reshape(x2.shape).contiguous()
    dx = torch.empty_like(x2)
    (dg, dg_direct), (db, db_direct) = _grad_target(gamma_p, gamma), _grad_target(beta_p, gamma)
    direct = dg_direct and db_direct
    if not direct:
      dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
    d = _desc(x2, ctx.eps)
    d.accumulate = int(direct)
    d.defer_reduce = int(_defer_colsum_ok(direct, gamma_p, beta_p))
    ws = _ws(d, x2)
    with torch.cuda.device(x2.device):
      _lib.check(_lib.lib().mmt_ln_bwd(d, _p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(dx), _p(dg),
                                       _p(db), _p(ws), ws.numel(), _stream(x2)))
    if d.defer_reduce:
      _defer_colsum(ws, x2.shape[0], x2.shape[1], 0, (dg, db))
    return dx.view(ctx.shape), _finish(gamma_p, dg, direct), _finish(beta_p, db, direct), None


def layer_norm(x, gamma, beta, eps=1e-12):
  """LayerNorm over the last dim (gamma/beta fp32)."""
  return _LayerNormFn.apply(x, gamma, beta, eps)


class _LayerNormKeepFn(torch.autograd.Function):
  """(x, LayerNorm(x)): the input is handed on as an alias so that a consumer of BOTH (the first pre-activation
  block: residual sum and LayerNorm of the embedding output) sends its two gradients back through ONE node --
  dx = dx_alias + LayerNormBackward(dh) in one kernel (`mmt_ln_bwd_add`) instead of autograd's add over [B*S, H]."""

  @staticmethod
  def forward(ctx, x, gamma, beta, eps):
    _check(x, gamma, beta)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).contiguous()
    gamma_p, beta_p = gamma, beta
    gamma, beta = _f32(gamma), _f32(beta)
    y = torch.empty_like(x2)
    mean = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    d = _desc(x2, eps)
    with torch.cuda.device(x.device):
      _lib.check(_lib.lib().mmt_ln_fwd(d, _p(x2), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _stream(x)))
    ctx.save_for_backward(x2, gamma, mean, rstd)
    ctx.eps, ctx.shape = eps, shape
    ctx.params = (gamma_p, beta_p)
    return x.view_as(x), y.view(shape)

  @staticmethod
  def backward(ctx, dx_alias, dy):
    x2, gamma, mean, rstd = ctx.saved_tensors
    gamma_p, beta_p = ctx.params
    if dy is None:
      return dx_alias, None, None, None
    dy2 = dy.reshape(x2.shape).contiguous()
    din = None if dx_alias is None else dx_alias.reshape(x2.shape).contiguous()
    dx = torch.empty_like(x2)
    (dg, dg_direct), (db, db_direct) = _grad_target(gamma_p, gamma), _grad_target(beta_p, gamma)
    direct = dg_direct and db_direct
    if not direct:
      dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
    d = _desc(x2, ctx.eps)
    d.accumulate = int(direct)
    d.defer_reduce = int(_defer_colsum_ok(direct, gamma_p, beta_p))
    ws = _ws(d, x2)
    with torch.cuda.device(x2.device):
      _lib.check(_lib.lib().mmt_ln_bwd_add(d, _p(dy2), _p(x2), _p(gamma), _p(mean), _p(rstd), _p(din), _p(dx), _p(dg),
                                           _p(db), _p(ws), ws.numel(), _stream(x2)))
    if d.defer_reduce:
      _defer_colsum(ws, x2.shape[0], x2.shape[1], 0, (dg, db))
    return dx.view(ctx.shape), _finish(gamma_p, dg, direct), _finish(beta_p, db, direct), None


def layer_norm_keep(x, gamma, beta, eps=1e-12):
  """(x_alias, LayerNorm(x)) -- use x_alias for every other consumer of x (see `_LayerNormKeepFn`)."""
  return _LayerNormKeepFn.apply(x, gamma, beta, eps)


class _ResidualBlockFn(torch.autograd.Function):
  """(x_new, h) = (x + Dropout(o + bias), LayerNorm(x_new)); h is None without gamma."""

  @staticmethod
  def forward(ctx, o, bias, x, gamma, beta, eps, p, seed):
    _check(o, bias, x, gamma, beta)
    shape = x.shape
    o2, x2 = o.reshape(-1, shape[-1]).contiguous(), x.reshape(-1, shape[-1]).contiguous()
    ctx.params = (bias, gamma, beta)
    bias = _f32(bias)
    has_ln = gamma is not None
    if has_ln:
      gamma, beta = _f32(gamma), _f32(beta)
    x_new = torch.empty_like(x2)
    h = torch.empty_like(x2) if has_ln else None
    mean = torch.empty(x2.shape[0], dtype=torch.float32, device=x.device) if has_ln else None
    rstd = torch.empty_like(mean) if has_ln else None
    d = _desc(x2, eps, p, seed)
    with torch.cuda.device(x.device):
      _lib.check(_lib.lib().mmt_residual_block_fwd(d, _p(o2), _p(bias), _p(x2), _p(gamma), _p(beta),
                                                   _p(x_new), _p(h), _p(mean), _p(rstd), _stream(x)))
    ctx.save_for_backward(x_new, gamma, mean, rstd, bias)
    ctx.cfg = (eps, p, seed, shape, has_ln)
    if has_ln:
      return x_new.view(shape), h.view(shape)
    return x_new.view(shape), None

  @staticmethod
  def backward(ctx, dx_new, dh):
    x_new, gamma, mean, rstd, bias = ctx.saved_tensors
    eps, p, seed, shape, has_ln = ctx.cfg
    dxn = None if dx_new is None else dx_new.reshape(x_new.shape).contiguous()
    dh2 = None if (dh is None or not has_ln) else dh.reshape(x_new.shape).contiguous()
    if has_ln and dh2 is None:
      dh2 = torch.zeros_like(x_new)
    d_o, dx = torch.empty_like(x_new), torch.empty_like(x_new)
    bias_p, gamma_p, beta_p = ctx.params
    dbias, direct = _grad_target(bias_p, bias)
    dg = db = None
    if has_ln:
      (dg, d1), (db, d2) = _grad_target(gamma_p, gamma), _grad_target(beta_p, gamma)
      direct = direct and d1 and d2
    if not direct:
      dbias = torch.empty_like(bias)
      dg = torch.empty_like(gamma) if has_ln else None
      db = torch.empty_like(gamma) if has_ln else None
    d = _desc(x_new, eps, p, seed)
    d.accumulate = int(direct)
    d.defer_reduce = int(_defer_colsum_ok(direct, bias_p, gamma_p, beta_p))
    ws = _ws(d, x_new)
    with torch.cuda.device(x_new.device):
      _lib.check(_lib.lib().mmt_residual_block_bwd(
          d, _p(dxn), _p(dh2), _p(x_new), _p(gamma), _p(mean), _p(rstd), _p(d_o), _p(dx), _p(dbias),
          _p(dg), _p(db), _p(ws), ws.numel(), _stream(x_new)))
    if d.defer_reduce:
      _defer_colsum(ws, x_new.shape[0], x_new.shape[1], 1 if has_ln else 2, (dbias, dg, db) if has_ln else (dbias,))
    return (d_o.view(shape), _finish(bias_p, dbias, direct), dx.view(shape),
            _finish(gamma_p, dg, direct) if has_ln else None,
            _finish(beta_p, db, direct) if has_ln else None, None, None, None)


def residual_block(o, bias, x, gamma=None, beta=None, eps=1e-12, p=0.0, seed=0):
  """x_new = x + Dropout(o + bias); h = LayerNorm(x_new) (None if gamma is None)."""
  return _ResidualBlockFn.apply(o, bias, x, gamma, beta, eps, p, seed)


class _BiasGeluFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, u, bias):
    _check(u, bias)
    shape = u.shape
    u2 = u.reshape(-1, shape[-1]).contiguous()
    ctx.param = bias
    bias = _f32(bias)
    y = torch.empty_like(u2)
    d = _desc(u2)
    with torch.cuda.device(u.device):
      _lib.check(_lib.lib().mmt_bias_gelu_fwd(d, _p(u2), _p(bias), _p(y), _stream(u)))
    ctx.save_for_backward(u2, bias)
    ctx.shape = shape
    return y.view(shape)

  @staticmethod
  def backward(ctx, dy):
    u2, bias = ctx.saved_tensors
    dy2 = dy.reshape(u2.shape).contiguous()
    du = torch.empty_like(u2)
    dbias, direct = _grad_target(ctx.param, bias)
    d = _desc(u2)
    d.accumulate = int(direct)
    d.defer_reduce = int(_defer_colsum_ok(direct, ctx.param))
    ws = _ws(d, u2)
    with torch.cuda.device(u2.device):
      _lib.check(_lib.lib().mmt_bias_gelu_bwd(d, _p(dy2), _p(u2), _p(bias), _p(du), _p(dbias), _p(ws),
                                              ws.numel(), _stream(u2)))
    if d.defer_reduce:
      _defer_colsum(ws, u2.shape[0], u2.shape[1], 3, (dbias,))
    return du.view(ctx.shape), _finish(ctx.param, dbias, direct)


def bias_gelu_forward_(u2: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
  """gelu_tanh(u2 + bias) for a contiguous [rows, H] u2, outside autograd (callers own the backward)."""
  _check(u2, bias)
  y = torch.empty_like(u2)
  with torch.cuda.device(u2.device):
    _lib.check(_lib.lib().mmt_bias_gelu_fwd(_desc(u2), _p(u2), _p(_f32(bias)), _p(y), _stream(u2)))
  return y


def bias_gelu(u, bias):
  """gelu_tanh(u + bias)."""
  return _BiasGeluFn.apply(u, bias)



def _ffn_gemm_ok(a: torch.Tensor, w: torch.Tensor, M: int, N: int, K: int) -> bool:
  return (a.is_cuda and a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and M % 256 == 0 and N % 256 == 0
          and K % 64 == 0 and a.stride(1) == 1 and w.stride(1) == 1 and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0
          and a.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)


def ffn_gelu_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]):
  """(u, g) with u = x[M,K] @ w[N,K]^T + bias (bf16), g = gelu_tanh(u): one hand-written GEMM with the
  activation in its epilogue (`mmt_ffn_gelu_gemm`).  Returns None when the shape is outside the kernel's."""
  M, K = x.shape
  N = w.shape[0]
  if not (w.shape[1] == K and _ffn_gemm_ok(x, w, M, N, K)
          and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N))):
    return None
  u = torch.empty(M, N, device=x.device, dtype=torch.bfloat16)
  g = torch.empty_like(u)
  with torch.cuda.device(x.device):
    _lib.check(_lib.lib().mmt_ffn_gelu_gemm(_p(x), x.stride(0), _p(w), w.stride(0), _p(bias) if bias is not None else None,
                                            _p(u), N, _p(g), N, M, N, K, _stream(x)))
  return u, g


def ffn_dgelu_gemm(dy: torch.Tensor, w: torch.Tensor, u: torch.Tensor, bias: Optional[torch.Tensor] = None):
  """du = (dy[M,K] @ w[K,N]) * gelu_tanh'(u[M,N] (+ bias)) in one GEMM (`mmt_ffn_dgelu_gemm`); None when the
  shape is outside the kernel's."""
  M, K = dy.shape
  N = w.shape[1]
  if not (w.shape[0] == K and u.shape == (M, N) and u.dtype == torch.bfloat16 and u.stride(1) == 1
          and u.stride(0) % 8 == 0 and u.data_ptr() % 16 == 0 and _ffn_gemm_ok(dy, w, M, N, K)
          and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == N))):
    return None
  du = torch.empty(M, N, device=dy.device, dtype=torch.bfloat16)
  with torch.cuda.device(dy.device):
    _lib.check(_lib.lib().mmt_ffn_dgelu_gemm(_p(dy), dy.stride(0), _p(w), w.stride(0), _p(u), u.stride(0),
                                             _p(bias) if bias is not None else None, _p(du), N, M, N, K, _stream(dy)))
  return du

def colsum(x2: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
  """fp32 [C] column sums of a 2-D fp32 | bf16 tensor with unit column stride (any row stride / alignment): the bias
  gradient `dy.sum(0)`.  With `out` (fp32, contiguous [C]) the sums are ADDED to it."""
  if not (x2.is_cuda and x2.dim() == 2 and x2.stride(1) == 1 and x2.dtype in (torch.float32, torch.bfloat16)
          and x2.shape[0] > 0 and x2.shape[1] > 0):
    s = x2.sum(0, dtype=torch.float32)
    return s if out is None else out.add_(s)
  rows, C = x2.shape
  L = _lib.lib()
  ws = torch.empty(max(L.mmt_colsum_workspace_bytes(rows, C), 16), dtype=torch.uint8, device=x2.device)
  acc = out is not None
  if acc and (out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != C):
    raise ValueError('colsum: out must be a contiguous fp32 [C] tensor')
  res = out if acc else torch.empty(C, dtype=torch.float32, device=x2.device)
  with torch.cuda.device(x2.device):
    _lib.check(L.mmt_colsum(rows, C, _dtype_code(x2.dtype), _p(x2), x2.stride(0), _p(res), int(acc), _p(ws), ws.numel(),
                            _stream(x2)))
  return res


def accumulate_grad_(acc: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
  """acc (fp32, contiguous) += g (fp32 | bf16), one streaming kernel."""
  if (not acc.is_cuda or acc.dtype != torch.float32 or not acc.is_contiguous() or not g.is_contiguous()
      or g.dtype not in (torch.float32, torch.bfloat16) or acc.numel() != g.numel()
      or acc.data_ptr() % 16 or g.data_ptr() % 16):
    return acc.add_(g)
  code = _lib.MMT_F32 if g.dtype == torch.float32 else _lib.MMT_BF16
  with torch.cuda.device(acc.device):
    _lib.check(_lib.lib().mmt_accumulate_grad(acc.data_ptr(), g.data_ptr(), code, acc.numel(), _stream(acc)))
  return acc


def wgrad_accumulate_(dw: torch.Tensor, dy: torch.Tensor, x: torch.Tensor, dbias: Optional[torch.Tensor] = None) -> bool:
  """dw (fp32 [M,N]) += dy[K,M]^T @ x[K,N] (bf16) with the hand-written split-K kernel; with
  `dbias` (fp32 [M], contiguous) also dbias += dy.sum(0) from the same pass over dy.
  Returns False (nothing done) when the shape/layout is outside what the kernel is built for."""
  K, M = dy.shape
  N = x.shape[1]
  ok = (dw.is_cuda and dw.dtype == torch.float32 and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
        and dw.shape == (M, N) and x.shape[0] == K and M % 128 == 0 and N % 256 == 0 and K > 0
        and dw.stride(1) == 1 and dy.stride(1) == 1 and x.stride(1) == 1
        and dy.stride(0) % 8 == 0 and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
        and dw.stride(0) % 4 == 0 and dw.data_ptr() % 16 == 0
        and (dbias is None or (dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == M
                               and dbias.device == dw.device)))
  if not ok:
    return False
  L = _lib.lib()
  ws = _wgrad_ws(dw.device, L.mmt_wgrad_workspace_bytes(M, N, K))
  with torch.cuda.device(dw.device):
    _lib.check(L.mmt_wgrad_bias_accumulate(dw.data_ptr(), dw.stride(0), None if dbias is None else dbias.data_ptr(),
                                           dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, N, K,
                                           ws.data_ptr(), ws.numel(), _stream(dw)))
  return True


# ---- weight gradients: grouped launches -----------------------------------------------------------------
_GRAPH_TASK_ID = getattr(torch._C, '_current_graph_task_id', None)


def _graph_task_id():
  return _GRAPH_TASK_ID() if _GRAPH_TASK_ID is not None else -1


def grouping_is_safe() -> bool:
  """The queues of grouped weight-gradient products tell a new backward pass from a stale one by autograd's graph-task
  id.  Without that symbol every pass would look alike and products left behind by a backward that raised would be
  launched (with dead operands) by the next one: fail closed -- no grouping, every product launched on its own."""
  return _GRAPH_TASK_ID is not None


def _wgrad_groupable(dw, dy, x, dbias) -> bool:
  K, M = dy.shape
  N = x.shape[1]
  if not grouping_is_safe():
    return False
  return (dw.is_cuda and dw.dtype == torch.float32 and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
          and dw.shape == (M, N) and x.shape[0] == K and M % 256 == 0 and N % 256 == 0 and K % 64 == 0 and K > 0
          and dw.stride(1) == 1 and dy.stride(1) == 1 and x.stride(1) == 1 and dy.stride(0) % 8 == 0
          and x.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and dw.stride(0) % 4 == 0
          and dw.data_ptr() % 16 == 0
          and (dbias is None or (dbias.dtype == torch.float32 and dbias.is_contiguous() and dbias.numel() == M
                                 and dbias.device == dw.device)))


# Weight-gradient products waiting to be launched together (per device): the four Dense layers of an encoder
# block contract the same rows, and ONE grouped launch (`mmt_wgrad_grouped`) needs a 2-way instead of a 6- to
# 24-way split of K -- 4x less fp32 slab traffic (~0.2 GB per block).  Nothing but the optimizer reads dW, so
# waiting for the block's last product costs nothing.  (Until round 3 a second stream carried these products,
# MMT_WGRAD_SIDE_STREAM; with the grouped launch the two measured the same -- 15.93 vs 15.98 ms per step over
# four same-box pairs -- and the form was removed in round 4.)
_WG_GROUP = 4
# Without a gradient exchange to overlap (no parameter carries a reducer's hooks) nothing needs a block's gradients before
# the backward pass ends, and the more products one launch holds the better its tiles fill the chip: 28 products (seven
# blocks, 756 tiles = 2.95 rounds of 256 workgroups with K unsplit) or what is left when backward ends (the other five
# blocks of a 12-layer encoder: 540 tiles = 2 rounds + 28 tiles split 8-way) -- against 216 workgroups on 256 CUs and two
# fp32 slabs per tile for one block alone (csrc/wgrad_gemm.hip: wgrad_dma_big_kernel).
_WG_GROUP_ALONE = 28


def _launch_wgrad_group(items, device, stream) -> None:
  """The queued products on `stream` (the current stream of the caller's `torch.cuda.stream` context)."""
  L = _lib.lib()
  if len(items) == 1:
    dw, dy, x, dbias = items[0][:4]
    if not wgrad_accumulate_(dw, dy, x, dbias):
      raise RuntimeError('mmt_wgrad_accumulate refused a shape _wgrad_groupable admitted')
    return
  K = items[0][1].shape[0]
  arr = (_lib.WgradProblem * len(items))()
  for q, (dw, dy, x, dbias, *_) in zip(arr, items):
    q.dw, q.ldw, q.dbias = dw.data_ptr(), dw.stride(0), (None if dbias is None else dbias.data_ptr())
    q.dy, q.ldy, q.x, q.ldx = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0)
    q.M, q.N = dy.shape[1], x.shape[1]
  ws = _wgrad_ws(device, max(16, L.mmt_wgrad_group_workspace_bytes(len(items), arr, K)))
  with torch.cuda.device(device):
    _lib.check(L.mmt_wgrad_grouped(len(items), arr, K, ws.data_ptr(), ws.numel(), stream.cuda_stream))


# The products are queued and launched together on the CURRENT stream, and only then are the
# parameters reported ready -- the reducer ignores autograd's own post-accumulate notification for a parameter
# while `_mmt_grad_deferred` is set, so a bucket's all-reduce can never be enqueued ahead of its last product.
_wg_deferred = {}


def _flush_wgrad_deferred(device) -> None:
  entry = _wg_deferred.pop(device, None)
  if not entry or not entry[1]:
    return
  items = entry[1]
  _launch_wgrad_group(items, device, torch.cuda.current_stream(device))
  for item in items:
    for prm in item[4]:
      for hook in getattr(prm, '_mmt_grad_ready_hooks', ()):
        hook(prm)


def reset_host_queues() -> None:
  """Forgets every weight-gradient product a backward pass has queued but not launched.  For a step that was ABANDONED half-way (a graph capture that raised inside backward, `graphed.py`): its
  queued products point at activations of a step that never ran; the retry must not launch them."""
  _wg_deferred.clear()
  _cs_deferred.clear()


def _flush_all_deferred():
  for device in list(_wg_deferred):
    _flush_wgrad_deferred(device)


def wgrad_accumulate_deferred_(dw, dy, x, dbias, notify) -> bool:
  """Queues dw += dy^T x (dbias += column sums) for a grouped launch on the current stream; `notify` = the
  parameters whose gradient-ready hooks run once the launch is enqueued.  False: not a shape the grouped kernel
  takes (the caller launches and notifies at once)."""
  device = dw.device
  if _WG_GROUP <= 1 or not _wgrad_groupable(dw, dy, x, dbias):
    return False
  gid = _graph_task_id()
  entry = _wg_deferred.get(device)
  if entry is not None and entry[0] != gid:              # left behind by a backward pass that raised
    for item in _wg_deferred.pop(device)[1]:
      for prm in item[4]:
        prm._mmt_grad_deferred = False
    entry = None
  if entry is not None and entry[1] and entry[1][0][1].shape[0] != dy.shape[0]:
    _flush_wgrad_deferred(device)
    entry = None
  if entry is None:
    entry = _wg_deferred[device] = (gid, [])
    torch.autograd.Variable._execution_engine.queue_callback(_flush_all_deferred)    # end of this backward pass
  for prm in notify:
    prm._mmt_grad_deferred = True
  entry[1].append((dw, dy, x, dbias, tuple(notify)))
  exchanged = any(getattr(prm, '_mmt_grad_ready_hooks', ()) for item in entry[1] for prm in item[4])
  if len(entry[1]) >= (_WG_GROUP if exchanged else _WG_GROUP_ALONE):
    _flush_wgrad_deferred(device)
  return True


_WGRAD_WS = {}


def _wgrad_ws(device, nbytes):
  """One grow-only scratch buffer per (device, stream) for the split-K slabs: a buffer is only ever used on the stream
  it was allocated on, so its reuse -- and the release of the one it replaces when it grows -- is stream-ordered."""
  key = (device, torch.cuda.current_stream(device).cuda_stream)
  buf = _WGRAD_WS.get(key)
  if buf is None or buf.numel() < nbytes:
    buf = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=device)
    _WGRAD_WS[key] = buf
  return buf


# ---- embedding assembly (MmtEncoder.call, mmt_encoder.py:189-218) --------------------------------
_ARANGE = {}


def _arange_i32(n: int, device) -> torch.Tensor:
  """Cached [n] int32 ramp (a constant of the model, not re-generated every step)."""
  key = (n, str(device))
  t = _ARANGE.get(key)
  if t is None:
    t = _ARANGE[key] = torch.arange(n, dtype=torch.int32, device=device)
  return t


def _dtype_code(dt):
  if dt == torch.float32:
    return _lib.MMT_F32
  if dt == torch.bfloat16:
    return _lib.MMT_BF16
  raise TypeError(f'embed_assemble supports float32 and bfloat16 outputs, got {dt}')


def _mm_f32(a, b):
  """a @ b with fp32 output for low-precision operands (fp32 accumulation either way)."""
  if a.dtype == torch.float32:
    return torch.mm(a, b)
  try:
    return torch.mm(a, b, out_dtype=torch.float32)
  except (TypeError, RuntimeError):
    return torch.mm(a.float(), b.float())


class _EmbedAssembleFn(torch.autograd.Function):
  """Dropout(LN(WordEmb[ids])) + SegEmb[seg] (+ PosEmb) (+ projected patches) in one kernel.

  Backward: the word-table gradient (a scatter-add) is added straight into `word_table.grad`
  (fp32, created if missing) by `mmt_embed_bwd` and `None` is returned for it -- the table's own
  AccumulateGrad (tied MaskedLM logits) still runs after this node, so gradient-ready hooks fire
  once; dgamma / dbeta go straight into existing fp32 `.grad`s when possible."""

  @staticmethod
  def forward(ctx, word_ids, seg_ids, word_table, seg_table, pos_table, gamma, beta, patch_proj, cfg):
    eps, p, seed, patch_start, out_dtype = cfg
    _check(word_ids, seg_ids, word_table, seg_table, gamma, beta, patch_proj, pos_table)
    B, S = word_ids.shape
    V, H = word_table.shape
    ids = word_ids.reshape(-1).to(torch.int32).contiguous()
    seg = seg_ids.reshape(-1).to(torch.int32).contiguous()
    wt, st = _f32(word_table.detach()), _f32(seg_table.detach())
    pt = None if pos_table is None else _f32(pos_table.detach())
    if pt is not None and pt.shape[0] < S:
      raise ValueError(f'position table has {pt.shape[0]} rows, sequence length is {S}')
    g32, b32 = _f32(gamma.detach()), _f32(beta.detach())
    n_patch = 0
    if patch_proj is not None:
      if patch_proj.dim() != 3 or patch_proj.shape[0] != B or patch_proj.shape[2] != H or patch_proj.dtype != out_dtype:
        raise ValueError('patch_proj must be [B, n_patch, H] in the output dtype')
      patch_proj = patch_proj.contiguous()
      n_patch = patch_proj.shape[1]
    d = _lib.EmbedDesc()
    d.rows, d.S, d.H, d.dtype = B * S, S, H, _dtype_code(out_dtype)
    d.vocab, d.seg_vocab, d.patch_start, d.n_patch = V, st.shape[0], int(patch_start), n_patch
    d.eps, d.dropout_p, d.dropout_seed = float(eps), float(p), (int(seed) + step_scalars.host_epoch(word_table.device)) & ((1 << 64) - 1)
    d.dropout_epoch = step_scalars.epoch_ptr(word_table.device) if p else None
    out = torch.empty((B, S, H), dtype=out_dtype, device=word_table.device)
    mean = torch.empty(B * S, dtype=torch.float32, device=out.device)
    rstd = torch.empty_like(mean)
    with torch.cuda.device(out.device):
      _lib.check(_lib.lib().mmt_embed_fwd(d, _p(ids), _p(seg), _p(wt), _p(st), _p(pt), _p(g32), _p(b32),
                                          _p(patch_proj), None, _p(out), _p(mean), _p(rstd), _stream(out)))
    ctx.save_for_backward(ids, seg, wt, g32, mean, rstd)
    ctx.desc, ctx.params = d, (word_table, seg_table, pos_table, gamma, beta)
    ctx.has_patch = patch_proj is not None
    return out

  @staticmethod
  def backward(ctx, dout):
    ids, seg, wt, g32, mean, rstd = ctx.saved_tensors
    word_table, seg_table, pos_table, gamma, beta = ctx.params
    d = ctx.desc
    B, S, H = dout.shape
    dout2 = dout.reshape(B * S, H).contiguous()
    srt = torch.sort(ids, stable=True)                                # equal ids adjacent, ties in row order
    sorted_ids, order = srt.values.contiguous(), srt.indices.to(torch.int32)
    if isinstance(word_table, torch.nn.Parameter) and word_table.requires_grad:
      if word_table.grad is None:
        word_table.grad = torch.zeros_like(word_table, dtype=torch.float32)
      dword = word_table.grad
      if dword.dtype != torch.float32 or not dword.is_contiguous():
        raise RuntimeError('embed_assemble: word_table.grad must be a contiguous fp32 tensor')
    else:
      dword = torch.zeros(wt.shape, dtype=torch.float32, device=wt.device)
    (dg, dg_direct), (db, db_direct) = _grad_target(gamma, g32), _grad_target(beta, g32)
    direct = dg_direct and db_direct
    if not direct:
      dg, db = torch.empty_like(g32), torch.empty_like(g32)
    d.accumulate = int(direct)
    need_patch = ctx.has_patch and ctx.needs_input_grad[7]
    dpatch = torch.empty((B, d.n_patch, H), dtype=dout.dtype, device=dout.device) if need_patch else None
    L = _lib.lib()
    n = L.mmt_embed_workspace_bytes(d)
    ws = torch.empty((max(n, 16),), dtype=torch.uint8, device=dout.device)
    with torch.cuda.device(dout.device):
      _lib.check(L.mmt_embed_bwd(d, _p(dout2), _p(sorted_ids), _p(order), _p(wt), _p(g32), _p(mean), _p(rstd), _p(dword),
                                 _p(dg), _p(db), _p(dpatch), _p(ws), ws.numel(), _stream(dout)))
    dword_ret = None
    if dword is not getattr(word_table, 'grad', None) and ctx.needs_input_grad[2]:
      dword_ret = dword.to(word_table.dtype)
    dseg = dpos = None
    if ctx.needs_input_grad[3]:     # segment table: one-hot^T @ dout (the reference's one-hot lookup, transposed)
      # rows with an id outside the table are all-zero, as under the reference's one-hot lookup
      onehot = (seg.unsqueeze(1) == _arange_i32(seg_table.shape[0], seg.device)).to(dout2.dtype)
      dseg = _mm_f32(onehot.t(), dout2).to(seg_table.dtype)
    if pos_table is not None and ctx.needs_input_grad[4]:
      dpos = torch.zeros(pos_table.shape, dtype=torch.float32, device=dout.device)
      dpos[:S] = dout.sum(0, dtype=torch.float32)
      dpos = dpos.to(pos_table.dtype)
    return (None, None, dword_ret, dseg, dpos, _finish(gamma, dg, direct), _finish(beta, db, direct), dpatch, None)


def embed_assemble(word_ids, seg_ids, word_table, seg_table, gamma, beta, pos_table=None, patch_proj=None,
                   eps=1e-12, p=0.0, seed=0, patch_start=2, out_dtype=torch.bfloat16):
  """[B,S] ids -> [B,S,H] embeddings in `out_dtype` (see `_EmbedAssembleFn`); `patch_proj`
  [B, n_patch, H] (projection bias included) is added at positions [patch_start, patch_start+n_patch)."""
  return _EmbedAssembleFn.apply(word_ids, seg_ids, word_table, seg_table, pos_table, gamma, beta, patch_proj,
                                (eps, p, seed, patch_start, out_dtype))


# ---- softmax cross-entropy over wide rows (MLM / MPP heads) ---------------------------------------
class _SoftmaxXentFn(torch.autograd.Function):
  """loss[row] = logsumexp(logits[row]) - logits[row, label[row]] (fp32), one pass over the logits in
  their storage dtype; backward writes dlogits = (softmax - onehot) * dloss[row] in that dtype."""

  @staticmethod
  def forward(ctx, logits, labels):
    _check(logits, labels)
    rows, C = logits.shape
    lab = labels.reshape(-1).to(torch.int32).contiguous()
    loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
    lse = torch.empty_like(loss)
    with torch.cuda.device(logits.device):
      _lib.check(_lib.lib().mmt_xent_fwd(rows, C, _dtype_code(logits.dtype), _p(logits), logits.stride(0), _p(lab),
                                         _p(loss), _p(lse), _stream(logits)))
    ctx.save_for_backward(logits, lab, lse)
    return loss

  @staticmethod
  def backward(ctx, dloss):
    logits, lab, lse = ctx.saved_tensors
    rows, C = logits.shape
    coef = dloss.to(torch.float32).contiguous()
    dlogits = torch.empty_like(logits)
    with torch.cuda.device(logits.device):
      _lib.check(_lib.lib().mmt_xent_bwd(rows, C, _dtype_code(logits.dtype), _p(logits), logits.stride(0), _p(lab),
                                         _p(lse), _p(coef), _p(dlogits), dlogits.stride(0), _stream(logits)))
    return dlogits, None


class _WeightedXentFn(torch.autograd.Function):
  """divide_no_nan(sum_i w_i l_i, sum_i w_i) with l_i the softmax cross-entropy of row i (times `lmul`), w_i the
  label weight (times `mask[i // mask_div]`): per-row losses in one pass over the logits (`mmt_xent_fwd`), the
  two sums, the guarded division and d loss / d l_i in one more launch (`mmt_weighted_loss`); backward is
  `mmt_xent_bwd_scaled` with the upstream gradient read on the device.  Replaces ~11 + ~8 framework kernels
  per loss term (`pretraining.py:95-140`)."""

  @staticmethod
  def forward(ctx, logits, labels, weight, lmul, mask, mask_div, want_argmax=False):
    rows, C = logits.shape
    dev = logits.device
    lab = labels.reshape(-1).to(torch.int32).contiguous()
    w = weight.reshape(-1).to(torch.float32).contiguous()
    lm = None if lmul is None else lmul.reshape(-1).to(torch.float32).contiguous()
    mk = None if mask is None else mask.reshape(-1).to(torch.float32).contiguous()
    buf = torch.empty(3 * rows + 3, dtype=torch.float32, device=dev)      # loss | lse | coef | {loss, num, den}
    loss, lse, coef, out3 = buf[:rows], buf[rows:2 * rows], buf[2 * rows:3 * rows], buf[3 * rows:]
    amax = torch.empty(rows, dtype=torch.int32, device=dev) if want_argmax else None
    with torch.cuda.device(dev):
      _lib.check(_lib.lib().mmt_xent_fwd_argmax(rows, C, _dtype_code(logits.dtype), _p(logits), logits.stride(0), _p(lab),
                                                _p(loss), _p(lse), _p(amax) if want_argmax else None, _stream(logits)))
      _lib.check(_lib.lib().mmt_weighted_loss(rows, _p(loss), _p(w), _p(lm) if lm is not None else None,
                                              _p(mk) if mk is not None else None, int(mask_div), _p(out3), _p(coef),
                                              _stream(logits)))
    ctx.save_for_backward(logits, lab, buf)
    if want_argmax:
      ctx.mark_non_differentiable(amax)
      return out3[0], amax
    return out3[0]

  @staticmethod
  def backward(ctx, dloss, *_unused):
    logits, lab, buf = ctx.saved_tensors
    rows, C = logits.shape
    lse, coef = buf[rows:2 * rows], buf[2 * rows:3 * rows]
    g = dloss.reshape(1).to(torch.float32)
    dlogits = torch.empty_like(logits)
    with torch.cuda.device(logits.device):
      _lib.check(_lib.lib().mmt_xent_bwd_scaled(rows, C, _dtype_code(logits.dtype), _p(logits), logits.stride(0), _p(lab),
                                                _p(lse), _p(coef), _p(g), _p(dlogits), dlogits.stride(0), _stream(logits)))
    return dlogits, None, None, None, None, None, None


def weighted_softmax_cross_entropy(logits, labels, weight, lmul=None, mask=None, mask_div=1, return_argmax=False):
  """Scalar `divide_no_nan(sum w l, sum w)` over the rows of 2-D `logits` (fp32 | bf16, unit column stride, at most
  65535 rows); `mask` (one value per `mask_div` consecutive rows) multiplies the weights, `lmul` the row losses.
  `return_argmax`: also the int32 [rows] first-occurrence arg-max of every row, taken in the same pass over the
  logits (for the accuracy metrics of `process_metrics`, src/tasks/pretraining.py:198-222)."""
  if logits.dim() != 2 or logits.stride(1) != 1 or not 0 < logits.shape[0] <= 65535:
    raise ValueError('weighted_softmax_cross_entropy expects [0 < rows <= 65535, C] logits with unit column stride')
  _check(logits, labels)
  return _WeightedXentFn.apply(logits, labels, weight, lmul, mask, mask_div, bool(return_argmax))


def softmax_cross_entropy(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
  """Per-row -log softmax(logits)[label] (fp32 [rows]) for 2-D logits (fp32 | bf16, unit column stride,
  at most 65535 rows); labels outside [0, C) give loss 0 and no gradient."""
  if logits.dim() != 2 or logits.stride(1) != 1 or logits.shape[0] > 65535:
    raise ValueError('softmax_cross_entropy expects [rows <= 65535, C] logits with unit column stride')
  return _SoftmaxXentFn.apply(logits, labels)
