"""Layers of the hot path, mirroring the pieces of `etcmodel.layers` / TF-Model-Garden that
`MmtEncoder` and the two model wrappers instantiate.

Reference call sites: `src/modeling/models/mmt_encoder.py:90-135` (EmbeddingLookup,
RelativeTransformerLayers), `mmt_pretraining_model.py:91-103` (MaskedLM, MaskedPP),
`tasks/pretraining.py:75-78` (ClassificationHead).  Semantics per SURVEY.md App. A.1-A.3.
torch is the buffer carrier / autograd glue; the attention core runs in the HIP kernels.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, ops


def truncated_normal_(t: torch.Tensor, stddev: float) -> torch.Tensor:
  """tf.keras.initializers.TruncatedNormal: resampled beyond two standard deviations."""
  return nn.init.trunc_normal_(t, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev)


def _GELU_TANH(x):
  return F.gelu(x, approximate='tanh')


def get_activation(name_or_fn) -> Optional[Callable]:
  """'gelu' is the tanh approximation, as in the reference encoder
  (`mmt_encoder.py:53-54`; TFM `tf_utils.get_activation('gelu')` is also approximate)."""
  if name_or_fn is None or callable(name_or_fn):
    return name_or_fn
  table = {'gelu': _GELU_TANH, 'relu': F.relu, 'tanh': torch.tanh,
           'linear': None, 'identity': None}
  if name_or_fn not in table:
    raise ValueError(f'unknown activation {name_or_fn!r}')
  return table[name_or_fn]


def _current_shadow(param: torch.Tensor, dtype: torch.dtype):
  """The low-precision copy `optimization.FusedAdamW` keeps next to a master parameter, or None.  The
  optimizer's kernel updates master and shadow together without touching torch's version counter; any
  OTHER in-place write to the master (load_state_dict, checkpoint restore, manual init -- all bump
  `param._version`) is noticed here and the shadow is re-synchronised before it is used."""
  shadow = getattr(param, '_mmt_shadow', None)
  if shadow is None or shadow.dtype != dtype:
    return None
  if param._version != getattr(param, '_mmt_shadow_version', param._version):
    with torch.no_grad():
      shadow.copy_(param.detach())
    param._mmt_shadow_version = param._version
  return shadow


class _CastParamFn(torch.autograd.Function):
  """fp32 master parameter -> compute dtype.  The backward accumulates the low-precision
  gradient straight into `param.grad` (one mixed-precision add instead of a cast kernel plus an
  add kernel) and then runs the parameter's gradient-ready hooks (the DP bucket reducer)."""

  @staticmethod
  def forward(ctx, param, dtype):
    ctx.param = param
    shadow = _current_shadow(param, dtype)
    return shadow if shadow is not None else param.detach().to(dtype)

  @staticmethod
  def backward(ctx, g):
    param = ctx.param
    if param.grad is None:
      param.grad = g.to(param.dtype)
    else:
      fused.accumulate_grad_(param.grad, g)
    for hook in getattr(param, '_mmt_grad_ready_hooks', ()):
      hook(param)
    return None, None


def cast_param(param: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
  if param.dtype == dtype:
    return param
  if not (torch.is_grad_enabled() and param.requires_grad):
    return param.to(dtype)
  return _CastParamFn.apply(param, dtype)




def _param_weight(param: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
  shadow = _current_shadow(param, dtype)
  return shadow if shadow is not None else param.detach().to(dtype)


def _notify(param):
  for hook in getattr(param, '_mmt_grad_ready_hooks', ()):
    hook(param)


def _add_bias_grad(bias, dy2):
  """bias.grad += dy2.sum(0): the HIP column-sum kernel, straight into an fp32 master gradient when there is one."""
  g = bias.grad
  if dy2.is_cuda and g is not None and g.dtype == torch.float32 and g.is_contiguous():
    fused.colsum(dy2, out=g)
    return
  db = fused.colsum(dy2) if dy2.is_cuda else dy2.sum(0, dtype=torch.float32)
  if g is None:
    bias.grad = db.to(bias.dtype)
  else:
    g.add_(db)


def _accumulate_dense_grads(weight, bias, dy2, x2):
  """weight.grad (fp32) += dy2^T @ x2 and bias.grad += dy2.sum(0) for a Dense layer y = x W^T + b, written
  straight into the master gradients: the hand-written split-K kernel (bias column sums fused, grouped with the
  block's other products), else library GEMM + accumulate.  Runs the parameters'
  gradient-ready hooks once the writes are enqueued."""
  want_b = bias is not None and bias.requires_grad
  b_done = False
  if weight.requires_grad:
    if weight.grad is None:
      weight.grad = torch.zeros_like(weight, dtype=torch.float32)
    fuse_b = want_b and bias.grad is not None and bias.grad.dtype == torch.float32 and bias.grad.is_contiguous()
    if (weight.is_cuda and weight.grad.dtype == torch.float32
        and fused.wgrad_accumulate_deferred_(weight.grad, dy2, x2, bias.grad if fuse_b else None,
                                             (weight, bias) if fuse_b else (weight,))):
      # queued with the block's other weight gradients; the parameters are reported ready at the launch
      if want_b and not fuse_b:
        _add_bias_grad(bias, dy2)
        _notify(bias)
      return
    wgrad = fused.wgrad_accumulate_
    if weight.grad.dtype == torch.float32 and wgrad(weight.grad, dy2, x2, bias.grad if fuse_b else None):
      b_done = fuse_b
    elif weight.grad.dtype == torch.float32:
      fused.accumulate_grad_(weight.grad, torch.mm(dy2.t(), x2))
    else:
      weight.grad.add_(torch.mm(dy2.t(), x2))
    _notify(weight)
  if want_b:
    if not b_done:
      _add_bias_grad(bias, dy2)
    _notify(bias)


class _LinearFn(torch.autograd.Function):
  """y = x @ W^T (+ b) with fp32 master parameters and bf16 compute.  Backward: dx by a library
  GEMM; dW by the hand-written split-K kernel `mmt_wgrad_accumulate`, accumulated in fp32
  straight into `weight.grad` (library GEMM + accumulate kernel when the shape is outside what
  the kernel is built for); db as an fp32 column sum into `bias.grad`."""

  @staticmethod
  def forward(ctx, x, weight, bias):
    w = _param_weight(weight, x.dtype)
    b = None if bias is None else _param_weight(bias, x.dtype)
    ctx.save_for_backward(x, w)
    ctx.params = (weight, bias)
    return F.linear(x, w, b)

  @staticmethod
  def backward(ctx, dy):
    x, w = ctx.saved_tensors
    weight, bias = ctx.params
    dy2 = dy.reshape(-1, dy.shape[-1])
    x2 = x.reshape(-1, x.shape[-1])
    if not dy2.is_contiguous():
      dy2 = dy2.contiguous()
    dx = torch.mm(dy2, w).view(x.shape) if ctx.needs_input_grad[0] else None
    _accumulate_dense_grads(weight, bias, dy2, x2)
    return dx, None, None


class _FfnFn(torch.autograd.Function):
  """f = gelu_tanh(x @ W1^T + b1) @ W2^T: the feed-forward pair of one encoder block with the activation inside
  backward GEMM's epilogue (`mmt_ffn_dgelu_gemm`), so the [B*S, 4H] gradient is not swept by a separate activation
  pass.  Forward: library GEMM + one bias/GELU pass (the fused forward GEMM, `mmt_ffn_gelu_gemm`, measured slower than
  that pair and is not used by the model).  Parameter gradients as `_LinearFn`."""

  @staticmethod
  def forward(ctx, x, w1, b1, w2):
    w1s, w2s = _param_weight(w1, x.dtype), _param_weight(w2, x.dtype)
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
      x2 = x2.contiguous()
    u = F.linear(x2, w1s)                # library GEMM + activation pass; u is kept WITHOUT the bias
    g = fused.bias_gelu_forward_(u, b1.detach())
    ctx.save_for_backward(x2, u, g, w1s, w2s)
    ctx.params = (w1, b1, w2)
    ctx.shape = x.shape
    return F.linear(g, w2s).view(*x.shape[:-1], w2.shape[0])

  @staticmethod
  def backward(ctx, df):
    x2, u, g, w1s, w2s = ctx.saved_tensors
    w1, b1, w2 = ctx.params
    df2 = df.reshape(-1, df.shape[-1])
    if not df2.is_contiguous():
      df2 = df2.contiguous()
    _accumulate_dense_grads(w2, None, df2, g)
    du = fused.ffn_dgelu_gemm(df2, w2s, u, b1.detach())
    if du is None:
      raise RuntimeError('mmt_ffn_dgelu_gemm refused a shape _ffn_ok admitted')
    _accumulate_dense_grads(w1, b1, du, x2)
    dx = torch.mm(du, w1s).view(ctx.shape) if ctx.needs_input_grad[0] else None
    return dx, None, None, None


def _ffn_ok(x, w1, b1, w2) -> bool:
  return (x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled()
          and all(isinstance(t, nn.Parameter) and t.dtype == torch.float32 for t in (w1, b1, w2))
          and x.numel() // x.shape[-1] % 256 == 0 and w1.shape[0] % 256 == 0 and x.shape[-1] % 64 == 0)


def _linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
  """x @ W^T + b in x's dtype (fp32 master weights, bf16 compute)."""
  if (x.is_cuda and x.dtype == torch.bfloat16 and torch.is_grad_enabled() and isinstance(weight, nn.Parameter)
      and weight.dtype == torch.float32 and (bias is None or isinstance(bias, nn.Parameter))):
    return _LinearFn.apply(x, weight, bias)
  return F.linear(x, cast_param(weight, x.dtype), None if bias is None else cast_param(bias, x.dtype))




class _TiedLogitsFn(torch.autograd.Function):
  """logits = x @ table^T + bias against the (tied) word table: the forward reads the bf16 shadow
  table the optimizer maintains (no 94 MB fp32 -> bf16 cast per step), the backward adds
  dlogits^T @ x into the fp32 master gradient with one mixed-precision accumulate (no cast + add
  pair).  It returns None for the table and does NOT run the gradient-ready hooks: the table's own
  AccumulateGrad node fires them once, after the embedding lookup's backward has added its part."""

  @staticmethod
  def forward(ctx, x, table, bias):
    w = _param_weight(table, x.dtype)
    b = None if bias is None else _param_weight(bias, x.dtype)
    ctx.save_for_backward(x, w)
    ctx.params = (table, bias)
    return F.linear(x, w, b)

  @staticmethod
  def backward(ctx, dy):
    x, w = ctx.saved_tensors
    table, bias = ctx.params
    dx = None
    if ctx.needs_input_grad[0]:
      M, V = dy.shape
      if (dy.is_cuda and dy.dtype == torch.bfloat16 and dy.is_contiguous() and w.is_contiguous()
          and V % 3 == 0 and V >= 8192 and M * w.shape[1] <= (1 << 21)):
        # [M, V] . [V, H] with a long V and a small M x H has 12-64 output tiles for 256 CUs and no split-K solution in
        # the tuned library set (149 us at 0.32 PFLOP/s at M = 1024): three K slices as one batched product (103 us,
        # tools/tied_dgrad_probe.py); 30522 = 3 x 10174.  bmm rounds each slice's partial to bf16 before the three are
        # added in fp32, so dx carries three roundings of partial sums where the single product has one of the total:
        # <= 3 bf16 ulps of the largest partial, inside the 3e-2 gradient bar (tests/test_gpu_encoder.py)
        k = V // 3
        dx = torch.bmm(dy.as_strided((3, M, k), (k, V, 1)), w.view(3, k, w.shape[1])).sum(0, dtype=torch.float32).to(dy.dtype)
      else:
        dx = torch.mm(dy, w)
    if table.requires_grad:
      if table.grad is None:
        table.grad = torch.zeros_like(table, dtype=torch.float32)
      fused.accumulate_grad_(table.grad, torch.mm(dy.t(), x))
    db = None
    if bias is not None and bias.requires_grad:
      db = (fused.colsum(dy) if dy.is_cuda else dy.sum(0, dtype=torch.float32)).to(bias.dtype)
    return dx, None, db


class EmbeddingLookup(nn.Module):
  """etc_layers.EmbeddingLookup (ctor args as at `mmt_encoder.py:90-95`): table [V,E], an
  optional projection when projection_size != embedding_size.  One-hot and gather lookups
  agree for in-range ids (App. A.4 (7)), so `use_one_hot_lookup` only documents intent."""

  def __init__(self, vocab_size: int, embedding_size: int, projection_size: Optional[int] = None,
               initializer_range: float = 0.02, use_one_hot_lookup: bool = False, name: str = ''):
    super().__init__()
    self.vocab_size, self.embedding_size = vocab_size, embedding_size
    self.projection_size = projection_size
    self.initializer_range = initializer_range
    self.use_one_hot_lookup = use_one_hot_lookup
    self.name = name
    self.embedding_table = nn.Parameter(torch.empty(vocab_size, embedding_size))
    truncated_normal_(self.embedding_table, initializer_range)
    if projection_size is not None and projection_size != embedding_size:
      self.embedding_projection = nn.Parameter(torch.empty(projection_size, embedding_size))
      truncated_normal_(self.embedding_projection, initializer_range)
    else:
      self.embedding_projection = None

  def forward(self, ids: torch.Tensor) -> torch.Tensor:
    out = F.embedding(ids.long(), self.embedding_table)
    if self.embedding_projection is not None:
      out = F.linear(out, self.embedding_projection)
    return out


class RelativeAttention(nn.Module):
  """etc_layers.RelativeAttention: Q/K/V projections (one fused [3H,H] GEMM), the
  QkvRelativeAttention core (HIP kernels), output projection.  App. A.3 `inner_att`."""

  def __init__(self, hidden_size: int, num_heads: int, relative_vocab_size: Optional[int],
               att_dropout_prob: float, initializer_range: float, use_relative_bias: bool = True):
    super().__init__()
    if hidden_size % num_heads:
      raise ValueError('`hidden_size` must be a multiple of `num_heads`.')
    self.hidden_size, self.num_heads = hidden_size, num_heads
    self.head_size = hidden_size // num_heads
    self.att_dropout_prob = att_dropout_prob
    self.qkv_weight = nn.Parameter(torch.empty(3 * hidden_size, hidden_size))
    self.qkv_bias = nn.Parameter(torch.zeros(3 * hidden_size))
    self.output_weight = nn.Parameter(torch.empty(hidden_size, hidden_size))
    self.output_bias = nn.Parameter(torch.zeros(hidden_size))
    truncated_normal_(self.qkv_weight, initializer_range)
    truncated_normal_(self.output_weight, initializer_range)
    if relative_vocab_size:
      self.relative_emb_table = nn.Parameter(
          torch.empty(relative_vocab_size, num_heads, self.head_size))
      truncated_normal_(self.relative_emb_table, initializer_range)
      if use_relative_bias:
        self.relative_bias_table = nn.Parameter(torch.empty(relative_vocab_size, num_heads))
        truncated_normal_(self.relative_bias_table, initializer_range)
      else:
        self.relative_bias_table = None
    else:
      self.relative_emb_table = self.relative_bias_table = None

  def forward(self, x, att_mask=None, relative_att_ids=None, pattern=None, valid_len=None,
              training=False, dropout_seed=0, add_output_bias=True):
    B, S, H = x.shape
    qkv = _linear(x, self.qkv_weight, self.qkv_bias).view(B, S, 3, self.num_heads, self.head_size)
    emb_p, bias_p = self.relative_emb_table, self.relative_bias_table
    if emb_p is None or (relative_att_ids is None and (pattern is None or pattern.id_mode == 0)):
      emb = bias = sinks = None
    elif (x.is_cuda and x.dtype != torch.float32 and torch.is_grad_enabled() and emb_p.requires_grad
          and emb_p.dtype == torch.float32 and (bias_p is None or (bias_p.requires_grad and bias_p.dtype == torch.float32))):
      # fp32 masters, low-precision compute: the backward adds the fp32 table gradients into .grad
      emb = _param_weight(emb_p, x.dtype)
      bias = None if bias_p is None else _param_weight(bias_p, x.dtype)
      sinks = (emb_p, bias_p)
    else:
      emb = cast_param(emb_p, x.dtype)
      bias = None if bias_p is None else cast_param(bias_p, x.dtype)
      sinks = None
    p_drop = self.att_dropout_prob if training else 0.0
    out = ops.relative_attention_qkv(
        qkv, emb, bias, rel_grad_sinks=sinks, att_mask=att_mask, relative_att_ids=relative_att_ids,
        pattern=pattern, valid_len=valid_len, dropout_p=p_drop,
        dropout_seed=int(dropout_seed) if p_drop > 0 else 0)
    return _linear(out.reshape(B, S, H), self.output_weight, self.output_bias if add_output_bias else None)


class RelativeTransformerLayer(nn.Module):
  """ResidualBlock(RelativeAttention) + ResidualBlock(FFN) (App. A.3 `layer`)."""

  def __init__(self, hidden_size, num_heads, intermediate_size, activation, hidden_dropout_prob,
               att_dropout_prob, initializer_range, relative_vocab_size, use_pre_activation_order):
    super().__init__()
    self.attention = RelativeAttention(hidden_size, num_heads, relative_vocab_size,
                                       att_dropout_prob, initializer_range)
    self.attention_layer_norm = nn.LayerNorm(hidden_size, eps=1e-12)
    self.ffn_layer_norm = nn.LayerNorm(hidden_size, eps=1e-12)
    self.intermediate_weight = nn.Parameter(torch.empty(intermediate_size, hidden_size))
    self.intermediate_bias = nn.Parameter(torch.zeros(intermediate_size))
    self.ffn_output_weight = nn.Parameter(torch.empty(hidden_size, intermediate_size))
    self.ffn_output_bias = nn.Parameter(torch.zeros(hidden_size))
    truncated_normal_(self.intermediate_weight, initializer_range)
    truncated_normal_(self.ffn_output_weight, initializer_range)
    self.activation = activation
    self.hidden_dropout_prob = hidden_dropout_prob
    self.use_pre_activation_order = use_pre_activation_order

  def _ln(self, ln: nn.LayerNorm, x):
    return F.layer_norm(x, ln.normalized_shape, ln.weight.to(x.dtype), ln.bias.to(x.dtype), ln.eps)

  def _ffn(self, x):
    y = _linear(x, self.intermediate_weight, self.intermediate_bias)
    if self.activation is not None:
      y = self.activation(y)
    return _linear(y, self.ffn_output_weight, self.ffn_output_bias)

  def forward(self, x, training=False, **att_kw):
    drop = lambda t: F.dropout(t, self.hidden_dropout_prob, training)
    if self.use_pre_activation_order:   # y = x + Dropout(inner(LN(x)))
      x = x + drop(self.attention(self._ln(self.attention_layer_norm, x), training=training, **att_kw))
      x = x + drop(self._ffn(self._ln(self.ffn_layer_norm, x)))
    else:                               # y = LN(x + Dropout(inner(x)))
      x = self._ln(self.attention_layer_norm, x + drop(self.attention(x, training=training, **att_kw)))
      x = self._ln(self.ffn_layer_norm, x + drop(self._ffn(x)))
    return x


class RelativeTransformerLayers(nn.Module):
  """etc_layers.RelativeTransformerLayers with the ctor args of `mmt_encoder.py:124-135` and
  the call contract of `:220-224`: (inputs, att_mask, relative_att_ids, training).  The
  `pattern` / `valid_len` keywords select the structured fast path instead of dense side
  inputs (the build's extension; equal to the dense operator on the materialised mask)."""

  def __init__(self, hidden_size, num_hidden_layers, num_attention_heads, intermediate_size=None,
               hidden_act=None, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
               initializer_range=0.02, relative_vocab_size=None, use_pre_activation_order=False,
               use_one_hot_lookup=True):
    super().__init__()
    intermediate_size = intermediate_size or 4 * hidden_size
    self.use_one_hot_lookup = use_one_hot_lookup
    self.use_fused_kernels = True
    self.layers = nn.ModuleList([
        RelativeTransformerLayer(hidden_size, num_attention_heads, intermediate_size, hidden_act,
                                 hidden_dropout_prob, attention_probs_dropout_prob,
                                 initializer_range, relative_vocab_size, use_pre_activation_order)
        for _ in range(num_hidden_layers)])

  def _fused_ok(self, x):
    l0 = self.layers[0] if len(self.layers) else None
    return (l0 is not None and x.is_cuda and l0.use_pre_activation_order and l0.activation is _GELU_TANH
            and x.dtype in (torch.float32, torch.bfloat16) and x.shape[-1] % 8 == 0 and x.shape[-1] <= 2048
            and l0.intermediate_weight.shape[0] % 8 == 0 and l0.intermediate_weight.shape[0] <= 8192)

  def _forward_fused(self, x, training, dropout_seed, att_kw):
    """Pre-activation stack with the fused HIP residual-block kernels: per layer two
    `residual_block` calls (bias + dropout + residual + the NEXT LayerNorm) and one `bias_gelu`;
    the GEMMs run without bias epilogues."""
    L = len(self.layers)
    p = self.layers[0].hidden_dropout_prob if training else 0.0
    first = self.layers[0].attention_layer_norm
    x, h = fused.layer_norm_keep(x, first.weight, first.bias, first.eps)    # x: alias, both gradients come back through one node
    for i, layer in enumerate(self.layers):
      o = layer.attention(h, training=training, dropout_seed=dropout_seed * 131 + i,
                          add_output_bias=False, **att_kw)
      ln2 = layer.ffn_layer_norm
      x, h2 = fused.residual_block(o, layer.attention.output_bias, x, ln2.weight, ln2.bias, ln2.eps,
                                   p, fused.next_seed(dropout_seed) if p else 0)
      if _ffn_ok(h2, layer.intermediate_weight, layer.intermediate_bias, layer.ffn_output_weight):
        f = _FfnFn.apply(h2, layer.intermediate_weight, layer.intermediate_bias, layer.ffn_output_weight)
      else:
        u = _linear(h2, layer.intermediate_weight, None)
        y = fused.bias_gelu(u, layer.intermediate_bias)
        f = _linear(y, layer.ffn_output_weight, None)
      nxt = self.layers[i + 1].attention_layer_norm if i + 1 < L else None
      x, h = fused.residual_block(f, layer.ffn_output_bias, x,
                                  None if nxt is None else nxt.weight, None if nxt is None else nxt.bias,
                                  1e-12 if nxt is None else nxt.eps, p,
                                  fused.next_seed(dropout_seed) if p else 0)
    return x

  def forward(self, inputs, att_mask=None, relative_att_ids=None, training=False, pattern=None,
              valid_len=None, dropout_seed=0):
    x = inputs
    if self.use_fused_kernels and self._fused_ok(x):
      return self._forward_fused(x, training, dropout_seed,
                                 dict(att_mask=att_mask, relative_att_ids=relative_att_ids,
                                      pattern=pattern, valid_len=valid_len))
    for i, layer in enumerate(self.layers):
      x = layer(x, training=training, att_mask=att_mask, relative_att_ids=relative_att_ids,
                pattern=pattern, valid_len=valid_len, dropout_seed=dropout_seed * 131 + i)
    return x


_ROW_OFFSETS = {}


def _row_offsets(B: int, S: int, device, plus: int = 0, dtype=torch.long) -> torch.Tensor:
  """Cached [B,1] column b * S + plus: constants of the batch geometry, not rebuilt every step (kept in the
  positions' own integer dtype: a mixed int32 + int64 add runs torch's slow casting kernel, 12 us for 1 K elements)."""
  key = (B, S, plus, str(device), dtype)
  t = _ROW_OFFSETS.get(key)
  if t is None:
    t = _ROW_OFFSETS[key] = (torch.arange(B, device=device, dtype=torch.long) * S + plus).to(dtype).view(-1, 1)
  return t


def gather_rows_merged(sequence_tensor: torch.Tensor, position_sets):
  """One index_select for several heads: `position_sets` is a list of [B,M_i] position tensors
  (gather_indexes semantics each) or plain ints (the same position in every example, e.g. a
  classification head's [CLS] index); returns the list of [B*M_i, W] row blocks.  One gather forward
  and ONE dense scatter (zero-fill + index_add) backward instead of one per head."""
  B, S, W = sequence_tensor.shape
  dev = sequence_tensor.device
  tensors = [p for p in position_sets if not isinstance(p, int)]
  idt = tensors[0].dtype if tensors and all(t.dtype == tensors[0].dtype for t in tensors) and \
      tensors[0].dtype in (torch.int32, torch.int64) and B * S < 2 ** 31 else torch.long
  flat = [_row_offsets(B, S, dev, p, idt).reshape(-1) if isinstance(p, int)
          else (p.to(idt) + _row_offsets(B, S, dev, 0, idt)).reshape(-1) for p in position_sets]
  rows = sequence_tensor.reshape(B * S, W).index_select(0, torch.cat(flat))
  return list(torch.split(rows, [f.numel() for f in flat]))


def gather_indexes(sequence_tensor: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
  """`src/tensor_utils.py:27-44`: [B,S,W], [B,M] -> [B*M, W]."""
  B, S, W = sequence_tensor.shape
  flat_offsets = (torch.arange(B, device=positions.device) * S).view(-1, 1)
  flat_positions = (positions.long() + flat_offsets).reshape(-1)
  return sequence_tensor.reshape(B * S, W).index_select(0, flat_positions)


def _head_layer_norm(ln: nn.LayerNorm, x: torch.Tensor) -> torch.Tensor:
  """LayerNorm of a head: the HIP row kernel (fp32 gamma / beta read as they are, dgamma / dbeta added straight into
  the fp32 master gradients) on the GPU, torch's elsewhere."""
  H = x.shape[-1]
  if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and H % 8 == 0 and H <= 2048
      and ln.weight.dtype == torch.float32 and ln.bias.dtype == torch.float32):
    return fused.layer_norm(x, ln.weight, ln.bias, ln.eps)
  return F.layer_norm(x, ln.normalized_shape, ln.weight.to(x.dtype), ln.bias.to(x.dtype), ln.eps)


class MaskedLM(nn.Module):
  """TFM keras_nlp.layers.MaskedLM (output='logits'): gather -> dense(act) -> LN -> tied
  logits + bias (`mmt_pretraining_model.py:91-96`)."""

  def __init__(self, embedding_layer: EmbeddingLookup, activation=None, bind: bool = True):
    super().__init__()
    V, E = embedding_layer.vocab_size, embedding_layer.embedding_size
    self._embedding_layer = [embedding_layer] if bind else None   # not a submodule: tied table
    if not bind:
      self.mlm_embedding_table = nn.Parameter(torch.empty(V, E))
      truncated_normal_(self.mlm_embedding_table, embedding_layer.initializer_range)
    self.dense_weight = nn.Parameter(torch.empty(E, E))
    nn.init.xavier_uniform_(self.dense_weight)          # glorot_uniform
    self.dense_bias = nn.Parameter(torch.zeros(E))
    self.layer_norm = nn.LayerNorm(E, eps=1e-12)
    self.output_bias = nn.Parameter(torch.zeros(V))
    self.activation = activation

  @property
  def embedding_table(self):
    return self._embedding_layer[0].embedding_table if self._embedding_layer else self.mlm_embedding_table

  def forward(self, sequence_data, masked_positions, gathered=None):
    # `gathered`: rows already picked by the caller's single merged gather (models.py)
    x = gather_indexes(sequence_data, masked_positions) if gathered is None else gathered
    if (self.activation is _GELU_TANH and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2
        and torch.is_grad_enabled()):
      # dense without its bias, then bias + tanh-GELU in the encoder's fused kernel (forward and backward one pass
      # each, the bias gradient by the column-sum kernel) instead of the framework's GELU pair
      # (`mmt_pretraining_model.py:91-103`: MaskedLM = dense + activation + LayerNorm + tied logits)
      x = fused.bias_gelu(_linear(x, self.dense_weight, None), self.dense_bias)
    else:
      x = _linear(x, self.dense_weight, self.dense_bias)
      if self.activation is not None:
        x = self.activation(x)
    x = _head_layer_norm(self.layer_norm, x)
    # the tied table also receives the embedding-lookup gradient through plain autograd, so it
    # must not use the accumulate-in-backward cast (one gradient-ready event per parameter)
    table = self.embedding_table
    if (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and torch.is_grad_enabled()
        and isinstance(table, nn.Parameter) and table.dtype == torch.float32 and table.requires_grad):
      logits = _TiedLogitsFn.apply(x, table, self.output_bias)
    else:
      logits = F.linear(x, table.to(x.dtype), cast_param(self.output_bias, x.dtype))
    return logits.view(masked_positions.shape[0], masked_positions.shape[1], -1)


class _AddRowBiasFn(torch.autograd.Function):
  """x[rows, C] + bias[C]; the bias gradient is one pass of the HIP column-sum kernel (torch's reduction needs 22 us
  for the [392, 512] logits of the patch-prediction head)."""

  @staticmethod
  def forward(ctx, x, bias):
    ctx.bias_dtype = bias.dtype
    return x + bias.to(x.dtype)

  @staticmethod
  def backward(ctx, dy):
    dy2 = dy.reshape(-1, dy.shape[-1])
    return dy, fused.colsum(dy2 if dy2.stride(1) == 1 else dy2.contiguous()).to(ctx.bias_dtype)


def _add_row_bias(x, bias):
  if x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and bias.requires_grad:
    return _AddRowBiasFn.apply(x, bias)
  return x + bias.to(x.dtype)


class MaskedPP(nn.Module):
  """`src/modeling/layers/masked_patch_prediction_layer.py:58-98`: gather -> LN -> dense -> bias."""

  def __init__(self, hidden_size: int, output_num_classes: int, activation=None, output: str = 'logits'):
    super().__init__()
    if output not in ('predictions', 'logits'):
      raise ValueError(f'Unknown `output` value "{output}". `output` can be either "logits"'
                       f'or "predictions"')
    self._output_type = output
    self.layer_norm = nn.LayerNorm(hidden_size, eps=1e-12)
    self.dense_weight = nn.Parameter(torch.empty(output_num_classes, hidden_size))
    nn.init.xavier_uniform_(self.dense_weight)
    self.dense_bias = nn.Parameter(torch.zeros(output_num_classes))
    self.bias = nn.Parameter(torch.zeros(output_num_classes))
    self.activation = activation

  def forward(self, sequence_data, masked_positions, gathered=None):
    x = gather_indexes(sequence_data, masked_positions) if gathered is None else gathered
    x = _head_layer_norm(self.layer_norm, x)
    x = _linear(x, self.dense_weight, self.dense_bias)
    if self.activation is not None:
      x = self.activation(x)
    logits = _add_row_bias(x, self.bias).view(masked_positions.shape[0], masked_positions.shape[1], -1)
    return logits if self._output_type == 'logits' else F.log_softmax(logits.float(), -1)


class ClassificationHead(nn.Module):
  """TFM nlp ClassificationHead built from `ClsHeadConfig` (`src/configs/mmt.py:24-31`)."""

  def __init__(self, hidden_size: int, inner_dim: int = 0, num_classes: int = 2,
               activation: Optional[str] = 'tanh', dropout_rate: float = 0.0,
               cls_token_idx: int = 0, name: Optional[str] = None):
    super().__init__()
    self.name = name or 'cls_head'
    self.cls_token_idx, self.dropout_rate = cls_token_idx, dropout_rate
    self.activation = get_activation(activation)
    in_dim = hidden_size
    if inner_dim:
      self.dense_weight = nn.Parameter(torch.empty(inner_dim, hidden_size))
      nn.init.xavier_uniform_(self.dense_weight)
      self.dense_bias = nn.Parameter(torch.zeros(inner_dim))
      in_dim = inner_dim
    else:
      self.dense_weight = None
    self.out_proj_weight = nn.Parameter(torch.empty(num_classes, in_dim))
    nn.init.xavier_uniform_(self.out_proj_weight)
    self.out_proj_bias = nn.Parameter(torch.zeros(num_classes))

  def forward(self, features, training=False, gathered=None):
    x = features[:, self.cls_token_idx] if gathered is None else gathered
    x = F.dropout(x, self.dropout_rate, training)
    if self.dense_weight is not None:
      x = _linear(x, self.dense_weight, self.dense_bias)
      if self.activation is not None:
        x = self.activation(x)
      x = F.dropout(x, self.dropout_rate, training)
    return _linear(x, self.out_proj_weight, self.out_proj_bias)

  @property
  def checkpoint_items(self):
    # TFM ClassificationHead.checkpoint_items: the inner dense layer and the output projection
    items = {'out_proj': {'kernel': self.out_proj_weight, 'bias': self.out_proj_bias}}
    if self.dense_weight is not None:
      items['dense'] = {'kernel': self.dense_weight, 'bias': self.dense_bias}
    return items


def weighted_sparse_categorical_crossentropy_loss(logits, labels, label_weights, metrics=None,
                                                  name='', pos_weights=None, example_mask=None, aux=None):
  """`src/modeling/losses/weighted_sparse_categorical_crossentropy_loss.py:17-43` incl.
  `divide_no_nan` (an all-zero weight vector gives loss 0, App. B q13).  `example_mask` ([B], optional) multiplies
  the weights of every row of its example (the ITM label that masks the MLM / MPP terms, `pretraining.py:101-109`).
  `metrics`: the task's metric objects (list, as the reference passes them; `{name}_loss` is updated, `:42`) or the
  legacy dict that collects the loss values.  `aux` (dict, optional): receives `{name}_argmax`, the first-occurrence
  arg-max of every logits row from the loss kernel's own pass, for `process_metrics`."""
  from . import metrics as metrics_lib
  flat = logits.reshape(-1, logits.shape[-1])
  want_amax = aux is not None and f'{name}_accuracy' in metrics_lib.by_name(metrics)
  if flat.is_cuda and flat.dtype in (torch.float32, torch.bfloat16) and flat.stride(1) == 1 and 0 < flat.shape[0] <= 65535:
    # per-row losses in one HIP pass over the logits in their storage dtype, then sums + guarded division +
    # per-row derivative in one more launch
    div = flat.shape[0] // example_mask.numel() if example_mask is not None else 1
    loss = fused.weighted_softmax_cross_entropy(flat, labels.reshape(-1), label_weights, lmul=pos_weights,
                                                mask=example_mask, mask_div=div, return_argmax=want_amax)
    if want_amax:
      loss, aux[f'{name}_argmax'] = loss[0], loss[1].view(labels.shape)
  else:
    unweighted = F.cross_entropy(flat.float(), labels.reshape(-1).long(), reduction='none').view(labels.shape)
    if pos_weights is not None:
      unweighted = unweighted * pos_weights.to(unweighted.dtype)
    w = label_weights.to(unweighted.dtype)
    if example_mask is not None:
      w = w * example_mask.to(w.dtype).reshape(-1, *([1] * (w.dim() - 1)))
    num, den = (w * unweighted).sum(), w.sum()
    loss = torch.where(den != 0, num / torch.where(den != 0, den, torch.ones_like(den)),
                       torch.zeros_like(num))
  if metrics is not None:
    named = metrics_lib.by_name(metrics)
    if f'{name}_loss' in named:
      named[f'{name}_loss'].update_state(loss.detach())
    elif isinstance(metrics, dict):
      metrics.setdefault(f'{name}_loss', []).append(loss.detach())
  return loss
