"""Host side of the hot path: torch tensors as buffer carriers over the C ABI.

`relative_attention` is the operator `etcmodel.layers.attention.QkvRelativeAttention`
computes inside `RelativeTransformerLayers` (reference call site
src/modeling/models/mmt_encoder.py:220-224; math SURVEY.md App. A.3).  Two input forms:

  * dense   : `att_mask`, `relative_att_ids` int32 [B,S,S] exactly as the reference feeds
              them (literal operator);
  * pattern : an `AttentionPattern` descriptor -- mask and ids are generated in-kernel and
              never materialised (the long-sequence fast path).
"""
from __future__ import annotations

import dataclasses
import math
import weakref
from typing import Optional

import torch

from . import _lib, step_scalars


@dataclasses.dataclass(frozen=True)
class AttentionPattern:
  """Structured mask + relative-id generator (mirrors `mmt_mask_desc`).

  local_radius >= seq_len with n_global == 0 is the reference's segmented mask
  (src/data/data_utils.py:321-322); id_mode 1 is the etcmodel 1-D generator
  (data_utils.py:300-301), id_mode 2 `MmtRelativePositionGenerator`
  (src/feature_utils.py:29).

  Global tokens: the contiguous range [global_start, global_start + n_global), or `global_index`, any set of
  positions (a tuple of ints).  A listed set that is in fact a contiguous run takes the structured kernels like
  the range form; any other set is served through the dense operator with the materialised mask (correct, but
  O(S^2) work and memory: there is no structured kernel for scattered global tokens).
  """
  local_radius: int = 1 << 30
  global_start: int = 0
  n_global: int = 0
  id_mode: int = _lib.MMT_IDS_1D
  max_dist: int = 12
  patches_per_row: int = 0
  core_layers: int = 0
  global_index: Optional[tuple] = None

  def normalized(self) -> 'AttentionPattern':
    """The same pattern with a listed global set sorted, de-duplicated and -- when it is a contiguous run -- turned
    into the range form."""
    if self.global_index is None:
      return self
    idx = tuple(sorted({int(i) for i in self.global_index}))
    if any(i < 0 for i in idx):
      raise ValueError('global_index must hold non-negative positions')
    if not idx:
      return dataclasses.replace(self, global_index=None, global_start=0, n_global=0)
    if idx[-1] - idx[0] + 1 == len(idx):
      return dataclasses.replace(self, global_index=None, global_start=idx[0], n_global=len(idx))
    return dataclasses.replace(self, global_index=idx, global_start=0, n_global=len(idx))

  def to_desc(self, valid_len: Optional[torch.Tensor] = None, device=None) -> _lib.MaskDesc:
    m = _lib.MaskDesc()
    m.valid_len = valid_len.data_ptr() if valid_len is not None else None
    m.local_radius = min(int(self.local_radius), (1 << 31) - 1)
    m.global_start, m.n_global = int(self.global_start), int(self.n_global)
    m.id_mode, m.max_dist = int(self.id_mode), int(self.max_dist)
    m.patches_per_row, m.core_layers = int(self.patches_per_row), int(self.core_layers)
    m.global_index = None
    if self.global_index is not None:
      if device is None:
        raise ValueError('a pattern with global_index needs the device its index list lives on')
      m.n_global = len(self.global_index)
      m.global_index = _index_list(self.global_index, device).data_ptr()
    return m


_INDEX_LISTS = {}


def _index_list(idx: tuple, device) -> torch.Tensor:
  """Device copy of a listed global set (kept alive here: descriptors carry raw pointers)."""
  key = (idx, str(device))
  t = _INDEX_LISTS.get(key)
  if t is None:
    if len(_INDEX_LISTS) > 64:
      _INDEX_LISTS.clear()
    t = _INDEX_LISTS[key] = torch.tensor(idx, dtype=torch.int32, device=device)
  return t


_DENSE_CACHE = {}


def _materialized(pattern: 'AttentionPattern', valid_len, B: int, S: int, device):
  """(att_mask, relative_att_ids) int32 [B,S,S] of a pattern with a listed global set, through `mmt_side_inputs`;
  the last result is kept, so that the layers of one encoder pass (same pattern, same valid_len tensor) share it.
  The entry is tied to the valid_len tensor OBJECT (a weak reference + its version counter), never to its address:
  a later batch's tensor that the caching allocator places at the same address is a different object and misses;
  when the tensor dies the entry (two B*S*S int32 tensors) is dropped with it."""
  key = (pattern, B, S, str(device))
  hit = _DENSE_CACHE.get('last')
  if hit is not None and hit[0] == key:
    ref, ver = hit[3]
    if (valid_len is None and ref is None) or (ref is not None and valid_len is not None and ref() is valid_len
                                               and ver == valid_len._version):
      return hit[1], hit[2]
  if any(i >= S for i in pattern.global_index):
    raise ValueError('global_index position outside the sequence')
  img = valid_len if valid_len is not None else torch.full((B,), S, dtype=torch.int32, device=device)
  txt = torch.zeros(B, dtype=torch.int32, device=device)
  mask = torch.empty((B, S, S), dtype=torch.int32, device=device)
  ids = torch.empty((B, S, S), dtype=torch.int32, device=device) if pattern.id_mode != _lib.MMT_IDS_NONE else None
  desc = pattern.to_desc(None, device)
  with torch.cuda.device(device):
    _lib.check(_lib.lib().mmt_side_inputs(desc, B, S, img.data_ptr(), txt.data_ptr(), 1, mask.data_ptr(),
                                          None if ids is None else ids.data_ptr(), None, _stream_ptr(device)))
  if valid_len is None:
    tie = (None, 0)
  else:
    tie = (weakref.ref(valid_len, lambda _r: _DENSE_CACHE.pop('last', None) if _DENSE_CACHE.get('last', (None,) * 4)[3][0] is _r else None),
           valid_len._version)
  _DENSE_CACHE['last'] = (key, mask, ids, tie)
  return mask, ids


def clear_pattern_cache() -> None:
  """Drops the cached dense side inputs of listed global sets (and the device copies of their index lists)."""
  _DENSE_CACHE.clear()
  _INDEX_LISTS.clear()


def _resolve_pattern(pattern, att_mask, rel_ids, valid_len, q):
  """Listed global sets: contiguous runs become the range form, anything else the dense operator's inputs."""
  if pattern is None or pattern.global_index is None:
    return pattern, att_mask, rel_ids
  pattern = pattern.normalized()
  if pattern.global_index is None:
    return pattern, att_mask, rel_ids
  if att_mask is not None or rel_ids is not None:
    raise ValueError('pass either dense att_mask/relative_att_ids or a pattern, not both')
  mask, ids = _materialized(pattern, valid_len, q.shape[0], q.shape[1], q.device)
  return None, mask, ids


def _stream_ptr(device) -> int:
  return torch.cuda.current_stream(device).cuda_stream


def _dtype_code(t: torch.Tensor) -> int:
  if t.dtype == torch.float32:
    return _lib.MMT_F32
  if t.dtype == torch.bfloat16:
    return _lib.MMT_BF16
  raise TypeError(f'relative_attention supports float32 and bfloat16, got {t.dtype}')


def _strides(t: torch.Tensor):
  if t.dim() != 4 or t.stride(3) != 1:
    raise ValueError('q/k/v/out must be [B,S,N,D] views with a contiguous head dimension')
  return (t.stride(0), t.stride(1), t.stride(2))


_SYNC = {}


def _sync_words(device, stream_ptr: int, words: int) -> torch.Tensor:
  """The zero-initialised arrival counters kernels that combine partial results inside one launch need
  (`mmt_attn_desc.sync`): one buffer per (device, stream) -- calls that share one must be stream-ordered -- created
  zeroed once; every call leaves it zeroed."""
  key = (str(device), int(stream_ptr))
  t = _SYNC.get(key)
  if t is None or t.numel() < words:
    if len(_SYNC) > 64:
      _SYNC.clear()
    t = _SYNC[key] = torch.zeros(max(int(words), 1024), dtype=torch.int32, device=device)
  return t


def _make_desc(q, k, v, out, R, pattern, valid_len, scale, mask_value, scale_before_add,
               dropout_p, dropout_seed, tuning=0) -> _lib.AttnDesc:
  B, S, N, D = q.shape
  d = _lib.AttnDesc()
  d.B, d.S, d.N, d.D, d.R = B, S, N, D, R
  d.dtype = _dtype_code(q)
  for name, t in (('q_stride', q), ('k_stride', k), ('v_stride', v), ('o_stride', out)):
    getattr(d, name)[:] = _strides(t)
  d.scale = float(scale if scale is not None else 1.0 / math.sqrt(D))
  d.mask_value = float(mask_value)
  d.flags = _lib.MMT_FLAG_SCALE_BEFORE_ADD if scale_before_add else 0
  d.dropout_p = float(dropout_p)
  d.dropout_seed = (int(dropout_seed) + step_scalars.host_epoch(q.device)) & ((1 << 64) - 1)
  d.dropout_epoch = step_scalars.epoch_ptr(q.device) if dropout_p else None
  d.mask = (pattern or AttentionPattern(id_mode=_lib.MMT_IDS_NONE)).to_desc(valid_len, q.device)
  d.tuning = int(tuning)
  sync = _sync_words(q.device, _stream_ptr(q.device), B * N)
  d.sync, d.sync_words = sync.data_ptr(), sync.numel()
  return d


def _check_inputs(q, k, v, rel_emb, rel_bias, att_mask, rel_ids, valid_len):
  if not q.is_cuda:
    raise RuntimeError('relative_attention runs on the GPU only (no CPU fallback)')
  for t in (k, v):
    if t.shape != q.shape or t.dtype != q.dtype or t.device != q.device:
      raise ValueError('q, k, v must agree in shape, dtype and device')
  B, S, N, D = q.shape
  R = 0
  if rel_emb is not None:
    if rel_emb.dim() != 3 or rel_emb.shape[1:] != (N, D) or rel_emb.dtype != q.dtype:
      raise ValueError('rel_emb must be [R,N,D] in the dtype of q')
    R = rel_emb.shape[0]
    if not rel_emb.is_contiguous():
      raise ValueError('rel_emb must be contiguous')
    if rel_bias is not None and (rel_bias.shape != (R, N) or rel_bias.dtype != q.dtype
                                 or not rel_bias.is_contiguous()):
      raise ValueError('rel_bias must be contiguous [R,N] in the dtype of q')
  for t in (att_mask, rel_ids):
    if t is not None and (t.dtype != torch.int32 or t.shape != (B, S, S) or not t.is_contiguous()):
      raise ValueError('att_mask / relative_att_ids must be contiguous int32 [B,S,S]')
  if valid_len is not None and (valid_len.dtype != torch.int32 or valid_len.shape != (B,)
                                or not valid_len.is_contiguous()):
    raise ValueError('valid_len must be contiguous int32 [B]')
  return R


def relative_attention_forward(q, k, v, rel_emb=None, rel_bias=None, *, att_mask=None,
                               relative_att_ids=None, pattern: Optional[AttentionPattern] = None,
                               valid_len=None, scale=None, mask_value=-10000.0,
                               scale_before_add=False, dropout_p=0.0, dropout_seed=0,
                               return_lse=True, tuning=0):
  """Forward only.  Returns (out [B,S,N,D] in q.dtype, lse fp32 [B,N,S]).  `tuning`: `_lib.MMT_TUNE_*` kernel-selection
  switches (0 = the library's defaults; the parity tests use them to reach every kernel)."""
  R = _check_inputs(q, k, v, rel_emb, rel_bias, att_mask, relative_att_ids, valid_len)
  pattern, att_mask, relative_att_ids = _resolve_pattern(pattern, att_mask, relative_att_ids, valid_len, q)
  dense = att_mask is not None or relative_att_ids is not None
  if dense and pattern is not None:
    raise ValueError('pass either dense att_mask/relative_att_ids or a pattern, not both')
  B, S, N, D = q.shape
  out = torch.empty((B, S, N, D), dtype=q.dtype, device=q.device)
  lse = torch.empty((B, N, S), dtype=torch.float32, device=q.device) if return_lse else None
  desc = _make_desc(q, k, v, out, R, pattern, valid_len, scale, mask_value, scale_before_add,
                    dropout_p, dropout_seed, tuning)
  L = _lib.lib()
  ws_bytes = 0 if dense else L.mmt_workspace_bytes(desc)
  ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=q.device)
  ptr = lambda t: None if t is None else t.data_ptr()
  with torch.cuda.device(q.device):
    _lib.check(L.mmt_attn_fwd(desc, ptr(q), ptr(k), ptr(v), ptr(rel_emb), ptr(rel_bias),
                              ptr(att_mask), ptr(relative_att_ids), ptr(out), ptr(lse), ptr(ws),
                              ws.numel(), _stream_ptr(q.device)))
  return out, lse


def side_inputs(pattern: AttentionPattern, num_image_wordpieces: torch.Tensor,
                num_text_wordpieces: torch.Tensor, max_seq_len: int, *, want_mask=True,
                want_ids=True, want_segment_ids=True, materialize_pattern=False):
  """Device-side `add_side_input_features` (src/data/data_utils.py:335-379), batched.

  Returns dict(segment_ids [B,S], att_mask [B,S,S], relative_att_ids [B,S,S]) int32.
  """
  if not num_image_wordpieces.is_cuda:
    raise RuntimeError('side_inputs runs on the GPU only (no CPU fallback)')
  dev = num_image_wordpieces.device
  B, S = num_image_wordpieces.shape[0], int(max_seq_len)
  img = num_image_wordpieces.to(torch.int32).contiguous()
  txt = num_text_wordpieces.to(torch.int32).contiguous()
  want_ids = want_ids and pattern.id_mode != _lib.MMT_IDS_NONE and pattern.max_dist > 0
  mask = torch.empty((B, S, S), dtype=torch.int32, device=dev) if want_mask else None
  ids = torch.empty((B, S, S), dtype=torch.int32, device=dev) if want_ids else None
  seg = torch.empty((B, S), dtype=torch.int32, device=dev) if want_segment_ids else None
  pattern = pattern.normalized()
  desc = pattern.to_desc(None, dev)
  ptr = lambda t: None if t is None else t.data_ptr()
  with torch.cuda.device(dev):
    _lib.check(_lib.lib().mmt_side_inputs(desc, B, S, ptr(img), ptr(txt), int(materialize_pattern),
                                          ptr(mask), ptr(ids), ptr(seg), _stream_ptr(dev)))
  return {'segment_ids': seg, 'att_mask': mask, 'relative_att_ids': ids}


def relative_attention_backward(dout, q, k, v, rel_emb, rel_bias, out, lse, *, att_mask=None,
                                relative_att_ids=None, pattern: Optional[AttentionPattern] = None,
                                valid_len=None, scale=None, mask_value=-10000.0,
                                scale_before_add=False, dropout_p=0.0, dropout_seed=0, grads_out=None,
                                rel_grads_accum=None, tuning=0):
  """Backward of `relative_attention_forward` (recomputes P from `lse`).

  Returns (dq, dk, dv, drel_emb, drel_bias); the table gradients are fp32.  `grads_out` may
  give preallocated (dq, dk, dv) with the strides of (q, k, v) -- e.g. the three slices of one
  fused [B,S,3,N,D] gradient buffer.  `rel_grads_accum` = (demb [R,N,D], dbias [R,N] | None), fp32
  and contiguous: the table gradients are ADDED to these buffers (the fp32 master gradients) and
  returned as such."""
  R = _check_inputs(q, k, v, rel_emb, rel_bias, att_mask, relative_att_ids, valid_len)
  pattern, att_mask, relative_att_ids = _resolve_pattern(pattern, att_mask, relative_att_ids, valid_len, q)
  B, S, N, D = q.shape
  dout = dout if dout.stride() == out.stride() else dout.contiguous()
  if out.stride() != dout.stride():
    out = out.contiguous()
  if grads_out is not None:
    dq, dk, dv = grads_out
  else:
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
  if dq.stride() != q.stride() or dk.stride() != k.stride() or dv.stride() != v.stride():
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
  desc = _make_desc(q, k, v, out, R, pattern, valid_len, scale, mask_value, scale_before_add,
                    dropout_p, dropout_seed, tuning)
  if rel_grads_accum is not None and R:
    drel_emb, drel_bias = rel_grads_accum
    for t, shape in ((drel_emb, (R, N, D)), (drel_bias, (R, N))):
      if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous()
                            or t.device != q.device):
        raise ValueError(f'rel_grads_accum buffers must be contiguous fp32 {shape} on {q.device}')
    if rel_bias is not None and drel_bias is None:
      raise ValueError('rel_grads_accum needs a dbias buffer when rel_bias is given')
    desc.flags |= _lib.MMT_FLAG_ACCUM_REL_GRADS
  else:
    drel_emb = torch.empty((R, N, D), dtype=torch.float32, device=q.device) if R else None
    drel_bias = torch.empty((R, N), dtype=torch.float32, device=q.device) if (R and rel_bias is not None) else None
  L = _lib.lib()
  ws_bytes = L.mmt_workspace_bytes(desc)
  ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=q.device)
  ptr = lambda t: None if t is None else t.data_ptr()
  with torch.cuda.device(q.device):
    _lib.check(L.mmt_attn_bwd(desc, ptr(q), ptr(k), ptr(v), ptr(rel_emb), ptr(rel_bias),
                              ptr(att_mask), ptr(relative_att_ids), ptr(out), ptr(dout), ptr(lse),
                              ptr(dq), ptr(dk), ptr(dv), ptr(drel_emb), ptr(drel_bias), ptr(ws),
                              ws.numel(), _stream_ptr(q.device)))
  return dq, dk, dv, drel_emb, drel_bias


class _RelativeAttentionFn(torch.autograd.Function):

  @staticmethod
  def forward(ctx, q, k, v, rel_emb, rel_bias, kw):
    out, lse = relative_attention_forward(q, k, v, rel_emb, rel_bias, **kw)
    ctx.save_for_backward(q, k, v, rel_emb, rel_bias, out, lse)
    ctx.kw = kw
    return out

  @staticmethod
  def backward(ctx, dout):
    q, k, v, rel_emb, rel_bias, out, lse = ctx.saved_tensors
    kw = {a: b for a, b in ctx.kw.items() if a != 'return_lse'}
    dq, dk, dv, de, db = relative_attention_backward(dout, q, k, v, rel_emb, rel_bias, out, lse, **kw)
    de = None if de is None else de.to(rel_emb.dtype)
    db = None if db is None else db.to(rel_bias.dtype)
    return dq, dk, dv, de, db, None


def relative_attention(q, k, v, rel_emb=None, rel_bias=None, **kw):
  """Differentiable QkvRelativeAttention (see module docstring); kwargs as
  `relative_attention_forward` (att_mask / relative_att_ids or pattern / valid_len, scale,
  mask_value, scale_before_add, dropout_p, dropout_seed)."""
  kw.pop('return_lse', None)
  return _RelativeAttentionFn.apply(q, k, v, rel_emb, rel_bias, kw)


class _RelativeAttentionQkvFn(torch.autograd.Function):
  """Same operator on the fused projection output qkv [B,S,3,N,D]: the backward writes dq, dk, dv
  into the three slices of ONE gradient buffer (no per-slice zero-fill / copy in autograd)."""

  @staticmethod
  def forward(ctx, qkv, rel_emb, rel_bias, kw, sinks):
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    out, lse = relative_attention_forward(q, k, v, rel_emb, rel_bias, **kw)
    ctx.save_for_backward(qkv, rel_emb, rel_bias, out, lse)
    ctx.kw, ctx.sinks = kw, sinks
    return out

  @staticmethod
  def backward(ctx, dout):
    qkv, rel_emb, rel_bias, out, lse = ctx.saved_tensors
    kw = {a: b for a, b in ctx.kw.items() if a != 'return_lse'}
    dqkv = torch.empty_like(qkv)
    accum = None
    if ctx.sinks is not None and rel_emb is not None:     # fp32 master tables: add straight into .grad
      for p in ctx.sinks:
        if p is not None and p.grad is None:
          p.grad = torch.zeros_like(p, dtype=torch.float32)
      accum = (ctx.sinks[0].grad, None if ctx.sinks[1] is None else ctx.sinks[1].grad)
    _, _, _, de, db = relative_attention_backward(
        dout, qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], rel_emb, rel_bias, out, lse,
        grads_out=(dqkv[:, :, 0], dqkv[:, :, 1], dqkv[:, :, 2]), rel_grads_accum=accum, **kw)
    if accum is not None:
      for p in ctx.sinks:
        for hook in getattr(p, '_mmt_grad_ready_hooks', ()) if p is not None else ():
          hook(p)
      return dqkv, None, None, None, None
    de = None if de is None else de.to(rel_emb.dtype)
    db = None if db is None else db.to(rel_bias.dtype)
    return dqkv, de, db, None, None


def relative_attention_qkv(qkv, rel_emb=None, rel_bias=None, rel_grad_sinks=None, **kw):
  """`relative_attention` on a fused, contiguous qkv [B,S,3,N,D] tensor.

  `rel_grad_sinks` = (emb_master, bias_master | None): fp32 master parameters whose low-precision
  copies are `rel_emb` / `rel_bias` (passed detached).  The backward then adds the fp32 table
  gradients straight into their `.grad` (no bf16 round trip, no separate accumulate kernels) and
  runs their `_mmt_grad_ready_hooks`."""
  if qkv.dim() != 5 or qkv.shape[2] != 3 or not qkv.is_contiguous():
    raise ValueError('qkv must be a contiguous [B,S,3,N,D] tensor')
  kw.pop('return_lse', None)
  if rel_grad_sinks is not None:
    for p in rel_grad_sinks:
      if p is not None and (p.dtype != torch.float32 or not p.is_contiguous()):
        raise ValueError('rel_grad_sinks must be contiguous fp32 parameters')
  return _RelativeAttentionQkvFn.apply(qkv, rel_emb, rel_bias, kw, rel_grad_sinks)
