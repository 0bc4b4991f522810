"""Data-parallel training over RCCL/xGMI: one process per GPU, gradients all-reduced in
buckets that overlap the backward pass.

Replaces `src/distribute_utils.py:97-188` (tf.distribute strategy factory; the gradient
all-reduce is implicit in `optimizer.apply_gradients`, `src/tasks/pretraining.py:273`).
The reference SUMs per-replica gradients (and divides the loss by `num_replicas_in_sync`
only with `scale_loss=True`, pretraining.py:286-296); `reduce='mean'` is this build's default
and `reduce='sum'` reproduces the reference default -- SURVEY.md 8(e).
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

_VALID = ('off', 'one_device', 'mirrored', 'multi_worker_mirrored', 'tpu', 'parameter_server')


class DataParallelStrategy:
  """What `get_distribution_strategy` returns: rank/world bookkeeping + the gradient reducer."""

  def __init__(self, backend: Optional[str], bucket_mb: float = 48.0):
    self.backend = backend
    self.bucket_bytes = int(bucket_mb * (1 << 20))
    self.rank = dist.get_rank() if backend else 0
    self.num_replicas_in_sync = dist.get_world_size() if backend else 1

  def make_reducer(self, params: List[torch.nn.Parameter], reduce: str = 'mean'):
    return GradientBucketReducer(params, self, reduce)


def get_distribution_strategy(distribution_strategy='mirrored', num_gpus=0, all_reduce_alg=None,
                              num_packs=1, tpu_address=None, zone=None, bucket_mb=48.0, **kwargs):
  """Same argument checks as `distribute_utils.get_distribution_strategy` (:136-188); every
  multi-device strategy maps to one-process-per-GPU data parallelism over RCCL (`nccl` backend
  on ROCm) or gloo on CPU."""
  del kwargs, num_packs, tpu_address, zone
  if num_gpus < 0:
    raise ValueError('`num_gpus` can not be negative.')
  if not isinstance(distribution_strategy, str):
    msg = 'distribution_strategy must be a string but got: %s.' % (distribution_strategy,)
    if distribution_strategy == False:  # noqa: E712  (yaml `off` -> False, as in the reference)
      msg += (" If you meant to pass the string 'off', make sure you add quotes around 'off' "
              'so that yaml interprets it as a string instead of a bool.')
    raise ValueError(msg)
  name = distribution_strategy.lower()
  if name not in _VALID:
    raise ValueError('Unrecognized Distribution Strategy: %r' % distribution_strategy)
  if name in ('off', 'one_device'):
    if num_gpus > 1:
      raise ValueError('When {} GPUs are specified, distribution_strategy flag cannot be set '
                       'to `off`.'.format(num_gpus) if name == 'off' else
                       '`OneDeviceStrategy` can not be used for more than one device.')
    return DataParallelStrategy(None, bucket_mb)
  if name in ('tpu', 'parameter_server'):
    raise ValueError(f'{name} strategy is not available on MI355X; use `mirrored`.')
  if all_reduce_alg not in (None, 'nccl', 'ring', 'hierarchical_copy'):
    raise ValueError(f'unknown all_reduce_alg {all_reduce_alg!r}')
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world == 1:
    return DataParallelStrategy(None, bucket_mb)
  if not dist.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    kw = {}
    if backend == 'nccl':
      kw['device_id'] = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend, **kw)
  return DataParallelStrategy(dist.get_backend(), bucket_mb)


class GradientBucketReducer:
  """Flat fp32 gradient buckets, filled in reverse parameter order; each bucket's all-reduce is
  launched (async, on RCCL's stream) as soon as its last gradient has been accumulated, so the
  exchange overlaps the rest of backward.  `param.grad` are views into the buckets: no copies."""

  def __init__(self, params, strategy: DataParallelStrategy, reduce: str = 'mean'):
    if reduce not in ('mean', 'sum'):
      raise ValueError("reduce must be 'mean' or 'sum'")
    self.strategy, self.reduce = strategy, reduce
    self.world = strategy.num_replicas_in_sync
    # MMT_FORCE_DIST=1: run the multi-rank machinery (ready hooks, bucketed async all-reduce, main-stream weight
    # gradients, CU budget) with ONE rank -- the only way to execute the RCCL path on a one-GPU box
    self.exchange = self.world > 1 or (strategy.backend is not None and os.environ.get('MMT_FORCE_DIST') == '1')
    self.params = [p for p in params if p.requires_grad]
    self.buckets: List[torch.Tensor] = []
    self.layout = []            # (param, bucket index, element offset) for flat optimizers
    self.buckets_are_zero = True
    self.pending_scale = 1.0
    self._bucket_of, self._pending, self._handles = {}, [], []
    self._ready = set()         # parameters already counted in this (armed) backward
    order = list(reversed(self.params))
    cur, cur_bytes = [], 0
    groups = []
    for p in order:
      nbytes = p.numel() * 4
      if cur and cur_bytes + nbytes > strategy.bucket_bytes:
        groups.append(cur); cur, cur_bytes = [], 0
      cur.append(p); cur_bytes += nbytes
    if cur:
      groups.append(cur)
    pad4 = lambda n: (n + 1023) & ~1023   # every parameter owns whole 1024-element chunks (fused AdamW)
    for gi, group in enumerate(groups):
      flat = torch.zeros(sum(pad4(p.numel()) for p in group), dtype=torch.float32, device=group[0].device)
      off = 0
      for p in group:
        p.grad = flat[off:off + p.numel()].view_as(p)
        self.layout.append((p, gi, off))
        off += pad4(p.numel())
        self._bucket_of[p] = gi
      self.buckets.append(flat)
      self._pending.append(len(group))
    self._group_sizes = list(self._pending)
    self.armed = True
    if self.exchange:
      if self.params and self.params[0].is_cuda:
        # leave compute units to the RCCL kernels that run under backward (csrc/wgrad_gemm.hip)
        from . import _lib
        _lib.lib().mmt_wgrad_set_cu_budget(int(os.environ.get('MMT_WGRAD_CUS', '224')))
        _lib.lib().mmt_ffn_set_cu_budget(int(os.environ.get('MMT_FFN_CUS', '224')))
      for p in self.params:
        p.register_post_accumulate_grad_hook(self._on_grad_ready)
        # Kernels that add a gradient straight into `.grad` (weight-gradient GEMM, fused-layer column
        # sums, relative tables, `_CastParamFn`) return None to autograd and call these hooks themselves
        # once the write is enqueued.  Depending on the torch version the parameter's AccumulateGrad
        # node still runs its post-accumulate hook for such an undefined gradient (it does on 2.10), so
        # `_on_grad_ready` counts every parameter at most once per backward.
        p._mmt_grad_ready_hooks = (self._on_grad_ready_from_kernel,)

  def zero_grad(self):
    if not self.buckets_are_zero:      # a fused optimizer step may already have cleared them
      for b in self.buckets:
        b.zero_()
    self.buckets_are_zero = False
    self._pending = list(self._group_sizes)
    self._handles = []
    self._ready = set()
    self.armed = True
    # a backward that raised may have left parameters marked "product still queued": start every step clean
    for p in self._bucket_of:
      p._mmt_grad_deferred = False

  def set_armed(self, armed: bool):
    """With gradient accumulation over micro-batches only the LAST backward may launch the
    bucket all-reduces; earlier ones just accumulate locally."""
    self.armed = bool(armed)

  def _on_grad_ready_from_kernel(self, p):
    """A kernel that wrote p.grad itself reports that the write is enqueued."""
    p._mmt_grad_deferred = False
    self._on_grad_ready(p)

  def _on_grad_ready(self, p):
    # a parameter whose gradient product is still queued (fused.wgrad_accumulate_deferred_: launched together with
    # its encoder block's other weight gradients) is NOT ready when autograd fires the post-accumulate hook for the
    # None its backward returned -- only when the kernel-side notification arrives
    if not self.armed or id(p) in self._ready or getattr(p, '_mmt_grad_deferred', False):
      return
    self._ready.add(id(p))
    gi = self._bucket_of[p]
    self._pending[gi] -= 1
    if self._pending[gi] == 0:
      self._launch(gi)

  def _launch(self, gi):
    # (every gradient kernel runs on the current stream, and so is the collective ordered: nothing to wait for)
    self._handles.append(dist.all_reduce(self.buckets[gi], op=dist.ReduceOp.SUM, async_op=True))

  def finish(self, defer_mean: bool = False):
    """Waits for the outstanding all-reduces and applies the mean (if requested).  With `defer_mean`
    the buckets keep the SUM over replicas and the 1/world factor is remembered instead
    (`pending_scale`): `clip_by_global_norm(apply=False)` folds it into the factor a fused optimizer
    multiplies the gradients with, which saves one read-modify-write pass over every gradient."""
    self.pending_scale = 1.0
    if self.exchange:
      for gi, left in enumerate(self._pending):     # parameters that received no gradient
        assert left >= 0, 'gradient-ready accounting went negative'
        if left > 0:
          self._launch(gi)
      for h in self._handles:
        h.wait()
      if self.reduce == 'mean':
        if defer_mean:
          self.pending_scale = 1.0 / self.world
        else:
          for b in self.buckets:
            b.mul_(1.0 / self.world)
    self._handles = []
    self._ready = set()

  def clip_by_global_norm(self, max_norm: float, apply: bool = True) -> torch.Tensor:
    """Returns the clip factor min(1, max_norm / ||g||) as a device scalar; with apply=False the
    gradients are left alone (a fused optimizer multiplies them while it reads them)."""
    ps = getattr(self, 'pending_scale', 1.0)                  # buckets hold ps^-1 x the true gradient
    if self.buckets[0].is_cuda and len(self.buckets) <= 16:
      # one streaming HIP pass over the slabs + a fixed-order sum; the clip factor is written on the device
      import ctypes
      from . import _lib
      dev = self.buckets[0].device
      if getattr(self, '_clip_ws', None) is None:
        self._clip_ws = torch.empty(2048 + 2, dtype=torch.float32, device=dev)
        self._clip_ptrs = (ctypes.c_void_p * len(self.buckets))(*[b.data_ptr() for b in self.buckets])
        self._clip_sizes = (ctypes.c_int64 * len(self.buckets))(*[b.numel() for b in self.buckets])
      scale = self._clip_ws[2048:2049].view(())
      with torch.cuda.device(dev):
        _lib.check(_lib.lib().mmt_grad_clip_scale(
            len(self.buckets), self._clip_ptrs, self._clip_sizes, float(max_norm), float(ps), scale.data_ptr(),
            self._clip_ws[2049:].data_ptr(), self._clip_ws.data_ptr(), 2048 * 4, torch.cuda.current_stream(dev).cuda_stream))
    else:
      norms = torch._foreach_norm(self.buckets)               # one pass, no temporaries
      total = torch.linalg.vector_norm(torch.stack(norms)) * ps
      scale = torch.clamp(max_norm / (total + 1e-6), max=1.0).float() * ps
    if apply:
      torch._foreach_mul_(self.buckets, scale)
      self.pending_scale = 1.0
    return scale

  def pending_scale_tensor(self):
    """The deferred mean factor as a device scalar (for a fused optimizer when no clipping is asked)."""
    ps = getattr(self, 'pending_scale', 1.0)
    return None if ps == 1.0 else torch.full((), ps, dtype=torch.float32, device=self.buckets[0].device)
