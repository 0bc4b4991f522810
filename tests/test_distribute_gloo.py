"""N>1 path on CPU: world_size-2 gloo run of the bucketed gradient reducer (the same code RCCL
runs on the GPUs), checked against a single-process computation on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__  # noqa: F401


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


class _DirectScale(torch.autograd.Function):
  """y = x * w with the product kernels' gradient protocol: dw is added straight into w.grad, the
  parameter's gradient-ready hooks are called by hand and autograd gets None for it (its
  AccumulateGrad node may still fire the post-accumulate hook: the reducer must count w once)."""

  @staticmethod
  def forward(ctx, x, w):
    ctx.save_for_backward(x, w.detach())
    ctx.w = w
    return x * w.detach()

  @staticmethod
  def backward(ctx, g):
    x, wv = ctx.saved_tensors
    ctx.w.grad.add_((g * x).sum(0))
    for hook in getattr(ctx.w, '_mmt_grad_ready_hooks', ()):
      hook(ctx.w)
    return g * wv, None


def _worker(rank, world, port, reduce, out):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  from mmt_amd import distribute
  strategy = distribute.get_distribution_strategy('mirrored', num_gpus=0, bucket_mb=3000 / (1 << 20))
  assert strategy.num_replicas_in_sync == world and strategy.backend == 'gloo'
  torch.manual_seed(0)
  model = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8),
                              torch.nn.Linear(8, 1))
  unused = torch.nn.Parameter(torch.ones(3))           # receives no gradient
  direct = torch.nn.Parameter(torch.full((64,), 0.5))   # gradient written directly (see _DirectScale)
  # `direct` sits mid-network and shares a bucket with the first layer's bias, whose gradient arrives
  # later: counting `direct` twice would launch that bucket's all-reduce too early
  mp_ = list(model.parameters())
  params = [unused] + mp_[:2] + [direct] + mp_[2:]
  net = lambda t: model[3](model[2](_DirectScale.apply(model[1](model[0](t)), direct)))
  reducer = strategy.make_reducer(params, reduce=reduce)
  assert len(reducer.buckets) >= 2                      # several buckets -> overlap path exercised
  gi = reducer._bucket_of[direct]
  assert reducer._bucket_of[mp_[1]] == gi and reducer._group_sizes[gi] > 2
  g = torch.Generator().manual_seed(100 + rank)
  x = torch.randn(4, 16, generator=g)
  for _ in range(2):                                    # two steps: zero_grad / re-arm
    reducer.zero_grad()
    reducer.set_armed(False)                            # micro-step 1 of 2: accumulate locally only
    (net(x[:2]).pow(2).sum() / 4).backward()
    reducer.set_armed(True)                             # last micro-step launches the all-reduces
    (net(x[2:]).pow(2).sum() / 4).backward()
    # every parameter counted exactly once: only the bucket of the gradient-less parameter is still open
    assert sorted(reducer._pending) == [0] * (len(reducer.buckets) - 1) + [1], reducer._pending
    reducer.finish()
  out[rank] = [p.grad.clone() for p in params]
  dist.destroy_process_group()


@pytest.mark.parametrize('reduce', ['mean', 'sum'])
def test_bucketed_allreduce_world2(reduce):
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_worker, args=(world, _free_port(), reduce, out), nprocs=world, join=True)
  torch.manual_seed(0)
  model = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.Tanh(), torch.nn.Linear(64, 8),
                              torch.nn.Linear(8, 1))
  grads = None
  direct = torch.nn.Parameter(torch.full((64,), 0.5))
  for rank in range(world):
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(4, 16, generator=g)
    model.zero_grad(); direct.grad = None
    model[3](model[2](model[1](model[0](x)) * direct)).pow(2).mean().backward()
    mp_ = list(model.parameters())
    cur = [p.grad.clone() for p in mp_[:2] + [direct] + mp_[2:]]
    grads = cur if grads is None else [a + b for a, b in zip(grads, cur)]
  if reduce == 'mean':
    grads = [gr / world for gr in grads]
  for rank in range(world):
    got = out[rank]
    assert float(got[0].abs().max()) == 0.0             # unused parameter: zero gradient everywhere
    for a, b in zip(got[1:], grads):
      assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()
  for a, b in zip(out[0], out[1]):
    assert torch.equal(a, b)
