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


# ---- scale_loss=True (pretraining.py:286-296): each replica differentiates loss / num_replicas and the
# ---- optimizer SUMs the replicas -> same update as one process running both shards as micro-batches
class _TinyTaskConfig:
  def __init__(self, scale_loss):
    self.scale_loss = scale_loss


class _TinyModel(torch.nn.Module):
  def __init__(self):
    super().__init__()
    self.net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))

  def forward(self, x, training=False):
    return {'y': self.net(x)}


def _tiny_task(scale_loss, num_replicas):
  from mmt_amd import tasks

  class T(tasks._TaskBase):
    def build_losses(self, labels, outputs, metrics=None):
      return (outputs['y'][:, 0] - labels['t']).pow(2).mean()
  return T(_TinyTaskConfig(scale_loss), num_replicas=num_replicas)


def _tiny_batch(rank):
  g = torch.Generator().manual_seed(7 + rank)
  return {'word_ids': torch.zeros(4, 1), 'x': torch.randn(4, 6, generator=g)}, {'t': torch.randn(4, generator=g)}


class _DropWordIds(torch.nn.Module):     # train_step reads the batch size off inputs['word_ids']
  def __init__(self, m):
    super().__init__(); self.m = m

  def forward(self, word_ids, x, training=False):
    return self.m(x, training)


def _scale_loss_worker(rank, world, port, out):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  from mmt_amd import distribute, tasks
  strategy = distribute.get_distribution_strategy('mirrored', num_gpus=0)
  task = _tiny_task(True, world)
  assert tasks.gradient_reduce_mode(task.task_config) == 'sum'
  torch.manual_seed(3)
  model = _DropWordIds(_TinyModel())
  reducer = strategy.make_reducer(list(model.parameters()), reduce=tasks.gradient_reduce_mode(task.task_config))
  opt = torch.optim.SGD(model.parameters(), lr=0.5)
  task.train_step(_tiny_batch(rank), model, opt, reducer=reducer, micro_batch_size=4, step=1)
  out[rank] = [p.detach().clone() for p in model.parameters()]
  dist.destroy_process_group()


def test_scale_loss_world2_equals_micro_batched_single_process():
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_scale_loss_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  from mmt_amd import distribute, tasks
  task = _tiny_task(False, 1)
  torch.manual_seed(3)
  model = _DropWordIds(_TinyModel())
  init = [p.detach().clone() for p in model.parameters()]
  reducer = distribute.DataParallelStrategy(None).make_reducer(list(model.parameters()))
  opt = torch.optim.SGD(model.parameters(), lr=0.5)
  (i0, l0), (i1, l1) = _tiny_batch(0), _tiny_batch(1)
  both = ({k: torch.cat([i0[k], i1[k]]) for k in i0}, {k: torch.cat([l0[k], l1[k]]) for k in l0})
  task.train_step(both, model, opt, reducer=reducer, micro_batch_size=4, step=1)
  want = [p.detach() for p in model.parameters()]
  assert max(float((a - b).abs().max()) for a, b in zip(want, init)) > 1e-3      # the step moved them
  for rank in range(world):
    for a, b in zip(out[rank], want):
      assert torch.allclose(a, b, atol=1e-6), float((a - b).abs().max())


def test_bench_launches_its_own_ranks():
  """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (children; the parent never
  touches a GPU), runs the gradient exchange over gloo here, and relays ONE JSON line."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
  env['MMT_DIST_BACKEND'] = 'gloo'
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--mode', 'allreduce',
                      '--allreduce-mb', '4', '--steps', '3', '--warmup', '1'],
                     env=env, capture_output=True, text=True, timeout=240)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['config']['ranks_seen'] == 2 and d['config']['backend'] == 'gloo'
  assert d['config']['parallelism'] == 'dp2' and d['steps'] == 3 and d['value'] > 0
  # asking for more ranks than were launched is an error, not a silent 1-rank run
  r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--mode', 'allreduce'],
                     env=dict(env, WORLD_SIZE='1', RANK='0'), capture_output=True, text=True, timeout=120)
  assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)
