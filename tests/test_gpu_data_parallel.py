"""N>1 path on the GPU: two ranks share cuda:0 (gloo carries the collectives; RCCL needs one device
per rank) and run the REAL model -- fused kernels writing parameter gradients straight into the flat
fp32 buckets and firing the reducer's gradient-ready hooks themselves (weight-gradient GEMM, embedding
scatter, relative tables, LayerNorm/bias column sums).  Checked against each rank's own un-reduced
gradients: all_reduce(mean) must give (g0 + g1) / 2 on both ranks, bit-identical across ranks."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import __graft_entry__  # noqa: F401
from tests.test_gpu_encoder import tiny_experiment

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  import torch.distributed as dist
  import mmt_amd
  from mmt_amd import distribute
  dist.init_process_group('gloo')
  torch.cuda.set_device(0)
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.bfloat16, num_replicas=world)
  torch.manual_seed(5)                                   # identical replicas
  model = task.build_model().cuda()
  data = task.build_inputs(exp.task.train_data, device='cuda', rank=rank, batch_size=2)
  inputs, labels = next(data)

  def backward(reducer):
    loss = task.build_losses(labels, model(**inputs, training=False))
    loss.backward()
    if reducer is not None:
      reducer.finish()
    return float(loss)

  # pass 1: local gradients through a single-replica reducer (same direct-write code path)
  local = distribute.DataParallelStrategy(None).make_reducer(list(model.parameters()))
  local.zero_grad()
  backward(local)
  g_local = [b.clone().cpu() for b in local.buckets]
  # pass 2: the distributed reducer (small buckets: several all-reduces in flight during backward)
  strategy = distribute.DataParallelStrategy('gloo', bucket_mb=0.25)
  red = strategy.make_reducer(list(model.parameters()), reduce='mean')
  assert red.world == world and len(red.buckets) > 2
  red.zero_grad()
  backward(red)
  flat = torch.cat([b.reshape(-1) for b in red.buckets]).cpu()
  out[rank] = (torch.cat([g.reshape(-1) for g in g_local]), flat, [tuple(b.shape) for b in red.buckets])
  dist.destroy_process_group()


def test_two_ranks_on_one_gpu_average_their_bucket_gradients():
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  (l0, r0, _), (l1, r1, _) = out[0], out[1]
  assert torch.equal(r0, r1)                              # every replica holds the same reduced gradient
  # every parameter owns whole 1024-element chunks in reverse parameter order whatever the bucket size,
  # so the two reducers' concatenated buckets line up element for element
  assert l0.numel() == r0.numel() and float(r0.abs().max()) > 0
  assert torch.allclose(r0.double(), (l0.double() + l1.double()) / 2, atol=1e-6, rtol=1e-5)


def _step_worker(rank, world, port, out, scale_loss=False):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  import torch.distributed as dist
  import mmt_amd
  from mmt_amd import distribute, optimization
  dist.init_process_group('gloo')
  torch.cuda.set_device(0)
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  exp.task.model.encoder.mmt.hidden_dropout_prob = 0.0
  exp.task.model.encoder.mmt.attention_probs_dropout_prob = 0.0

  def make(num_replicas, strategy):
    # scale_loss=True (pretraining.py:286-296) on the data-parallel side only: gradient of loss / replicas,
    # SUM over replicas -- the single-process reference below keeps the plain micro-batch mean
    exp.task.scale_loss = bool(scale_loss and num_replicas > 1)
    task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.bfloat16, num_replicas=num_replicas)
    torch.manual_seed(6)
    model = task.build_model().cuda()
    reduce = mmt_amd.tasks.gradient_reduce_mode(exp.task)
    assert reduce == ('sum' if exp.task.scale_loss else 'mean')
    reducer = strategy.make_reducer(list(model.parameters()), reduce=reduce)
    opt = optimization.create_optimizer(model, exp.trainer.optimizer_config, reducer=reducer)
    optimization.set_learning_rate(opt, 1e-3)
    return task, model, reducer, opt

  batches = [next(mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.bfloat16).build_inputs(
      exp.task.train_data, device='cuda', rank=r, batch_size=2)) for r in range(world)]
  # data-parallel step: this rank's half, mean folded into the fused optimizer's gradient scale
  task, model, reducer, opt = make(world, distribute.DataParallelStrategy('gloo', bucket_mb=0.25))
  task.train_step(batches[rank], model, opt, reducer=reducer, clip_norm=1.0, micro_batch_size=2)
  # reference: one process, both halves as two micro-batches
  cat = lambda vs: torch.cat(vs, 0) if torch.is_tensor(vs[0]) else vs[0]
  both = tuple({k: cat([b[i][k] for b in batches]) for k in batches[0][i]} for i in (0, 1))
  task1, model1, reducer1, opt1 = make(1, distribute.DataParallelStrategy(None))
  task1.train_step(both, model1, opt1, reducer=reducer1, clip_norm=1.0, micro_batch_size=2)
  worst = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(model.parameters(), model1.parameters()))
  moved = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(model1.parameters(), make(1, distribute.DataParallelStrategy(None))[1].parameters()))
  out[rank] = (worst, moved, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
  dist.destroy_process_group()


@pytest.mark.parametrize('scale_loss', [False, True], ids=['mean', 'scale_loss-sum'])
def test_data_parallel_train_step_equals_micro_batched_single_process(scale_loss):
  """Two ranks x 2 samples with the deferred 1/world mean (or scale_loss + SUM) == one process x 2
  micro-batches of 2."""
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_step_worker, args=(world, _free_port(), out, scale_loss), nprocs=world, join=True)
  (w0, m0, p0), (w1, m1, p1) = out[0], out[1]
  assert torch.equal(p0, p1)                      # replicas stay in sync
  assert m0 > 1e-4                                 # the step moved the parameters ...
  assert max(w0, w1) < 1e-6 + 1e-3 * m0            # ... to the same place as the single-process reference


def _dropout_worker(rank, world, port, out):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  import torch.distributed as dist
  import mmt_amd
  dist.init_process_group('gloo')
  torch.cuda.set_device(0)
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  exp.task.model.encoder.mmt.hidden_dropout_prob = 0.5
  exp.task.model.encoder.mmt.attention_probs_dropout_prob = 0.5
  task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.bfloat16, num_replicas=world)
  torch.manual_seed(5)
  model = task.build_model().cuda()
  inputs, _ = next(task.build_inputs(exp.task.train_data, device='cuda', rank=0, batch_size=2))   # SAME data on both ranks
  enc = {k: v for k, v in inputs.items() if k not in ('mlm_positions', 'mpp_positions')}
  seqs = []
  for step in (1, 1, 2):
    mmt_amd.fused.set_seed_stream(step, 0, rank)
    with torch.no_grad():
      seqs.append(model.encoder(**enc, training=True)['sequence_output'].float().cpu())
  out[rank] = seqs
  dist.destroy_process_group()


def test_replicas_draw_different_dropout_masks():
  """Same weights, same inputs, same step: the ranks' dropout masks differ (seeded by rank), a rank
  repeats its own masks for the same step and changes them with the step."""
  world = 2
  mgr = mp.Manager()
  out = mgr.dict()
  mp.spawn(_dropout_worker, args=(world, _free_port(), out), nprocs=world, join=True)
  a, b = out[0], out[1]
  assert torch.equal(a[0], a[1]) and torch.equal(b[0], b[1])
  assert not torch.equal(a[0], a[2])
  assert not torch.equal(a[0], b[0])


def test_rccl_path_executes_with_one_rank():
  """`MMT_FORCE_DIST=1 python bench.py`: ONE rank, but through `init_process_group('nccl')`, the multi-rank reducer
  (ready hooks, one async RCCL all-reduce per 48 MB bucket under backward, waits), the main-stream weight gradients
  and the reduced CU budget -- the RCCL code path of BASELINE config 4 (`distribute_utils.py:97-188`,
  `pretraining.py:273`) executed on real hardware, which the one-GPU box cannot do with more ranks.  The all-reduce
  of ones must see exactly this rank, and the step must produce a finite throughput."""
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MMT_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
  env.pop('MMT_DIST_BACKEND', None)
  out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '3', '--warmup', '2', '--no-cpu-baseline'],
                       env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  line = json.loads(out.stdout.strip().splitlines()[-1])
  assert line['config']['backend'] == 'nccl' and line['config']['ranks_seen'] == 1
  assert 'RCCL' in line['config']['grad_allreduce']
  assert line['value'] > 0 and line['ms_per_step'] > 0
  ar = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--mode', 'allreduce', '--steps', '3', '--warmup', '1'],
                      env=env, capture_output=True, text=True, timeout=600)
  assert ar.returncode == 0, ar.stderr[-2000:]           # asserts inside that every bucket holds the SUM over the ranks
  assert json.loads(ar.stdout.strip().splitlines()[-1])['config']['backend'] == 'nccl'
