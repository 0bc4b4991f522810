"""The train step recorded as a HIP graph (mmt_amd/graphed.py) and the device-resident step scalars behind it
(named by the descriptors since ABI 4: `dropout_epoch`, `mmt_adamw_desc.hyper`): same numbers as the eager step -- the kernels are deterministic and a dropout seed is
descriptor seed + epoch wherever the addition happens."""
import numpy as np
import pytest
import torch

from tests._cases import attention_inputs, bf16_round

pytestmark = pytest.mark.gpu


def test_dropout_epoch_on_the_device_equals_the_host_sum():
  """Attention forward + backward and a residual block with dropout: seed S and epoch E read by the kernels from device
  memory give bit for bit what seed S + E in the descriptor gives."""
  import mmt_amd
  from mmt_amd import fused, step_scalars
  B, S, N, R = 2, 512, 2, 32
  q, k, v, emb, bias = (torch.from_numpy(bf16_round(x)).cuda().bfloat16() for x in attention_inputs(B, S, N, R, seed=4))
  dout = torch.randn_like(q)
  pat = mmt_amd.AttentionPattern(local_radius=64, global_start=300, n_global=8, id_mode=1, max_dist=12)
  x = torch.randn(B * S, 768, device='cuda', dtype=torch.bfloat16)
  o = torch.randn_like(x)
  b = torch.randn(768, device='cuda')
  gam, bet = torch.randn(768, device='cuda'), torch.randn(768, device='cuda')
  seed, epoch = 0x1234_5678_9ABC, step_scalars.epoch_of(77)
  assert not step_scalars.device_active() and step_scalars.host_epoch() == 0      # nothing left behind by other tests

  def run(s):
    out, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, pattern=pat, dropout_p=0.2, dropout_seed=s)
    grads = mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, pattern=pat, dropout_p=0.2, dropout_seed=s)
    xn, hh = fused.residual_block(o, b, x, gam, bet, p=0.1, seed=s & ((1 << 63) - 1))
    return [out, lse, *grads, xn, hh]

  want = run((seed + epoch) & ((1 << 64) - 1))
  sc = step_scalars.DeviceStepScalars(q.device)
  sc.enable()
  try:
    sc.write(77, 1e-4, 1, 0.9, 0.999)
    got = run(seed)
  finally:
    sc.disable()
  plain = run(seed)
  for a, b2 in zip(want[:7], got[:7]):
    assert torch.equal(a, b2)
  assert not torch.equal(plain[0], got[0])          # (the epoch does change the mask)


def test_residual_block_epoch():
  from mmt_amd import fused, step_scalars
  x = torch.randn(1024, 768, device='cuda', dtype=torch.bfloat16)
  o = torch.randn_like(x)
  b = torch.randn(768, device='cuda')
  g, be = torch.randn(768, device='cuda'), torch.randn(768, device='cuda')
  seed, epoch = 0x7654_3210, step_scalars.epoch_of(5)
  want = fused.residual_block(o, b, x, g, be, p=0.1, seed=(seed + epoch) & ((1 << 63) - 1))
  sc = step_scalars.DeviceStepScalars(x.device)
  sc.enable()
  try:
    sc.write(5, 1e-4, 1, 0.9, 0.999)
    got = fused.residual_block(o, b, x, g, be, p=0.1, seed=seed)
  finally:
    sc.disable()
  if (seed + epoch) < (1 << 63):                       # (the host-side seed is a 63-bit value)
    for a, c in zip(want, got):
      assert torch.equal(a, c)


def _small_step(graph):
  import bench
  from mmt_amd import benchmarks
  cfg = dict(bench.config3(), S=256, P=14, B=2, g0=2 + 14 * 14, ng=8)
  return benchmarks.make_train_step_bench(cfg, torch.device('cuda:0'), 0, 1, dtype=torch.bfloat16, graph=graph)[0]


def test_graphed_train_step_matches_the_eager_one():
  """Ten steps each way on a small configuration (dropout 0.1 on): identical loss sequences, identical parameters."""
  res = {}
  for graph in (False, True):
    step = _small_step(graph)
    losses = [float(step()['loss']) for _ in range(10)]
    params = torch.cat([s['param'] for s in step.optimizer.slabs])
    res[graph] = (losses, params.clone())
    step.close()
  assert res[False][0] == res[True][0]
  assert torch.equal(res[False][1], res[True][1])
  assert len(set(res[True][0])) > 5                   # (the loss does move: the replays are not one frozen step)


@pytest.mark.parametrize('micro', [8, 4], ids=['one-micro-step', 'two-micro-steps'])
def test_trainer_loop_graphed_equals_eager(tmp_path, monkeypatch, micro):
  """`train.run_experiment` (new batch every step, metrics updated inside the step, checkpoint at the end): the logs of
  the graphed loop equal the eager loop's, and both checkpoints carry the same step count and parameters -- also with
  the reference's micro-batch accumulation inside the recorded step (two micro-steps, the reducer armed for the last)."""
  from tests.test_gpu_encoder import tiny_experiment
  from mmt_amd import checkpoint, train
  logs, ckpts = {}, {}
  for mode in ('0', '1'):
    monkeypatch.setenv('MMT_STEP_GRAPH', mode)
    exp = tiny_experiment(S=256, radius=32, n_global=8)
    exp.override({'task': {'model': {'encoder': {'mmt': {'hidden_dropout_prob': 0.1, 'attention_probs_dropout_prob': 0.1}}},
                           'train_data': {'global_batch_size': 8}, 'micro_batch_size': micro},
                  'runtime': {'mixed_precision_dtype': 'bfloat16'},
                  'trainer': {'train_steps': 9, 'checkpoint_interval': 0}})
    d = tmp_path / mode
    d.mkdir()
    model, lg = train.run_experiment(exp, 'train', str(d), log_every=1)
    assert train.run_experiment.last_step_launch == ('graph' if mode == '1' else 'eager')
    logs[mode] = [{k: v for k, v in e.items() if k != 'elapsed_s'} for e in lg]
    ckpts[mode] = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()]).cpu()
    assert checkpoint.latest_checkpoint(str(d))
  assert logs['0'] == logs['1']
  assert torch.equal(ckpts['0'], ckpts['1'])
  assert len({e['loss'] for e in logs['1']}) > 4


def test_a_step_that_cannot_be_recorded_stays_eager(monkeypatch):
  """If the recording raises, the step object warns once, drops its device scalars and keeps running eager steps."""
  from mmt_amd import step_scalars
  step = _small_step(True)
  for _ in range(3):
    step()
  gs = [c.cell_contents for c in step.__closure__ if hasattr(c.cell_contents, '_record')][0]
  task = gs.task
  real = task.train_step
  calls = {'n': 0}

  def flaky(*a, **k):
    calls['n'] += 1
    if calls['n'] == 1:
      raise RuntimeError('not capturable')
    return real(*a, **k)
  monkeypatch.setattr(task, 'train_step', flaky)
  with pytest.warns(UserWarning, match='not recorded as a HIP graph'):
    l4 = float(step()['loss'])
  assert gs.graph is None and not step_scalars.device_active()
  l5 = float(step()['loss'])
  assert np.isfinite(l4) and np.isfinite(l5) and gs.optimizer.t == 5
  step.close()


def _twin_steps(B=4, micro=2):
  """Two identically initialised (task, model, optimizer) sets behind GraphedTrainStep objects -- the second one never
  records (all eager) -- plus the resident batch."""
  import bench
  from mmt_amd import benchmarks
  cfg = dict(bench.config3(), S=256, P=14, B=B, g0=2 + 14 * 14, ng=8)
  out = []
  for _ in range(2):
    step = benchmarks.make_train_step_bench(cfg, torch.device('cuda:0'), 0, 1, dtype=torch.bfloat16, graph=True)[0]
    cells = [c.cell_contents for c in step.__closure__]
    gs = [c for c in cells if hasattr(c, '_record')][0]
    batch = [c for c in cells if isinstance(c, tuple) and len(c) == 2 and isinstance(c[0], dict)][0]
    gs.task.task_config.micro_batch_size = micro
    gs.static_inputs = False
    out.append((gs, batch))
  out[1][0].eager_left = 1 << 62
  return out


def test_a_batch_that_does_not_fit_the_graph_runs_eagerly():
  """A short batch in the middle of a graphed run (the last batch of an epoch): that step runs eagerly with the step's
  own dropout masks and learning rate, the replays before and after are untouched -- losses and parameters equal the
  all-eager twin's.  (Copied into the recorded buffers it would broadcast and train on duplicated rows.)"""
  (gs, batch), (eg, batch2) = _twin_steps()
  short = lambda b: tuple({k: (v[:2] if torch.is_tensor(v) else v) for k, v in tree.items()} for tree in b)
  from mmt_amd import step_scalars
  losses = {0: [], 1: []}
  for step in range(1, 10):          # the two models take turns in ONE process: neither may see the other's step scalars
    for i, (st, bt) in enumerate(((gs, batch), (eg, batch2))):
      b = short(bt) if step == 6 else bt
      if i == 0 and step == 6:
        with pytest.warns(UserWarning, match='does not match the recorded HIP graph'):
          losses[i].append(float(st(b, step)['loss']))
      else:
        losses[i].append(float(st(b, step)['loss']))
      assert not step_scalars.device_active()       # device-resident scalars are named only while a step is recorded
  assert gs.graph is not None and eg.graph is None
  assert losses[0] == losses[1]
  assert torch.equal(torch.cat([s['param'] for s in gs.optimizer.slabs]), torch.cat([s['param'] for s in eg.optimizer.slabs]))
  assert gs.optimizer.param_groups[0]['lr'] == eg.optimizer.param_groups[0]['lr']      # the host's copy follows the schedule
  # a batch with a key missing, or another constant, is refused the same way
  assert not gs._same_signature(({k: v for k, v in batch[0].items() if k != 'word_ids'}, batch[1]))
  gs.close(); eg.close()


def test_a_capture_that_fails_inside_backward_leaves_no_queued_work(monkeypatch):
  """The recording raises in the middle of backward (after weight-gradient products of the upper layers were queued
  and the reducer had counted gradients): the step object warns, resets the host-side queues and the reducer, and the
  eager retry and every later step equal the all-eager twin's."""
  from mmt_amd import fused, ops, step_scalars
  (gs, batch), (eg, batch2) = _twin_steps()
  real = ops.relative_attention_backward
  seen = {'queued': 0, 'ready': 0, 'calls': 0, 'raised': 0}

  def flaky(*a, **k):
    if torch.cuda.is_current_stream_capturing() and not seen['raised']:
      seen['calls'] += 1
      if seen['calls'] == 2:            # the second attention backward of the pass: layers above it have run theirs
        seen['raised'] = 1
        seen['queued'] = sum(len(e[1]) for e in fused._wg_deferred.values())
        seen['ready'] = len(gs.reducer._ready) + sum(1 for p in gs.model.parameters() if getattr(p, '_mmt_grad_deferred', False))
        raise RuntimeError('not capturable (test)')
    return real(*a, **k)
  monkeypatch.setattr(ops, 'relative_attention_backward', flaky)
  la, lb = [], []
  for step in range(1, 8):
    if step == 4:
      with pytest.warns(UserWarning, match='not recorded as a HIP graph'):
        la.append(float(gs(batch, step)['loss']))
    else:
      la.append(float(gs(batch, step)['loss']))
    lb.append(float(eg(batch2, step)['loss']))
  assert seen['raised'] == 1 and seen['queued'] + seen['ready'] > 0      # the failure did strand host-side state of the abandoned pass
  assert gs.graph is None and not step_scalars.device_active()
  assert not fused._wg_deferred
  assert la == lb
  assert torch.equal(torch.cat([s['param'] for s in gs.optimizer.slabs]), torch.cat([s['param'] for s in eg.optimizer.slabs]))
  gs.close(); eg.close()


def test_classification_trainer_loop_graphed_equals_eager(tmp_path, monkeypatch):
  """The same comparison for `mmt/classification` (its own train_step surface: logits of one head, AUC / accuracy
  metrics updated inside the recorded step)."""
  from tests.test_gpu_encoder import tiny_experiment
  from mmt_amd import configs, train
  pre = tiny_experiment(S=256, radius=32, n_global=8)
  logs, params = {}, {}
  for mode in ('0', '1'):
    monkeypatch.setenv('MMT_STEP_GRAPH', mode)
    exp = configs.get_exp_config('mmt/classification')
    enc = pre.task.model.encoder.as_dict()
    enc['mmt'].update(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    exp.override({'task': {'model': {'encoder': enc, 'cls_heads': [{'inner_dim': 64, 'num_classes': 2, 'name': 'itm'}]},
                           'train_data': dict(pre.task.train_data.as_dict(), global_batch_size=8), 'micro_batch_size': 8},
                  'runtime': {'mixed_precision_dtype': 'bfloat16'},
                  'trainer': {'train_steps': 8, 'checkpoint_interval': 0}}, strict=False)
    d = tmp_path / mode
    d.mkdir()
    model, lg = train.run_experiment(exp, 'train', str(d), log_every=1)
    assert train.run_experiment.last_step_launch == ('graph' if mode == '1' else 'eager')
    logs[mode] = [{k: v for k, v in e.items() if k != 'elapsed_s'} for e in lg]
    params[mode] = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()]).cpu()
  assert logs['0'] == logs['1']
  assert torch.equal(params['0'], params['1'])
