"""The train step recorded as a HIP graph (mmt_amd/graphed.py) and the device-resident step scalars behind it
(mmt_set_step_scalars): same numbers as the eager step -- the kernels are deterministic and a dropout seed is
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
