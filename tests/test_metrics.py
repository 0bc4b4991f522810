"""Metrics of the train step (mmt_amd/metrics.py, tasks.build_metrics / process_metrics) against the numpy
restatement of the Keras 2.5 update rules (oracle/metrics.py); the SUM over replicas with 2 gloo ranks; on the GPU
(-m gpu) the arg-max the loss kernel reports and the metrics of a real train step."""
import os
import socket

import numpy as np
import pytest
import torch

import __graft_entry__  # noqa: F401
from oracle import metrics as om


def _run_metric_checks(device):
  from mmt_amd import metrics as M
  rng = np.random.default_rng(0)
  # ---- Mean: unweighted, weighted with broadcasting, scalar updates (the loss metrics), reset
  m = M.Mean('x_loss')
  st = np.zeros(2, np.float32)
  for shape, wshape in (((7, 3), None), ((7, 3), (7, 1)), ((), None), ((5,), (5,))):
    v = rng.standard_normal(shape).astype(np.float32)
    w = None if wshape is None else rng.random(wshape).astype(np.float32)
    m.update_state(torch.from_numpy(np.asarray(v)).to(device), None if w is None else torch.from_numpy(w).to(device))
    st = om.mean_update(st, v, w)
  assert abs(float(m.result()) - float(om.divide_no_nan(st[0], st[1]))) < 1e-5
  m.reset_state()
  assert float(m.result()) == 0.0                                   # divide_no_nan(0, 0)
  # ---- SparseCategoricalAccuracy: ties go to the first index, weights of 0 drop rows, [B, L] shaped inputs
  acc = M.SparseCategoricalAccuracy('mlm_accuracy')
  st = np.zeros(2, np.float32)
  for _ in range(3):
    logits = rng.integers(-3, 4, (4, 6, 11)).astype(np.float32)      # small integers: many ties
    y = rng.integers(0, 11, (4, 6))
    w = (rng.random((4, 6)) > 0.3).astype(np.float32) * rng.random((4, 6)).astype(np.float32)
    acc.update_state(torch.from_numpy(y).to(device), torch.from_numpy(logits).to(device), torch.from_numpy(w).to(device))
    st = om.sparse_categorical_accuracy_update(st, y, logits, w)
  assert abs(float(acc.result()) - float(om.divide_no_nan(st[0], st[1]))) < 1e-6
  pre = M.SparseCategoricalAccuracy('a')                             # a precomputed arg-max is used as given
  pre.update_state(torch.tensor([1, 2, 3], device=device), None, torch.tensor([1.0, 1.0, 2.0], device=device),
                   argmax=torch.tensor([1, 0, 3], device=device, dtype=torch.int32))
  assert abs(float(pre.result()) - 0.75) < 1e-7
  # ---- AUC(PR): weighted, several updates; degenerate inputs give 0 through divide_no_nan
  auc = M.AUC('auc', curve='PR')
  st = np.zeros((4, 200), np.float32)
  for n in (50, 33):
    y = (rng.random(n) > 0.6).astype(np.int32)
    p = np.clip(0.35 * y + rng.random(n) * 0.7, 0, 1).astype(np.float32)
    w = rng.random(n).astype(np.float32) + 0.5
    auc.update_state(torch.from_numpy(y).to(device), torch.from_numpy(p).to(device), torch.from_numpy(w).to(device))
    st = om.auc_update(st, y, p, w)
  got, want = float(auc.result()), float(om.auc_pr_result(st))
  assert 0.5 < want < 1.0 and abs(got - want) < 2e-5, (got, want)
  assert float(M.AUC('auc').result()) == 0.0
  with pytest.raises(ValueError):
    M.AUC('auc', curve='XX')


def test_metrics_match_the_keras_restatement_cpu():
  _run_metric_checks('cpu')


@pytest.mark.gpu
def test_metrics_match_the_keras_restatement_gpu():
  _run_metric_checks('cuda')


def test_task_metric_sets_and_process_metrics_cpu():
  """`build_metrics` names (pretraining.py:183-196, classification.py:132-148) and `process_metrics` with the ITM
  masking of the MLM / MPP weights (pretraining.py:198-222), on CPU tensors."""
  from mmt_amd import configs, tasks
  from mmt_amd import metrics as M
  cfg = configs.get_exp_config('mmt/pretraining').task
  cfg.model.cls_heads = [configs.ClsHeadConfig(inner_dim=8, num_classes=2, name='itm')]
  task = tasks.get_task(cfg)
  ms = task.build_metrics()
  assert [m.name for m in ms] == ['mlm_accuracy', 'mlm_loss', 'mpp_accuracy', 'mpp_loss', 'itm_accuracy', 'itm_loss']
  rng = np.random.default_rng(3)
  B, L, P, V, C = 4, 5, 3, 13, 8
  labels = {'mlm_label_ids': rng.integers(0, V, (B, L)), 'mlm_label_weights': rng.random((B, L)).astype(np.float32),
            'mpp_label_ids': rng.integers(0, C, (B, P)), 'mpp_label_weights': rng.random((B, P)).astype(np.float32),
            'itm_label_ids': np.array([1, 0, 1, 0]), 'itm_label_weights': np.ones(B, np.float32)}
  outs = {'mlm_logits': rng.standard_normal((B, L, V)).astype(np.float32),
          'mpp_logits': rng.standard_normal((B, P, C)).astype(np.float32),
          'itm_logits': rng.standard_normal((B, 2)).astype(np.float32)}
  t = lambda d: {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}
  task.process_metrics(ms, t(labels), t(outs))
  named = M.by_name(ms)
  itm = labels['itm_label_ids'][:, None].astype(np.float32)
  for head, wkey, mask in (('mlm', 'mlm_label_weights', itm), ('mpp', 'mpp_label_weights', itm), ('itm', 'itm_label_weights', 1.0)):
    st = om.sparse_categorical_accuracy_update(np.zeros(2, np.float32), labels[f'{head}_label_ids'], outs[f'{head}_logits'],
                                               labels[wkey] * mask)
    assert abs(float(named[f'{head}_accuracy'].result()) - float(om.divide_no_nan(st[0], st[1]))) < 1e-6, head
  assert float(named['mlm_loss'].result()) == 0.0                  # losses are updated by the loss function, not here
  # classification: metric sets by number of classes, AUC fed the probability of True
  ccfg = configs.get_exp_config('mmt/classification').task
  ccfg.model.num_classes = 2
  ctask = tasks.get_task(ccfg)
  cms = ctask.build_metrics()
  assert [m.name for m in cms] == ['cls_accuracy', 'auc', 'classification_loss']
  y = rng.integers(0, 2, 40); lg = rng.standard_normal((40, 2)).astype(np.float32); w = rng.random(40).astype(np.float32)
  ctask.process_metrics(cms, {'label_ids': torch.from_numpy(y), 'label_weights': torch.from_numpy(w)}, {'logits': torch.from_numpy(lg)})
  e = np.exp(lg - lg.max(1, keepdims=True)); prob = (e / e.sum(1, keepdims=True))[:, 1]
  want = float(om.auc_pr_result(om.auc_update(np.zeros((4, 200), np.float32), y, prob, w)))
  assert abs(float(M.by_name(cms)['auc'].result()) - want) < 2e-5
  ccfg.model.num_classes = 1
  assert [m.name for m in tasks.get_task(ccfg).build_metrics()] == ['auc', 'classification_loss']
  ccfg.model.num_classes = 5
  assert [m.name for m in tasks.get_task(ccfg).build_metrics()] == ['cls_accuracy', 'classification_loss']


def _free_port():
  s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _metric_worker(rank, world, port, out):
  import torch.distributed as dist
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from mmt_amd import metrics as M
  g = torch.Generator().manual_seed(10 + rank)
  acc, mean = M.SparseCategoricalAccuracy('a'), M.Mean('l')
  y = torch.randint(0, 5, (6,), generator=g); lg = torch.randn(6, 5, generator=g); w = torch.rand(6, generator=g)
  acc.update_state(y, lg, w)
  mean.update_state(torch.tensor(float(rank + 1)))
  out[rank] = (float(acc.result()), float(mean.result()), acc.state.tolist())
  dist.destroy_process_group()


def test_metrics_sum_over_replicas_gloo():
  """metric variables are SUM-aggregated over the replicas when read."""
  import torch.multiprocessing as mp
  world, port = 2, _free_port()
  with mp.Manager() as mgr:
    out = mgr.dict()
    mp.spawn(_metric_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
  assert r0[0] == r1[0] and r0[1] == r1[1] == 1.5                    # both ranks read the same, aggregated, values
  tot = r0[2][0] + r1[2][0]; cnt = r0[2][1] + r1[2][1]
  assert abs(r0[0] - tot / cnt) < 1e-6


@pytest.mark.gpu
def test_loss_kernel_reports_the_first_argmax():
  """`mmt_xent_fwd_argmax`: the arg-max of every logits row from the loss pass, first occurrence on ties (bf16 logits
  of a 30522-way head tie often), odd and even widths, both dtypes; losses unchanged."""
  from mmt_amd import fused
  g = torch.Generator(device='cuda').manual_seed(0)
  for C, dt in ((30522, torch.bfloat16), (513, torch.bfloat16), (512, torch.float32), (7, torch.float32)):
    rows = 37
    lg = (torch.randn(rows, C, device='cuda', generator=g) * 2).to(dt)
    lg[3, 5] = lg[3].max() ; lg[3, C - 1] = lg[3, 5]                 # explicit ties: first index must win
    lg[4, C - 1] = lg[4].float().max() + 1                           # the (odd) last element is the maximum
    lab = torch.randint(0, C, (rows,), device='cuda', generator=g)
    w = torch.rand(rows, device='cuda', generator=g)
    loss, am = fused.weighted_softmax_cross_entropy(lg, lab, w, return_argmax=True)
    ref = torch.argmax(lg.float(), dim=-1)
    # torch.argmax does not promise the first index on ties: build the first-occurrence arg-max explicitly
    mx = lg.float().max(dim=-1, keepdim=True).values
    first = torch.where(lg.float() == mx, torch.arange(C, device='cuda')[None, :], torch.full((1, 1), C, device='cuda')).min(dim=-1).values
    assert torch.equal(am.long(), first), (C, dt)
    assert int(first[4]) == C - 1 and (lg.float()[torch.arange(rows), ref] == mx[:, 0]).all()
    plain = fused.weighted_softmax_cross_entropy(lg, lab, w)
    assert torch.equal(loss, plain)


@pytest.mark.gpu
def test_train_step_updates_the_metrics_on_the_device():
  """One pretraining train step with `task.build_metrics()`: the accuracies equal the restatement applied to the
  step's own logits (recomputed with the same dropout seeds), the loss metrics equal the step's loss terms."""
  from mmt_amd import configs, tasks, fused
  from mmt_amd import metrics as M
  cfg = configs.get_exp_config('mmt/pretraining')
  enc = cfg.task.model.encoder.mmt
  enc.num_hidden_layers, enc.hidden_size, enc.num_attention_heads, enc.intermediate_size = 2, 128, 2, 512
  cfg.task.model.cls_heads = [configs.ClsHeadConfig(inner_dim=128, num_classes=2, name='itm')]
  d = cfg.task.train_data
  d.max_seq_len, d.image_size, d.global_batch_size = 256, 224, 4
  d.mlm_max_selections_per_seq, d.mpp_max_selections_per_seq, d.tasks = 20, 10, 'mlm,itm'
  cfg.task.micro_batch_size = 4
  torch.manual_seed(0)
  task = tasks.get_task(cfg.task)
  model = task.build_model().cuda()
  opt = torch.optim.SGD(model.parameters(), lr=0.0)
  batch = next(task.build_inputs(d, device='cuda'))
  ms = task.build_metrics()
  out = task.train_step(batch, model, opt, metrics=ms, step=7)
  torch.cuda.synchronize()
  named = M.by_name(ms)
  inputs, labels = batch
  fused.set_seed_stream(7, 0, 0)
  with torch.no_grad():
    outs = model(**inputs, training=True)
  itm = labels['itm_label_ids'].float().cpu().numpy()[:, None] if 'itm_label_weights' in labels else 1.0
  for head, mask in (('mlm', itm), ('mpp', itm)):
    st = om.sparse_categorical_accuracy_update(np.zeros(2, np.float32), labels[f'{head}_label_ids'].cpu().numpy(),
                                               outs[f'{head}_logits'].float().cpu().numpy(),
                                               labels[f'{head}_label_weights'].float().cpu().numpy() * mask)
    assert abs(float(named[f'{head}_accuracy'].result()) - float(om.divide_no_nan(st[0], st[1]))) < 1e-6, head
  total = sum(float(named[f'{h}_loss'].result()) for h in ('mlm', 'mpp', 'itm') if f'{h}_loss' in named and named[f'{h}_loss'].state is not None)
  assert abs(total - float(out[task.loss])) < 1e-4 * max(1.0, abs(total))
