"""Checkpoint / resume host logic (mmt_amd/checkpoint.py; reference semantics: classification.py:229-253,
pretraining.py:341-351, model.checkpoint_items).  CPU part: by-name items, latest/max_to_keep, warm start.
GPU part: an interrupted run restored from its checkpoint continues bit-identically."""
import os

import pytest
import torch

import __graft_entry__  # noqa: F401
from tests.test_gpu_encoder import tiny_experiment


def _models():
  import mmt_amd
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.float32)
  torch.manual_seed(1)
  a = task.build_model()
  torch.manual_seed(2)
  b = task.build_model()
  return exp, task, a, b


def test_save_restore_roundtrip_and_housekeeping(tmp_path):
  from mmt_amd import checkpoint
  exp, task, a, b = _models()
  assert set(checkpoint.model_items(a)) >= {'encoder', 'masked_lm', 'masked_pp'}
  opt = torch.optim.AdamW(a.parameters(), lr=1e-3)
  for step in (10, 20, 30, 40):
    path = checkpoint.save(str(tmp_path), step, a, opt, max_to_keep=2)
  assert sorted(os.listdir(tmp_path)) == ['ckpt-30.pt', 'ckpt-40.pt']          # max_to_keep
  assert checkpoint.latest_checkpoint(str(tmp_path)) == path and path.endswith('ckpt-40.pt')
  assert checkpoint.latest_checkpoint(path) == path and checkpoint.latest_checkpoint('') is None
  assert any(not torch.equal(p, q) for p, q in zip(a.parameters(), b.parameters()))
  step = checkpoint.restore(path, b, torch.optim.AdamW(b.parameters(), lr=1e-3))
  assert step == 40
  for (n, p), q in zip(a.named_parameters(), b.parameters()):
    assert torch.equal(p, q), n


def test_finetune_warm_start_takes_encoder_only(tmp_path):
  """classification.py:229-253: the encoder comes from the pretraining checkpoint; the fine-tuning heads
  (whose checkpoint key does not carry the head name) keep their initialisation."""
  import mmt_amd
  from mmt_amd import checkpoint, configs
  exp, task, pre, _ = _models()
  path = checkpoint.save(str(tmp_path), 5, pre)
  cexp = configs.get_exp_config('mmt/classification')
  cexp.override({'task': {'init_checkpoint': str(tmp_path),
                          'model': {'encoder': exp.task.model.encoder.as_dict(),
                                    'cls_heads': [{'inner_dim': 64, 'num_classes': 2, 'name': 'itm'}]},
                          'train_data': exp.task.train_data.as_dict()}}, strict=False)
  ctask = mmt_amd.tasks.get_task(cexp.task, compute_dtype=torch.float32)
  torch.manual_seed(7)
  model = ctask.build_model()
  heads_before = [p.clone() for p in model.classification_heads.parameters()]
  found = ctask.initialize(model)
  assert found == ['encoder']
  for (n, p), (_, q) in zip(pre.encoder.named_parameters(), model.encoder.named_parameters()):
    assert torch.equal(p, q), n
  for p, q in zip(heads_before, model.classification_heads.parameters()):
    assert torch.equal(p, q)
  # pretraining task: init_checkpoint restores every matching item, missing ones are tolerated
  exp.task.init_checkpoint = path
  exp2_task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.float32)
  torch.manual_seed(9)
  fresh = exp2_task.build_model()
  assert set(exp2_task.initialize(fresh)) >= {'encoder', 'masked_lm', 'masked_pp'}
  assert all(torch.equal(p, q) for p, q in zip(pre.parameters(), fresh.parameters()))


@pytest.mark.gpu
def test_resume_continues_bit_identically(tmp_path):
  import mmt_amd
  from mmt_amd import checkpoint, distribute, optimization
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  # dropout ON: the masks are a function of (train step, micro step, rank), so the resumed run draws the
  # masks the uninterrupted run drew for steps 4 and 5
  exp.task.model.encoder.mmt.hidden_dropout_prob = 0.1
  exp.task.model.encoder.mmt.attention_probs_dropout_prob = 0.1

  def fresh():
    task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.bfloat16)
    torch.manual_seed(4)
    model = task.build_model().cuda()
    reducer = distribute.DataParallelStrategy(None).make_reducer(list(model.parameters()))
    opt = optimization.create_optimizer(model, exp.trainer.optimizer_config, reducer=reducer)
    optimization.set_learning_rate(opt, 1e-3)
    batch = next(task.build_inputs(exp.task.train_data, device='cuda', batch_size=4))
    return task, model, reducer, opt, batch

  task, model, reducer, opt, batch = fresh()
  for step in (1, 2, 3):
    task.train_step(batch, model, opt, reducer=reducer, clip_norm=1.0, step=step)
  path = checkpoint.save(str(tmp_path), 3, model, opt)
  for step in (4, 5):
    task.train_step(batch, model, opt, reducer=reducer, clip_norm=1.0, step=step)
  want = [p.detach().clone() for p in model.parameters()]

  task2, model2, reducer2, opt2, batch2 = fresh()
  start = checkpoint.restore(path, model2, opt2)     # no refresh_shadow(): the shadows re-sync by themselves
  assert start == 3
  for step in (start + 1, start + 2):
    task2.train_step(batch2, model2, opt2, reducer=reducer2, clip_norm=1.0, step=step)
  for (n, p), q in zip(model2.named_parameters(), want):
    assert torch.equal(p, q), n
