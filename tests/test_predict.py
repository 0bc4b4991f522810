"""Retrieval predict + recall@k host logic (mmt_amd/predict.py; prediction_helper.py:30-118)."""
import json
import os

import numpy as np
import pytest
import torch

import __graft_entry__  # noqa: F401


def brute_recall(score, gt, k, axis):
  """Definition: a query (row for i2t, column for t2i) with at least one ground-truth partner counts as a
  hit when one of its partners is among its k highest-scored candidates."""
  if axis == 0:
    score, gt = score.T, gt.T
  hits = valid = 0
  for s, g in zip(score, gt):
    if g.sum() == 0:
      continue
    valid += 1
    top = np.argsort(-s, kind='stable')[:k]
    hits += int(g[top].any())
  return hits / valid


def test_recall_at_k_matches_definition_and_handles_missing_pairs(tmp_path):
  from mmt_amd import predict as P
  rng = np.random.default_rng(0)
  n_img, n_txt = 7, 9
  score = rng.permutation(n_img * n_txt).reshape(n_img, n_txt) / (n_img * n_txt)      # distinct scores
  owner = rng.integers(0, n_img, n_txt)                                                # each text belongs to one image
  results = []
  for i in range(n_img):
    for t in range(n_txt):
      if (i, t) in ((0, 0), (3, 5)) and owner[t] != i:
        continue                                                                       # pairs never scored
      results.append(P.RawResult(10 + i, 100 + t, 10 + int(owner[t]), float(score[i, t])))
  results.append(results[0])                                                           # duplicate pair: mean
  rec = P.get_recall_at_k(results, topks=(1, 3))
  full = np.full((n_img, n_txt), -1.0)
  gt = np.zeros((n_img, n_txt))
  for r in results:
    full[r.image_index - 10, r.text_index - 100] = r.output
    gt[r.image_index - 10, r.text_index - 100] = float(r.image_index == r.gt_image_index)
  for k in (1, 3):
    assert rec[f'i2t @ {k:>2}'] == f'{brute_recall(full, gt, k, 1):.4f}'
    assert rec[f't2i @ {k:>2}'] == f'{brute_recall(full, gt, k, 0):.4f}'
  out = P.write_results(results + [P.RawResult(10, 100, 10, 1.7)], str(tmp_path), topks=(1,))
  assert set(os.listdir(tmp_path)) == {'results.csv', 'recall.json'}
  lines = open(tmp_path / 'results.csv').read().splitlines()
  assert lines[0] == 'image_index,text_index,gt_image_index,output' and lines[-1].endswith(',1.00000000')
  assert json.load(open(tmp_path / 'recall.json')) == dict(out)


@pytest.mark.gpu
def test_predict_runs_the_classification_model(tmp_path):
  import mmt_amd
  from mmt_amd import configs, predict as P
  from tests.test_gpu_encoder import tiny_experiment
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  cexp = configs.get_exp_config('mmt/retrieval')
  cexp.override({'task': {'model': {'encoder': exp.task.model.encoder.as_dict(),
                                    'cls_heads': [{'inner_dim': 64, 'num_classes': 2, 'name': 'itm'}]},
                          'train_data': exp.task.train_data.as_dict()}}, strict=False)
  task = mmt_amd.tasks.get_task(cexp.task, compute_dtype=torch.bfloat16)
  torch.manual_seed(0)
  model = task.build_model().cuda()
  data = task.build_inputs(cexp.task.train_data, device='cuda', batch_size=6)
  batches = []
  for s in range(2):
    inputs, labels = next(data)
    inputs = dict(inputs)
    inputs['image_index'] = torch.arange(6, device='cuda') // 2 + 3 * s
    inputs['text_index'] = torch.arange(6, device='cuda') + 6 * s
    inputs['gt_image_index'] = inputs['image_index'].clone()
    batches.append((inputs, labels))
  res = P.predict(task, batches, model)
  assert len(res) == 12 and all(0.0 <= r.output <= 1.0 for r in res)
  rec = P.write_results(res, str(tmp_path))
  assert rec['i2t @ 10'] == '1.0000'           # every image's own texts are among its (<= 10) candidates
