"""Retrieval prediction + recall@k (`src/tasks/classification.py:256-334`,
`src/prediction_helper.py:30-118`).

`predict` runs the classification model over batches carrying `image_index`, `text_index` and
`gt_image_index` next to the encoder inputs and turns the head's logits into one score per pair with the
reference's rule (1 class: sigmoid; 2 classes: softmax probability of class 1; more: argmax).
`get_recall_at_k` rebuilds the reference's image x text score matrix (mean over duplicate pairs, missing
pairs = -1 / not ground truth) and reports image-to-text and text-to-image recall@k; `write_results`
writes the same two files, `results.csv` and `recall.json`."""
from __future__ import annotations

import collections
import csv
import json
import os
from typing import Iterable, List, Sequence

import numpy as np
import torch

RawResult = collections.namedtuple('RawResult', ['image_index', 'text_index', 'gt_image_index', 'output'])


@torch.no_grad()
def predict(task, batches: Iterable, model, logits_key: str = 'itm_logits') -> List[RawResult]:
  """`classification.predict`: one RawResult per example of every batch."""
  num_classes = None
  results: List[RawResult] = []
  was_training = model.training
  model.eval()
  for batch in batches:
    inputs = dict(batch[0] if isinstance(batch, (tuple, list)) else batch)
    image_index = inputs.pop('image_index')
    text_index = inputs.pop('text_index')
    gt_image_index = inputs.pop('gt_image_index')
    outputs = model(**inputs, training=False)
    key = logits_key if logits_key in outputs else next(k for k in outputs if k.endswith('_logits'))
    logits = outputs[key].float()
    num_classes = logits.shape[-1] if logits.dim() > 1 else 1
    if num_classes == 1:
      out = torch.sigmoid(logits.reshape(-1))
    elif num_classes == 2:
      out = torch.softmax(logits, dim=1)[:, 1]
    else:
      out = torch.argmax(logits, dim=1)
    for a, b, c, d in zip(image_index.tolist(), text_index.tolist(), gt_image_index.tolist(), out.tolist()):
      results.append(RawResult(a, b, c, d))
  model.train(was_training)
  return results


def _pivot(rows, cols, values):
  """pandas pivot_table(values, index, columns) with the default mean aggregation: sorted unique labels,
  NaN where a pair never occurs."""
  ri, r_inv = np.unique(rows, return_inverse=True)
  ci, c_inv = np.unique(cols, return_inverse=True)
  total = np.zeros((len(ri), len(ci)))
  count = np.zeros((len(ri), len(ci)))
  np.add.at(total, (r_inv, c_inv), values)
  np.add.at(count, (r_inv, c_inv), 1)
  with np.errstate(invalid='ignore', divide='ignore'):
    return np.where(count > 0, total / count, np.nan)


def get_recall_at_k(results: Sequence[RawResult], topks=(1, 3, 5, 10)) -> 'collections.OrderedDict[str, str]':
  """`prediction_helper.get_recall_at_k_from_dataframe` on a list of RawResult."""
  img = np.array([r.image_index for r in results])
  txt = np.array([r.text_index for r in results])
  gt = np.array([r.gt_image_index for r in results])
  out = np.array([r.output for r in results], dtype=np.float64)
  score = np.nan_to_num(_pivot(img, txt, out), nan=-1.0)             # missing pairs sort last
  gt_matrix = np.nan_to_num(_pivot(img, txt, (img == gt).astype(np.float64)), nan=0.0)
  rank = lambda x, axis: np.argsort(np.argsort(x, axis=axis, kind='stable'), axis=axis, kind='stable')
  m, n = score.shape
  i2t_rank = (rank(score, 1) - n) * -1                               # 1 = best text of an image
  t2i_rank = (rank(score, 0) - m) * -1
  recall = collections.OrderedDict()
  for name, rk, axis in (('i2t', i2t_rank, 1), ('t2i', t2i_rank, 0)):
    for k in topks:
      at_gt = rk * gt_matrix
      match = np.clip(((at_gt <= k) & (at_gt > 0)).sum(axis=axis).astype(float), 0, 1)
      valid = np.clip(gt_matrix.sum(axis=axis), 0, 1)
      r = match.sum() / valid.sum() if valid.sum() > 0 else 0.0
      recall[f'{name} @ {k:>2}'] = f'{r:.4f}'
  return recall


def write_results(results: Sequence[RawResult], output_dir: str, topks=(1, 3, 5, 10)):
  """`prediction_helper._write_results`: results.csv (scores clipped to [0, 1], %.8f) and recall.json."""
  os.makedirs(output_dir, exist_ok=True)
  clipped = [r._replace(output=min(1.0, max(0.0, float(r.output)))) for r in results]
  with open(os.path.join(output_dir, 'results.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(RawResult._fields)
    for r in clipped:
      w.writerow([r.image_index, r.text_index, r.gt_image_index, f'{r.output:.8f}'])
  recall = get_recall_at_k(clipped, topks)
  with open(os.path.join(output_dir, 'recall.json'), 'w') as f:
    json.dump(recall, f, indent=4)
  return recall
