"""Checkpoint / resume.

The reference checkpoints through `tf.train.Checkpoint` objects keyed by `model.checkpoint_items`
(`src/modeling/models/mmt_pretraining_model.py:155-166`, `mmt_classification_model.py:79-81`), restores
the latest checkpoint of `model_dir` when training restarts (TFM `run_experiment`), keeps `max_to_keep`
files, and warm-starts fine-tuning from a pretraining checkpoint by name (`src/tasks/classification.py:
229-253`: the encoder plus every classification head whose name also exists in the fine-tuning model;
`src/tasks/pretraining.py:341-351`: the whole model, partial matches allowed).  Here a checkpoint is one
`torch.save` file `ckpt-<step>.pt`:
  {'step', 'items': {<checkpoint_items key>: state_dict or tensor}, 'optimizer': state_dict | None}
so the same by-name semantics carry over."""
from __future__ import annotations

import glob
import os
import re
from typing import Dict, Optional

import torch


def _item_state(item) -> Dict[str, torch.Tensor]:
  if isinstance(item, torch.nn.Module):
    return {k: v.detach().cpu() for k, v in item.state_dict().items()}
  if isinstance(item, torch.nn.ModuleList):
    return {k: v.detach().cpu() for k, v in item.state_dict().items()}
  if torch.is_tensor(item):
    return {'': item.detach().cpu()}
  if isinstance(item, dict):                  # a group of named tensors (e.g. a head's kernel + bias)
    return {k: v.detach().cpu() for k, v in item.items()}
  raise TypeError(f'cannot checkpoint {type(item)}')


def _load_item(item, state: Dict[str, torch.Tensor], strict: bool) -> None:
  if torch.is_tensor(item):
    with torch.no_grad():
      item.copy_(state[''].to(item.device, item.dtype))
    return
  if isinstance(item, dict):
    with torch.no_grad():
      for k, t in item.items():
        if k in state:
          t.copy_(state[k].to(t.device, t.dtype))
        elif strict:
          raise RuntimeError(f'checkpoint item lacks tensor {k!r}')
    return
  missing, unexpected = item.load_state_dict(state, strict=False)
  if strict and (missing or unexpected):
    raise RuntimeError(f'checkpoint mismatch: missing {missing}, unexpected {unexpected}')


def model_items(model) -> Dict[str, object]:
  """`model.checkpoint_items` (the reference's by-name checkpoint surface), else {'model': model}."""
  return dict(getattr(model, 'checkpoint_items', None) or {'model': model})


def latest_checkpoint(ckpt_dir_or_file: str) -> Optional[str]:
  """File itself, or the highest-step `ckpt-<step>.pt` of a directory (tf.train.latest_checkpoint)."""
  if not ckpt_dir_or_file:
    return None
  if os.path.isfile(ckpt_dir_or_file):
    return ckpt_dir_or_file
  best, best_step = None, -1
  for f in glob.glob(os.path.join(ckpt_dir_or_file, 'ckpt-*.pt')):
    m = re.search(r'ckpt-(\d+)\.pt$', f)
    if m and int(m.group(1)) > best_step:
      best, best_step = f, int(m.group(1))
  return best


def save(model_dir: str, step: int, model, optimizer=None, max_to_keep: int = 5) -> str:
  os.makedirs(model_dir, exist_ok=True)
  payload = {'step': int(step), 'items': {k: _item_state(v) for k, v in model_items(model).items()},
             'optimizer': None if optimizer is None else optimizer.state_dict()}
  path = os.path.join(model_dir, f'ckpt-{int(step)}.pt')
  tmp = path + '.tmp'
  torch.save(payload, tmp)
  os.replace(tmp, path)                       # a crash never leaves a truncated latest checkpoint
  if max_to_keep and max_to_keep > 0:
    files = sorted(glob.glob(os.path.join(model_dir, 'ckpt-*.pt')),
                   key=lambda f: int(re.search(r'ckpt-(\d+)\.pt$', f).group(1)))
    for old in files[:-max_to_keep]:
      os.remove(old)
  return path


def restore(path: str, model, optimizer=None, strict: bool = True) -> int:
  """Full restore (resume): every checkpoint item of the model, then the optimizer state."""
  payload = torch.load(path, map_location='cpu', weights_only=True)
  items = model_items(model)
  for key, item in items.items():
    if key not in payload['items']:
      if strict:
        raise RuntimeError(f'checkpoint {path} has no item {key!r}')
      continue
    _load_item(item, payload['items'][key], strict)
  if optimizer is not None and payload.get('optimizer') is not None:
    optimizer.load_state_dict(payload['optimizer'])
  return int(payload.get('step', 0))


def restore_items(path: str, mapping: Dict[str, object]) -> list:
  """Partial, by-name restore (`expect_partial`): loads the listed items that exist in the file and
  returns the keys that were found."""
  payload = torch.load(path, map_location='cpu', weights_only=True)
  found = []
  for key, item in mapping.items():
    if key in payload['items']:
      _load_item(item, payload['items'][key], strict=False)
      found.append(key)
  return found
