"""Task objects of the reference for the hot path: `PretrainingTask`
(`src/tasks/pretraining.py:51-351`) and `ClassificationTask` (`src/tasks/classification.py:55-254`)
-- build_model / build_inputs / build_losses / train_step with the reference's micro-batch
accumulation semantics (pretraining.py:224-298)."""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import torch

from . import step_scalars, configs, fused, gemm_tuning, input_utils, layers, models

_TASKS = {}


def register_task_cls(config_cls):
  def deco(cls):
    _TASKS[config_cls] = cls
    return cls
  return deco


def get_task(task_config, logging_dir=None, **kw):
  """`task_factory.get_task`."""
  for cfg_cls, task_cls in _TASKS.items():
    if type(task_config) is cfg_cls:
      return task_cls(task_config, logging_dir=logging_dir, **kw)
  raise KeyError(f'no task registered for {type(task_config).__name__}')


def gradient_reduce_mode(task_config) -> str:
  """How the data-parallel reducer combines per-replica gradients for a task config.

  The reference's optimizer SUMs the replicas' gradients (`pretraining.py:273`).  With
  `scale_loss=True` each replica differentiates `loss / num_replicas` (`pretraining.py:286-296`), so
  the SUM is the mean-of-replicas gradient: 'sum'.  With the default `scale_loss=False` the reference
  applies world x the mean; this build averages instead unless the task config says
  `gradient_reduction: sum` -- a field of `PretrainingTaskConfig` / `ClassificationConfig` (YAML key
  `task.gradient_reduction`), the reference's literal behaviour."""
  if getattr(task_config, 'scale_loss', False):
    return 'sum'
  mode = getattr(task_config, 'gradient_reduction', 'mean')
  if mode not in ('mean', 'sum'):
    raise ValueError(f"task.gradient_reduction must be 'mean' or 'sum', got {mode!r}")
  return mode


def _compute_dtype(runtime_dtype: Optional[str]) -> torch.dtype:
  return torch.bfloat16 if runtime_dtype in ('bfloat16', 'mixed_bfloat16', 'bf16') else torch.float32


class _TaskBase:
  loss = 'loss'

  def __init__(self, params, logging_dir=None, name=None, compute_dtype=torch.float32,
               num_replicas: int = 1):
    self.task_config = params
    self.logging_dir, self.name = logging_dir, name
    self.compute_dtype = compute_dtype
    self.num_replicas = num_replicas

  def _build_encoder(self, encoder_cfg):
    gemm_tuning.ensure()            # measured library-GEMM selections for the BASELINE shapes (no-op on CPU)
    data_cfg = self.task_config.train_data
    return configs.build_encoder(encoder_cfg, compute_dtype=self.compute_dtype,
                                 patch_embedding_size=data_cfg.patch_size ** 2 * 3)

  def build_inputs(self, params, device='cuda', rank: int = 0, dense_side_inputs: bool = False,
                   batch_size: Optional[int] = None, ragged: bool = False):
    """Synthetic stand-in for the tf.data loaders: an endless iterator of (inputs, labels)
    with the reference's feature contract; per-replica batch = global_batch_size / replicas
    (`pretrain_dataloader.py:107-108`), seeded per rank."""
    per_replica = batch_size or max(1, params.global_batch_size // self.num_replicas)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    vocab = self.task_config.model.encoder.get().vocab_size
    task = 'pretrain' if isinstance(self, PretrainingTask) else 'classification'
    def it():
      while True:
        yield input_utils.synthetic_batch(params, per_replica, device, gen, vocab_size=vocab,
                                          dense_side_inputs=dense_side_inputs, ragged=ragged, task=task)
    return it()

  # ---- the reference's gradient-accumulation train step (pretraining.py:224-298) ----------
  def train_step(self, inputs, model, optimizer, metrics: Optional[Dict] = None, reducer=None,
                 micro_batch_size: Optional[int] = None, clip_norm: Optional[float] = None,
                 step: Optional[int] = None):
    """`step` = global train step (from the trainer loop / checkpoint): with the micro-step index and
    the data-parallel rank it determines every dropout mask of this call (`fused.set_seed_stream`)."""
    inputs, labels = inputs
    if step is None:
      step = self._auto_step = getattr(self, '_auto_step', 0) + 1
    rank = torch.distributed.get_rank() if (torch.distributed.is_available() and
                                            torch.distributed.is_initialized()) else 0
    try:
      micro = micro_batch_size or getattr(self.task_config, 'micro_batch_size', None)
      batch_size = inputs['word_ids'].shape[0]
      micro = micro or batch_size
      num_small_steps = batch_size // micro
      if num_small_steps == 0:
        warnings.warn('per-replica batch smaller than the micro batch: zero micro-steps run '
                      '(reference behaviour, SURVEY.md App. B q7)')
      if reducer is not None:
        reducer.zero_grad()
      else:
        optimizer.zero_grad(set_to_none=True)
      all_loss = torch.zeros((), device=inputs['word_ids'].device)
      is_t = lambda v: torch.is_tensor(v)
      for i in range(num_small_steps):
        # the reference takes the leading `micro` examples, then rotates them to the end
        if reducer is not None:
          reducer.set_armed(i == num_small_steps - 1)
        fused.set_seed_stream(step, i, rank)
        sl = slice(i * micro, (i + 1) * micro)
        small_inputs = {k: (v[sl] if is_t(v) else v) for k, v in inputs.items()}
        small_labels = {k: v[sl] for k, v in labels.items()}
        outputs = model(**small_inputs, training=True)
        loss = self.build_losses(small_labels, outputs, metrics)
        if metrics:                        # pretraining.py:297 / classification.py:211 (after the gradient in the reference;
          self.process_metrics(metrics, small_labels, outputs)      # nothing here depends on that order)
        if self.task_config.scale_loss:   # pretraining.py:286-296: gradient of loss / replicas
          grad_loss = loss / self.num_replicas
        else:
          grad_loss = loss / num_small_steps
        grad_loss.backward()
        all_loss = all_loss + (loss / num_small_steps).detach()
      fused_opt = hasattr(optimizer, 'slabs')          # optimization.FusedAdamW
      scale = None
      if reducer is not None:
        # gradient all-reduce == optimizer.apply_gradients(:273); a fused optimizer applies 1/world itself
        reducer.finish(defer_mean=fused_opt)
        if clip_norm:
          scale = reducer.clip_by_global_norm(clip_norm, apply=not fused_opt)
        elif fused_opt:
          scale = reducer.pending_scale_tensor()
      elif clip_norm:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
      if fused_opt:
        optimizer.step(grad_scale=scale)
      else:
        optimizer.step()
      return {self.loss: all_loss}
    finally:
      step_scalars.set_step(0)          # calls outside a train step take their dropout seeds as given -- also when the step raised

  @torch.no_grad()
  def initialize(self, model):
    """`pretraining.py:341-351`: load `task_config.init_checkpoint` (file, or the latest checkpoint of a
    directory) into the whole model by checkpoint-item name; missing items are tolerated."""
    from . import checkpoint
    path = checkpoint.latest_checkpoint(getattr(self.task_config, 'init_checkpoint', '') or '')
    if not path:
      return []                        # 'init_checkpoint is empty. Train from scratch.'
    return checkpoint.restore_items(path, checkpoint.model_items(model))

  def validation_step(self, inputs, model, metrics=None):
    """`pretraining.py:300-313` / `classification.py:214-227`: forward, losses, metrics."""
    inputs, labels = inputs
    outputs = model(**inputs, training=False)
    loss = self.build_losses(labels, outputs, metrics)
    if metrics:
      self.process_metrics(metrics, labels, outputs)
    return {self.loss: loss}

  def process_metrics(self, metrics, labels, model_outputs):
    del metrics, labels, model_outputs


@register_task_cls(configs.PretrainingTaskConfig)
class PretrainingTask(_TaskBase):
  """MLM + MPP + ITM pretraining."""

  def build_model(self, params=None):
    config = params or self.task_config.model
    encoder = self._build_encoder(config.encoder)
    data_cfg = self.task_config.train_data
    mpp_output_num_classes = (2 ** data_cfg.output_channel_bits) ** 3        # pretraining.py:69
    hidden = config.encoder.get().hidden_size
    heads = [layers.ClassificationHead(hidden, **c.as_dict()) for c in config.cls_heads]
    return models.MmtPretrainingModel(
        encoder=encoder, mpp_output_num_classes=mpp_output_num_classes,
        mlm_activation=config.mlm_activation, mlm_initializer=config.mlm_initializer,
        mpp_activation=config.mpp_activation, mpp_initializer=config.mpp_initializer,
        classification_heads=heads, bind_word_embedding_table=config.bind_word_embedding_table)

  def build_metrics(self, training=None):
    """`pretraining.py:183-196`: accuracy + mean loss for MLM, MPP and every classification head."""
    del training
    from . import metrics as M
    out = [M.SparseCategoricalAccuracy('mlm_accuracy'), M.Mean('mlm_loss'),
           M.SparseCategoricalAccuracy('mpp_accuracy'), M.Mean('mpp_loss')]
    for cfg in self.task_config.model.cls_heads:
      out += [M.SparseCategoricalAccuracy(f'{cfg.name}_accuracy'), M.Mean(f'{cfg.name}_loss')]
    return out

  def process_metrics(self, metrics, labels, model_outputs):
    """`pretraining.py:198-222`: the accuracies, MLM / MPP weights masked by the example's ITM label.  The arg-max
    of each logits row comes from the loss kernel's pass when `build_losses` left it in `model_outputs`."""
    from . import metrics as M
    named = M.by_name(metrics)
    if not named:
      return
    if 'itm_label_weights' in labels:
      itm = labels['itm_label_ids'].to(torch.float32).unsqueeze(1)
      mlm_w, mpp_w = labels['mlm_label_weights'] * itm, labels['mpp_label_weights'] * itm
    else:
      mlm_w, mpp_w = labels['mlm_label_weights'], labels['mpp_label_weights']
    if 'mlm_accuracy' in named:
      named['mlm_accuracy'].update_state(labels['mlm_label_ids'], model_outputs['mlm_logits'], mlm_w,
                                         argmax=model_outputs.get('mlm_argmax'))
    if 'mpp_accuracy' in named:
      named['mpp_accuracy'].update_state(labels['mpp_label_ids'], model_outputs['mpp_logits'], mpp_w,
                                         argmax=model_outputs.get('mpp_argmax'))
    if 'itm_accuracy' in named and 'itm_logits' in model_outputs:
      named['itm_accuracy'].update_state(labels['itm_label_ids'], model_outputs['itm_logits'],
                                         labels['itm_label_weights'], argmax=model_outputs.get('itm_argmax'))

  def build_losses(self, labels, model_outputs, metrics=None, aux_losses=None):
    """`pretraining.py:95-140`."""
    wsce = layers.weighted_sparse_categorical_crossentropy_loss
    # MLM / MPP losses are masked on negative pairs: the example's ITM label multiplies its rows' weights
    itm = labels['itm_label_ids'].float() if 'itm_label_weights' in labels else None
    total = wsce(model_outputs['mlm_logits'], labels['mlm_label_ids'], labels['mlm_label_weights'], metrics, 'mlm',
                 example_mask=itm, aux=model_outputs)
    total = total + wsce(model_outputs['mpp_logits'], labels['mpp_label_ids'], labels['mpp_label_weights'], metrics,
                         'mpp', example_mask=itm, aux=model_outputs)
    if 'itm_label_weights' in labels:
      total = total + wsce(model_outputs['itm_logits'], labels['itm_label_ids'],
                           labels['itm_label_weights'], metrics, 'itm', aux=model_outputs)
    if aux_losses:
      total = total + sum(aux_losses)
    return total


@register_task_cls(configs.ClassificationConfig)
class ClassificationTask(_TaskBase):
  """ITM / retrieval fine-tuning."""
  METRIC_TYPES = frozenset(['accuracy', 'auc'])

  def __init__(self, params, logging_dir=None, name=None, **kw):
    super().__init__(params, logging_dir, name, **kw)
    if params.metric_type not in self.METRIC_TYPES:
      raise ValueError(f'Invalid metric_type: {params.metric_type}')
    d = params.train_data
    self.label_field = d.label_field or 'label_ids'
    self.logits_field = d.logits_field or 'logits'
    self.label_weights_field = d.label_weights_field or 'label_weights'
    self.pos_weights_field = d.pos_weights_field or 'pos_weights'
    self.task_name = 'classification'

  def build_model(self):
    config = self.task_config.model
    encoder = self._build_encoder(config.encoder)
    hidden = config.encoder.get().hidden_size
    heads = [layers.ClassificationHead(hidden, **c.as_dict()) for c in config.cls_heads]
    return models.MmtClassificationModel(encoder=encoder, classification_heads=heads)

  def initialize(self, model):
    """`classification.py:229-253`: from `init_checkpoint` restore the encoder, plus every checkpoint item
    of the fine-tuning model whose key contains the name of one of its classification heads (the
    reference's own matching rule; with its `checkpoint_items` that is the encoder only unless a key
    carries the head's name)."""
    from . import checkpoint
    path = checkpoint.latest_checkpoint(self.task_config.init_checkpoint or '')
    if not path:
      return []
    items = checkpoint.model_items(model)
    mapping = {'encoder': items['encoder']}
    for head_cfg in self.task_config.model.cls_heads:
      for key, item in items.items():
        if head_cfg.name and head_cfg.name in key:
          mapping[key] = item
    return checkpoint.restore_items(path, mapping)

  def _logits_key(self, model_outputs):
    if self.logits_field in model_outputs:
      return self.logits_field
    return next(k for k in model_outputs if k.endswith('_logits'))      # e.g. head named 'itm' -> 'itm_logits'

  def build_metrics(self, training=None):
    """`classification.py:132-148`: AUC(PR) for one class, accuracy + AUC(PR) for two, accuracy beyond; plus the
    mean of the task's loss."""
    del training
    from . import metrics as M
    n = self.task_config.model.num_classes
    if n == 1:
      out = [M.AUC('auc', curve='PR')]
    elif n == 2:
      out = [M.SparseCategoricalAccuracy('cls_accuracy'), M.AUC('auc', curve='PR')]
    else:
      out = [M.SparseCategoricalAccuracy('cls_accuracy')]
    out.append(M.Mean(f'{self.task_name}_loss'))
    return out

  def process_metrics(self, metrics, labels, model_outputs):
    """`classification.py:150-170`."""
    from . import metrics as M
    named = M.by_name(metrics)
    if not named:
      return
    label_ids, label_weights = labels[self.label_field], labels[self.label_weights_field]
    logits = model_outputs[self._logits_key(model_outputs)]
    if 'auc' in named:
      n = self.task_config.model.num_classes
      if n == 1:
        probs = torch.sigmoid(logits.float().reshape(-1))
      elif n == 2:
        probs = torch.softmax(logits.float(), dim=-1)[:, 1]          # the probability of True
      else:
        raise ValueError('auc requires # classes either 1 or 2.')
      named['auc'].update_state(label_ids, probs, label_weights)
    if 'cls_accuracy' in named:
      named['cls_accuracy'].update_state(label_ids, logits, label_weights,
                                         argmax=model_outputs.get(f'{self.task_name}_argmax'))

  def build_losses(self, labels, model_outputs, metrics=None, aux_losses=None):
    """`classification.py:100-126` (the num_classes == 1 branch is dead in the reference, q6)."""
    logits_key = self._logits_key(model_outputs)
    loss = layers.weighted_sparse_categorical_crossentropy_loss(
        model_outputs[logits_key], labels[self.label_field], labels[self.label_weights_field],
        metrics, self.task_name, pos_weights=labels.get(self.pos_weights_field))
    if aux_losses:
      loss = loss + sum(aux_losses)
    return loss
