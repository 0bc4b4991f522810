"""Config dataclasses, experiment registry and the encoder factory hook of the reference,
without TF-Model-Garden / gin.

Field names, defaults and YAML keys follow `src/configs/encoders.py:32-109`,
`src/configs/mmt.py:24-50`, `src/data/configs.py:20-55`,
`src/data/pretrain_dataloader.py:29-42`, `src/tasks/pretraining.py:42-48`,
`src/tasks/classification.py:40-52` and the experiment factories
`src/configs/pretraining_experiments.py:50`, `src/configs/finetuning_experiments.py:25,63`
so the reference's `--experiment`, `--config_file` YAMLs and dotted `--params_override`
keep working for the hot path.  Unknown YAML keys warn and are ignored by default
(SURVEY.md App. B q11); `strict=True` raises.
"""
from __future__ import annotations

import copy
import dataclasses
import warnings
from typing import Any, Callable, Dict, List, Optional

import yaml


class Config:
  """Tiny stand-in for `hyperparams.Config`: nested override from dicts / dotted keys."""

  def override(self, values: Dict[str, Any], strict: bool = False, _path: str = ''):
    for key, val in values.items():
      if '.' in key:
        head, rest = key.split('.', 1)
        self.override({head: {rest: val}}, strict, _path)
        continue
      if not hasattr(self, key):
        msg = f'unknown config key `{_path}{key}`'
        if strict:
          raise KeyError(msg)
        warnings.warn(msg + ' (ignored)')
        continue
      cur = getattr(self, key)
      if isinstance(cur, Config) and isinstance(val, dict):
        cur.override(val, strict, f'{_path}{key}.')
      elif isinstance(cur, list) and isinstance(val, list) and key == 'cls_heads':
        setattr(self, key, [ClsHeadConfig(**v) if isinstance(v, dict) else v for v in val])
      else:
        setattr(self, key, val)
    return self

  def as_dict(self):
    return dataclasses.asdict(self)


@dataclasses.dataclass
class MmtEncoderConfig(Config):     # encoders.py:32-101
  vocab_size: int = 30522
  segment_vocab_size: int = 16
  embedding_size: Optional[int] = None
  hidden_size: int = 768
  num_hidden_layers: int = 12
  num_attention_heads: int = 12
  relative_pos_max_distance: int = 12
  relative_vocab_size: int = 32
  relative_att_num_core_layers: int = 0
  max_absolute_position_embeddings: Optional[int] = None
  intermediate_size: int = 3072
  hidden_activation: str = 'gelu'
  hidden_dropout_prob: float = 0.1
  attention_probs_dropout_prob: float = 0.1
  initializer_range: float = 0.02
  use_pre_activation_order: bool = True
  use_one_hot_lookup: bool = True
  use_pooler_layer: bool = False


@dataclasses.dataclass
class EncoderConfig(Config):        # encoders.py:104-109 (OneOfConfig keyed by `type`)
  type: Optional[str] = 'mmt'
  mmt: MmtEncoderConfig = dataclasses.field(default_factory=MmtEncoderConfig)

  def get(self):
    return getattr(self, self.type)


@dataclasses.dataclass
class ClsHeadConfig(Config):        # mmt.py:24-31
  inner_dim: int = 0
  num_classes: int = 2
  activation: Optional[str] = 'tanh'
  dropout_rate: float = 0.0
  cls_token_idx: int = 0
  name: Optional[str] = None


@dataclasses.dataclass
class PretrainModelConfig(Config):  # mmt.py:34-42
  encoder: EncoderConfig = dataclasses.field(default_factory=EncoderConfig)
  mlm_activation: str = 'gelu'
  mlm_initializer: str = 'glorot_uniform'
  mpp_activation: str = 'gelu'
  mpp_initializer: str = 'glorot_uniform'
  cls_heads: List[ClsHeadConfig] = dataclasses.field(default_factory=list)
  bind_word_embedding_table: bool = True   # read at pretraining.py:88


@dataclasses.dataclass
class ClassificationModelConfig(Config):   # mmt.py:45-50
  encoder: EncoderConfig = dataclasses.field(default_factory=EncoderConfig)
  num_classes: int = 0
  cls_heads: List[ClsHeadConfig] = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class MmtDataConfig(Config):        # data/configs.py:20-55 (+ TFM DataConfig basics)
  seed: int = 128
  input_path: str = ''
  num_examples: int = 0
  vocab_filename: str = ''
  is_training: bool = True
  global_batch_size: int = 256
  image_data_field: str = 'image_data'
  text_special_token_field_dict: str = (
      '{"caption_attribution_description": "[ATT]",'
      ' "caption_reference_description":"[REF]"}')
  image_key_field: str = 'image_key'
  tasks: str = ''
  patch_size: int = 16
  image_size: int = 224
  patch_order: str = 'raster_scan'
  max_pixel_val: int = 256
  max_seq_len: int = 512
  relative_pos_max_distance: int = 12
  relative_att_num_core_layers: int = 0
  label_field: Optional[str] = None
  label_weights_field: Optional[str] = None
  logits_field: Optional[str] = None
  pos_weights_field: Optional[str] = None
  min_shift: int = 5
  use_rand_aug: bool = False
  cycle_length: int = 8
  deterministic: bool = True
  drop_remainder: bool = True
  # Build-defined long-sequence pattern; defaults = reference mask.
  local_radius: int = 1 << 30
  num_global_tokens: int = 0


@dataclasses.dataclass
class MmtPretrainDataConfig(MmtDataConfig):  # pretrain_dataloader.py:29-42
  mlm_use_whole_word: bool = True
  mlm_fraction_to_mask: float = 0.15
  mpp_fraction_to_mask: float = 0.5
  mlm_max_selections_per_seq: int = 256
  mpp_max_selections_per_seq: int = 98
  output_channel_bits: int = 3
  input_channels: int = 3
  use_patch_mask_token_id: bool = False


@dataclasses.dataclass
class MmtClassificationDataConfig(MmtDataConfig):   # classification_dataloader.py:30-34
  negative_positive_ratio: int = 1
  pos_weight: float = 1.0


@dataclasses.dataclass
class MmtRetrievalDataConfig(MmtDataConfig):        # retrieval_dataloader.py:31-43
  # image and text records stored separately: all combinations are enumerated on the fly
  image_input_path: str = ''
  text_input_path: str = ''
  num_image_examples: int = 0
  num_text_examples: int = 0
  negative_positive_ratio: int = 1
  pos_weight: float = 1.0
  drop_remainder: bool = False
  include_image_text_index: bool = True


@dataclasses.dataclass
class PretrainingTaskConfig(Config):         # pretraining.py:42-48
  model: PretrainModelConfig = dataclasses.field(default_factory=PretrainModelConfig)
  scale_loss: bool = False
  train_data: MmtDataConfig = dataclasses.field(default_factory=MmtPretrainDataConfig)
  validation_data: MmtDataConfig = dataclasses.field(
      default_factory=lambda: MmtPretrainDataConfig(is_training=False))
  init_checkpoint: str = ''
  micro_batch_size: int = 64     # BATCH_SIZE_PER_REPLICA, pretraining.py:39 (App. B q7)
  # Build-defined: how data-parallel replicas' gradients combine when scale_loss is False -- 'mean' (default) or
  # 'sum' = what the reference's optimizer literally applies (pretraining.py:273).  With scale_loss
  # the reference's own loss / replicas makes the SUM the mean, and this field is not consulted.
  gradient_reduction: str = 'mean'


@dataclasses.dataclass
class ClassificationConfig(Config):          # classification.py:40-52
  model: ClassificationModelConfig = dataclasses.field(default_factory=ClassificationModelConfig)
  scale_loss: bool = False
  train_data: MmtDataConfig = dataclasses.field(default_factory=MmtDataConfig)
  validation_data: MmtDataConfig = dataclasses.field(
      default_factory=lambda: MmtDataConfig(is_training=False))
  init_checkpoint: str = ''
  init_cls_pooler: bool = False
  metric_type: str = 'accuracy'
  gradient_reduction: str = 'mean'   # as PretrainingTaskConfig.gradient_reduction (classification.py:200-210)


@dataclasses.dataclass
class OptimizerConfig(Config):   # the adamw / polynomial / warmup block of `_TRAINER`
  weight_decay_rate: float = 0.01
  exclude_from_weight_decay: List[str] = dataclasses.field(
      default_factory=lambda: ['LayerNorm', 'layer_norm', 'bias'])
  beta_1: float = 0.9
  beta_2: float = 0.999
  epsilon: float = 1e-7
  gradient_clip_norm: float = 1.0
  initial_learning_rate: float = 1e-4
  end_learning_rate: float = 0.0
  decay_steps: int = 1000000
  power: float = 1.0
  warmup_steps: int = 0
  warmup_power: float = 1.0


@dataclasses.dataclass
class TrainerConfig(Config):
  train_steps: int = 1000000
  steps_per_loop: int = 100
  summary_interval: int = 100
  checkpoint_interval: int = 1000
  max_to_keep: int = 5
  validation_interval: int = 1000
  validation_steps: int = -1
  # TFM `TrainerConfig` keys the fine-tune YAMLs set (itm_2d_from_vit.yaml:84-86): accepted and carried; exporting a
  # best checkpoint needs the validation loop's metric of that name
  best_checkpoint_export_subdir: str = ''
  best_checkpoint_eval_metric: str = ''
  best_checkpoint_metric_comp: str = 'higher'
  optimizer_config: OptimizerConfig = dataclasses.field(default_factory=OptimizerConfig)

  def override(self, values, strict=False, _path=''):
    # accept the nested TFM layout optimizer_config.{optimizer.adamw, learning_rate.polynomial,
    # warmup.polynomial}.* by flattening it onto OptimizerConfig
    oc = values.get('optimizer_config')
    if isinstance(oc, dict):
      flat = {}
      for block, inner in (('optimizer', 'adamw'), ('learning_rate', 'polynomial'), ('warmup', 'polynomial')):
        sub = (oc.get(block) or {}).get(inner) or {}
        for k, v in sub.items():
          flat['warmup_power' if (block == 'warmup' and k == 'power') else k] = v
      values = dict(values, optimizer_config={**{k: v for k, v in oc.items()
                                                 if k not in ('optimizer', 'learning_rate', 'warmup')}, **flat})
    return super().override(values, strict, _path)


@dataclasses.dataclass
class RuntimeConfig(Config):
  distribution_strategy: str = 'mirrored'
  mixed_precision_dtype: Optional[str] = None   # 'bfloat16' | 'float32' | None
  num_gpus: int = 0
  all_reduce_alg: Optional[str] = None
  enable_xla: bool = False
  tpu: Optional[str] = None


@dataclasses.dataclass
class ExperimentConfig(Config):
  task: Config = None
  trainer: TrainerConfig = dataclasses.field(default_factory=TrainerConfig)
  runtime: RuntimeConfig = dataclasses.field(default_factory=RuntimeConfig)
  restrictions: List[str] = dataclasses.field(default_factory=list)


# ------------------------------ experiment registry ---------------------------------------
_EXPERIMENTS: Dict[str, Callable[[], ExperimentConfig]] = {}


def register_config_factory(name: str):
  def deco(fn):
    if name in _EXPERIMENTS:
      raise KeyError(f'experiment {name!r} registered twice')
    _EXPERIMENTS[name] = fn
    return fn
  return deco


def get_exp_config(name: str) -> ExperimentConfig:
  if name not in _EXPERIMENTS:
    raise KeyError(f'experiment {name!r} is not registered; known: {sorted(_EXPERIMENTS)}')
  return _EXPERIMENTS[name]()


@register_config_factory('mmt/pretraining')
def mmt_pretraining() -> ExperimentConfig:       # pretraining_experiments.py:50-63
  return ExperimentConfig(
      task=PretrainingTaskConfig(),
      trainer=TrainerConfig(train_steps=1000000,
                            optimizer_config=OptimizerConfig(initial_learning_rate=1e-4)),
      restrictions=['task.train_data.is_training != None',
                    'task.validation_data.is_training != None'])


def _finetune(data_cls, lr=3e-5) -> ExperimentConfig:
  cfg = ExperimentConfig(
      task=ClassificationConfig(train_data=data_cls(), validation_data=data_cls(is_training=False)),
      trainer=TrainerConfig(optimizer_config=OptimizerConfig(initial_learning_rate=lr)),
      restrictions=['task.train_data.is_training != None',
                    'task.validation_data.is_training != None'])
  cfg.task.model.encoder.type = 'mmt'
  return cfg


@register_config_factory('mmt/classification')
def mmt_classification() -> ExperimentConfig:    # finetuning_experiments.py:25-60
  return _finetune(MmtClassificationDataConfig)


@register_config_factory('mmt/retrieval')
def mmt_retrieval() -> ExperimentConfig:         # finetuning_experiments.py:63-98
  return _finetune(MmtRetrievalDataConfig)


def parse_configuration(experiment: str, config_files=(), params_override: Optional[str] = None,
                        strict: bool = False) -> ExperimentConfig:
  """`train_utils.parse_configuration`: registry default -> YAML files -> dotted overrides."""
  cfg = get_exp_config(experiment)
  for path in config_files or ():
    with open(path) as f:
      cfg.override(yaml.safe_load(f) or {}, strict)
  if params_override:
    if '=' in params_override and ':' not in params_override.split('=')[0]:
      pairs = [kv for kv in params_override.split(',') if kv]
      cfg.override({k.strip(): yaml.safe_load(v) for k, v in (p.split('=', 1) for p in pairs)}, strict)
    else:
      cfg.override(yaml.safe_load(params_override), strict)
  return cfg


# ------------------------------ encoder factory hook ----------------------------------------
def build_encoder(config: EncoderConfig, encoder_cls=None, bypass_config: bool = False, **kwargs):
  """`encoders.build_encoder` (encoders.py:112-158): any class with the MmtEncoder surface can
  be injected through `encoder_cls`."""
  from .encoder import MmtEncoder
  if bypass_config:
    return encoder_cls()
  if config.type != 'mmt':
    raise ValueError('Only MmtEncoder is supported now')
  c = config.get()
  cls = encoder_cls or MmtEncoder
  return cls(vocab_size=c.vocab_size, segment_vocab_size=c.segment_vocab_size,
             embedding_size=c.embedding_size, hidden_size=c.hidden_size,
             num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads,
             intermediate_size=c.intermediate_size, inner_activation=c.hidden_activation,
             hidden_dropout_prob=c.hidden_dropout_prob,
             attention_probs_dropout_prob=c.attention_probs_dropout_prob,
             max_absolute_position_embeddings=c.max_absolute_position_embeddings,
             relative_vocab_size=c.relative_vocab_size,
             relative_pos_max_distance=c.relative_pos_max_distance,
             initializer_range=c.initializer_range,
             use_pre_activation_order=c.use_pre_activation_order,
             use_one_hot_lookup=c.use_one_hot_lookup, use_pooler_layer=c.use_pooler_layer, **kwargs)
