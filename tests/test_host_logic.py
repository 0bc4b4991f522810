"""Host-side mirror of the reference's registry / config / optimiser / strategy surface -- CPU."""
import os
import warnings

import pytest
import torch

import __graft_entry__  # noqa: F401
from mmt_amd import configs, distribute, optimization, tasks

YAML = """
task:
  model:
    encoder:
      type: mmt
      mmt:
        relative_att_num_core_layers: 2
        relative_pos_max_distance: 12
        relative_vocab_size: 49
    cls_heads:
      - inner_dim: 768
        num_classes: 2
        name: 'itm'
  train_data:
    global_batch_size: 4096
    max_seq_len: 256
    tasks: 'mlm,itm'
    mpp_fraction_to_mask: 0.0
    use_image_text_matching_label: true      # key the dataclasses do not have (App. B q11)
trainer:
  train_steps: 40000
  optimizer_config:
    learning_rate:
      polynomial:
        initial_learning_rate: 0.0005
        decay_steps: 40000
    warmup:
      polynomial:
        warmup_steps: 4000
"""


def test_registry_names_and_defaults():
  for name in ('mmt/pretraining', 'mmt/classification', 'mmt/retrieval'):
    assert configs.get_exp_config(name).task is not None
  with pytest.raises(KeyError):
    configs.get_exp_config('mmt/unknown')
  pre = configs.get_exp_config('mmt/pretraining')
  e = pre.task.model.encoder.get()
  # src/configs/encoders.py:32-101
  assert (e.vocab_size, e.hidden_size, e.num_hidden_layers, e.num_attention_heads) == (30522, 768, 12, 12)
  assert (e.relative_pos_max_distance, e.relative_vocab_size, e.intermediate_size) == (12, 32, 3072)
  assert e.use_pre_activation_order is True and e.use_one_hot_lookup is True and e.use_pooler_layer is False
  o = pre.trainer.optimizer_config
  assert (o.weight_decay_rate, o.initial_learning_rate, o.end_learning_rate) == (0.01, 1e-4, 0.0)
  assert o.exclude_from_weight_decay == ['LayerNorm', 'layer_norm', 'bias']
  assert configs.get_exp_config('mmt/retrieval').trainer.optimizer_config.initial_learning_rate == 3e-5
  assert pre.task.micro_batch_size == 64 and pre.task.scale_loss is False
  d = pre.task.train_data
  assert (d.mlm_max_selections_per_seq, d.mpp_max_selections_per_seq, d.output_channel_bits) == (256, 98, 3)
  assert isinstance(tasks.get_task(pre.task), tasks.PretrainingTask)
  assert isinstance(tasks.get_task(configs.get_exp_config('mmt/classification').task), tasks.ClassificationTask)


def test_yaml_and_dotted_overrides(tmp_path):
  path = tmp_path / 'exp.yaml'
  path.write_text(YAML)
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    cfg = configs.parse_configuration('mmt/pretraining', [str(path)],
                                      'task.train_data.max_seq_len=1024,runtime.mixed_precision_dtype=bfloat16')
  assert any('use_image_text_matching_label' in str(x.message) for x in w)
  assert cfg.task.train_data.max_seq_len == 1024 and cfg.task.train_data.global_batch_size == 4096
  assert cfg.task.model.encoder.mmt.relative_vocab_size == 49
  assert cfg.task.model.cls_heads[0].name == 'itm' and cfg.task.model.cls_heads[0].inner_dim == 768
  oc = cfg.trainer.optimizer_config
  assert (oc.initial_learning_rate, oc.decay_steps, oc.warmup_steps) == (0.0005, 40000, 4000)
  assert cfg.runtime.mixed_precision_dtype == 'bfloat16'
  with pytest.raises(KeyError):
    configs.parse_configuration('mmt/pretraining', [str(path)], strict=True)


REF_YAMLS = '/root/reference/src/exp_yamls'


@pytest.mark.skipif(not os.path.isdir(REF_YAMLS), reason='the reference tree is not on this machine (GPU box)')
def test_reference_yamls_parse_in_place():
  """Every experiment YAML the reference ships (read where it lies, never copied) loads through the registry; the
  only keys left unknown are the ones the reference's own dataclasses lack too."""
  import glob
  q11 = ('use_image_text_matching_label',)
  paths = sorted(glob.glob(os.path.join(REF_YAMLS, '*', '*', '*.yaml')))
  assert len(paths) >= 9
  for path in paths:
    exp = 'mmt/pretraining' if os.sep + 'pretrain' + os.sep in path else 'mmt/classification'
    with warnings.catch_warnings(record=True) as w:
      warnings.simplefilter('always')
      cfg = configs.parse_configuration(exp, [path])
    extra = [str(x.message) for x in w if 'unknown config key' in str(x.message) and not any(k in str(x.message) for k in q11)]
    assert not extra, (path, extra)
    assert cfg.task.model.encoder.type == 'mmt'
    if exp == 'mmt/classification':
      d = cfg.task.train_data
      assert isinstance(d, configs.MmtClassificationDataConfig) and d.negative_positive_ratio >= 1 and d.pos_weight > 0
      assert cfg.trainer.best_checkpoint_metric_comp in ('higher', 'lower')
  # the retrieval experiment carries the retrieval loader's fields (retrieval_dataloader.py:31-43)
  r = configs.get_exp_config('mmt/retrieval').task.train_data
  assert isinstance(r, configs.MmtRetrievalDataConfig)
  assert (r.image_input_path, r.num_text_examples, r.drop_remainder, r.include_image_text_index) == ('', 0, False, True)
  assert configs.get_exp_config('mmt/retrieval').task.validation_data.is_training is False


def test_classification_data_config_drives_the_matching_step():
  """`negative_positive_ratio` / `min_shift` of the data config reach the in-batch negatives
  (classification_dataloader.py:132-140); `pos_weight` the retrieval labels (data_utils.py:744-760)."""
  from mmt_amd import feature_pipeline as fp
  cfg = configs.get_exp_config('mmt/classification')
  cfg.override({'task.train_data.negative_positive_ratio': 3, 'task.train_data.min_shift': 2})
  d = cfg.task.train_data
  assert fp.matching_batch_size(d, 4) == (5 // 4 + 2) * 4
  B = 12
  feats = {'text_token_ids': torch.arange(B * 2).reshape(B, 2), 'num_text_wordpieces': torch.arange(B),
           'patch_token_ids': torch.arange(B * 3).reshape(B, 3), 'patch_embeddings': torch.zeros(B, 3, 4),
           'num_image_wordpieces': torch.full((B,), 5)}
  out = fp.make_matching_features_from_config(d, feats, torch.arange(B))
  assert out['itm_label_ids'].shape == (B * 4,) and int(out['itm_label_ids'].sum()) == B
  assert torch.equal(out['itm_pos_weights'][:B], torch.full((B,), 3.0)) and torch.equal(out['itm_pos_weights'][B:], torch.ones(3 * B))
  for i in range(1, 4):                                        # copy i: text rolled by min_shift + i
    assert torch.equal(out['num_text_wordpieces'][i * B:(i + 1) * B], torch.roll(torch.arange(B), 2 + i))
  lab = fp.make_retrieval_labels({'image_index': torch.tensor([0, 1, 2, 3]), 'gt_image_index': torch.tensor([0, 2, 2, 0])},
                                 pos_weight=4.0)
  assert lab['label_ids'].tolist() == [1, 0, 1, 0] and lab['label_ids'].dtype == torch.int32
  assert lab['label_weights'].tolist() == [4.0, 1.0, 4.0, 1.0]


def test_learning_rate_schedule_and_decay_groups():
  oc = configs.OptimizerConfig(initial_learning_rate=5e-4, decay_steps=40000, warmup_steps=4000)
  assert optimization.learning_rate_at(oc, 0) == 0.0
  assert abs(optimization.learning_rate_at(oc, 2000) - 5e-4 * (1 - 2000 / 40000) * 0.5) < 1e-12
  assert abs(optimization.learning_rate_at(oc, 20000) - 2.5e-4) < 1e-12
  assert optimization.learning_rate_at(oc, 50000) == 0.0
  m = torch.nn.Module()
  m.dense_weight = torch.nn.Parameter(torch.zeros(2, 2))
  m.dense_bias = torch.nn.Parameter(torch.zeros(2))
  m.attention_layer_norm = torch.nn.LayerNorm(2)
  decay, no_decay = optimization.split_decay_groups(m.named_parameters(), oc.exclude_from_weight_decay)
  assert len(decay) == 1 and len(no_decay) == 3


def test_encoder_argument_errors():
  from mmt_amd import MmtEncoder
  with pytest.raises(ValueError, match='too small'):
    MmtEncoder(vocab_size=100, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
               relative_vocab_size=20, relative_pos_max_distance=12)
  with pytest.raises(ValueError, match='must be 0'):
    MmtEncoder(vocab_size=100, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
               relative_vocab_size=None, relative_pos_max_distance=12)
  enc = MmtEncoder(vocab_size=100, hidden_size=64, num_hidden_layers=1, num_attention_heads=1,
                   intermediate_size=128)
  assert enc.get_word_embedding_layer().vocab_size == 100
  assert enc.get_word_embedding_table().shape == (100, 64)
  assert enc.get_config()['relative_vocab_size'] == 32
  with pytest.raises(ValueError):
    enc.pooler_layer
  c = configs.EncoderConfig(type='bert')
  with pytest.raises(ValueError, match='Only MmtEncoder'):
    configs.build_encoder(c)


def test_embedding_assembly_matches_oracle_on_cpu():
  """A.1 runs on plain torch ops, so it can be checked without a GPU."""
  from mmt_amd import MmtEncoder
  from oracle import encoder as oenc
  torch.manual_seed(0)
  enc = MmtEncoder(vocab_size=50, hidden_size=64, num_hidden_layers=0, num_attention_heads=1,
                   intermediate_size=64, max_absolute_position_embeddings=40, patch_embedding_size=12)
  word = torch.randint(0, 50, (2, 20)); seg = torch.randint(0, 3, (2, 20)); pe = torch.randn(2, 9, 12)
  got = enc.embed(word, seg, pe, training=False)
  sd = {'encoder.' + k: v.detach() for k, v in enc.state_dict().items()}
  want = oenc.encoder_forward(sd, dict(enc.get_config(), num_hidden_layers=0), word, seg, None, None, pe)
  assert float((got.double() - want).abs().max()) < 1e-5
  assert float(got[:, 11:].sub(enc.embed(word, seg, None)[:, 11:]).abs().max()) == 0  # patches only at [2, 11)


def test_distribution_strategy_argument_errors():
  with pytest.raises(ValueError, match='can not be negative'):
    distribute.get_distribution_strategy('mirrored', num_gpus=-1)
  with pytest.raises(ValueError, match="quotes around 'off'"):
    distribute.get_distribution_strategy(False)
  with pytest.raises(ValueError, match='Unrecognized'):
    distribute.get_distribution_strategy('bogus')
  with pytest.raises(ValueError):
    distribute.get_distribution_strategy('off', num_gpus=2)
  s = distribute.get_distribution_strategy('mirrored', num_gpus=1)
  assert s.num_replicas_in_sync == 1 and s.rank == 0


def test_losses_divide_no_nan():
  from mmt_amd.layers import weighted_sparse_categorical_crossentropy_loss as wsce
  logits = torch.randn(2, 3, 5); labels = torch.randint(0, 5, (2, 3))
  assert float(wsce(logits, labels, torch.zeros(2, 3))) == 0.0
  w = torch.tensor([[1., 0., 1.], [0., 0., 1.]])
  ref = (torch.nn.functional.cross_entropy(logits.view(-1, 5), labels.view(-1), reduction='none') * w.view(-1)).sum() / 3
  assert abs(float(wsce(logits, labels, w)) - float(ref)) < 1e-6


def test_stale_bf16_shadow_is_resynchronised_after_a_master_write():
  """`layers._param_weight` hands the forward the optimizer's bf16 shadow of an fp32 master; a torch-side
  write to the master (load_state_dict, checkpoint restore) must not leave the forward on old weights."""
  import torch
  from mmt_amd import layers
  p = torch.nn.Parameter(torch.arange(8, dtype=torch.float32))
  p._mmt_shadow = p.detach().to(torch.bfloat16)
  p._mmt_shadow_version = p._version
  assert layers._param_weight(p, torch.bfloat16) is p._mmt_shadow
  with torch.no_grad():
    p.copy_(torch.full((8,), 3.0))
  w = layers._param_weight(p, torch.bfloat16)
  assert w is p._mmt_shadow and torch.equal(w.float(), torch.full((8,), 3.0))
  m = torch.nn.Linear(4, 2, bias=False)
  m.weight._mmt_shadow = m.weight.detach().to(torch.bfloat16)
  m.weight._mmt_shadow_version = m.weight._version
  m.load_state_dict({'weight': torch.ones(2, 4)})
  assert torch.equal(layers._param_weight(m.weight, torch.bfloat16).float(), torch.ones(2, 4))


def test_dropout_seed_stream_depends_on_step_micro_step_and_rank():
  from mmt_amd import fused, step_scalars
  def draw(step, micro, rank, n=3):       # the seeds the descriptors carry: host seed + the step's epoch
    fused.set_seed_stream(step, micro, rank)
    return [(fused.next_seed(0) + step_scalars.host_epoch()) % (1 << 64) for _ in range(n)]
  a = draw(5, 0, 0)
  assert a == draw(5, 0, 0)                          # resumable: a pure function of its arguments
  assert len(set(a)) == 3
  for other in (draw(6, 0, 0), draw(5, 1, 0), draw(5, 0, 1)):
    assert not set(a) & set(other)
  step_scalars.set_step(0)


def test_gradient_reduce_mode_follows_the_task_config():
  from mmt_amd import tasks
  pre = configs.get_exp_config('mmt/pretraining').task
  assert tasks.gradient_reduce_mode(pre) == 'mean'       # build default with scale_loss False
  pre.override({'gradient_reduction': 'sum'})
  assert tasks.gradient_reduce_mode(pre) == 'sum'        # the reference's literal SUM (pretraining.py:273)
  pre.override({'gradient_reduction': 'mean', 'scale_loss': True})
  assert tasks.gradient_reduce_mode(pre) == 'sum'        # pretraining.py:286-296: loss / replicas, then SUM
  cls = configs.parse_configuration('mmt/classification', params_override='task.gradient_reduction=sum').task
  assert tasks.gradient_reduce_mode(cls) == 'sum'
  cls.gradient_reduction = 'median'
  with pytest.raises(ValueError):
    tasks.gradient_reduce_mode(cls)


def test_attention_pattern_normalizes_listed_global_sets():
  """Host logic of `mmt_mask_desc.global_index` (ABI 2): sorted, de-duplicated; a contiguous run becomes the range
  form (structured kernels), a scattered list stays a list; negative positions are refused."""
  import mmt_amd
  P = mmt_amd.AttentionPattern
  run = P(local_radius=8, global_index=(12, 10, 11, 11)).normalized()
  assert run.global_index is None and (run.global_start, run.n_global) == (10, 3)
  scat = P(local_radius=8, global_index=(40, 3, 3, 17)).normalized()
  assert scat.global_index == (3, 17, 40) and scat.n_global == 3
  empty = P(local_radius=8, global_index=()).normalized()
  assert empty.global_index is None and empty.n_global == 0
  plain = P(local_radius=8, global_start=5, n_global=2)
  assert plain.normalized() is plain
  with pytest.raises(ValueError):
    P(global_index=(-1, 4)).normalized()
  with pytest.raises(ValueError):
    scat.to_desc(None)                      # a listed set needs the device its index list lives on


def test_step_scalars_host_side():
  """The step's share of a dropout seed (its epoch) is a pure function of the step, is what the descriptors add on the
  host, and is left at zero after a train step so that explicit seeds outside one are taken as given."""
  from mmt_amd import fused, step_scalars
  assert step_scalars.epoch_of(0) == 0
  assert step_scalars.epoch_of(3) == (3 * 0x9E3779B97F4A7C15) % (1 << 64)
  fused.set_seed_stream(7, 1, 2)
  a = fused.next_seed(0)
  assert step_scalars.host_epoch() == step_scalars.epoch_of(7)
  fused.set_seed_stream(8, 1, 2)
  assert fused.next_seed(0) == a                     # the seeds handed around on the host do not depend on the step
  assert step_scalars.host_epoch() == step_scalars.epoch_of(8)
  step_scalars.set_step(0)
  assert step_scalars.host_epoch() == 0 and not step_scalars.device_active()
