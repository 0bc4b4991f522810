"""Input contract of the models (`src/input_utils.py:21-103`) and the synthetic feature
generator used instead of the reference's tf.data pipeline (out of scope, SURVEY.md section 2
row 15; its OUTPUT feature contract is what is kept here)."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .ops import AttentionPattern, side_inputs

CLS_ID, PATCH_ID, SEP_ID, ATT_ID = 101, 1, 102, 2   # [CLS], [PATCH]=[unused0], [SEP], [ATT]=[unused1]
MASK_ID = 103                                         # [MASK] of the BERT vocabulary
PATCH_START_UNUSED_INDEX = 104                        # src/data/data_utils.py:31


def encoder_input_spec(data_cfg) -> Dict[str, Tuple[tuple, torch.dtype]]:
  """`create_mmt_encoder_inputs` (`input_utils.py:21-53`): names, per-example shapes, dtypes."""
  P = data_cfg.image_size // data_cfg.patch_size
  S = data_cfg.max_seq_len
  return {
      'word_ids': ((S,), torch.int32),
      'segment_ids': ((S,), torch.int32),
      'relative_att_ids': ((S, S), torch.int32),
      'att_mask': ((S, S), torch.int32),
      'patch_embeddings': ((P * P, data_cfg.patch_size ** 2 * 3), torch.float32),
  }


def mtm_input_and_label_spec(data_cfg):
  """`create_mtm_inputs_and_labels` (`input_utils.py:56-103`)."""
  a, b = data_cfg.mlm_max_selections_per_seq, data_cfg.mpp_max_selections_per_seq
  inputs = {'mlm_positions': ((a,), torch.int32), 'mpp_positions': ((b,), torch.int32)}
  labels = {'mlm_label_ids': ((a,), torch.int32), 'mlm_label_weights': ((a,), torch.int32),
            'mpp_label_ids': ((b,), torch.int32), 'mpp_label_weights': ((b,), torch.int32)}
  return inputs, labels


def attention_pattern_from_config(data_cfg, encoder_cfg=None) -> AttentionPattern:
  """Pattern descriptor equivalent to the side inputs `get_add_side_input_features_fn` builds
  (`src/data/data_utils.py:285-332`) plus the build's band/global extension."""
  P = data_cfg.image_size // data_cfg.patch_size
  m = data_cfg.relative_pos_max_distance
  r = data_cfg.relative_att_num_core_layers
  n_img = 2 + P * P
  g = int(getattr(data_cfg, 'num_global_tokens', 0))
  return AttentionPattern(
      local_radius=int(getattr(data_cfg, 'local_radius', 1 << 30)),
      global_start=n_img if g else 0, n_global=g,     # the [ATT] marker and the tokens after it
      id_mode=_lib.MMT_IDS_NONE if m <= 0 else (_lib.MMT_IDS_2D if r > 0 else _lib.MMT_IDS_1D),
      max_dist=m, patches_per_row=P if r > 0 else 0, core_layers=r)


def synthetic_batch(data_cfg, batch_size: int, device, generator: Optional[torch.Generator] = None,
                    vocab_size: int = 30522, dense_side_inputs: bool = False,
                    ragged: bool = False, task: str = 'pretrain'):
  """One (inputs, labels) pair with the shapes/dtypes of the reference's feature contract.
  Layout `[CLS][PATCH] patch_1..patch_{P^2} [ATT] text ... [SEP] pad` (data_utils.py:224-231)."""
  S = data_cfg.max_seq_len
  P = data_cfg.image_size // data_cfg.patch_size
  n_patch = P * P
  n_img = 2 + n_patch
  if n_img >= S:
    raise ValueError('max_seq_len leaves no room for text')
  g = generator
  B = batch_size
  ri = lambda lo, hi, shape: torch.randint(lo, hi, shape, device=device, generator=g, dtype=torch.int32)
  word_ids = ri(1000, vocab_size, (B, S))
  word_ids[:, 0], word_ids[:, 1] = CLS_ID, PATCH_ID
  word_ids[:, 2:n_img] = PATCH_START_UNUSED_INDEX + torch.arange(n_patch, device=device, dtype=torch.int32)
  word_ids[:, n_img] = ATT_ID
  max_text = S - n_img
  if ragged:
    lo = max(2, int(0.75 * S) - n_img)
    n_text = ri(lo, max_text + 1, (B,))
  else:
    n_text = torch.full((B,), max_text, device=device, dtype=torch.int32)
  n_image = torch.full((B,), n_img, device=device, dtype=torch.int32)
  valid_len = (n_image + n_text).to(torch.int32)
  pos = torch.arange(S, device=device)[None]
  word_ids = torch.where(pos < valid_len[:, None], word_ids, torch.zeros_like(word_ids))
  pattern = attention_pattern_from_config(data_cfg)
  inputs = {
      'word_ids': word_ids,
      'patch_embeddings': torch.randn(B, n_patch, data_cfg.patch_size ** 2 * 3, device=device, generator=g),
  }
  if dense_side_inputs:     # exactly what the reference feeds: int32 [B,S,S] tensors
    si = side_inputs(pattern, n_image, n_text, S,
                     materialize_pattern=pattern.local_radius < S or pattern.n_global > 0)
    inputs.update(si)
    if si['relative_att_ids'] is None:
      inputs.pop('relative_att_ids')
  else:                     # structured fast path: descriptor + valid lengths only
    si = side_inputs(pattern, n_image, n_text, S, want_mask=False, want_ids=False)
    inputs['segment_ids'] = si['segment_ids']
    inputs['attention_pattern'] = pattern
    inputs['valid_len'] = valid_len
  labels = {}
  if task == 'pretrain':
    # MLM / MPP masking exactly as the reference's data pipeline applies it (`get_masking_fn`,
    # data_utils.py:383-639), on the device: feature_pipeline.make_mlm_and_mpp_features
    from . import feature_pipeline as fp
    rf = lambda *shape: torch.rand(*shape, device=device, generator=g)
    word_ids[torch.arange(B, device=device), (valid_len - 1).long()] = SEP_ID
    feats = {'patch_token_ids': word_ids[:, :n_img].contiguous(), 'text_token_ids': word_ids[:, n_img:].contiguous(),
             'num_text_wordpieces': n_text, 'patch_embeddings': inputs['patch_embeddings'],
             'unnormalized_patch_embeddings': rf(B, n_patch, data_cfg.patch_size ** 2 * 3)}
    randoms = {'mlm_item_keys': rf(B, max_text), 'mlm_value_u': rf(B, max_text), 'mlm_random_ids': ri(0, vocab_size, (B, max_text)),
               'mpp_item_keys': rf(B, n_img), 'mpp_value_u': rf(B, n_img), 'mpp_random_ids': ri(0, vocab_size, (B, n_img))}
    m = fp.make_mlm_and_mpp_features(
        feats, randoms, max_seq_len=S, num_patches=n_patch, patch_size=data_cfg.patch_size, vocab_size=vocab_size,
        mask_token_id=MASK_ID, unselectable_ids=(CLS_ID, SEP_ID, PATCH_ID, ATT_ID),
        mlm_fraction_to_mask=data_cfg.mlm_fraction_to_mask, mpp_fraction_to_mask=data_cfg.mpp_fraction_to_mask,
        mlm_max_selections_per_seq=data_cfg.mlm_max_selections_per_seq,
        mpp_max_selections_per_seq=data_cfg.mpp_max_selections_per_seq,
        output_channel_bits=data_cfg.output_channel_bits)
    inputs.update(word_ids=m['word_ids'], patch_embeddings=m['patch_embeddings'],
                  mlm_positions=m['mlm_positions'], mpp_positions=m['mpp_positions'])
    labels.update(mlm_label_ids=m['mlm_label_ids'], mlm_label_weights=m['mlm_label_weights'],
                  mpp_label_ids=m['mpp_label_ids'], mpp_label_weights=m['mpp_label_weights'])
    if 'itm' in (data_cfg.tasks or 'mlm,itm'):
      labels.update(itm_label_ids=ri(0, 2, (B,)),
                    itm_label_weights=torch.ones(B, device=device, dtype=torch.float32))
  else:
    labels.update(label_ids=ri(0, 2, (B,)), label_weights=torch.ones(B, device=device),
                  pos_weights=torch.ones(B, device=device))
  return inputs, labels
