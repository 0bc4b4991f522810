"""Model wrappers of the reference: `MmtPretrainingModel`
(`src/modeling/models/mmt_pretraining_model.py:23-173`) and `MmtClassificationModel`
(`src/modeling/models/mmt_classification_model.py:23-93`) around `MmtEncoder`."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import layers
from .encoder import MmtEncoder


def _check_unique(heads):
  if len({h.name for h in heads}) != len(heads):
    raise ValueError('Classification heads should have unique names.')


class MmtPretrainingModel(nn.Module):

  def __init__(self, encoder: MmtEncoder, mpp_output_num_classes: Optional[int] = None,
               mlm_activation=None, mlm_initializer: str = 'glorot_uniform', mpp_activation=None,
               mpp_initializer: str = 'glorot_uniform',
               classification_heads: Optional[List[layers.ClassificationHead]] = None,
               bind_word_embedding_table: bool = True, name: str = 'mmt_pretraining_model'):
    super().__init__()
    self.name = name
    self.encoder = encoder
    self.classification_heads = nn.ModuleList(classification_heads or [])
    _check_unique(self.classification_heads)
    hidden = encoder.get_config()['hidden_size']
    self.masked_lm = layers.MaskedLM(encoder.get_word_embedding_layer(),
                                     layers.get_activation(mlm_activation),
                                     bind=bind_word_embedding_table)
    self.masked_pp = layers.MaskedPP(hidden, mpp_output_num_classes,
                                     layers.get_activation(mpp_activation), output='logits')

  def forward(self, word_ids, segment_ids=None, att_mask=None, relative_att_ids=None,
              patch_embeddings=None, mlm_positions=None, mpp_positions=None, training=None,
              attention_pattern=None, valid_len=None):
    outputs = dict(self.encoder(word_ids=word_ids, segment_ids=segment_ids, att_mask=att_mask,
                                relative_att_ids=relative_att_ids,
                                patch_embeddings=patch_embeddings, training=training,
                                attention_pattern=attention_pattern, valid_len=valid_len))
    seq = outputs['sequence_output']
    # every head reads a few rows of the sequence output: pick them with one merged gather
    B = seq.shape[0]
    sets, names = [], []
    if mlm_positions is not None:
      sets.append(mlm_positions); names.append('/mlm')
    if mpp_positions is not None:
      sets.append(mpp_positions); names.append('/mpp')
    for head in self.classification_heads:
      sets.append(int(head.cls_token_idx))
      names.append(head.name)
    rows = dict(zip(names, layers.gather_rows_merged(seq, sets))) if sets else {}
    if mlm_positions is not None:
      outputs['mlm_logits'] = self.masked_lm(seq, masked_positions=mlm_positions, gathered=rows['/mlm'])
    if mpp_positions is not None:
      outputs['mpp_logits'] = self.masked_pp(seq, masked_positions=mpp_positions, gathered=rows['/mpp'])
    for head in self.classification_heads:
      outputs[f'{head.name}_logits'] = head(seq, training=bool(training), gathered=rows[head.name])
    return outputs

  @property
  def checkpoint_items(self):
    items = dict(encoder=self.encoder, masked_lm=self.masked_lm, masked_pp=self.masked_pp)
    for head in self.classification_heads:
      for key, item in head.checkpoint_items.items():
        items[f'{head.name}.{key}'] = item
    return items


class MmtClassificationModel(nn.Module):

  def __init__(self, encoder: MmtEncoder, classification_heads: List[layers.ClassificationHead],
               name: str = 'mmt_classification_model'):
    super().__init__()
    self.name = name
    self.encoder = encoder
    self.classification_heads = nn.ModuleList(classification_heads)
    _check_unique(self.classification_heads)

  def forward(self, word_ids, segment_ids=None, att_mask=None, relative_att_ids=None,
              patch_embeddings=None, training=None, attention_pattern=None, valid_len=None):
    outputs = dict(self.encoder(word_ids=word_ids, segment_ids=segment_ids, att_mask=att_mask,
                                relative_att_ids=relative_att_ids,
                                patch_embeddings=patch_embeddings, training=training,
                                attention_pattern=attention_pattern, valid_len=valid_len))
    for head in self.classification_heads:
      outputs[f'{head.name}_logits'] = head(outputs['sequence_output'], training=bool(training))
    return outputs

  @property
  def checkpoint_items(self):
    return dict(encoder=self.encoder, classification_heads=self.classification_heads)
