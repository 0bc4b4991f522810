"""`MmtEncoder` on MI355X: the reference's encoder surface (`src/modeling/models/mmt_encoder.py`)
over the HIP relative-attention path.

Same constructor arguments (`mmt_encoder.py:45-65`), same argument errors (`:69-80`), same
call contract `encoder(word_ids, segment_ids, att_mask, relative_att_ids, patch_embeddings,
training) -> {'sequence_output'}` (`:166-172,226-227`), same accessors (`:239-261`).
Extension: `attention_pattern` / `valid_len` run the structured fast path (mask and relative
ids generated in-kernel, nothing [S,S]-shaped is ever materialised).
"""
from __future__ import annotations

import collections
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused, layers
from .ops import AttentionPattern

_NUM_OTHER_RELATIVE_IDS = 3   # mmt_encoder.py:25


class MmtEncoder(nn.Module):

  def __init__(self, vocab_size: int, segment_vocab_size: int = 16, embedding_size: int = None,
               hidden_size: int = 768, num_hidden_layers: int = 12, num_attention_heads: int = 12,
               intermediate_size: int = 3072, inner_activation='gelu',
               hidden_dropout_prob: float = 0.1, attention_probs_dropout_prob: float = 0.1,
               max_absolute_position_embeddings: Optional[int] = None,
               relative_vocab_size: int = 32, relative_pos_max_distance: int = 12,
               initializer_range: float = 0.02, use_pre_activation_order: bool = False,
               use_one_hot_lookup: bool = True, use_pooler_layer: bool = False,
               patch_embedding_size: int = 768, compute_dtype: torch.dtype = torch.float32,
               name: str = 'mmt_encoder'):
    super().__init__()
    if relative_vocab_size is None:
      if relative_pos_max_distance != 0:
        raise ValueError('`relative_pos_max_distance` must be 0 when `relative_vocab_size` '
                         'is None.')
    elif relative_vocab_size < 2 * relative_pos_max_distance + 1 + _NUM_OTHER_RELATIVE_IDS:
      raise ValueError(f'`relative_vocab_size` ({relative_vocab_size}) too small for '
                       f'`relative_pos_max_distance` ({relative_pos_max_distance}')
    if embedding_size is None:
      embedding_size = hidden_size
    self.name = name
    self.compute_dtype = compute_dtype
    activation = layers.get_activation(inner_activation)

    self._word_embedding_layer = layers.EmbeddingLookup(
        vocab_size, embedding_size, projection_size=hidden_size,
        initializer_range=initializer_range, name='word_embeddings')
    if max_absolute_position_embeddings is None:
      self._position_embeddings = None
    else:
      self._position_embeddings = nn.Parameter(torch.empty(max_absolute_position_embeddings, hidden_size))
      layers.truncated_normal_(self._position_embeddings, initializer_range)
    self._segment_embedding_layer = layers.EmbeddingLookup(
        segment_vocab_size, embedding_size, projection_size=hidden_size,
        initializer_range=initializer_range, use_one_hot_lookup=use_one_hot_lookup,
        name='segment_embeddings')
    self._patch_projection_weight = nn.Parameter(torch.empty(hidden_size, patch_embedding_size))
    layers.truncated_normal_(self._patch_projection_weight, initializer_range)
    self._patch_projection_bias = nn.Parameter(torch.zeros(hidden_size))
    self._embedding_norm_layer = nn.LayerNorm(hidden_size, eps=1e-12)
    self._hidden_dropout_prob = hidden_dropout_prob
    self._transformer_layers = layers.RelativeTransformerLayers(
        hidden_size=hidden_size, num_hidden_layers=num_hidden_layers,
        num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
        hidden_act=activation, hidden_dropout_prob=hidden_dropout_prob,
        attention_probs_dropout_prob=attention_probs_dropout_prob,
        initializer_range=initializer_range, relative_vocab_size=relative_vocab_size,
        use_pre_activation_order=use_pre_activation_order, use_one_hot_lookup=use_one_hot_lookup)
    if use_pooler_layer:
      self._pooler_weight = nn.Parameter(torch.empty(hidden_size, hidden_size))
      layers.truncated_normal_(self._pooler_weight, initializer_range)
      self._pooler_bias = nn.Parameter(torch.zeros(hidden_size))
    config_dict = {
        'vocab_size': vocab_size, 'segment_vocab_size': segment_vocab_size,
        'hidden_size': hidden_size, 'num_hidden_layers': num_hidden_layers,
        'num_attention_heads': num_attention_heads, 'intermediate_size': intermediate_size,
        'inner_activation': inner_activation if isinstance(inner_activation, str) else 'custom',
        'hidden_dropout_prob': hidden_dropout_prob,
        'attention_probs_dropout_prob': attention_probs_dropout_prob,
        'max_absolute_position_embeddings': max_absolute_position_embeddings,
        'relative_vocab_size': relative_vocab_size,
        'relative_pos_max_distance': relative_pos_max_distance,
        'initializer_range': initializer_range,
        'use_pre_activation_order': use_pre_activation_order,
        'use_one_hot_lookup': use_one_hot_lookup, 'use_pooler_layer': use_pooler_layer,
    }
    self._config = collections.namedtuple('Config', config_dict.keys())(**config_dict)
    self.use_fused_embedding = True

  def _fused_embed_ok(self, word_ids):
    w, sg = self._word_embedding_layer, self._segment_embedding_layer
    H = w.embedding_table.shape[1]
    return (word_ids.is_cuda and w.embedding_projection is None and sg.embedding_projection is None
            and self.compute_dtype in (torch.float32, torch.bfloat16) and H % 8 == 0 and H <= 2048
            and w.embedding_table.dtype == torch.float32 and sg.embedding_table.dtype == torch.float32
            and sg.embedding_table.shape[1] == H and self.use_fused_embedding)

  def embed(self, word_ids, segment_ids=None, patch_embeddings=None, training=False):
    """Embedding assembly, `mmt_encoder.py:189-218`: LayerNorm + dropout on
    the WORD embeddings only; segment / position / projected patches are added afterwards.
    On the GPU this is one HIP kernel (`fused.embed_assemble`) writing the compute dtype; the
    patch projection runs in the compute dtype like every other Dense layer of the stack."""
    if segment_ids is None:
      segment_ids = torch.ones_like(word_ids)
    ln = self._embedding_norm_layer
    if self._fused_embed_ok(word_ids):
      cd = self.compute_dtype
      pe = None
      if patch_embeddings is not None:
        pe = layers._linear(patch_embeddings.to(cd), self._patch_projection_weight, self._patch_projection_bias)
        if word_ids.shape[1] < 2 + pe.shape[1]:
          raise ValueError('sequence too short for [CLS], [PATCH] and the patches')
      p = self._hidden_dropout_prob if training else 0.0
      return fused.embed_assemble(word_ids, segment_ids, self._word_embedding_layer.embedding_table,
                                  self._segment_embedding_layer.embedding_table, ln.weight, ln.bias,
                                  pos_table=self._position_embeddings, patch_proj=pe, eps=ln.eps, p=p,
                                  seed=fused.next_seed(0) if p else 0, patch_start=2, out_dtype=cd)
    word = self._word_embedding_layer(word_ids)
    seg = self._segment_embedding_layer(segment_ids)
    word = F.layer_norm(word, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
    word = F.dropout(word, self._hidden_dropout_prob, training)
    emb = word + seg
    S = word.shape[1]
    if self._position_embeddings is not None:
      emb = emb + self._position_embeddings[:S]
    if patch_embeddings is not None:
      pe = F.linear(patch_embeddings.to(emb.dtype), self._patch_projection_weight,
                    self._patch_projection_bias)
      n_patch = pe.shape[1]
      # 2 is for CLS and [PATCH] (`:213-218`): patches live at [2, 2 + n_patch)
      emb = emb + F.pad(pe, (0, 0, 2, S - 2 - n_patch))
    return emb

  def forward(self, word_ids, segment_ids=None, att_mask=None, relative_att_ids=None,
              patch_embeddings=None, training: Optional[bool] = None,
              attention_pattern: Optional[AttentionPattern] = None, valid_len=None):
    training = bool(training)
    emb = self.embed(word_ids, segment_ids, patch_embeddings, training).to(self.compute_dtype)
    out = self._transformer_layers(inputs=emb, att_mask=att_mask, relative_att_ids=relative_att_ids,
                                   training=training, pattern=attention_pattern, valid_len=valid_len,
                                   dropout_seed=(fused.next_seed(0) >> 24) if training else 0)
    outputs = {'sequence_output': out}
    if hasattr(self, '_pooler_weight'):
      first = out[:, 0]
      outputs['pooled_output'] = torch.tanh(layers._linear(first, self._pooler_weight, self._pooler_bias))
    return outputs

  # accessors, `mmt_encoder.py:239-261`
  def get_word_embedding_table(self):
    return self._word_embedding_layer.embedding_table

  def get_word_embedding_layer(self):
    return self._word_embedding_layer

  def get_config(self):
    return dict(self._config._asdict())

  @property
  def transformer_layers(self):
    return self._transformer_layers

  @property
  def pooler_layer(self):
    if hasattr(self, '_pooler_weight'):
      return self._pooler_weight
    raise ValueError('pooler layers is not initialized.')

  @classmethod
  def from_config(cls, config, custom_objects=None):
    return cls(**config)
