"""Device-side feature pipeline of the data layer's contract: the per-example
tensor transforms the reference runs inside tf.data on the host, as batched torch ops on the GPU
(integer / byte work: gathers, reshapes, bucketize -- nothing here is a GEMM).

  convert_image_to_patches   src/data/data_utils.py:147-180  (tf.image.extract_patches 16x16 VALID + raster order)
  make_mpp_label_ids         src/data/data_utils.py:448-481  (mean colour per channel -> 2^bits bins -> base-2^bits id)
  make_matching_features     src/data/data_utils.py:642-712  (in-batch negatives for ITM: tile images, roll texts)

File formats, JPEG decode, WordPiece and RandAugment stay out of scope."""
from __future__ import annotations

from typing import Dict

import torch


def convert_image_to_patches(images: torch.Tensor, patch_size: int) -> torch.Tensor:
  """[B, H, W, C] -> [B, P*P, patch_size*patch_size*C], P = H // patch_size, raster scan; inside a
  patch the order is (row, column, channel), as tf.image.extract_patches flattens it."""
  if images.dim() != 4:
    raise ValueError('images must be [batch, height, width, channels]')
  B, H, W, C = images.shape
  if H != W:
    raise ValueError('square images expected (image_size x image_size)')
  P = H // patch_size
  x = images[:, :P * patch_size, :P * patch_size]           # VALID padding: the remainder is dropped
  x = x.reshape(B, P, patch_size, P, patch_size, C).permute(0, 1, 3, 2, 4, 5)
  return x.reshape(B, P * P, patch_size * patch_size * C)


def make_mpp_label_ids(mpp_embeddings: torch.Tensor, patch_size: int, channels: int = 3,
                       output_channel_bits: int = 3, max_pixel_val: int = 256) -> torch.Tensor:
  """[..., patch_size^2 * channels] pixel values in [0, 1] -> int32 class ids in [0, 2^(bits*channels)):
  the mean of each channel, scaled to 0..255, is bucketized into 2^bits equal bins and the per-channel
  bins are combined as digits of base 2^bits (channel 0 least significant)."""
  lead = mpp_embeddings.shape[:-1]
  bin_size = max_pixel_val // (2 ** output_channel_bits)
  x = mpp_embeddings.to(torch.float32) * (max_pixel_val - 1)
  avg = x.reshape(*lead, patch_size * patch_size, channels).mean(dim=-2)
  bins = torch.arange(bin_size, max_pixel_val, bin_size, device=avg.device, dtype=torch.float32)
  digit = torch.bucketize(avg, bins, right=True)             # boundaries[i-1] <= x < boundaries[i]
  weight = (2 ** output_channel_bits) ** torch.arange(channels, device=avg.device)
  return (digit * weight).sum(-1).to(torch.int32)


def make_matching_features(features: Dict[str, torch.Tensor], image_keys: torch.Tensor,
                           negative_positive_ratio: int = 1, min_shift: int = 5) -> Dict[str, torch.Tensor]:
  """In-batch negatives for image-text matching.  The batch is sorted by image (equal images adjacent),
  image-side features are tiled `ratio + 1` times, text-side features of copy i are rolled by
  `min_shift + i` examples, and copy 0 is the positive set: itm_label_ids = 1 for the first batch_size
  examples, itm_pos_weights = 1 + label * (ratio - 1)."""
  B = image_keys.shape[0]
  if not B > negative_positive_ratio + 1 + min_shift:
    raise ValueError('batch_size must exceed negative_positive_ratio + 1 + min_shift')
  if negative_positive_ratio <= 0:
    raise ValueError('negative_positive_ratio must be positive')
  _, idx = torch.unique(image_keys, return_inverse=True)     # ids in order of first appearance are not
  first = torch.full((int(idx.max()) + 1,), B, dtype=torch.long, device=idx.device)   # guaranteed: re-rank
  first.scatter_reduce_(0, idx, torch.arange(B, device=idx.device), reduce='amin')
  rank = torch.argsort(torch.argsort(first))[idx]            # tf.unique numbering: by first appearance
  order = torch.argsort(rank, stable=True)
  out = {k: v[order] for k, v in features.items()}
  copies = negative_positive_ratio + 1
  for k in ('patch_token_ids', 'patch_embeddings', 'num_image_wordpieces'):
    if k in out:
      out[k] = out[k].repeat(copies, *([1] * (out[k].dim() - 1)))
  base = torch.arange(B, device=image_keys.device)
  perm = torch.cat([base] + [torch.roll(base, shifts=min_shift + i) for i in range(1, copies)])
  for k in ('text_token_ids', 'num_text_wordpieces', 'mlm_positions', 'mlm_label_ids', 'mlm_label_weights',
            'mpp_positions', 'mpp_label_ids', 'mpp_label_weights'):
    if k in out:
      out[k] = out[k][perm]
  label = torch.zeros(B * copies, device=image_keys.device)
  label[:B] = 1.0
  out['itm_label_ids'] = label.to(torch.int32)
  out['itm_label_weights'] = torch.ones_like(label)
  out['itm_pos_weights'] = 1.0 + label * (negative_positive_ratio - 1)
  return out


def matching_batch_size(data_config, batch_size_per_replica: int) -> int:
  """Examples the matching step needs in hand so that no roll creates a false negative
  (`MmtClassificationDataLoader.load`, src/data/classification_dataloader.py:132-137)."""
  max_shift = int(data_config.negative_positive_ratio) + int(data_config.min_shift)
  return (max_shift // batch_size_per_replica + 2) * batch_size_per_replica


def make_matching_features_from_config(data_config, features: Dict[str, torch.Tensor],
                                       image_keys: torch.Tensor) -> Dict[str, torch.Tensor]:
  """`get_matching_fn(config, batch, config.negative_positive_ratio)` of the classification loader
  (classification_dataloader.py:138-140): the data config's `negative_positive_ratio` and `min_shift` drive the
  in-batch negatives (pretraining fixes the ratio at 1, pretrain_dataloader.py:184)."""
  return make_matching_features(features, image_keys,
                                negative_positive_ratio=int(getattr(data_config, 'negative_positive_ratio', 1)),
                                min_shift=int(data_config.min_shift))


def make_retrieval_labels(features: Dict[str, torch.Tensor], pos_weight: float = 1.0) -> Dict[str, torch.Tensor]:
  """`get_retrieval_label_fn` (src/data/data_utils.py:744-760): label 1 where the image is the text's ground-truth
  image, weight `pos_weight` on those pairs and 1 elsewhere (`MmtRetrievalDataConfig.pos_weight`)."""
  label = (features['image_index'] == features['gt_image_index']).to(torch.int32)
  out = dict(features)
  out['label_ids'] = label
  out['label_weights'] = label.to(torch.float32) * (float(pos_weight) - 1.0) + 1.0
  return out


# ---------------------------------------------------------------------------------------------------
# MLM / MPP masking (`get_masking_fn` -> `make_mlm_and_mpp_features`, src/data/data_utils.py:383-639),
# batched on the device.  The reference calls tf_text.mask_language_model with a RandomItemSelector and a
# MaskValuesChooser (tensorflow_text 2.5.0, src/requirements.txt); their published algorithm is restated
# here with the RANDOM DRAWS AS EXPLICIT INPUTS (one shuffle key and one value-choice uniform per item, one
# replacement id per token), so that the index outputs are a deterministic, bit-exact function of them.
# ---------------------------------------------------------------------------------------------------
def random_item_masking(token_ids: torch.Tensor, n_tokens: torch.Tensor, *, selection_rate: float,
                        max_selections: int, unselectable_ids, mask_token_id: int, item_keys: torch.Tensor,
                        value_u: torch.Tensor, random_ids: torch.Tensor, word_start: torch.Tensor = None,
                        mask_token_rate: float = 0.8, random_token_rate: float = 0.1):
  """tf_text.mask_language_model(ids, RandomItemSelector(max_selections, selection_rate, unselectable_ids),
  MaskValuesChooser(vocab, mask_token_id, mask_token_rate)) for a padded batch.

  token_ids [B,T] int32 (entries at or beyond n_tokens[b] are padding); word_start [B,T] bool marks the first
  wordpiece of every item (None: every token is an item -- `mlm_use_whole_word=False`, data_utils.py:598-600).
  item_keys / value_u [B,T] float32 in [0,1): entry i belongs to item i of the row; random_ids [B,T] int32.
  Items with any unselectable wordpiece are never chosen; of the n selectable items
  min(ceil(n * selection_rate), max_selections) are chosen -- those with the smallest shuffle keys; a chosen
  item becomes [MASK] when its value_u < mask_token_rate, random ids when < mask_token_rate +
  random_token_rate, and stays as it is otherwise.
  Returns (masked_token_ids [B,T], positions [B,T] (ascending, first n_masked valid, rest 0), n_masked [B],
  original ids at those positions [B,T])."""
  B, T = token_ids.shape
  dev = token_ids.device
  pos = torch.arange(T, device=dev)[None]
  valid = pos < n_tokens[:, None]
  if word_start is None:
    item = pos.expand(B, T)
  else:
    item = torch.cumsum((word_start & valid).to(torch.int64), 1) - 1
  item = torch.where(valid, item, torch.full_like(item, T - 1)).clamp_(min=0)
  unsel_tok = torch.zeros_like(valid)
  for u in unselectable_ids:
    unsel_tok |= token_ids == int(u)
  unsel_tok &= valid
  item_unsel = torch.zeros(B, T, dtype=torch.int32, device=dev).scatter_reduce_(
      1, item, unsel_tok.to(torch.int32), reduce='amax', include_self=True).bool()
  item_exists = torch.zeros(B, T, dtype=torch.int32, device=dev).scatter_reduce_(
      1, item, valid.to(torch.int32), reduce='amax', include_self=True).bool()
  selectable = item_exists & ~item_unsel
  n_selectable = selectable.sum(1)
  n_sel = torch.minimum(torch.ceil(n_selectable.to(torch.float32) * torch.tensor(selection_rate, dtype=torch.float32, device=dev)),
                        torch.tensor(float(max_selections), device=dev)).to(torch.int64)
  keys = torch.where(selectable, item_keys.to(torch.float32), torch.full_like(item_keys, float('inf'), dtype=torch.float32))
  rank = torch.argsort(torch.argsort(keys, dim=1, stable=True), dim=1, stable=True)
  item_chosen = selectable & (rank < n_sel[:, None])
  tok_chosen = torch.gather(item_chosen, 1, item) & valid
  u = torch.gather(value_u.to(torch.float32), 1, item)
  to_mask = u < mask_token_rate
  to_rand = ~to_mask & (u < mask_token_rate + random_token_rate)
  new_tok = torch.where(to_mask, torch.full_like(token_ids, mask_token_id), torch.where(to_rand, random_ids.to(token_ids.dtype), token_ids))
  masked = torch.where(tok_chosen, new_tok, token_ids)
  order = torch.argsort((~tok_chosen).to(torch.int8), dim=1, stable=True)      # chosen positions first, ascending
  n_masked = tok_chosen.sum(1)
  live = pos < n_masked[:, None]
  positions = torch.where(live, order, torch.zeros_like(order)).to(torch.int32)
  labels = torch.where(live, torch.gather(token_ids, 1, order), torch.zeros_like(token_ids))
  return masked, positions, n_masked, labels


def _pad_or_fail(x: torch.Tensor, n_live: torch.Tensor, width: int, what: str) -> torch.Tensor:
  """tensor_utils.pad_to_max_seq_len on the compacted [B,T] rows: tf.pad cannot shorten, so more live entries
  than `width` is an error in the reference as well."""
  if x.shape[1] >= width:
    if int(n_live.max()) > width:
      raise ValueError(f'{what}: more masked positions than max selections ({int(n_live.max())} > {width})')
    return x[:, :width]
  return torch.nn.functional.pad(x, (0, width - x.shape[1]))


def make_mlm_and_mpp_features(features: Dict[str, torch.Tensor], randoms: Dict[str, torch.Tensor], *,
                              max_seq_len: int, num_patches: int, patch_size: int, vocab_size: int,
                              mask_token_id: int, unselectable_ids, mlm_fraction_to_mask: float = 0.15,
                              mpp_fraction_to_mask: float = 0.5, mlm_max_selections_per_seq: int = 256,
                              mpp_max_selections_per_seq: int = 98, patch_mask_token_id: int = None,
                              channels: int = 3, output_channel_bits: int = 3, max_pixel_val: int = 256):
  """`make_mlm_and_mpp_features` (data_utils.py:507-637) + `make_word_ids_features` (:728-741), batched.

  features: patch_token_ids [B, 2+P] ([CLS] [PATCH] patch tokens), text_token_ids [B,T] + num_text_wordpieces
  [B] (+ optional text_word_start [B,T] for whole-word masking), patch_embeddings [B,P,E],
  unnormalized_patch_embeddings [B,P,E].  randoms: {mlm,mpp}_{item_keys,value_u,random_ids}.
  Returns the new features: word_ids [B,S], patch_token_ids, text_token_ids [B, S-2-P], patch_embeddings (rows
  of [MASK]ed patches zeroed), mlm_/mpp_ positions, label_ids, label_weights."""
  out = dict(features)
  B = features['patch_token_ids'].shape[0]
  dev = features['patch_token_ids'].device
  pm = mask_token_id if patch_mask_token_id is None else patch_mask_token_id
  mlm_max = min(mlm_max_selections_per_seq, max_seq_len)
  # ---- patches (:521-588)
  ptok = features['patch_token_ids']
  n_ptok = torch.full((B,), ptok.shape[1], device=dev, dtype=torch.int64)
  mpp_tok, mpp_pos, n_mpp, _ = random_item_masking(
      ptok, n_ptok, selection_rate=mpp_fraction_to_mask, max_selections=mpp_max_selections_per_seq,
      unselectable_ids=unselectable_ids, mask_token_id=pm, item_keys=randoms['mpp_item_keys'],
      value_u=randoms['mpp_value_u'], random_ids=randoms['mpp_random_ids'])
  mpp_pos = _pad_or_fail(mpp_pos, n_mpp, mpp_max_selections_per_seq, 'mpp')
  live = torch.arange(mpp_max_selections_per_seq, device=dev)[None] < n_mpp[:, None]
  shifted = (mpp_pos.to(torch.int64) - 2).clamp_(min=0)                                  # -2: [CLS] and [PATCH]
  unnorm = features['unnormalized_patch_embeddings']
  emb = torch.gather(unnorm, 1, shifted[..., None].expand(-1, -1, unnorm.shape[-1]))
  lab = make_mpp_label_ids(emb, patch_size, channels, output_channel_bits, max_pixel_val)
  out['mpp_label_ids'] = torch.where(live, lab, torch.zeros_like(lab))
  num_real = (mpp_tok == pm).sum(1)                                                       # get_masked_weights (:483-505)
  idx = torch.arange(mpp_max_selections_per_seq, device=dev)[None]
  out['mpp_label_weights'] = ((idx < num_real[:, None]) & live).to(torch.float32)
  out['mpp_positions'] = mpp_pos
  keep = (mpp_tok[:, 2:2 + num_patches] != pm).to(features['patch_embeddings'].dtype)   # zero the masked patches (:577-585)
  out['patch_embeddings'] = features['patch_embeddings'] * keep[..., None]
  out['patch_token_ids'] = mpp_tok
  # ---- text (:590-636)
  ttok = features['text_token_ids']
  mlm_tok, mlm_pos, n_mlm, mlm_lab = random_item_masking(
      ttok, features['num_text_wordpieces'].to(torch.int64), selection_rate=mlm_fraction_to_mask,
      max_selections=mlm_max, unselectable_ids=unselectable_ids, mask_token_id=mask_token_id,
      item_keys=randoms['mlm_item_keys'], value_u=randoms['mlm_value_u'], random_ids=randoms['mlm_random_ids'],
      word_start=features.get('text_word_start'))
  tlive = torch.arange(ttok.shape[1], device=dev)[None] < n_mlm[:, None]
  mlm_pos = torch.where(tlive, mlm_pos + (2 + num_patches), torch.zeros_like(mlm_pos))     # offset before the padding
  out['mlm_positions'] = _pad_or_fail(mlm_pos, n_mlm, mlm_max, 'mlm')
  out['mlm_label_ids'] = _pad_or_fail(mlm_lab, n_mlm, mlm_max, 'mlm')
  valid_t = torch.arange(ttok.shape[1], device=dev)[None] < features['num_text_wordpieces'][:, None]
  num_real_t = ((mlm_tok == mask_token_id) & valid_t).sum(1)
  out['mlm_label_weights'] = (torch.arange(mlm_max, device=dev)[None] < num_real_t[:, None]).to(torch.float32)
  max_remaining = max_seq_len - num_patches - 2
  text = torch.where(valid_t, mlm_tok, torch.zeros_like(mlm_tok))
  text = text[:, :max_remaining] if text.shape[1] >= max_remaining else torch.nn.functional.pad(text, (0, max_remaining - text.shape[1]))
  out['text_token_ids'] = text
  out['word_ids'] = torch.cat([mpp_tok, text], dim=1)                                        # make_word_ids_features
  return out
