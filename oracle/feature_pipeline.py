"""ORACLE (test infrastructure, not product code) -- numpy restatement of the reference's host-side
feature transforms, per example / per batch, with plain loops where that is the clearest reading:

  convert_image_to_patches  src/data/data_utils.py:147-180
  make_mpp_label_ids        src/data/data_utils.py:448-481
  make_matching_features    src/data/data_utils.py:642-712

tf.image.extract_patches, math_ops._bucketize, tf.unique and tf.roll are TensorFlow ops that cannot run here
(TF is not installed); their documented semantics are restated: extract_patches flattens a patch as
(row, column, channel); _bucketize(x, b) = number of boundaries <= x; tf.unique numbers values by first
appearance; tf.roll shifts towards higher indices.  PARITY UNPINNED beyond those documented semantics: the
reference holds no test or golden vector for these functions."""
import numpy as np


def convert_image_to_patches(image, patch_size):
  H, W, C = image.shape
  P = H // patch_size
  out = np.zeros((P * P, patch_size * patch_size * C), dtype=image.dtype)
  for pi in range(P):
    for pj in range(P):
      patch = image[pi * patch_size:(pi + 1) * patch_size, pj * patch_size:(pj + 1) * patch_size, :]
      out[pi * P + pj] = patch.reshape(-1)            # (row, column, channel); raster scan over patches
  return out


def make_mpp_label_ids(mpp_embeddings, patch_size, channels=3, output_channel_bits=3, max_pixel_val=256):
  bin_size = max_pixel_val // (2 ** output_channel_bits)
  n = mpp_embeddings.shape[0]
  x = mpp_embeddings.astype(np.float32) * np.float32(max_pixel_val - 1)
  avg = x.reshape(n, patch_size ** 2, channels).mean(axis=1, dtype=np.float32)
  bins = list(range(bin_size, max_pixel_val, bin_size))
  digit = np.zeros((n, channels), dtype=np.int64)
  for b in bins:
    digit += (avg >= b)
  weight = (2 ** output_channel_bits) ** np.arange(channels)
  return (digit * weight[None]).sum(1).astype(np.int32)


def make_matching_features(features, image_keys, negative_positive_ratio=1, min_shift=5):
  B = len(image_keys)
  seen, idx = {}, []
  for k in image_keys.tolist():
    seen.setdefault(k, len(seen))
    idx.append(seen[k])
  order = np.argsort(np.asarray(idx), kind='stable')
  out = {k: v[order] for k, v in features.items()}
  copies = negative_positive_ratio + 1
  for k in ('patch_token_ids', 'patch_embeddings', 'num_image_wordpieces'):
    if k in out:
      out[k] = np.concatenate([out[k]] * copies, axis=0)
  perms = [np.arange(B)]
  for i in range(1, copies):
    perms.append(np.roll(np.arange(B), min_shift + i))
  perm = np.concatenate(perms)
  for k in ('text_token_ids', 'num_text_wordpieces', 'mlm_positions', 'mlm_label_ids', 'mlm_label_weights',
            'mpp_positions', 'mpp_label_ids', 'mpp_label_weights'):
    if k in out:
      out[k] = out[k][perm]
  label = np.zeros(B * copies, dtype=np.float32)
  label[:B] = 1.0
  out['itm_label_ids'] = label.astype(np.int32)
  out['itm_label_weights'] = np.ones_like(label)
  out['itm_pos_weights'] = 1.0 + label * (negative_positive_ratio - 1)
  return out


# ---------------------------------------------------------------------------------------------------
# MLM / MPP masking: src/data/data_utils.py:383-639 (+ make_word_ids_features :728-741), one example at a time.
# The selection itself lives in tensorflow_text 2.5.0 (src/requirements.txt), absent from /root/reference and
# from this image; its published algorithm is restated (PARITY UNPINNED for these details: the reference holds
# no test of its data pipeline):
#   RandomItemSelector.get_selection_mask : items (axis=1: words when whole-word masking, else wordpieces) that
#       contain an unselectable id are dropped; num_to_select = min(ceil(n_selectable * selection_rate),
#       max_selections_per_batch); the selectable items are shuffled and the first num_to_select taken;
#   MaskValuesChooser.get_mask_values      : ONE uniform draw per chosen item: < mask_token_rate (0.8) -> every
#       wordpiece becomes mask_token; < mask_token_rate + random_token_rate (0.1) -> random ids; else unchanged;
#   mask_language_model                    : returns the masked ids, the flat positions of the chosen wordpieces
#       in ascending order and the original ids at those positions.
# TF's random streams cannot be reproduced, so the random draws are explicit inputs (shuffle = ascending order
# of one key per item): the outputs are then a deterministic function of them and compared bit for bit.
# ---------------------------------------------------------------------------------------------------
def random_item_masking(tokens, word_start, selection_rate, max_selections, unselectable_ids, mask_token_id,
                        item_keys, value_u, random_ids, mask_token_rate=0.8, random_token_rate=0.1):
  n = len(tokens)
  items = []                              # lists of token indices: one per word (whole-word masking) or per wordpiece
  for i in range(n):
    if i == 0 or word_start is None or word_start[i]:
      items.append([])
    items[-1].append(i)
  selectable = [j for j, it in enumerate(items) if not any(int(tokens[i]) in unselectable_ids for i in it)]
  num_to_select = min(int(np.ceil(np.float32(len(selectable)) * np.float32(selection_rate))), max_selections)
  shuffled = sorted(selectable, key=lambda j: (np.float32(item_keys[j]), j))
  chosen = sorted(shuffled[:num_to_select])
  masked = np.array(tokens, copy=True)
  positions = []
  for j in chosen:
    u = np.float32(value_u[j])
    for i in items[j]:
      positions.append(i)
      if u < mask_token_rate:
        masked[i] = mask_token_id
      elif u < mask_token_rate + random_token_rate:
        masked[i] = random_ids[i]
  positions = np.array(sorted(positions), dtype=np.int32)
  return masked, positions, np.asarray(tokens)[positions] if len(positions) else np.zeros((0,), tokens.dtype)


def _pad_to(x, width):
  """tensor_utils.pad_to_max_seq_len: tf.pad with max_seq_len - len zeros (a negative amount is an error)."""
  if len(x) > width:
    raise ValueError('pad_to_max_seq_len cannot shorten')
  return np.concatenate([x, np.zeros((width - len(x),), x.dtype)])


def make_mlm_and_mpp_features(ex, rnd, max_seq_len, num_patches, patch_size, vocab_size, mask_token_id,
                              unselectable_ids, mlm_fraction_to_mask=0.15, mpp_fraction_to_mask=0.5,
                              mlm_max_selections_per_seq=256, mpp_max_selections_per_seq=98,
                              patch_mask_token_id=None, channels=3, output_channel_bits=3, max_pixel_val=256):
  """ex: patch_token_ids [2+P], text_token_ids [n_text] (+ text_word_start), patch_embeddings [P,E],
  unnormalized_patch_embeddings [P,E]; rnd: {mlm,mpp}_{item_keys,value_u,random_ids} for this example."""
  pm = mask_token_id if patch_mask_token_id is None else patch_mask_token_id
  mlm_max = min(mlm_max_selections_per_seq, max_seq_len)
  out = {}
  mpp_tok, mpp_pos, _ = random_item_masking(ex['patch_token_ids'], None, mpp_fraction_to_mask, mpp_max_selections_per_seq,
                                            unselectable_ids, pm, rnd['mpp_item_keys'], rnd['mpp_value_u'], rnd['mpp_random_ids'])
  n_masked = len(mpp_pos)
  emb = ex['unnormalized_patch_embeddings'][mpp_pos - 2] if n_masked else np.zeros((0, ex['unnormalized_patch_embeddings'].shape[1]), np.float32)
  lab = make_mpp_label_ids(emb, patch_size, channels, output_channel_bits, max_pixel_val) if n_masked else np.zeros((0,), np.int32)
  num_real = int((mpp_tok == pm).sum())
  w = (np.arange(n_masked) < num_real).astype(np.float32)            # get_masked_weights over patch_masked_seq_len
  out['mpp_positions'] = _pad_to(mpp_pos, mpp_max_selections_per_seq)
  out['mpp_label_ids'] = _pad_to(lab, mpp_max_selections_per_seq)
  out['mpp_label_weights'] = _pad_to(w, mpp_max_selections_per_seq)
  keep = (mpp_tok[2:2 + num_patches] != pm).astype(ex['patch_embeddings'].dtype)
  out['patch_embeddings'] = ex['patch_embeddings'] * keep[:, None]
  out['patch_token_ids'] = mpp_tok
  mlm_tok, mlm_pos, mlm_lab = random_item_masking(ex['text_token_ids'], ex.get('text_word_start'), mlm_fraction_to_mask, mlm_max,
                                                  unselectable_ids, mask_token_id, rnd['mlm_item_keys'], rnd['mlm_value_u'],
                                                  rnd['mlm_random_ids'])
  out['mlm_positions'] = _pad_to((mlm_pos + 2 + num_patches).astype(np.int32), mlm_max)
  out['mlm_label_ids'] = _pad_to(mlm_lab, mlm_max)
  num_real_t = int((mlm_tok == mask_token_id).sum())
  out['mlm_label_weights'] = (np.arange(mlm_max) < num_real_t).astype(np.float32)     # over the PADDED length (:627-631)
  out['text_token_ids'] = _pad_to(mlm_tok, max_seq_len - num_patches - 2)
  out['word_ids'] = _pad_to(np.concatenate([out['patch_token_ids'], out['text_token_ids']]), max_seq_len)
  return out
