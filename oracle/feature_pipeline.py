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
