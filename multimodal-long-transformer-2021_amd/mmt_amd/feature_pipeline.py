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
