"""ORACLE (test infrastructure, not product code) -- integer side inputs.

CPU/numpy restatement of the reference's attention side-input generators.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import this module; the product path never routes through it.

Parity status: PINNED.  `make_relative_att_ids` reproduces both golden
matrices of the reference's own test (`src/feature_utils_test.py:64-72` and
`:95-108`, committed under `tests/golden/`), and those matrices also pin the
recalled 1-D convention of the un-vendored `etcmodel` generator through their
text blocks.

Reference files followed (read as text; TF is not importable here):
  * `src/feature_utils.py:29-255`     MmtRelativePositionGenerator
  * `src/data/data_utils.py:285-380`  get_add_side_input_features_fn
  * etcmodel.feature_utils.RelativePositionGenerator / make_segmented_att_mask
    (third party, absent; call sites `data_utils.py:300-301,322`,
    `feature_utils.py:86-87,178-180`) -- SURVEY.md App. A.2.
"""
from __future__ import annotations

import numpy as np

# Order of the eight coarse directions; ids d*d .. d*d+7 are handed out in this
# order (`src/feature_utils.py:107-110` iterates `direction_config.values()`,
# declared at `:221-254`).
_DIRECTIONS = ('top', 'top_right', 'right', 'right_bottom', 'bottom',
               'bottom_left', 'left', 'top_left')


class RelativePositionGenerator1D:
  """etcmodel.feature_utils.RelativePositionGenerator, restated.

  id(i, j) = min(j - i, m) for j >= i, m + min(i - j, m) for j < i, so ids
  live in [0, 2m] (confirmed by the text blocks of
  `src/feature_utils_test.py:69-71,105-107`).
  """

  def __init__(self, max_distance: int):
    if max_distance < 0:
      raise ValueError('`max_distance` must not be negative.')
    self._max_distance = int(max_distance)

  @property
  def max_distance(self) -> int:
    return self._max_distance

  @property
  def relative_vocab_size(self) -> int:
    return 2 * self._max_distance + 1

  def make_relative_att_ids(self, seq_len: int, batch_size: int = 1) -> np.ndarray:
    m = self._max_distance
    pos = np.arange(seq_len, dtype=np.int64)
    dist = pos[None, :] - pos[:, None]          # j - i
    fwd = np.minimum(dist, m)
    bwd = m + np.minimum(-dist, m)
    ids = np.where(dist >= 0, fwd, bwd).astype(np.int32)
    return np.broadcast_to(ids[None], (batch_size, seq_len, seq_len)).copy()


def direction_config(num_patch_per_row: int, num_core_layers: int) -> dict:
  """Fill shape + paddings of the 8 coarse regions (`src/feature_utils.py:186-255`)."""
  d = 2 * num_core_layers + 1
  m = num_patch_per_row + num_core_layers + 1
  n = num_patch_per_row - num_core_layers
  boxes = {
      'top':          ((n, d), ((0, m), (n, n))),
      'top_right':    ((n, n), ((0, m), (m, 0))),
      'right':        ((d, n), ((n, n), (m, 0))),
      'right_bottom': ((n, n), ((m, 0), (m, 0))),
      'bottom':       ((n, d), ((m, 0), (n, n))),
      'bottom_left':  ((n, n), ((m, 0), (0, m))),
      'left':         ((d, n), ((n, n), (0, m))),
      'top_left':     ((n, n), ((0, m), (0, m))),
  }
  return {k: {'fill': list(boxes[k][0]), 'paddings': [list(p) for p in boxes[k][1]]}
          for k in _DIRECTIONS}


class MmtRelativePositionGenerator:
  """`src/feature_utils.py:29` restated with numpy (same ctor, same errors)."""

  def __init__(self, num_patch_per_row: int, num_core_layers: int,
               text_relative_pos_max_distance: int):
    # `src/feature_utils.py:60-65`
    if num_patch_per_row <= 0:
      raise ValueError('`num_patch_per_row` must be positive.')
    if num_core_layers <= 0:
      raise ValueError('`num_core_layers` must be positive.')
    if text_relative_pos_max_distance < 0:
      raise ValueError('`text_relative_pos_max_distance` must be positive.')
    self._num_patch_per_row = int(num_patch_per_row)
    self._num_core_layers = int(num_core_layers)
    self._core_layer_diameter = 2 * self._num_core_layers + 1
    text_max_id = 2 * int(text_relative_pos_max_distance) + 1
    # `:78-82` -- note P**2 (not d**2): the part ids can exceed the relative
    # vocabulary; reproduced on purpose.
    self._image_part_id = self._num_patch_per_row ** 2 + len(_DIRECTIONS) + text_max_id
    self._text_part_id = self._image_part_id + 1
    self._base_tensor = self.create_base_tensor()
    self._text_relative_generator = RelativePositionGenerator1D(
        text_relative_pos_max_distance)

  @property
  def direction_config(self) -> dict:
    return direction_config(self._num_patch_per_row, self._num_core_layers)

  def create_base_tensor(self) -> np.ndarray:
    """(d + 2n)^2 helper tensor, `src/feature_utils.py:89-112`."""
    r, d = self._num_core_layers, self._core_layer_diameter
    n = self._num_patch_per_row - r
    if n < 0:
      raise ValueError('`num_core_layers` larger than `num_patch_per_row`.')
    center = np.roll(np.arange(d * d, dtype=np.int32), d * r + r).reshape(d, d)
    base = np.pad(center, ((n, n), (n, n)))
    for idx, cfg in enumerate(self.direction_config.values(), start=d * d):
      block = np.full(cfg['fill'], idx, dtype=np.int32)
      base = base + np.pad(block, cfg['paddings'])
    return base.astype(np.int32)

  def make_relative_att_ids(self, seq_len: int, batch_size: int = 1) -> np.ndarray:
    """[1, S, S] int32 ids, `src/feature_utils.py:114-184`."""
    P = self._num_patch_per_row
    image_seq_len = P * P
    text_seq_len = int(seq_len) - image_seq_len
    if text_seq_len < 0:
      raise ValueError('`seq_len` shorter than the image part.')
    rows = []
    for x in range(P):            # `:164-170`: slide a P x P window over base
      for y in range(P):
        rows.append(self._base_tensor[P - x:2 * P - x, P - y:2 * P - y].reshape(-1))
    image_ids = np.stack(rows).astype(np.int32)
    image_ids = np.pad(image_ids, ((0, 0), (0, text_seq_len)),
                       constant_values=self._text_part_id)[None]          # `:172-176`
    text_ids = self._text_relative_generator.make_relative_att_ids(
        text_seq_len, batch_size=batch_size)
    text_ids = np.pad(text_ids, ((0, 0), (0, 0), (image_seq_len, 0)),
                      constant_values=self._image_part_id)                # `:178-183`
    return np.concatenate([image_ids, text_ids], axis=1).astype(np.int32)  # `:184`


def make_segmented_att_mask(example_ids: np.ndarray) -> np.ndarray:
  """etcmodel make_segmented_att_mask: 1 where two positions share an example id."""
  example_ids = np.asarray(example_ids)
  return (example_ids[..., :, None] == example_ids[..., None, :]).astype(np.int32)


def make_segment_ids(num_image_wordpieces: int, num_text_wordpieces: int,
                     max_seq_len: int) -> np.ndarray:
  """`src/data/data_utils.py:350-361`: 1 image part, 2 text part, 0 boundary / pad."""
  pos = np.arange(max_seq_len, dtype=np.int32)
  img = np.where(pos < num_image_wordpieces, 1, 0)
  txt_mask = (pos > num_image_wordpieces) & (pos < num_image_wordpieces + num_text_wordpieces)
  return (img + np.where(txt_mask, 2, 0)).astype(np.int32)


def add_side_input_features(num_image_wordpieces: int, num_text_wordpieces: int,
                            max_seq_len: int, relative_pos_max_distance: int,
                            relative_att_num_core_layers: int = 0,
                            image_size: int = 224, patch_size: int = 16) -> dict:
  """One example's `segment_ids`, `att_mask`, `relative_att_ids`
  (`src/data/data_utils.py:285-380`)."""
  if relative_att_num_core_layers > 0:
    generator = MmtRelativePositionGenerator(
        image_size // patch_size, relative_att_num_core_layers, relative_pos_max_distance)
  else:
    generator = RelativePositionGenerator1D(relative_pos_max_distance)
  seq_len = num_image_wordpieces + num_text_wordpieces
  out = {'segment_ids': make_segment_ids(num_image_wordpieces, num_text_wordpieces, max_seq_len)}
  breakpoints = np.zeros((1, max_seq_len), np.int32)           # one_hot(seq_len - 1), `:364-368`
  if 0 <= seq_len - 1 < max_seq_len:
    breakpoints[0, seq_len - 1] = 1
  example_ids = np.cumsum(breakpoints[:, ::-1], axis=-1)[:, ::-1]  # reverse cumsum, `:321`
  out['att_mask'] = make_segmented_att_mask(example_ids)[0]
  rel = None
  if relative_pos_max_distance > 0:                             # `:325-329`
    rel = generator.make_relative_att_ids(max_seq_len, 1)[0]
  out['relative_att_ids'] = rel
  return out


# --------------------------------------------------------------------------
# Build-defined sparse pattern.  Not in the reference: it
# is one particular `att_mask` fed to the reference's dense operator, and this
# function materialises it so the dense oracle can check the sparse kernels.
# --------------------------------------------------------------------------
def sparse_pattern_mask(seq_len: int, valid_len: int, local_radius: int,
                        global_start: int = 0, n_global: int = 0, global_index=None) -> np.ndarray:
  """mask(q,k) = segmented(q,k) & (|q-k| <= radius | global[q] | global[k]); the global tokens are the range
  [global_start, global_start + n_global) or, when given, the listed positions `global_index`."""
  pos = np.arange(seq_len)
  ex = (pos < valid_len)
  seg = ex[:, None] == ex[None, :]
  band = np.abs(pos[:, None] - pos[None, :]) <= local_radius
  if global_index is not None:
    is_g = np.isin(pos, np.asarray(list(global_index), dtype=np.int64))
  else:
    is_g = (pos >= global_start) & (pos < global_start + n_global)
  return (seg & (band | is_g[:, None] | is_g[None, :])).astype(np.int32)


def relative_ids_from_desc(seq_len: int, id_mode: int, max_dist: int,
                           patches_per_row: int = 0, core_layers: int = 0) -> np.ndarray:
  """[S,S] ids for a mask descriptor (id_mode 1 = 1-D over the sequence, 2 = 2-D+1-D)."""
  if id_mode == 1:
    return RelativePositionGenerator1D(max_dist).make_relative_att_ids(seq_len, 1)[0]
  if id_mode == 2:
    return MmtRelativePositionGenerator(patches_per_row, core_layers,
                                        max_dist).make_relative_att_ids(seq_len, 1)[0]
  raise ValueError('id_mode must be 1 or 2')
