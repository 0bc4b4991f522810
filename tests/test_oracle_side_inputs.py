"""Oracle (integer path) vs the reference's own golden vectors -- CPU only.

Mirrors `src/feature_utils_test.py` of the reference (same cases, same asserts).
"""
import json
import os

import numpy as np
import pytest

from oracle import c_port
from oracle import side_inputs as si

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), 'golden',
                                     'feature_utils_golden.json')))


def test_relative_position_generator_init():
  # feature_utils_test.py:25-35
  gen = si.MmtRelativePositionGenerator(num_patch_per_row=2, num_core_layers=1,
                                        text_relative_pos_max_distance=3)
  assert gen._num_patch_per_row == 2
  assert gen._num_core_layers == 1
  assert gen._core_layer_diameter == 3
  assert gen._image_part_id == 19
  assert gen._text_part_id == 20


@pytest.mark.parametrize('args', [(0, 1, 2), (1, 0, 2), (1, 1, -1)])
def test_relative_position_generator_init_invalid_arguments(args):
  # feature_utils_test.py:37-47 (each call checked on its own)
  with pytest.raises(ValueError):
    si.MmtRelativePositionGenerator(*args)


@pytest.mark.parametrize('case', GOLDEN['cases'], ids=lambda c: c['name'])
def test_make_relative_att_ids_golden(case):
  # feature_utils_test.py:49-74, :76-110
  gen = si.MmtRelativePositionGenerator(case['num_patch_per_row'], case['num_core_layers'],
                                        case['text_relative_pos_max_distance'])
  assert gen._image_part_id == case['image_part_id']
  assert gen._text_part_id == case['text_part_id']
  assert gen._core_layer_diameter == case['core_layer_diameter']
  if 'base_tensor' in case:
    np.testing.assert_array_equal(gen._base_tensor, np.array(case['base_tensor']))
  got = gen.make_relative_att_ids(case['seq_len'], 1)
  assert got.dtype == np.int32
  np.testing.assert_array_equal(got, np.array(case['expected']))


@pytest.mark.parametrize('case', GOLDEN['cases'], ids=lambda c: c['name'])
def test_c_port_matches_golden(case):
  got = c_port.relative_ids(case['seq_len'], 2, case['text_relative_pos_max_distance'],
                            case['num_patch_per_row'], case['num_core_layers'])
  np.testing.assert_array_equal(got[None], np.array(case['expected']))


@pytest.mark.parametrize('P,r,m,S', [(2, 1, 3, 7), (3, 2, 9, 12), (14, 2, 12, 256),
                                     (14, 1, 12, 200), (5, 5, 4, 40), (7, 3, 0, 60)])
def test_c_closed_form_equals_numpy_window_slicing(P, r, m, S):
  a = si.MmtRelativePositionGenerator(P, r, m).make_relative_att_ids(S, 1)[0]
  b = c_port.relative_ids(S, 2, m, P, r)
  np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('S,m', [(1, 0), (7, 3), (64, 12), (100, 200)])
def test_ids_1d(S, m):
  a = si.RelativePositionGenerator1D(m).make_relative_att_ids(S, 2)
  assert a.shape == (2, S, S) and a.dtype == np.int32
  np.testing.assert_array_equal(a[0], c_port.relative_ids(S, 1, m))
  assert a.min() >= 0 and a.max() <= 2 * m
  assert (np.diagonal(a[0]) == 0).all()
  if S > 1 and m >= 1:
    assert a[0, 0, 1] == 1 and a[0, 1, 0] == m + 1
  assert si.RelativePositionGenerator1D(m).relative_vocab_size == 2 * m + 1


def test_q1_part_ids_exceed_relative_vocab():
  # at P=14, m=12 the cross-modal ids are 229/230 >= R=49.
  gen = si.MmtRelativePositionGenerator(14, 2, 12)
  assert (gen._image_part_id, gen._text_part_id) == (229, 230)
  ids = gen.make_relative_att_ids(256, 1)
  assert ids.max() == 230


@pytest.mark.parametrize('img_wp,txt_wp,S', [(198, 30, 256), (6, 3, 12), (4, 0, 8), (5, 11, 16)])
def test_add_side_input_features(img_wp, txt_wp, S):
  f = si.add_side_input_features(img_wp, txt_wp, S, relative_pos_max_distance=3)
  seg = f['segment_ids']
  assert seg.shape == (S,)
  assert (seg[:img_wp] == 1).all()
  assert seg[img_wp] == 0 or txt_wp == 0                 # first text special token gets 0
  assert (seg[img_wp + 1:img_wp + txt_wp] == 2).all()
  assert (seg[img_wp + txt_wp:] == 0).all()
  np.testing.assert_array_equal(seg, c_port.segment_ids(S, img_wp, txt_wp))
  Lv = img_wp + txt_wp
  m = f['att_mask']
  assert m.shape == (S, S) and m.dtype == np.int32
  assert (m[:Lv, :Lv] == 1).all() and (m[Lv:, Lv:] == 1).all()
  assert (m[:Lv, Lv:] == 0).all() and (m[Lv:, :Lv] == 0).all()
  np.testing.assert_array_equal(m, c_port.att_mask(S, Lv))
  np.testing.assert_array_equal(f['relative_att_ids'], c_port.relative_ids(S, 1, 3))


def test_add_side_input_features_2d_and_none():
  f = si.add_side_input_features(6, 4, 12, relative_pos_max_distance=9,
                                 relative_att_num_core_layers=2, image_size=48, patch_size=16)
  np.testing.assert_array_equal(f['relative_att_ids'][None],
                                np.array(GOLDEN['cases'][1]['expected']))
  assert si.add_side_input_features(6, 4, 12, 0)['relative_att_ids'] is None


@pytest.mark.parametrize('S,Lv,rad,g0,ng', [(64, 64, 8, 40, 4), (96, 70, 16, 60, 8),
                                            (33, 20, 0, 0, 0), (48, 48, 100, 5, 3)])
def test_sparse_pattern_mask(S, Lv, rad, g0, ng):
  a = si.sparse_pattern_mask(S, Lv, rad, g0, ng)
  np.testing.assert_array_equal(a, c_port.att_mask(S, Lv, rad, g0, ng))
  assert (np.diagonal(a) == 1).all()                      # never a fully masked row
  assert (a == a.T).all()
  if rad >= S:
    np.testing.assert_array_equal(a, c_port.att_mask(S, Lv))  # radius >= S is the reference mask
