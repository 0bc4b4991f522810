"""K5 side-input kernel vs the oracle and the reference's golden matrices -- bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import side_inputs as si

pytestmark = pytest.mark.gpu

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), 'golden',
                                     'feature_utils_golden.json')))


def _run(pattern, img, txt, S, **kw):
  import mmt_amd
  dev = torch.device('cuda:0')
  out = mmt_amd.side_inputs(pattern, torch.tensor(img, dtype=torch.int32, device=dev),
                            torch.tensor(txt, dtype=torch.int32, device=dev), S, **kw)
  torch.cuda.synchronize()
  return {k: (None if v is None else v.cpu().numpy()) for k, v in out.items()}


@pytest.mark.parametrize('case', GOLDEN['cases'], ids=lambda c: c['name'])
def test_golden_matrices(case):
  import mmt_amd
  pat = mmt_amd.AttentionPattern(id_mode=2, max_dist=case['text_relative_pos_max_distance'],
                                 patches_per_row=case['num_patch_per_row'],
                                 core_layers=case['num_core_layers'])
  got = _run(pat, [case['seq_len']], [0], case['seq_len'])
  np.testing.assert_array_equal(got['relative_att_ids'], np.array(case['expected']))


@pytest.mark.parametrize('S,P,r,m', [(256, 14, 2, 12), (256, 14, 1, 12), (100, 7, 3, 5), (1024, 28, 2, 12),
                                     (77, 5, 5, 4)])
def test_ids_2d(S, P, r, m):
  import mmt_amd
  pat = mmt_amd.AttentionPattern(id_mode=2, max_dist=m, patches_per_row=P, core_layers=r)
  got = _run(pat, [S, S // 2], [0, 3], S)
  ref = si.MmtRelativePositionGenerator(P, r, m).make_relative_att_ids(S, 1)[0]
  for b in range(2):
    np.testing.assert_array_equal(got['relative_att_ids'][b], ref)


@pytest.mark.parametrize('S,m', [(64, 12), (257, 12), (1024, 3), (33, 100)])
def test_ids_1d_and_reference_features(S, m):
  import mmt_amd
  pat = mmt_amd.AttentionPattern(id_mode=1, max_dist=m)
  img = [S // 2, 5, S]
  txt = [S // 4, 0, 0]
  got = _run(pat, img, txt, S)
  for b in range(3):
    ref = si.add_side_input_features(img[b], txt[b], S, m)
    np.testing.assert_array_equal(got['segment_ids'][b], ref['segment_ids'])
    np.testing.assert_array_equal(got['att_mask'][b], ref['att_mask'])
    np.testing.assert_array_equal(got['relative_att_ids'][b], ref['relative_att_ids'])


@pytest.mark.parametrize('S,rad,g0,ng', [(128, 16, 100, 8), (130, 0, 0, 0), (96, 200, 3, 5)])
def test_materialised_pattern(S, rad, g0, ng):
  import mmt_amd
  pat = mmt_amd.AttentionPattern(local_radius=rad, global_start=g0, n_global=ng, id_mode=1, max_dist=4)
  img, txt = [S - 20, 10], [10, 7]
  got = _run(pat, img, txt, S, materialize_pattern=True)
  for b in range(2):
    ref = si.sparse_pattern_mask(S, img[b] + txt[b], rad, g0, ng)
    np.testing.assert_array_equal(got['att_mask'][b], ref)


def test_argument_errors():
  import mmt_amd
  from mmt_amd._lib import MmtError
  for bad in (dict(patches_per_row=0, core_layers=1), dict(patches_per_row=2, core_layers=0)):
    with pytest.raises(MmtError):
      _run(mmt_amd.AttentionPattern(id_mode=2, max_dist=3, **bad), [8], [0], 8)
  with pytest.raises(MmtError):
    _run(mmt_amd.AttentionPattern(id_mode=1, max_dist=-1), [8], [0], 8)


def test_full_size_properties():
  """S=4096 (BASELINE config 3 shape): closed-form invariants instead of a dense oracle."""
  import mmt_amd
  S, P, m = 4096, 63, 12
  pat = mmt_amd.AttentionPattern(id_mode=2, max_dist=m, patches_per_row=P, core_layers=2)
  dev = torch.device('cuda:0')
  out = mmt_amd.side_inputs(pat, torch.tensor([3971 + 100], dtype=torch.int32, device=dev),
                            torch.tensor([20], dtype=torch.int32, device=dev), S)
  ids, mask = out['relative_att_ids'][0], out['att_mask'][0]
  I = P * P
  assert int(ids.diagonal().abs().max()) == 0
  assert bool((ids[:I, I:] == I + 8 + 2 * m + 2).all()) and bool((ids[I:, :I] == I + 8 + 2 * m + 1).all())
  t = ids[I:, I:]
  assert int(t.max()) == 2 * m and bool((t[0, :m + 1] == torch.arange(m + 1, device=dev)).all())
  # translation invariance of the 2-D block: same (dx,dy) -> same id
  img = ids[:I, :I].reshape(P, P, P, P)
  assert bool((img[10, 10, 12, 9] == img[30, 40, 32, 39]).all())
  assert int(mask.sum()) == 4091 ** 2 + 5 ** 2
