"""Feature-pipeline transforms (mmt_amd/feature_pipeline.py) vs the numpy oracle: exact for the copies and
permutations, bit-exact for the integer label ids.  Runs on the CPU here and on the GPU under -m gpu."""
import numpy as np
import pytest
import torch

import __graft_entry__  # noqa: F401
from oracle import feature_pipeline as ofp


def _run(device):
  from mmt_amd import feature_pipeline as fp
  rng = np.random.default_rng(0)
  # patches: 5 images of 70x70 (VALID drops the 6-pixel remainder), patch 16, 3 channels
  imgs = rng.random((5, 70, 70, 3), dtype=np.float32)
  got = fp.convert_image_to_patches(torch.from_numpy(imgs).to(device), 16).cpu().numpy()
  for b in range(5):
    assert np.array_equal(got[b], ofp.convert_image_to_patches(imgs[b], 16))
  assert got.shape == (5, 16, 768)
  # MPP label ids: 512 classes, including values on bin boundaries
  emb = rng.random((40, 768), dtype=np.float32)
  emb[0] = 0.0; emb[1] = 1.0; emb[2] = 32.0 / 255.0; emb[3] = np.float32(63.999 / 255.0)
  ids = fp.make_mpp_label_ids(torch.from_numpy(emb).to(device), 16).cpu().numpy()
  want = ofp.make_mpp_label_ids(emb, 16)
  assert ids.dtype == np.int32 and np.array_equal(ids, want)
  assert ids[0] == 0 and ids[1] == 511 and 0 <= ids.min() and ids.max() <= 511
  # in-batch negatives
  B = 9
  keys = np.array([7, 3, 7, 9, 3, 11, 12, 13, 14])
  feats = {'patch_token_ids': rng.integers(0, 99, (B, 6)), 'patch_embeddings': rng.random((B, 6, 4), dtype=np.float32),
           'num_image_wordpieces': rng.integers(1, 8, (B,)), 'text_token_ids': rng.integers(0, 99, (B, 5)),
           'num_text_wordpieces': rng.integers(1, 6, (B,)), 'mlm_positions': rng.integers(0, 11, (B, 3)),
           'mlm_label_ids': rng.integers(0, 99, (B, 3)), 'mpp_label_weights': rng.random((B, 2), dtype=np.float32)}
  for ratio, shift in ((1, 5), (2, 1)):
    want = ofp.make_matching_features({k: v.copy() for k, v in feats.items()}, keys, ratio, shift)
    got = fp.make_matching_features({k: torch.from_numpy(v).to(device) for k, v in feats.items()},
                                    torch.from_numpy(keys).to(device), ratio, shift)
    assert set(got) == set(want)
    for k in want:
      assert np.array_equal(got[k].cpu().numpy(), want[k]), (k, ratio, shift)
    assert got['itm_label_ids'].dtype == torch.int32 and int(got['itm_label_ids'].sum()) == B
  with pytest.raises(ValueError):
    fp.make_matching_features({}, torch.arange(6, device=device), 1, 5)      # batch too small for the shift


def test_feature_pipeline_cpu():
  _run('cpu')


@pytest.mark.gpu
def test_feature_pipeline_gpu():
  _run('cuda')


# ---- MLM / MPP masking (data_utils.py:383-639): index outputs bit-exact against the per-example oracle ----
CLS, PATCH, SEP, ATT, MASK, VOCAB = 101, 1, 102, 2, 103, 30522


def _masking_case(rng, B, P, T, S, whole_word, rates, ragged=True, premasked=False):
  n_patch = P * P
  E = 12                                                     # patch 2 x 2 x 3 channels
  n_text = rng.integers(3, T + 1, B) if ragged else np.full(B, T)
  feats = {'patch_token_ids': np.tile(np.concatenate([[CLS, PATCH], 104 + np.arange(n_patch)]).astype(np.int32), (B, 1)),
           'text_token_ids': np.zeros((B, T), np.int32), 'num_text_wordpieces': n_text.astype(np.int32),
           'patch_embeddings': rng.standard_normal((B, n_patch, E)).astype(np.float32),
           'unnormalized_patch_embeddings': rng.random((B, n_patch, E), dtype=np.float32)}
  ws = np.zeros((B, T), bool)
  for b in range(B):
    t = rng.integers(1000, VOCAB, n_text[b]).astype(np.int32)
    t[0] = ATT; t[-1] = SEP
    if n_text[b] > 6:
      t[4] = SEP                                             # an unselectable id in the middle of the text
    if premasked and n_text[b] > 3:
      t[2] = MASK                                            # a [MASK] that was already there
    feats['text_token_ids'][b, :n_text[b]] = t
    ws[b, :n_text[b]] = rng.random(n_text[b]) < 0.6
    ws[b, 0] = True
  if whole_word:
    feats['text_word_start'] = ws
  rnd = {'mlm_item_keys': rng.random((B, T), dtype=np.float32), 'mlm_value_u': rng.random((B, T), dtype=np.float32),
         'mlm_random_ids': rng.integers(0, VOCAB, (B, T)).astype(np.int32),
         'mpp_item_keys': rng.random((B, 2 + n_patch), dtype=np.float32), 'mpp_value_u': rng.random((B, 2 + n_patch), dtype=np.float32),
         'mpp_random_ids': rng.integers(0, VOCAB, (B, 2 + n_patch)).astype(np.int32)}
  rnd['mlm_item_keys'][:, 3] = rnd['mlm_item_keys'][:, 1]    # a tie: broken by item order
  kw = dict(max_seq_len=S, num_patches=n_patch, patch_size=2, vocab_size=VOCAB, mask_token_id=MASK,
            unselectable_ids=[CLS, SEP, PATCH, ATT], mlm_fraction_to_mask=rates[0], mpp_fraction_to_mask=rates[1],
            mlm_max_selections_per_seq=rates[2], mpp_max_selections_per_seq=rates[3])
  return feats, rnd, kw


def _run_masking(device):
  from mmt_amd import feature_pipeline as fp
  rng = np.random.default_rng(3)
  cases = [dict(B=5, P=4, T=20, S=2 + 16 + 24, whole_word=False, rates=(0.15, 0.5, 20, 6)),
           dict(B=4, P=3, T=30, S=2 + 9 + 30, whole_word=True, rates=(0.3, 0.5, 41, 5)),
           dict(B=3, P=3, T=12, S=2 + 9 + 12, whole_word=False, rates=(0.0, 0.0, 8, 4)),      # nothing selected
           dict(B=3, P=3, T=12, S=2 + 9 + 12, whole_word=False, rates=(1.0, 1.0, 3, 2), premasked=True),   # capped by max
           dict(B=2, P=5, T=16, S=2 + 25 + 16, whole_word=True, rates=(0.5, 0.25, 43, 98), ragged=False)]
  for c in cases:
    feats, rnd, kw = _masking_case(rng, **c)
    got = fp.make_mlm_and_mpp_features({k: torch.from_numpy(v).to(device) for k, v in feats.items()},
                                       {k: torch.from_numpy(v).to(device) for k, v in rnd.items()}, **kw)
    for b in range(c['B']):
      nt = int(feats['num_text_wordpieces'][b])
      ex = {'patch_token_ids': feats['patch_token_ids'][b], 'text_token_ids': feats['text_token_ids'][b, :nt],
            'patch_embeddings': feats['patch_embeddings'][b], 'unnormalized_patch_embeddings': feats['unnormalized_patch_embeddings'][b]}
      if 'text_word_start' in feats:
        ex['text_word_start'] = feats['text_word_start'][b, :nt]
      want = ofp.make_mlm_and_mpp_features(ex, {k: v[b] for k, v in rnd.items()}, **kw)
      for k, w in want.items():
        g = got[k][b].cpu().numpy()
        assert g.shape == w.shape, (k, g.shape, w.shape)
        assert np.array_equal(g, w), (c, b, k, g, w)          # bit-exact, float outputs included
    assert got['word_ids'].shape[1] == kw['max_seq_len'] and got['mlm_positions'].dtype == torch.int32
  # the selection respects its contract: unselectable ids never change, positions ascend, labels are the originals
  feats, rnd, kw = _masking_case(rng, 6, 4, 24, 2 + 16 + 24, False, (0.5, 0.5, 24, 8))
  got = fp.make_mlm_and_mpp_features({k: torch.from_numpy(v).to(device) for k, v in feats.items()},
                                     {k: torch.from_numpy(v).to(device) for k, v in rnd.items()}, **kw)
  w = got['word_ids'].cpu().numpy()
  orig = np.concatenate([feats['patch_token_ids'], feats['text_token_ids']], 1)
  for tok in (CLS, SEP, PATCH, ATT):
    assert np.array_equal(w == tok, orig == tok)
  pos, lab, wt = (got[k].cpu().numpy() for k in ('mlm_positions', 'mlm_label_ids', 'mlm_label_weights'))
  for b in range(6):
    n = int((pos[b] > 0).sum())
    assert n > 0 and np.all(np.diff(pos[b, :n]) > 0) and np.array_equal(lab[b, :n], orig[b, pos[b, :n]])
    assert wt[b].sum() == (w[b, 18:] == MASK).sum()


def test_masking_cpu():
  _run_masking('cpu')


@pytest.mark.gpu
def test_masking_gpu():
  _run_masking('cuda')
