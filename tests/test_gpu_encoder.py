"""Encoder / train-step parity: the torch+HIP product vs the dense CPU oracle (oracle/encoder.py).

fp32, dropout off; forward 1e-3 (north_star), gradients 2e-3 relative to each tensor's max."""
import numpy as np
import pytest
import torch

from oracle import encoder as oenc
from oracle import side_inputs as si

pytestmark = pytest.mark.gpu


def tiny_experiment(S=256, image=224, m=12, R=32, core=0, radius=1 << 30, n_global=0, pre=True):
  from mmt_amd import configs
  exp = configs.get_exp_config('mmt/pretraining')
  exp.override({'task': {
      'model': {'encoder': {'mmt': dict(num_hidden_layers=2, hidden_size=128, num_attention_heads=2,
                                        intermediate_size=512, vocab_size=2000, relative_vocab_size=R,
                                        relative_pos_max_distance=m, relative_att_num_core_layers=core,
                                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                                        use_pre_activation_order=pre)},
                'cls_heads': [{'inner_dim': 128, 'num_classes': 2, 'name': 'itm'}]},
      'train_data': dict(max_seq_len=S, image_size=image, patch_size=16, relative_pos_max_distance=m,
                         relative_att_num_core_layers=core, mlm_max_selections_per_seq=8,
                         mpp_max_selections_per_seq=6, local_radius=radius, num_global_tokens=n_global,
                         tasks='mlm,itm')}})
  return exp


def dense_inputs_cpu(inputs, data_cfg):
  """CPU copies + the dense [B,S,S] side inputs the reference would feed (from the oracle)."""
  S = data_cfg.max_seq_len
  P = data_cfg.image_size // data_cfg.patch_size
  out = {k: v.detach().cpu() for k, v in inputs.items() if torch.is_tensor(v)}
  pat = inputs.get('attention_pattern')
  if pat is not None:
    vl = inputs['valid_len'].cpu().tolist()
    out['att_mask'] = torch.tensor(np.stack([si.sparse_pattern_mask(S, v, min(pat.local_radius, S), pat.global_start,
                                                                     pat.n_global) for v in vl]))
    if pat.id_mode:
      ids = si.relative_ids_from_desc(S, pat.id_mode, pat.max_dist, pat.patches_per_row, pat.core_layers)
      out['relative_att_ids'] = torch.tensor(ids)[None].expand(len(vl), S, S)
  return out


@pytest.mark.parametrize('kw', [dict(), dict(core=2, R=49), dict(radius=32, n_global=8), dict(pre=False)],
                         ids=['1d-dense-pattern', '2d', 'band-global', 'post-ln'])
@pytest.mark.parametrize('dense', [False, True], ids=['pattern', 'dense-inputs'])
def test_encoder_forward_matches_oracle(kw, dense):
  import mmt_amd
  exp = tiny_experiment(**kw)
  task = mmt_amd.tasks.get_task(exp.task)
  torch.manual_seed(0)
  model = task.build_model().cuda().eval()
  inputs, _ = next(task.build_inputs(exp.task.train_data, device='cuda', batch_size=2, ragged=True,
                                     dense_side_inputs=dense))
  enc_in = {k: v for k, v in inputs.items() if k not in ('mlm_positions', 'mpp_positions')}
  got = model.encoder(**enc_in, training=False)['sequence_output'].float().cpu()
  if dense:
    cpu = {k: v.cpu() for k, v in inputs.items() if torch.is_tensor(v)}
  else:
    cpu = dense_inputs_cpu(inputs, exp.task.train_data)
  sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
  want = oenc.encoder_forward(sd, model.encoder.get_config(), cpu['word_ids'], cpu.get('segment_ids'),
                              cpu.get('att_mask'), cpu.get('relative_att_ids'), cpu.get('patch_embeddings'))
  err = float((got.double() - want).abs().max())
  assert err < 1e-3, err


def test_train_step_gradients_match_oracle_autograd():
  import mmt_amd
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  task = mmt_amd.tasks.get_task(exp.task)
  torch.manual_seed(1)
  model = task.build_model().cuda()
  batch = next(task.build_inputs(exp.task.train_data, device='cuda', batch_size=2, ragged=True))
  inputs, labels = batch
  out = model(**inputs, training=False)
  loss = task.build_losses(labels, out)
  loss.backward()
  # oracle: float64 dense CPU model with autograd on copies of the same weights
  sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.named_parameters()}
  cpu_in = dense_inputs_cpu(inputs, exp.task.train_data)
  cpu_lab = {k: v.cpu() for k, v in labels.items()}
  ref_loss = oenc.pretraining_loss(sd, model.encoder.get_config(), cpu_in, cpu_lab)
  ref_loss.backward()
  assert abs(float(loss) - float(ref_loss)) < 1e-3
  for name, p in model.named_parameters():
    want = sd[name].grad
    if want is None:
      assert p.grad is None or float(p.grad.abs().max()) == 0, name
      continue
    got = p.grad.detach().cpu().double()
    err = float((got - want).abs().max()) / max(1e-3, float(want.abs().max()))
    assert err < 2e-3, (name, err)


def test_train_step_reduces_loss_bf16():
  import mmt_amd
  from mmt_amd import optimization
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  exp.task.micro_batch_size = 4
  exp.task.model.encoder.mmt.hidden_dropout_prob = 0.1
  exp.task.model.encoder.mmt.attention_probs_dropout_prob = 0.1
  task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.bfloat16)
  torch.manual_seed(2)
  model = task.build_model().cuda()
  opt = optimization.create_optimizer(model, exp.trainer.optimizer_config)
  optimization.set_learning_rate(opt, 1e-3)
  batch = next(task.build_inputs(exp.task.train_data, device='cuda', batch_size=8))
  losses = [float(task.train_step(batch, model, opt, clip_norm=1.0)['loss']) for _ in range(12)]
  assert all(np.isfinite(losses))
  assert losses[-1] < losses[0] - 0.5, losses


def test_bucketed_direct_gradients_equal_plain_autograd():
  """With the DP reducer, parameter gradients are written / accumulated straight into the flat
  fp32 buckets by the fused kernels and the cast backward; they must equal plain autograd's."""
  import mmt_amd
  from mmt_amd import distribute
  exp = tiny_experiment(S=256, radius=32, n_global=8)
  grads = []
  for use_reducer in (False, True):
    task = mmt_amd.tasks.get_task(exp.task, compute_dtype=torch.bfloat16)
    torch.manual_seed(3)
    model = task.build_model().cuda()
    batch = next(task.build_inputs(exp.task.train_data, device='cuda', batch_size=4))
    reducer = None
    if use_reducer:
      reducer = distribute.DataParallelStrategy(None).make_reducer(list(model.parameters()))
      reducer.zero_grad()
    inputs, labels = batch
    for micro in (slice(0, 2), slice(2, 4)):            # two micro-steps: accumulate path
      small = {k: (v[micro] if torch.is_tensor(v) else v) for k, v in inputs.items()}
      lab = {k: v[micro] for k, v in labels.items()}
      loss = task.build_losses(lab, model(**small, training=False)) / 2
      loss.backward()
    if reducer is not None:
      reducer.finish()
    grads.append({n: (None if p.grad is None else p.grad.detach().float().clone())
                  for n, p in model.named_parameters()})
  for name, g0 in grads[0].items():
    g1 = grads[1][name]
    if g0 is None:
      assert g1 is None or float(g1.abs().max()) == 0
      continue
    err = float((g0 - g1).abs().max()) / max(1e-3, float(g0.abs().max()))
    assert err < 2e-2, (name, err)        # bf16 rounding of the plain path's accumulated grads


def _full_config(cfg_over, dtype):
  import bench
  import mmt_amd
  cfg = dict(bench.config3(), **cfg_over)
  step, info = mmt_amd.make_train_step_bench(cfg, torch.device('cuda', 0), 0, 1, dtype=dtype)
  losses = [float(step()['loss']) for _ in range(6)]       # (three eager steps, the recording, two replays)
  torch.cuda.synchronize()
  step.close()
  assert all(np.isfinite(losses)), losses
  assert losses[-1] < losses[0], losses          # same batch every step: the loss must fall
  return losses


def test_config2_full_size_train_step_fp32():
  """BASELINE config 2 at full size (BERT-base dims, S=1024 = 2+28^2+238, radius 64, 8 globals [786,794),
  fp32 compute, B=8): the train step runs, stays finite and reduces the loss."""
  _full_config(dict(S=1024, P=28, B=8, g0=2 + 28 * 28, ng=8), torch.float32)


@pytest.mark.parametrize('ng', [8, 32, 128])
def test_config5_full_size_train_step_bf16(ng):
  """BASELINE config 5 at full size (S=8192 = 2+88^2+446, radius 64, g global tokens, bf16, B=2 per GPU)."""
  _full_config(dict(S=8192, P=88, B=2, g0=2 + 88 * 88, ng=ng), torch.bfloat16)


def test_config3_full_size_train_step_bf16():
  """BASELINE config 3 (the bench workload) through the same closure bench.py times."""
  _full_config(dict(), torch.bfloat16)


def test_config3_full_size_train_step_2d_ids_bf16():
  """The config-3 workload with the 2-D relative ids of the reference's `*_2d*.yaml` experiments (one core layer,
  relative_vocab_size 49: `MmtRelativePositionGenerator`, feature_utils.py:114-184) -- the lean 2-D attention kernels
  inside the full train step (`bench.py --ids2d`)."""
  _full_config(dict(R=49, core=1, P=63), torch.bfloat16)
