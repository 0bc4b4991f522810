"""Dev tool: a few hundred train steps of the bench workload on one fixed batch -- the loss must fall and stay finite
(all fused paths on: dropout, grouped weight gradients, fused losses, clip + AdamW)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import math, torch
from mmt_amd import benchmarks
import bench
cfg = bench.config3()
if len(sys.argv) > 1 and sys.argv[1] == '2d':
  cfg.update(R=49, core=1, P=63)
step, info = benchmarks.make_train_step_bench(cfg, torch.device('cuda:0'), 0, 1)
losses = []
for i in range(300):
  out = step()
  if i % 25 == 0 or i == 299:
    losses.append(float(out['loss']))
    print(i, round(losses[-1], 4), flush=True)
assert all(math.isfinite(x) for x in losses), losses
assert losses[-1] < 0.7 * losses[0], losses
print('soak ok', losses[0], '->', losses[-1])
