"""Dev tool: the train step eager and as a HIP graph: wall time per step, and the loss sequences side by side (they
must agree: same seeds, deterministic kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import benchmarks
import bench
dev = torch.device('cuda:0')
res = {}
for mode in (False, True):
  step, info = benchmarks.make_train_step_bench(bench.config3(), dev, 0, 1, dtype=torch.bfloat16, graph=mode)
  losses = []
  for _ in range(8):
    losses.append(float(step()['loss']))
  torch.cuda.synchronize(); t = time.perf_counter()
  for _ in range(20): step()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t) / 20 * 1e3
  losses.append(float(step()['loss']))
  res[mode] = losses
  print(f"{'graph' if mode else 'eager'}: {dt:.3f} ms/step   losses {[round(x, 5) for x in losses]}", flush=True)
  step.close()
  del step
  torch.cuda.empty_cache()
print('max |loss difference|', max(abs(a - b) for a, b in zip(res[False], res[True])))
