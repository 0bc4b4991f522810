import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import benchmarks
import bench
step, info = benchmarks.make_train_step_bench(bench.config3(), torch.device('cuda:0'), 0, 1)
def run(sleep_cycles, n=40):
  for _ in range(6):
    step()
    if sleep_cycles: torch.cuda._sleep(sleep_cycles)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n):
    step()
    if sleep_cycles: torch.cuda._sleep(sleep_cycles)
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e3
for cyc in (0, 200000, 1000000, 0, 200000, 1000000):
  print('sleep cycles', cyc, 'ms/step', round(run(cyc), 4))
