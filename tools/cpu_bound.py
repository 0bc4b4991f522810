"""Dev tool: is the train step host-bound?  Host enqueue time vs GPU completion time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import benchmarks
import bench
step, info = benchmarks.make_train_step_bench(bench.config3(), torch.device('cuda:0'), 0, 1)
for _ in range(5): step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step, gpu tail after last enqueue {1e3*(t2-t1):.2f} ms')
