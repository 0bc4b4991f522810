"""Dev tool: is the train step waiting for the host anywhere?  A step is timed as usual and with a GPU-side sleep queued at
its start (the host then runs ahead of the GPU by the sleep's length): sleep-inclusive time minus the sleep below the
plain time = GPU idle the host causes in a plain step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import benchmarks
import bench
dev = torch.device('cuda:0')
step, info = benchmarks.make_train_step_bench(bench.config3(), dev, 0, 1, dtype=torch.bfloat16)
for _ in range(5): step()
torch.cuda.synchronize()
# calibrate the sleep
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
cyc = 4_000_000
e0.record(); torch.cuda._sleep(cyc); e1.record(); torch.cuda.synchronize()
sleep_ms = e0.elapsed_time(e1)
def run(n, sleep):
  torch.cuda.synchronize(); t = time.perf_counter()
  for _ in range(n):
    if sleep: torch.cuda._sleep(cyc)
    step()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n * 1e3
for rep in range(3):
  a = run(20, False); b = run(20, True)
  print(f'plain {a:.3f} ms   with {sleep_ms:.2f} ms sleep {b:.3f} ms   -> step GPU time {b - sleep_ms:.3f} ms, host-caused idle {a - (b - sleep_ms):.3f} ms')
