"""Dev tool: host time of one graph replay call, and wall time per step with and without a GPU-side sleep ahead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import benchmarks
import bench
dev = torch.device('cuda:0')
step, info = benchmarks.make_train_step_bench(bench.config3(), dev, 0, 1, dtype=torch.bfloat16, graph=True)
for _ in range(8): step()
torch.cuda.synchronize()
ts = []
for _ in range(10):
  t = time.perf_counter(); step(); ts.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
print('host ms per graphed step call:', [round(x, 2) for x in ts])
def run(n, sleep):
  torch.cuda.synchronize(); t = time.perf_counter()
  for _ in range(n):
    if sleep: torch.cuda._sleep(4_000_000)
    step()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n * 1e3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(4_000_000); e1.record(); torch.cuda.synchronize()
sl = e0.elapsed_time(e1)
print('plain', run(20, False), 'with sleep', run(20, True) - sl)
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
def run2(n):
  torch.cuda.synchronize(); t = time.perf_counter()
  for i in range(n):
    cur, prev = ss[i % 2], ss[(i + 1) % 2]
    cur.wait_stream(prev)
    with torch.cuda.stream(cur):
      step()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n * 1e3
print('two streams alternating', run2(20), run2(20))
print('loss', float(step()['loss']))
