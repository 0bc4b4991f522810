"""Dev tool: torch.profiler view of one train step at BASELINE config 3 (op -> kernels), not part of the product."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
import mmt_amd
from mmt_amd import benchmarks
import bench
cfg = bench.config3()
step, info = benchmarks.make_train_step_bench(cfg, torch.device('cuda:0'), 0, 1)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
  for _ in range(3): step()
  torch.cuda.synchronize()
evs = prof.events()
# map each device kernel to the innermost CPU op that launched it (by correlation through linked events)
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for e in evs:
  if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
    # only leaf ops: skip if any child also has kernels
    if any(c.kernels for c in e.cpu_children): continue
    for k in e.kernels:
      key = (e.name[:50], k.name[:70])
      agg[key][0] += 1; agg[key][1] += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
out = os.path.join('gpurun_out', 'step_profile.txt'); os.makedirs('gpurun_out', exist_ok=True)
with open(out, 'w') as f:
  for (op, kn), (n, us) in rows[:120]:
    f.write(f"{op:50s} {kn:70s} n/step {n/3:6.1f} us/step {us/3:8.1f}\n")
print(open(out).read())
