"""Dev tool: which Python line launches each framework (torch) kernel of one EAGER train step at BASELINE config 3:
torch.profiler with stacks, grouped by (kernel, innermost mmt_amd / bench frame).  Not part of the product."""
import os, sys
os.environ['MMT_STEP_GRAPH'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from collections import defaultdict
import mmt_amd
from mmt_amd import benchmarks
import bench
cfg = bench.get_config(3)
step, info = benchmarks.make_train_step_bench(cfg, torch.device('cuda:0'), 0, 1, graph=False)
for _ in range(4): step()
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
  for _ in range(N): step()
  torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for e in prof.events():
  if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
    if any(c.kernels for c in e.cpu_children): continue
    frame = '?'
    for fr in (e.stack or []):
      if 'mmt_amd' in fr or 'bench.py' in fr:
        frame = fr.split('multimodal-long-transformer-2021_amd/')[-1][:70]
        break
    for k in e.kernels:
      if k.name.startswith(('mmt::', '_ZN3mmt', 'void mmt::', 'Cijk', 'Custom_Cijk')): continue
      agg[(k.name[:60], e.name[:28], frame)][0] += 1
      agg[(k.name[:60], e.name[:28], frame)][1] += k.duration
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/step_ops_probe.txt', 'w') as f:
  tot = 0.0
  for (kn, op, fr), (n, us) in rows:
    tot += us / N
    f.write(f'{us / N:8.1f} us/step  n {n / N:5.1f}  {kn:60s} {op:28s} {fr}\n')
  f.write(f'total framework kernels: {tot:.1f} us/step\n')
print(open('gpurun_out/step_ops_probe.txt').read()[-6000:])
