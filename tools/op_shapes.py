"""Dev tool: framework (aten) ops of one train step with input shapes and the python call site, sorted by device time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from torch.profiler import profile, ProfilerActivity
from mmt_amd import benchmarks
import bench
step, info = benchmarks.make_train_step_bench(bench.config3(), torch.device('cuda:0'), 0, 1)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
  step()
  torch.cuda.synchronize()
rows = []
for e in prof.events():
  if e.device_type == torch.autograd.DeviceType.CPU and e.kernels and not any(c.kernels for c in e.cpu_children):
    if e.name.startswith('aten::'):
      dev = sum(k.duration for k in e.kernels)
      site = next((s for s in (e.stack or []) if 'mmt_amd' in s or 'bench' in s), '')
      rows.append((dev, e.name, str(e.input_shapes)[:90], site.strip()[:110]))
rows.sort(reverse=True)
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/op_shapes.txt', 'w') as f:
  f.write(f'{len(rows)} aten ops with kernels, {sum(r[0] for r in rows):.0f} us\n')
  for dev, name, shp, site in rows:
    f.write(f'{dev:7.1f} {name:28s} {shp:90s} {site}\n')
print(open('gpurun_out/op_shapes.txt').read()[:200])
