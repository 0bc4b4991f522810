"""Dev tool: (re)measure the library-GEMM selections shipped in mmt_amd/tuned/ on an MI355X.
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tuned_.csv \\
  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 python tools/tune_gemms.py
runs a few train steps of BASELINE configs 3, 5 (g = 8, 32, 128) and 2 (fp32) so that TunableOp sees every GEMM
shape of those configurations; copy gpurun_out/tuned_0.csv over the shipped file afterwards."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import benchmarks
import bench
dev = torch.device('cuda:0')
base = bench.config3()
runs = [(dict(base), torch.bfloat16, 'config 3'),
        (dict(base, S=8192, P=88, B=2, g0=2 + 88 * 88, ng=8), torch.bfloat16, 'config 5 g8'),
        (dict(base, S=8192, P=88, B=2, g0=2 + 88 * 88, ng=32), torch.bfloat16, 'config 5 g32'),
        (dict(base, S=8192, P=88, B=2, g0=2 + 88 * 88, ng=128), torch.bfloat16, 'config 5 g128'),
        (dict(base, S=1024, P=28, B=8, g0=2 + 28 * 28, ng=8), torch.float32, 'config 2 fp32')]
for cfg, dt, name in runs:
  step, info = benchmarks.make_train_step_bench(cfg, dev, 0, 1, dtype=dt)
  for _ in range(3):
    step()
  torch.cuda.synchronize()
  print('tuned', name, flush=True)
  del step
  torch.cuda.empty_cache()
