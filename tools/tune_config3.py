"""Dev tool: re-measure the library-GEMM selections for BASELINE config 3 only, with TunableOp's rotating buffer (inputs not
L2 / Infinity-Cache resident between iterations, as inside a train step):
  PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tuned_c3.csv \\
  PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=512 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=50 python tools/tune_config3.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch
from mmt_amd import benchmarks
import bench
step, info = benchmarks.make_train_step_bench(bench.config3(), torch.device('cuda:0'), 0, 1, dtype=torch.bfloat16, graph=False)
for i in range(3):
  step()
  torch.cuda.synchronize()
  print('step', i, flush=True)
