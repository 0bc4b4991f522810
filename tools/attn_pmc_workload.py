"""Dev tool: the two attention calls bench.py's rooflines time (BASELINE config MMT_BENCH_CONFIG = 2 | 3 (default) | 5 with
MMT_BENCH_GLOBALS global tokens; contiguous q/k/v, dropout 0.1, fp32 table gradients accumulated), repeated, as the
workload of the rocprofv3 passes behind profiles/attn_traffic.json (config 3) and profiles/r04_cfg*_attn_traffic.json:
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/attn_pmc_workload.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/attn_pmc_workload.py
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn_ks -o a -- python3 tools/attn_pmc_workload.py
then (here, where git is available):  python tools/pmc_traffic.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'multimodal-long-transformer-2021_amd'))
import torch, mmt_amd, bench
cfg = bench.get_config(int(os.environ.get('MMT_BENCH_CONFIG', '3')), os.environ.get('MMT_BENCH_GLOBALS'))
td = torch.float32 if cfg['dtype'] == 'f32' else torch.bfloat16
B, S, N, D, R = cfg['B'], cfg['S'], cfg['N'], cfg['D'], cfg['R']
g = torch.Generator(device='cuda').manual_seed(1234)
q, k, v = (torch.randn(B, S, N, D, device='cuda', generator=g).to(td) for _ in range(3))
emb = (torch.randn(R, N, D, device='cuda', generator=g) * 0.02).to(td)
bias = (torch.randn(R, N, device='cuda', generator=g) * 0.02).to(td)
pat = mmt_amd.AttentionPattern(local_radius=cfg['radius'], global_start=cfg['g0'], n_global=cfg['ng'], id_mode=1, max_dist=cfg['m'])
kw = dict(pattern=pat, dropout_p=0.1, dropout_seed=12345)
out, lse = mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
dout = torch.randn_like(out)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
de = torch.zeros(R, N, D, device='cuda'); db = torch.zeros(R, N, device='cuda')
for _ in range(24):
  mmt_amd.relative_attention_forward(q, k, v, emb, bias, **kw)
  mmt_amd.relative_attention_backward(dout, q, k, v, emb, bias, out, lse, grads_out=(dq, dk, dv), rel_grads_accum=(de, db), **kw)
torch.cuda.synchronize()
