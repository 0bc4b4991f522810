#!/bin/bash
# Dev tool: the rocprofv3 passes behind the round-3 records in profiles/ (run on the GPU box; summaries are written by
# tools/pmc_traffic.py / copied by hand afterwards).
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/attn_ks gpurun_out/step_ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/attn_ks -o a -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/step_ks -o s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --mode train_step > gpurun_out/step_ks_bench.json 2>/dev/null
ls gpurun_out/attn_ks gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/step_ks
