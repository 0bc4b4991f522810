#!/bin/bash
# Dev tool (GPU box): bench line + the three rocprofv3 passes over the attention calls of one BASELINE configuration.
#   tools/r04_profiles.sh <config 2|3|5> <globals> [tag]     -> gpurun_out/<tag>/{bench.json, attn_ks, pmc_f, pmc_w}
# Summaries: here afterwards, `MMT_BENCH_CONFIG=.. MMT_BENCH_GLOBALS=.. python tools/pmc_traffic.py gpurun_out/<tag>/`.
export TMPDIR=/tmp
C=${1:-3}; G=${2:-8}; TAG=${3:-r04_cfg${C}_g${G}}
export MMT_BENCH_CONFIG=$C MMT_BENCH_GLOBALS=$G
D=gpurun_out/$TAG
rm -rf $D; mkdir -p $D
python3 bench.py --config $C --globals $G --steps 20 --warmup 5 > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $D/attn_ks -o a -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $D/pmc_f -o f -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $D/pmc_w -o w -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1 || exit 4
find $D -name '*_kernel_trace.csv' -delete; find $D -name '*agent_info.csv' -delete
cat $D/bench.json | head -c 600; echo; ls $D/attn_ks $D/pmc_f $D/pmc_w
