#!/bin/bash
# Dev tool (GPU box): bench line + rocprofv3 kernel stats of the config-3 train step.  tools/r04_step_profile.sh [tag]
export TMPDIR=/tmp
TAG=${1:-r04_step}
D=gpurun_out/$TAG
rm -rf $D; mkdir -p $D
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > $D/bench.json 2> $D/bench.err || { tail -5 $D/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks -o s -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --mode train_step > $D/profiled_bench.json 2>/dev/null || exit 2
python3 tools/step_trace.py $D/ks $D/last_step.csv > $D/last_step.txt; cat $D/last_step.txt | head -8
find $D -name '*_kernel_trace.csv' -delete; find $D -name '*agent_info.csv' -delete
python3 - <<PY
import json,csv
b=json.load(open('$D/bench.json')); print('step ms', b['ms_per_step'], 'fwd us', b['attention_fwd']['us_per_layer_call'], 'bwd us', b['attention_bwd']['us_per_layer_call'])
rows=list(csv.DictReader(open('$D/ks/s_kernel_stats.csv')))
steps=17   # 3 eager + 1 capture-less... (4 set-up + 3 warm-up + 10 timed)
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:22]:
  print(f"{float(r['TotalDurationNs'])/1e6/steps:7.3f} ms/step  {float(r['AverageNs'])/1e3:8.1f} us x{int(r['Calls'])//steps:4d}  {r['Name'][:90]}")
print('total ms/step', tot/1e6/steps)
PY
