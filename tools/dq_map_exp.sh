#!/bin/bash
# Dev tool: dQ-pass block placement experiment: kernel time + FETCH_SIZE for MMT_DQ_PLANE_MAJOR = 0 / 1.
export TMPDIR=/tmp
for m in 0 1; do
  export MMT_DQ_PLANE_MAJOR=$m
  rm -rf gpurun_out/dqm_k$m gpurun_out/dqm_f$m
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dqm_k$m -o a -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/dqm_f$m -o f -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1
  echo "MMT_DQ_PLANE_MAJOR=$m"
  python3 tools/kstat.py $(find gpurun_out/dqm_k$m -name '*kernel_stats.csv' | head -1) attn_bwd
  python3 - <<PY
import csv, collections
acc = collections.defaultdict(list)
import glob
for r in csv.DictReader(open(glob.glob('gpurun_out/dqm_f$m/**/f_counter_collection.csv', recursive=True)[0])):
  if r['Counter_Name'] == 'FETCH_SIZE' and 'attn_bwd' in r['Kernel_Name']:
    acc[r['Kernel_Name'].split('(')[0][-40:]].append(float(r['Counter_Value']))
for k, v in acc.items(): print('  FETCH_SIZE KB', k, round(sum(v[len(v)//4:]) / len(v[len(v)//4:]), 1))
PY
done
