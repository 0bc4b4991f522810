#!/bin/bash
# Dev tool: forward kernel time + HBM fetch of the attention workload (one configuration per call).
export TMPDIR=/tmp
rm -rf gpurun_out/ft_k gpurun_out/ft_f
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ft_k -o a -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/ft_f -o f -- python3 tools/attn_pmc_workload.py > /dev/null 2>&1
python3 tools/kstat.py $(find gpurun_out/ft_k -name '*kernel_stats.csv' | head -1) attn_fwd rows_combine
python3 - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for r in csv.DictReader(open(glob.glob('gpurun_out/ft_f/**/f_counter_collection.csv', recursive=True)[0])):
  if r['Counter_Name'] == 'FETCH_SIZE' and ('attn_fwd' in r['Kernel_Name'] or 'rows_combine' in r['Kernel_Name']):
    acc[r['Kernel_Name'].split('(')[0][-40:]].append(float(r['Counter_Value']))
for k, v in acc.items(): print('  FETCH_SIZE KB', k, round(sum(v[len(v)//4:]) / len(v[len(v)//4:]), 1))
PY
