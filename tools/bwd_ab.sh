#!/bin/bash
# Dev tool: backward call under rocprofv3 for a list of MMT_BWD_WIN settings (kernel durations).
export TMPDIR=/tmp PROBE_BWD=1   # arguments: labels only (one pass per argument)
i=0
for m in "$@"; do
  : # (the argument is a label; set the switches under test in the environment of the caller)
  d=gpurun_out/bw_$i; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o a -- python3 tools/attn_probe.py 2>&1 | grep "bwd us"
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "pass $m"; python3 tools/kstat.py $f attn_bwd drel
  i=$((i+1))
done
