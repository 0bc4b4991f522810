#!/bin/bash
# Dev tool: kernel durations of the forward call: "WIN NG" pairs.
export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  set -- $cfg
  export MMT_FWD_WIN=$1 PROBE_NG=$2
  d=gpurun_out/fy_$i; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o a -- python3 tools/attn_probe.py 2>&1 | grep "fwd us"
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python3 tools/kstat.py $f attn_
  i=$((i+1))
done
