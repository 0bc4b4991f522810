#!/bin/bash
# Dev tool (diagnostic build): ablations of the window forward kernel: "MODE SLEEP" pairs.
export TMPDIR=/tmp MMT_FWD_WIN=1
i=0
for cfg in "$@"; do
  set -- $cfg
  export MMT_DBG_MODE=$1 MMT_DBG_SLEEP=$2
  d=gpurun_out/fx_$i; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o a -- python3 tools/attn_probe.py > /dev/null 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "mode=$1 sleep=$2: $(python3 tools/kstat.py $f attn_fwd_win)"
  i=$((i+1))
done
