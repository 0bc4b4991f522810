#!/bin/bash
# Dev tool: kernel durations of the forward call under rocprofv3 for a list of "WIN TSTRIDE" settings.
export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  set -- $cfg
  export MMT_FWD_WIN=$1
  if [ "$2" != "-" ]; then export MMT_WIN_TSTRIDE=$2; else unset MMT_WIN_TSTRIDE; fi
  d=gpurun_out/fw_$i; rm -rf $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o a -- python3 tools/fwd_probe2.py 2>&1 | grep "fwd us"
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  python3 tools/kstat.py $f attn_ 
  i=$((i+1))
done
