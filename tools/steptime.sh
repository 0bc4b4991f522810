#!/bin/bash
# Dev tool: one bench.py run reduced to "label samples/s ms_per_step"; extra arguments are VAR=value settings.
#   ./tools/steptime.sh base MMT_FFN_FUSED=0
label=$1; shift
cd "$(dirname "$0")/.."
env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'])"
