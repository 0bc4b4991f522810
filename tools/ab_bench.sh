#!/bin/bash
# Dev tool: alternate the archived revision (_ab/A, see ab_setup.sh) and the working tree on ONE box:
#   gpurun -- 'tools/ab_bench.sh [pairs] [steps] [extra bench.py flags]'
pairs=${1:-3}; steps=${2:-80}; shift 2 2>/dev/null
get() { python "$1" --steps "$steps" --warmup 10 --no-cpu-baseline "${@:2}" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
a=(); b=()
for i in $(seq "$pairs"); do a+=("$(get _ab/A/bench.py "$@")"); b+=("$(get bench.py "$@")"); done
echo "A: ${a[*]}"; echo "B: ${b[*]}"
python - "${a[*]}" "${b[*]}" <<'PY'
import sys
a=[float(x) for x in sys.argv[1].split()]; b=[float(x) for x in sys.argv[2].split()]
print(f'mean A {sum(a)/len(a):.4f}  mean B {sum(b)/len(b):.4f}  B-A {sum(b)/len(b)-sum(a)/len(a):+.4f} ms')
PY
