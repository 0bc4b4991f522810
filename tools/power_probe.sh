#!/bin/bash
# Dev tool: samples GPU clock / power while bench.py runs (is the step power-limited?)
python bench.py --steps 1500 --warmup 10 --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 40); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" | sed -E 's/.*level: //; s/.*Power \(W\): /P=/' | tr '\n' ' '; echo
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
cat gpurun_out/power_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
