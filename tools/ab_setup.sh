#!/bin/bash
# Dev tool: build the package as of a git revision (default HEAD) under _ab/A so the working tree can
# be benchmarked against it on the SAME GPU box (box-to-box variance is ~1-2 %):
#   ./tools/ab_setup.sh [rev] && gpurun -- 'python _ab/A/bench.py --steps 100 --no-cpu-baseline; python bench.py --steps 100 --no-cpu-baseline'
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
rm -rf _ab/A && mkdir -p _ab/A
git archive "$rev" multimodal-long-transformer-2021_amd include bench.py oracle profiles/attn_traffic.json | tar -x -C _ab/A
make -C _ab/A/multimodal-long-transformer-2021_amd/csrc -j8 >/dev/null
echo "built $rev under _ab/A"
