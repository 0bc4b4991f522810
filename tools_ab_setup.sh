#!/bin/bash
# Dev tool: snapshot the package as built from HEAD into _ab/A_pkg (+ _ab/bench_A.py) so the working
# tree can be benchmarked against it on the SAME GPU box (box-to-box variance is ~1-2 %):
#   ./tools_ab_setup.sh && gpurun -- 'python _ab/bench_A.py --steps 100 ...; python bench.py --steps 100 ...'
set -e
cd "$(dirname "$0")"
rm -rf _ab && mkdir -p _ab
git stash -q
make -C multimodal-long-transformer-2021_amd/csrc -j8 >/dev/null
cp -r multimodal-long-transformer-2021_amd _ab/A_pkg && cp bench.py _ab/bench_A.py
git stash pop -q
touch multimodal-long-transformer-2021_amd/csrc/*.hip
make -C multimodal-long-transformer-2021_amd/csrc -j8 >/dev/null
sed -i "s#os.path.join(ROOT, 'multimodal-long-transformer-2021_amd')#os.path.join(ROOT, '_ab', 'A_pkg')#; s#^ROOT = .*#ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))#" _ab/bench_A.py
