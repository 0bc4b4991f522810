"""Dev tool: per-kernel means of the SQ / SQC counters collected by the rocprofv3 passes listed in DESIGN.md section 5
(gpurun_out/sq1..sq4/s_counter_collection.csv over tools/attn_pmc_workload.py) -> profiles/r02_attn_pmc_sq.json."""
import csv, json, collections, glob, subprocess
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob('gpurun_out/sq[0-9]/s_counter_collection.csv')):
  for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'].split('(')[0]
    if 'attn' in k or 'drel' in k:
      acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out = {'commit': subprocess.check_output(['git', 'rev-parse', '--short=12', 'HEAD']).decode().strip(),
       'workload': 'tools/attn_pmc_workload.py (config 3, B=4, dropout 0.1); means over the last 3/4 of 24-25 launches',
       'units': 'SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves (MI355X_MICROARCH.md)',
       'kernels': {}}
for k, cs in acc.items():
  out['kernels'][k] = {c: round(sum(v[len(v) // 4:]) / len(v[len(v) // 4:]), 1) for c, v in sorted(cs.items())}
json.dump(out, open('profiles/r02_attn_pmc_sq.json', 'w'), indent=1)
for k, cs in out['kernels'].items():
  print(k)
  wc = cs.get('SQ_WAVE_CYCLES', 0)
  for c, v in cs.items():
    extra = f'  ({100 * v / wc:.1f} % of wave cycles)' if wc and (c.startswith('SQ_WAIT') or c.startswith('SQ_ACTIVE')) else ''
    print(f'   {c:28s} {v:14.0f}{extra}')
