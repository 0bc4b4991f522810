"""Dev tool: ONE replayed train step out of a rocprofv3 kernel trace (CSV): the kernels between the last two
`mmt::write_step_scalars_kernel` launches (one per replayed step), summed per kernel name and per family.  A whole-run
`--stats` table divided by the number of steps also counts set-up work (parameter initialisation, GEMM tuning, the
eager warm-up steps' framework kernels) -- this does not.
  python tools/step_trace.py <dir with *_kernel_trace.csv> [out.csv]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'write_step_scalars' in r['Kernel_Name']]
if len(marks) < 3:
  sys.exit('fewer than three replayed steps in the trace')
a, b = marks[-3], marks[-2]          # the last COMPLETE step that is followed by another one
step = rows[a:b]
wall = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6
agg = defaultdict(lambda: [0, 0.0])
fam = defaultdict(lambda: [0, 0.0])
for r in step:
  n = r['Kernel_Name']
  us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
  agg[n][0] += 1; agg[n][1] += us
  k = ('hand-written (mmt::)' if 'mmt' in n else 'library GEMM (Cijk)' if 'Cijk' in n else 'framework (torch / rocprim / copies)')
  fam[k][0] += 1; fam[k][1] += us
busy = sum(v[1] for v in agg.values())
print(f'one replayed step: {len(step)} kernels, wall {wall:.3f} ms, kernel time {busy / 1e3:.3f} ms')
for k, (c, us) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
  print(f'  {k:40s} {c:4d} kernels {us / 1e3:8.3f} ms')
out = sys.argv[2] if len(sys.argv) > 2 else None
table = sorted(agg.items(), key=lambda kv: -kv[1][1])
for n, (c, us) in table[:24]:
  print(f'{us / 1e3:8.3f} ms {c:4d} x {us / c:8.1f} us  {n[:100]}')
if out:
  with open(out, 'w', newline='') as fo:
    w = csv.writer(fo)
    w.writerow(['Name', 'CallsPerStep', 'TotalUsPerStep', 'AverageUs'])
    for n, (c, us) in table:
      w.writerow([n, c, f'{us:.1f}', f'{us / c:.2f}'])
    w.writerow(['# one replayed step', len(step), f'{busy:.1f}', f'wall_ms={wall:.3f}'])
