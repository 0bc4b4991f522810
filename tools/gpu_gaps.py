"""Dev tool: GPU busy / idle time per train step from a rocprofv3 --kernel-trace CSV.
  rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --mode train_step
  python tools/gpu_gaps.py DIR/t_kernel_trace.csv
Steps are delimited by the AdamW kernel (one optimiser launch group per step)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
# step boundaries: first adamw kernel after a non-adamw kernel
bounds = [i for i, e in enumerate(ev) if 'adamw' in e[2] and (i == 0 or 'adamw' not in ev[i - 1][2])]
print('steps seen', len(bounds))
for a, b in list(zip(bounds[:-1], bounds[1:]))[-4:]:
  seg = ev[a:b]
  t0, t1 = seg[0][0], ev[b][0]
  busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
  for s, e, _ in seg[1:]:
    if s > cur_e:
      busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
      cur_e = max(cur_e, e)
  busy += cur_e - cur_s
  gaps = sorted(((seg[i + 1][0] - max(x[1] for x in seg[:i + 1][-4:]), seg[i][2][:40], seg[i + 1][2][:40]) for i in range(len(seg) - 1)), reverse=True)
  print(f'step wall {(t1 - t0) / 1e3:9.1f} us  busy(union) {busy / 1e3:9.1f} us  idle {(t1 - t0 - busy) / 1e3:8.1f} us  kernels {len(seg)}')
  print('   largest gaps (us):', [(round(g / 1e3, 1), a, b) for g, a, b in gaps[:6]])
