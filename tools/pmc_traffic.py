"""Dev tool: per-kernel means of FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc passes -> profiles/attn_fwd_traffic.json.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o f -- python3 tools/bwd_timing.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o w -- python3 tools/bwd_timing.py
  python tools/pmc_traffic.py
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2 for wide streaming reads, WRITE_SIZE as is; unit KB."""
import csv, json, collections
def means(path, counter):
  acc = collections.defaultdict(list)
  for r in csv.DictReader(open(path)):
    if r['Counter_Name'] == counter and ('attn' in r['Kernel_Name'] or 'drel' in r['Kernel_Name']):
      acc[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
  return {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in acc.items()}     # skip the first (cold) quarter
f = means('gpurun_out/pmc_f/f_counter_collection.csv', 'FETCH_SIZE')
w = means('gpurun_out/pmc_w/w_counter_collection.csv', 'WRITE_SIZE')
fwd = [k for k in f if 'attn_fwd_band' in k][0]
comb = [k for k in f if 'rows_combine' in k][0]
fetch, write = f[fwd] + f[comb], w[fwd] + w[comb]
out = {
  'kernel': 'one attention-forward call (config 3, B=4): attn_fwd_band_bf16_kernel<32,true> + attn_rows_combine_kernel',
  'FETCH_SIZE_KB': round(fetch, 1), 'WRITE_SIZE_KB': round(write, 1),
  'method': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (tools/bwd_timing.py, '
            'tools/pmc_traffic.py); gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE x2 for wide (16 B/lane) '
            'streaming reads, WRITE_SIZE as is; unit KB',
  'hbm_bytes_per_launch': int((2 * fetch + write) * 1024),
  'algorithmic_bytes_per_launch': 101649408,
  'note': 're-measured at round-1 v15; all_kernels_raw_KB = per-launch means of every attention kernel of one forward+backward call',
  'all_kernels_raw_KB': {k: {'FETCH_SIZE': round(f[k], 1), 'WRITE_SIZE': round(w.get(k, 0.0), 1)} for k in f},
}
json.dump(out, open('profiles/attn_fwd_traffic.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
