"""Dev tool: average duration (us) of the kernels whose name contains a pattern, from a rocprofv3 *_kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
  if any(p in r['Name'] for p in sys.argv[2:]):
    print(f"{float(r['AverageNs'])/1e3:8.1f} us  x{r['Calls']:>4}  {r['Name'][:70]}")
