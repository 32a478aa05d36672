"""Print a rocprofv3 --kernel-trace --stats kernel_stats.csv compactly (kernel names contain commas: csv module, not cut)."""
import csv
import sys

for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        name = r['Name']
        if name.startswith('__amd_rocclr'):
            continue
        short = name.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
        print(f"   {short[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}  max {float(r['MaxNs']) / 1e3:9.1f}")
