"""Sum the counters of one kernel (substring match) from a rocprofv3 counter_collection.csv."""
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(acc):
    print(f'{k:32s} launches {n[k]:4d}  per-launch {acc[k] / n[k]:.4e}')
