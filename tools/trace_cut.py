"""Cut a rocprofv3 kernel_trace.csv of bench.py to its timed region (the rows between the two xmem_trace_marker_kernel launches,
markers included), keeping the columns bench.parse_kernel_trace reads: the small file profiles/ keeps so that every figure of the
committed bench line can be recomputed (python tools/trace_table.py <cut.csv>).
usage: trace_cut.py kernel_trace.csv > cut.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marks = sorted(int(r['Start_Timestamp']) for r in rows if r['Kernel_Name'].startswith('xmem_trace_marker_kernel'))
if len(marks) < 2:
    raise SystemExit('trace markers not found')
t0, t1 = marks[0], marks[-1]
keep = [r for r in rows if t0 <= int(r['Start_Timestamp']) <= t1]
cols = ['Kernel_Name', 'Start_Timestamp', 'End_Timestamp']
extra = [c for c in ('Queue_Id', 'Stream_Id', 'VGPR_Count', 'LDS_Block_Size', 'Grid_Size', 'Workgroup_Size') if c in rows[0]]
w = csv.DictWriter(sys.stdout, fieldnames=cols + extra, extrasaction='ignore')
w.writeheader()
for r in sorted(keep, key=lambda r: int(r['Start_Timestamp'])):
    w.writerow(r)
