"""Host launch time vs GPU start time per kernel (rocprofv3 --kernel-trace --hip-trace CSVs): where does a queue wait although
its next kernel was submitted long ago?   python launch_lag.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv>"""
import csv, glob, sys, os
d = sys.argv[1]
kt = max(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getsize)
ht = max(glob.glob(os.path.join(d, '**', '*hip_api_trace.csv'), recursive=True), key=os.path.getsize)
api = {}
for r in csv.DictReader(open(ht)):
    api[r['Correlation_Id']] = (r['Function'], int(r['Start_Timestamp']), int(r['End_Timestamp']))
rows = list(csv.DictReader(open(kt)))
mk = [i for i, r in enumerate(rows) if 'trace_marker' in r['Kernel_Name']]
rows = rows[mk[0] + 1:mk[1]] if len(mk) >= 2 else rows
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
t0 = rows[0]['s']
prev_end = {}
n = 0
for r in rows:
    q = r['Queue_Id']
    gap = (r['s'] - prev_end.get(q, r['s'])) / 1e3
    a = api.get(r['Correlation_Id'])
    if gap > 300 and n < 40:
        n += 1
        lag = (r['s'] - a[1]) / 1e3 if a else float('nan')
        print(f"q{q} idle {gap:8.1f} us before {r['Kernel_Name'][:36]:36s} at {(r['s'] - t0) / 1e3:9.1f} us; "
              f"submitted by {a[0] if a else '?'} {lag:8.1f} us before it started")
    prev_end[q] = max(prev_end.get(q, 0), r['e'])
