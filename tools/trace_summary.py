"""Per-frame kernel summary of the TIMED region of a `rocprofv3 --kernel-trace` of bench.py.
The timed frames are located through the per-frame argmax kernel: the longest run of frames with a steady period."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
marks = [r['s'] for r in rows if r['Kernel_Name'].startswith('argmax_u8')]
gaps = [b - a for a, b in zip(marks, marks[1:])]
med = sorted(gaps)[len(gaps) // 2]
best, cur = (0, 0), 0
for i, g in enumerate(gaps):                       # longest run of gaps within 2x of the median period
    if g < 2.0 * med:
        cur += 1
        if cur > best[1]: best = (i - cur + 1, cur)
    else:
        cur = 0
i0, n = best
t0, t1 = marks[i0], marks[i0 + n]
sel = [r for r in rows if t0 <= r['s'] < t1]
frames = n
g = collections.defaultdict(lambda: [0, 0])
for r in sel:
    name = r['Kernel_Name'].split('(')[0]
    g[name][0] += 1; g[name][1] += r['e'] - r['s']
busy, last = 0, t0
for r in sel:                                       # union of kernel intervals (two streams overlap)
    a, b = max(r['s'], last), r['e']
    if b > a: busy += b - a; last = b
tot = sum(v[1] for v in g.values())
out = csv.writer(sys.stdout)
out.writerow(['# timed window', f'{frames} frames', f'{(t1 - t0) / frames / 1e3:.1f} us/frame wall', f'GPU busy (union) {busy / (t1 - t0):.3f}',
              f'sum of kernel durations {tot / frames / 1e3:.1f} us/frame', f'kernels/frame {len(sel) / frames:.1f}'])
out.writerow(['kernel', 'launches_per_frame', 'avg_us', 'us_per_frame', 'share_of_kernel_time'])
for k, v in sorted(g.items(), key=lambda kv: -kv[1][1]):
    out.writerow([k, f'{v[0] / frames:.2f}', f'{v[1] / v[0] / 1e3:.1f}', f'{v[1] / frames / 1e3:.1f}', f'{v[1] / tot:.4f}'])
