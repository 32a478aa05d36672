"""Per-queue timeline of the marker window of a `rocprofv3 --kernel-trace` of bench.py: runs of back-to-back kernels per HIP
queue (main stream / side stream), how long each queue is busy, how long BOTH are, and where the main queue idles.
    python tools/stream_timeline.py trace.csv [max_runs]"""
import csv
import sys
from collections import defaultdict


def merge(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    mk = [i for i, r in enumerate(rows) if 'trace_marker' in r['Kernel_Name']]
    if len(mk) >= 2:
        rows = rows[mk[0] + 1:mk[1]]
    for r in rows:
        r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    t0, t1 = rows[0]['s'], max(r['e'] for r in rows)
    frames = sum('argmax_u8' in r['Kernel_Name'] for r in rows)
    byq = defaultdict(list)
    for r in rows:
        byq[r['Queue_Id']].append(r)
    print(f'window {(t1 - t0) / 1e3:.1f} us, {frames} frames -> {(t1 - t0) / 1e3 / max(frames, 1):.1f} us per frame')
    merged = {}
    for q, rs in byq.items():
        merged[q] = merge((r['s'], r['e']) for r in rs)
        busy = sum(e - s for s, e in merged[q])
        print(f'queue {q}: {len(rs)} launches, busy {busy / 1e3:.1f} us ({100.0 * busy / (t1 - t0):.1f} %)')
    qs = sorted(merged, key=lambda q: -len(byq[q]))
    if len(qs) >= 2:
        A, B = merged[qs[0]], merged[qs[1]]
        i = j = 0; ov = 0
        while i < len(A) and j < len(B):
            s, e = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
            if e > s:
                ov += e - s
            if A[i][1] < B[j][1]:
                i += 1
            else:
                j += 1
        print(f'both queues busy {ov / 1e3:.1f} us ({100.0 * ov / (t1 - t0):.1f} %)')
    runs = []
    for q, rs in byq.items():
        cur = None
        for r in rs:
            if cur and r['s'] - cur[2] < 20000:
                cur[2] = max(cur[2], r['e']); cur[3] += 1
            else:
                if cur:
                    runs.append(cur)
                cur = [q, r['s'], r['e'], 1, r['Kernel_Name'][:34]]
        runs.append(cur)
    runs.sort(key=lambda x: x[1])
    lim = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    for x in runs[:lim]:
        print(f'q{x[0]} start {(x[1] - t0) / 1e3:9.1f} us  dur {(x[2] - x[1]) / 1e3:8.1f}  n={x[3]:3d}  first={x[4]}')


if __name__ == '__main__':
    main()
