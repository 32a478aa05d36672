"""Kernel-by-kernel sequence of the timed region of a marker-cut rocprofv3 kernel trace (tools/trace_cut.py output):
   python tools/frame_sequence.py profiles/r04_bench_b32_timed_region_kernel_trace.csv
(a) the main-stream frame with the shortest wall time (no key-encoder batch of the other stream beside it), (b) one key-encoder batch
of the side stream, (c) per-queue sums per frame.  The tracer's timestamps are contiguous on a queue: a kernel that only checks a flag
still shows ~4.6 us - that is the dispatch interval, not work."""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])


def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*\)$', '', n)[:72]


queues = defaultdict(list)
for r in rows:
    if 'xmem_trace_marker' not in r['Kernel_Name']:
        queues[r['Queue_Id']].append(r)
main_q = max(queues, key=lambda q: sum(1 for r in queues[q] if r['Kernel_Name'].startswith('affinity_hint_bound')))
main = queues[main_q]
starts = [i for i, r in enumerate(main) if r['Kernel_Name'].startswith('affinity_hint_bound')]
frames = [(main[a]['s'], main[b]['s'], a, b) for a, b in zip(starts[:-1], starts[1:])]
t0, t1, a, b = min(frames, key=lambda f: f[1] - f[0])
print(f'# (a) main stream (queue {main_q}): the frame with the shortest wall time of {len(frames)}: {(t1 - t0) / 1e3:.1f} us '
      f'(median frame {sorted(f[1] - f[0] for f in frames)[len(frames) // 2] / 1e3:.1f} us, mean {sum(f[1] - f[0] for f in frames) / len(frames) / 1e3:.1f} us)')
print('# start_us, duration_us, kernel')
for r in main[a:b]:
    print(f'{(r["s"] - t0) / 1e3:9.1f},{(r["e"] - r["s"]) / 1e3:8.1f},{short(r["Kernel_Name"])}')
for q, rs in queues.items():
    if q == main_q or not rs:
        continue
    grp, last = [], None
    for r in rs:
        if last is not None and r['s'] - last > 300000:
            break
        grp.append(r); last = r['e']
    print(f'\n# (b) side stream (queue {q}): one batched key-encoder pass (hints of the next frames + the decoder halves that depend on the '
          f'image only), {len(grp)} launches, span {(grp[-1]["e"] - grp[0]["s"]) / 1e3:.1f} us while the main stream keeps running')
    print('# start_us, duration_us, kernel')
    for r in grp:
        print(f'{(r["s"] - grp[0]["s"]) / 1e3:9.1f},{(r["e"] - r["s"]) / 1e3:8.1f},{short(r["Kernel_Name"])}')
nf = len(frames)
span = (main[starts[-1]]['s'] - main[starts[0]]['s']) / 1e3
print(f'\n# (c) per frame over {nf} frames ({span / nf:.1f} us wall per frame under the tracer):')
for q, rs in queues.items():
    inside = [r for r in rs if main[starts[0]]['s'] <= r['s'] < main[starts[-1]]['s']]
    print(f'# queue {q}: {len(inside) / nf:.1f} launches, {sum(r["e"] - r["s"] for r in inside) / nf / 1e3:.1f} us of kernel time per frame')
