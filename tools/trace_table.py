"""Per-frame kernel table of the TIMED region of a bench.py kernel trace (the window between the two marker kernels).
usage: trace_table.py kernel_trace.csv > table.csv"""
import csv, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tr = bench.parse_kernel_trace(sys.argv[1])
# frames in the window = launches of the per-frame argmax kernel
frames = max(1, next((v[0] for k, v in tr['kernels'].items() if k.startswith('argmax_u8')), 1))
w = csv.writer(sys.stdout)
tot = sum(v[1] for v in tr['kernels'].values())
w.writerow(['# timed window (marker kernels)', f'{frames} frames', f'{tr["window_ns"] / frames / 1e3:.1f} us/frame wall under the tracer',
            f'GPU busy (union over streams) {tr["busy_ns"] / tr["window_ns"]:.3f}', f'sum of kernel durations {tot / frames / 1e3:.1f} us/frame',
            f'launches/frame {tr["launches"] / frames:.1f}'])
w.writerow(['family', 'launches_per_frame', 'us_per_frame'])
for f, v in sorted(tr['families'].items(), key=lambda kv: -kv[1][1]):
    w.writerow([f, f'{v[0] / frames:.2f}', f'{v[1] / frames / 1e3:.1f}'])
w.writerow(['kernel', 'launches_per_frame', 'avg_us', 'median_us', 'us_per_frame', 'share_of_kernel_time'])
for k, v in sorted(tr['kernels'].items(), key=lambda kv: -kv[1][1]):
    w.writerow([k, f'{v[0] / frames:.2f}', f'{v[1] / v[0] / 1e3:.1f}', f'{tr["median_ns"].get(k, 0.0) / 1e3:.1f}', f'{v[1] / frames / 1e3:.1f}',
                f'{v[1] / tot:.4f}'])
