"""HBM-side traffic and MFMA-busy fraction per FRAME of bench.py's timed region, from separate rocprofv3 --pmc passes.

    cd /tmp && TMPDIR=/tmp python $REPO/tools/pmc_bench.py [--workload b32] [--steps 30] [--out $REPO/profiles/r02_..._pmc_per_frame.json]

Three passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are collected with --kernel-trace
only), each a child `bench.py --traced-child`, whose timed region is bracketed by `xmem_trace_marker_kernel` launches:
  1. FETCH_SIZE            read bytes  = 2 x FETCH_SIZE x 1024 (gfx950 tallies a 128-B request of a wide coalesced read as 64 B)
  2. WRITE_SIZE            write bytes = WRITE_SIZE x 1024 as reported (uncalibrated); Infinity-Cache hits are counted in both
  3. SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE     mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
The JSON carries the digest of the kernel sources it was measured on: bench.py quotes it only for that exact build.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                             # noqa: E402
from xmem2_amd.build import source_digest                # noqa: E402


def run_pass(counters, args):
    tmp = tempfile.mkdtemp(prefix='xmem_pmc_', dir=os.environ.get('TMPDIR', '/tmp'))
    cmd = ['rocprofv3', '--pmc'] + counters + ['--kernel-trace', '--output-format', 'csv', '-d', tmp, '--', sys.executable,
           os.path.join(ROOT, 'bench.py'), '--traced-child', '--no-cpu-baseline', '--steps', str(args.steps), '--warmup', '10',
           '--workload', args.workload, '--precision', args.precision]
    subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    files = glob.glob(os.path.join(tmp, '**', '*counter_collection.csv'), recursive=True)
    rows = []
    with open(max(files, key=os.path.getsize)) as f:
        for r in csv.DictReader(f):
            rows.append((int(r.get('Dispatch_Id', 0)), r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].strip(), r['Counter_Name'], float(r['Counter_Value'])))
    shutil.rmtree(tmp, ignore_errors=True)
    rows.sort()
    marks = [d for d, n, c, v in rows if n.startswith('xmem_trace_marker_kernel')]
    lo, hi = min(marks), max(marks)
    return [(n, c, v) for d, n, c, v in rows if lo < d < hi]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='b32')
    ap.add_argument('--precision', default='fp32')
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    per_k = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for counters in (['FETCH_SIZE'], ['WRITE_SIZE'], ['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE']):
        for n, c, v in run_pass(counters, args):
            per_k[n][c] += v
            if c == counters[0]:
                launches[(n, c)] += 1
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for n, cs in per_k.items():
        f = bench.family_of(n)
        for c, v in cs.items():
            fam[f][c] += v
    st = args.steps
    out = {'source_digest': source_digest(), 'workload': args.workload, 'precision': args.precision, 'frames': st,
           'note': 'per frame of the timed region (marker window); read = 2 x FETCH_SIZE KB (gfx950 correction), write = WRITE_SIZE KB as '
                   'reported (uncalibrated), Infinity-Cache hits counted; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024)',
           'families': {f: {'read_bytes': 2 * 1024.0 * c.get('FETCH_SIZE', 0.0) / st, 'write_bytes': 1024.0 * c.get('WRITE_SIZE', 0.0) / st}
                        for f, c in fam.items()},
           'mfma_busy': {f: (c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)) if c.get('GRBM_GUI_ACTIVE') else None
                         for f, c in fam.items()},
           'kernels': [{'kernel': n, 'launches_per_frame': launches.get((n, 'FETCH_SIZE'), 0) / st,
                        'read_bytes_per_frame': 2 * 1024.0 * cs.get('FETCH_SIZE', 0.0) / st,
                        'write_bytes_per_frame': 1024.0 * cs.get('WRITE_SIZE', 0.0) / st,
                        'mfma_busy': (cs['SQ_VALU_MFMA_BUSY_CYCLES'] / (cs['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)) if cs.get('GRBM_GUI_ACTIVE') else None}
                       for n, cs in sorted(per_k.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0.0))]}
    text = json.dumps(out, indent=1)
    if args.out:
        with open(args.out, 'w') as f:
            f.write(text)
    print(text if not args.out else f'wrote {args.out}')


if __name__ == '__main__':
    main()
