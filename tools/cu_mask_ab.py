#!/usr/bin/env python
"""A/B of CU-masked streams for the frame pipeline: bench.py B32 with the side stream (batched key encoder)
restricted to n CUs (XMEM_SIDE_CUS, n / 8 per XCD) and optionally the main stream on the complementary CUs (XMEM_MAIN_CUS).
Prints one line per variant; the baseline and the best variant are repeated with the marker-cut kernel trace so that the
mean / median durations of the latency-bound readout kernels can be compared.  Usage: python tools/cu_mask_ab.py [out.txt]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [('baseline', {}),
            ('side64', {'XMEM_SIDE_CUS': '64'}), ('side96', {'XMEM_SIDE_CUS': '96'}),
            ('side128', {'XMEM_SIDE_CUS': '128'}),
            ('side64+main192', {'XMEM_SIDE_CUS': '64', 'XMEM_MAIN_CUS': '64:192'}),
            ('side96+main160', {'XMEM_SIDE_CUS': '96', 'XMEM_MAIN_CUS': '96:160'}),
            ('baseline_again', {})]


def run(env_extra, trace):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-extra-modes', '--plain-steps', '0', '--steps', '200']
    if not trace:
        cmd.append('--no-kernel-trace')
    p = subprocess.run(cmd, env=dict(os.environ, **env_extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    try:
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        return {'error': p.stderr[-600:]}


def main():
    out = open(sys.argv[1], 'w') if len(sys.argv) > 1 else sys.stdout
    res = {}
    for name, env in VARIANTS:
        j = run(env, False)
        res[name] = j.get('value')
        print(f'{name:18s} {env}  frames/s {j.get("value")}  {j.get("error", "")}', file=out, flush=True)
    ok = {k: v for k, v in res.items() if v and not k.startswith('baseline')}
    best = max(ok, key=ok.get) if ok else None
    for name in ('baseline', best):
        if name is None:
            continue
        env = dict(VARIANTS)[name]
        j = run(env, True)
        print(f'--- traced: {name} {env}: frames/s {j.get("value")}', file=out)
        kt = j.get('kernel_trace', {})
        print('    families us/frame:', {k: round(v['us_per_frame'], 1) for k, v in kt.get('families', {}).items()}, file=out)
        for k, v in (j.get('roofline', {}).get('kernels') or {}).items():
            print(f'    {k:60s} mean {v["avg_us"]:7.1f} us  median {v["median_us"]:7.1f} us', file=out)
        out.flush()


if __name__ == '__main__':
    main()
