"""The trace-derived figures of the committed B32 bench line must follow from the committed trace (VERDICT round 3, item 2:
`roofline.frac` reproducible from `profiles/`): `profiles/r0N_bench_b32_timed_region_kernel_trace.csv` (rounds 4 - 6) is the rocprofv3 kernel trace
of the line's traced child cut to its timed region; this test re-derives `roofline.achieved / frac` (mean and median), the
conv-family time of `conv_roofline` and the family table from it with bench.py's own parser.  No GPU, no reference."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, 'profiles')


@pytest.mark.parametrize('rnd', ['r04', 'r05', 'r06'])
def test_b32_roofline_follows_from_the_committed_trace(rnd):
    import bench
    LINE, TRACE = os.path.join(PROF, f'{rnd}_bench_b32.json'), os.path.join(PROF, f'{rnd}_bench_b32_timed_region_kernel_trace.csv')
    if not (os.path.exists(LINE) and os.path.exists(TRACE)):
        pytest.skip(f'{rnd} profiles not present')
    line = json.loads(open(LINE).read().strip().splitlines()[-1])
    tr = bench.parse_kernel_trace(TRACE)
    r, c = line['roofline'], line['conv_roofline']
    assert r['source'] == 'timed_region_trace'
    fk = [k for k in tr['kernels'] if 'filter16' in k and '<false' in k]
    assert len(fk) == 1, fk
    n, total_ns = tr['kernels'][fk[0]]
    frames = r['kernel_launches_timed']
    assert n == frames                                          # one pass-1 filter launch per frame of the window
    mean_us, med_us = total_ns / n / 1e3, tr['median_ns'][fk[0]] / 1e3
    gf = r['algorithmic_gflop_per_call']
    assert gf == pytest.approx(4 * 64 * 51840 * 1620 / 1e9, rel=1e-6)       # F_sim = 4 C_k N HW at B32
    assert r['kernel_avg_us'] == pytest.approx(mean_us, rel=1e-6) and r['kernel_median_us'] == pytest.approx(med_us, rel=1e-6)
    assert r['achieved'] == pytest.approx(gf / (mean_us * 1e-3), rel=1e-6)
    assert r['frac'] == pytest.approx(gf / (mean_us * 1e-3) / bench.PEAK_F16_MFMA_TFLOPS, rel=1e-6)
    assert r['frac_median'] == pytest.approx(gf / (med_us * 1e-3) / bench.PEAK_F16_MFMA_TFLOPS, rel=1e-6)
    assert r['frac_of_sustained'] == pytest.approx(r['achieved'] / bench.SUSTAINED_F16_MFMA_TFLOPS, rel=1e-6)
    # the convolution family: kernel time per frame of the same window, executed MFMA FLOPs over it
    conv_us = tr['families']['conv'][1] / frames / 1e3
    assert c['us_per_frame'] == pytest.approx(conv_us, rel=1e-6)
    assert c['achieved'] == pytest.approx(c['executed_mfma_gflop_per_frame'] / (conv_us * 1e-3), rel=1e-6)
    assert c['frac'] == pytest.approx(c['achieved'] / bench.PEAK_FP32_MFMA_TFLOPS, rel=1e-6)
    assert sum(v['executed_mfma_gflop'] for v in c['executed_by_form'].values()) == pytest.approx(c['executed_mfma_gflop_per_frame'], rel=1e-6)
    if rnd >= 'r05':                                            # schema 5: the select as a whole against the pipe its contraction runs on
        aff_us = tr['families']['affinity'][1] / frames / 1e3
        assert r['call_frac'] == pytest.approx(r['algorithmic_gflop_per_frame'] / (aff_us * 1e-3) / bench.PEAK_F16_MFMA_TFLOPS, rel=1e-6)
        # schema 6 (end of round 6) changed no key: it made the timed region self-contained (bench.py, `schema_note`); round 5's line is schema 5
        assert line['schema'] in (5, bench.SCHEMA) and line['parity']['clear_margin'] == bench.CLEAR_MARGIN
        # round 5: the headline did not use the (then opt-in, then unexplained) early readout; round 6: root cause found, default on
        assert line['config']['early_readout'] is (rnd >= 'r06')
    # the per-frame table committed beside the trace is the same parse
    table = open(os.path.join(PROF, f'{rnd}_bench_b32_timed_region_per_frame.csv')).read()
    assert f'conv,{tr["families"]["conv"][0] / frames:.2f},{conv_us:.1f}' in table


@pytest.mark.skipif(not os.path.exists(os.path.join(PROF, 'r05_bench_b32_pmc_per_frame.json')), reason='round-5 profiles not present')
def test_pmc_figures_are_quoted_only_at_the_digest_they_were_measured_at():
    """bench.py quotes `traffic` only while the PMC file's digest matches the kernel sources + compiler flags of the tree; after a
    kernel edit it reports null until the passes are repeated (tools/pmc_bench.py)."""
    import bench
    from xmem2_amd import build
    pmc = json.load(open(os.path.join(PROF, 'r05_bench_b32_pmc_per_frame.json')))
    quoted = bench.committed_pmc('b32', 'fp32')
    if pmc['source_digest'] == build.source_digest():
        assert quoted is not None and quoted['families']['conv'] > 3e9 and quoted['families']['affinity'] > 5e7
    else:
        assert quoted is None or quoted['file'] != 'profiles/r05_bench_b32_pmc_per_frame.json'
