"""CPU check of the error bound behind the fp16 filter of the memory readout (xmem2_amd/csrc/affinity_filter.hip).

The kernels build one fp16 operand row per memory element and per query such that their fp32-accumulated dot product is an
UPPER estimate of the exact fp32 similarity.  This file restates that construction in numpy (float16 roundings included) and
checks the claim `estimate >= exact` on random and hostile inputs - the property the exactness of the GPU path rests on.
Not a product path: test infrastructure only (the GPU tests compare the real kernels with the fp32 select bit for bit)."""
import numpy as np
import pytest

KAPPA, ACC, ABS = np.float32(1.07e-3), np.float32(4.5e-5), np.float32(3e-7)      # affinity_common.hpp
F32 = np.float32


def f16_up(v):
    """affinity_common.hpp:f16_up - a float16 that is >= v for v >= 0."""
    with np.errstate(over='ignore'):
        return (v.astype(F32) * F32(1.001) + F32(6e-8)).astype(np.float16)


def memory_rows(x, ms):
    """affinity_rows16_kernel: [ms/8 x^2 | ms/8 x | msr_hi, msr_hi, msr_lo, kA, kB, |msr|^, zA, 0...] in float16."""
    msr = (ms * F32(0.125)).astype(F32)
    q = (x * x).astype(F32)
    with np.errstate(over='ignore', invalid='ignore'):
        a2 = (msr[:, None] * q).astype(F32); a1 = (msr[:, None] * x).astype(F32)
        An = np.sqrt((q * q).sum(1, dtype=F32)) * F32(1.0001); Bn = np.sqrt(q.sum(1, dtype=F32)) * F32(1.0001)
        am = np.abs(msr)
        mx = np.maximum(np.maximum(np.abs(a2).max(1), np.abs(a1).max(1)), am)
        z = ABS * (An + Bn) * am * F32(1024)
        mh = msr.astype(np.float16)
        aug = np.zeros((x.shape[0], 16), np.float16)
        aug[:, 0] = mh; aug[:, 1] = mh; aug[:, 2] = (msr - mh.astype(F32)).astype(np.float16)
        aug[:, 3] = f16_up(KAPPA * An * am); aug[:, 4] = f16_up(KAPPA * Bn * am); aug[:, 5] = f16_up(am)
        ok = (mx < F32(6.5e4)) & (z < F32(6.5e4))
        aug[:, 6] = np.where(ok, f16_up(z), np.float16(np.inf))
        return np.concatenate([a2.astype(np.float16), a1.astype(np.float16), aug], 1)


def query_rows(k, e):
    """prep part of affinity_hint_bound_kernel: [-e | 2ke | -bs_hi, -bs_lo, -bs_hi, C^, D^, mq^, 2^-10, 0...] and b_sq."""
    ke2 = (F32(2) * (k * e).astype(F32)).astype(F32)
    bsq = np.zeros(k.shape[0], F32)
    for c in range(k.shape[1]):                                    # the order does not matter for the bound (ACC covers it)
        bsq = (bsq + (e[:, c] * (k[:, c] * k[:, c]).astype(F32)).astype(F32)).astype(F32)
    with np.errstate(over='ignore', invalid='ignore'):
        C = np.sqrt((e * e).sum(1, dtype=F32)) * F32(1.0001); D = np.sqrt((ke2 * ke2).sum(1, dtype=F32)) * F32(1.0001)
        mx = np.maximum(np.maximum(np.abs(e).max(1), np.abs(ke2).max(1)), np.abs(bsq))
        bad = ~(mx < F32(6.5e4))
        C = np.where(bad, F32(np.inf), C); D = np.where(bad, F32(np.inf), D)
        bh = bsq.astype(np.float16)
        aug = np.zeros((k.shape[0], 16), np.float16)
        aug[:, 0] = -bh; aug[:, 2] = -bh; aug[:, 1] = -((bsq - bh.astype(F32)).astype(np.float16))
        aug[:, 3] = f16_up(C); aug[:, 4] = f16_up(D)
        aug[:, 5] = f16_up(ACC * np.abs(bsq) + ABS * (C + D)); aug[:, 6] = np.float16(2.0 ** -10)
        return np.concatenate([(-e).astype(np.float16), ke2.astype(np.float16), aug], 1), bsq


def exact_similarity(x, ms, k, e, bsq):
    """The value both pipelines output: ((sum x^2 (-e) + x (2ke)) - b_sq) * ms / 8 in fp32 (float64 here: the fp32 chains'
    own rounding is inside the bound's ACC / KAPPA terms, so the estimate must dominate this value plus that rounding)."""
    ke2 = (F32(2) * (k * e).astype(F32)).astype(np.float64)
    q = (x * x).astype(F32).astype(np.float64)
    t = q @ (-e.astype(np.float64)).T + x.astype(np.float64) @ ke2.T - bsq.astype(np.float64)[None, :]
    absum = q @ np.abs(e.astype(np.float64)).T + np.abs(x.astype(np.float64)) @ np.abs(ke2).T + np.abs(bsq.astype(np.float64))[None, :]
    msr = (ms * F32(0.125)).astype(np.float64)[:, None]
    return t * msr, absum * np.abs(msr)


CASES = {
    'network-like': dict(xs=0.9, qs=0.9, ms=(1.0, 4.0)),
    'small values (fp16 subnormals)': dict(xs=1e-3, qs=1e-3, ms=(1.0, 4.0), es=1e-2),
    'large shrinkage': dict(xs=0.9, qs=0.9, ms=(1.0, 4000.0)),
    'large keys': dict(xs=12.0, qs=9.0, ms=(0.5, 2.0)),
}


@pytest.mark.parametrize('case', list(CASES))
def test_fp16_estimate_dominates_the_exact_similarity(case):
    c = CASES[case]
    rng = np.random.default_rng(7)
    n, hw, ck = 3000, 200, 64
    x = (rng.standard_normal((n, ck)) * c['xs']).astype(F32)
    ms = rng.uniform(*c['ms'], n).astype(F32)
    k = (rng.standard_normal((hw, ck)) * c['qs']).astype(F32)
    e = (rng.uniform(0.05, 0.95, (hw, ck)) * c.get('es', 1.0)).astype(F32)
    R = memory_rows(x, ms); (Q, bsq) = query_rows(k, e)
    assert np.isfinite(R.astype(F32)).all() and np.isfinite(Q.astype(F32)).all()
    est = R.astype(F32) @ Q.astype(F32).T                              # fp16 x fp16 products are exact in fp32; fp32 accumulation
    S, mag = exact_similarity(x, ms, k, e, bsq)
    # the fp32 evaluation the estimate is compared with is itself within 256 * 2^-24 * sum|terms| of this float64 value, and
    # numpy's fp32 accumulation of `est` within 144 * 2^-24: both are part of KAPPA / ACC (2.3e-5 + 0.9e-5 < 4.5e-5)
    assert (est.astype(np.float64) >= S).all(), f'{case}: estimate below the exact value by {float((S - est).max()):.3e}'
    # and it is tight enough to be useful: the margin is a few percent of the spread of the similarities
    margin = est.astype(np.float64) - S
    assert np.median(margin) < 0.05 * S.std() + 1e-6, f'{case}: median margin {np.median(margin):.3e} vs spread {S.std():.3e}'
    assert (margin <= 3.2e-3 * mag + 1e-5).all()                       # 2 * KAPPA * sum|terms| (+ the rounded-up factors)


def test_operands_outside_the_fp16_range_keep_every_pair():
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((64, 64)) * 0.9).astype(F32); x[::4] *= 400.0          # x^2 ~ 1e5..1e6: inf in fp16
    ms = rng.uniform(1, 4, 64).astype(F32)
    k = (rng.standard_normal((32, 64)) * 0.9).astype(F32); k[::5] *= 300.0          # b_sq beyond the fp16 range
    e = rng.uniform(0.05, 0.95, (32, 64)).astype(F32)
    R = memory_rows(x, ms); (Q, _) = query_rows(k, e)
    with np.errstate(invalid='ignore', over='ignore'):
        est = R.astype(F32) @ Q.astype(F32).T
    bad_rows = ~np.isfinite(R[:, 128 + 6].astype(F32)); bad_q = ~np.isfinite(Q[:, 128 + 3].astype(F32))
    assert bad_rows[::4].all() and bad_q[::5].all()
    flagged = est[bad_rows][:, :] ; flagged_q = est[:, bad_q]
    # the filter keeps a pair unless `estimate < tau`: +inf and NaN both fail that test for every finite tau
    assert (~(flagged < 1e30)).all() and (~(flagged_q < 1e30)).all()
