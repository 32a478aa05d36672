"""The readout pipeline that serves every frame after the first (xmem_affinity_topk_hinted: bound from a hint -> fp16 filter
(which emits the per-query candidate lists itself)
-> [tighten -> second pass] -> exact fp32 refine) at the sizes the BASELINE configurations run it at:

    B32  N = 51 840   x HW = 1 620   (list capacity 2 048,  13 query tiles)
    C4   N = 921 600  x HW = 3 600   (list capacity 16 384, 29 query tiles)
    C5   N = 4 177 920 x HW = 8 160  (list capacity 16 384, 64 query tiles; rounds 2-5: a 4.3 GB bit matrix)

For every size: the hinted result must equal the un-hinted call (fp32 MFMA select) BIT FOR BIT for a perfect hint, a hint
shifted by one grid cell, a garbage hint (no bound: every pair is a candidate, every list overflows, the tighten pass
makes the bound) and a random hint (a loose bound: lists fill up to their capacity); one more memory holds more exact
duplicates of a query's best match than a list can take, so the second pass overflows as well and the refine scans the
whole memory for that tile (flag2).  Which path ran is read back from the workspace (list counters, per-tile flags), and
sampled queries of the hinted result are compared with the oracle's get_similarity + top-k (model/memory_util.py:7-65).
"""
import ctypes as C

import pytest
import torch

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu

SIZES = [(51840, 1620, 54, 3), (921600, 3600, 80, 3), (4177920, 8160, 120, 2)]
IDS = ['B32_480p_32frames', 'C4_720p_256frames', 'C5_1080p_512frames']


def _flags_and_counts(n, hw):
    """(pass-1 flags, pass-2 flags, list lengths) of the LAST hinted call, read from its workspace."""
    from xmem2_amd import ops
    from xmem2_amd._lib import load
    o = [C.c_size_t(), C.c_size_t(), C.c_size_t()]
    assert load().xmem_affinity_debug_offsets(n, hw, *[C.byref(x) for x in o]) == 0
    torch.cuda.synchronize()
    ws = ops.workspace(0, torch.device('cuda', torch.cuda.current_device()), 'affinity')
    nt = (hw + 127) // 128
    flags = ws[o[1].value:o[1].value + 8 * nt].view(torch.int32).cpu()
    cnt = ws[o[0].value:o[0].value + 4 * hw].view(torch.int32).cpu()
    return flags[:nt], flags[nt:], cnt


def _list_cap(n):
    c = 2048
    while c < n // 64 and c < 16384:
        c *= 2
    return c


def _list_stride(n):
    """Capacity of a list in the SECOND pass (= allocation stride): four times the first pass's, at most 16384."""
    c = _list_cap(n)
    return 16384 if c >= 4096 else 4 * c


def _make(n, hw, nseg, seed):
    gen = torch.Generator(device='cuda').manual_seed(seed)
    mk = torch.randn(n, 64, generator=gen, device='cuda') * 0.9
    ms = torch.rand(n, generator=gen, device='cuda') * 3 + 1
    qk = torch.randn(hw, 64, generator=gen, device='cuda') * 0.9
    qe = torch.rand(hw, 64, generator=gen, device='cuda') * 0.9 + 0.05
    # memory frames resemble the query frame (a video): every 7th query has a near copy in every "frame" of HW rows
    frames = n // hw
    for f in range(0, frames, max(1, frames // 16)):
        rows = torch.arange(0, hw, 7, device='cuda')
        mk[f * hw + rows] = qk[rows] + 0.05 * torch.randn(rows.numel(), 64, generator=gen, device='cuda')
    cuts = [0] + sorted(torch.randint(1, n, (nseg - 1,), generator=torch.Generator().manual_seed(seed)).tolist()) + [n]
    return mk, ms, qk, qe, cuts


def _oracle_check(mk, ms, qk, qe, idx, sims, n_pick, tag):
    hw = qk.shape[0]
    pick = torch.randperm(hw, generator=torch.Generator().manual_seed(5))[:n_pick]
    sim = R.get_similarity(mk.cpu().t().unsqueeze(0), ms.cpu().view(1, 1, -1), qk.cpu()[pick].t().unsqueeze(0),
                           qe.cpu()[pick].t().unsqueeze(0))
    vals_ref, idx_ref = torch.topk(sim[0], 30, dim=0)
    got_v, got_i = sims.cpu()[pick], idx.cpu().long()[pick]
    assert torch.allclose(got_v, vals_ref.t(), rtol=1e-4, atol=1e-4), f'{tag}: top-k values differ from the oracle'
    same = (torch.sort(got_i, 1)[0] == torch.sort(idx_ref.t(), 1)[0]).all(1).float().mean()
    assert float(same) > 0.9, f'{tag}: only {float(same):.2f} of the sampled queries have the oracle\'s index set'


@pytest.mark.parametrize('n,hw,gw,nseg', SIZES, ids=IDS)
def test_hinted_filter_equals_unhinted_at_served_sizes(n, hw, gw, nseg):
    from xmem2_amd import ops
    mk, ms, qk, qe, cuts = _make(n, hw, nseg, seed=n)
    segs = [(mk[a:b], ms[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    lcap = _list_cap(n)
    w0, i0, s0 = ops.affinity_topk(segs, qk, qe, 30, want_sim=True)               # un-hinted: fp32 MFMA select
    torch.cuda.synchronize()
    assert bool((s0[:, :-1] >= s0[:, 1:]).all()) and int(i0.min()) >= 0 and int(i0.max()) < n
    rnd = torch.randint(0, n, (hw, 30), generator=torch.Generator().manual_seed(3)).int().cuda()
    hints = {
        'perfect': (i0, sizes, gw),
        'shifted by one cell': (torch.roll(i0, 1, 0).contiguous(), sizes, gw),
        'shifted by one row': (torch.roll(i0, gw, 0).contiguous(), sizes, gw),
        'garbage': (torch.zeros_like(i0), sizes, gw),
        'random': (rnd, sizes, 0),
    }
    seen = {}
    for name, h in hints.items():
        w, i, sv = ops.affinity_topk(segs, qk, qe, 30, want_sim=True, hint=h)
        f1, f2, cnt = _flags_and_counts(n, hw)
        seen[name] = (int(f1.sum()), int(f2.sum()), int(cnt.max()))
        assert torch.equal(sv, s0), f'{name}: similarities differ from the un-hinted call'
        assert torch.equal(i, i0) and torch.equal(w, w0), f'{name}: indices / weights differ from the un-hinted call'
    print(f'N={n} HW={hw} lcap={lcap}: (flag1 tiles, flag2 tiles, longest list) per hint: {seen}')
    nt = (hw + 127) // 128
    assert seen['perfect'][0] == 0, 'a perfect hint must not need the second pass'
    assert seen['garbage'][0] == nt, 'no bound: every query tile must go through tighten + second pass'
    assert seen['random'][0] == nt, 'a bound from random elements keeps ~half the memory: every list must reach its capacity'
    _oracle_check(mk, ms, qk, qe, i, sv, 32 if n > 2_000_000 else 48, 'random hint')


@pytest.mark.parametrize('n,hw,gw,nseg', SIZES, ids=IDS)
def test_hinted_filter_full_scan_on_exact_ties_at_served_sizes(n, hw, gw, nseg):
    """More exact duplicates of a query's best match than a candidate list holds even in the second pass (whose lists are four
    times as long): the list overflows in pass 1 AND, with the tightened bound (= the tied value), in pass 2 - flag2 - and the
    refine evaluates every memory element for that tile.
    The result (lowest indices among the ties, as torch.topk on a stable sort would give) equals the un-hinted call."""
    from xmem2_amd import ops
    mk, ms, qk, qe, cuts = _make(n, hw, nseg, seed=n + 1)
    lcap = _list_stride(n)
    dup = lcap + 700
    targets = [5, hw // 2 + 3, hw - 2]                                          # three queries in three different 128-query tiles
    for j, q in enumerate(targets):
        lo = (j + 1) * (n // 5)
        mk[lo:lo + dup] = qk[q]
        ms[lo:lo + dup] = 2.0
    segs = [(mk[a:b], ms[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    w0, i0, s0 = ops.affinity_topk(segs, qk, qe, 30, want_sim=True)
    torch.cuda.synchronize()
    for j, q in enumerate(targets):                                               # the un-hinted call itself: 30 lowest duplicates
        lo = (j + 1) * (n // 5)
        assert sorted(i0[q].tolist()) == list(range(lo, lo + 30)), f'query {q}: ties must resolve to the lowest indices'
    for name, h in {'perfect': (i0, sizes, gw), 'shifted': (torch.roll(i0, 1, 0).contiguous(), sizes, gw)}.items():
        w, i, sv = ops.affinity_topk(segs, qk, qe, 30, want_sim=True, hint=h)
        f1, f2, cnt = _flags_and_counts(n, hw)
        assert torch.equal(sv, s0) and torch.equal(i, i0) and torch.equal(w, w0), f'{name}: differs from the un-hinted call'
        tiles = sorted({q // 128 for q in targets})
        assert all(int(f2[t]) != 0 for t in tiles), f'{name}: the tiles of the tied queries must take the full scan (flag2 = {f2.tolist()})'
        assert int(cnt.max()) >= lcap
    _oracle_check(mk, ms, qk, qe, i, sv, 24, 'ties')
