"""The fp16 filter's memory operand rows are kept by the stores (KeyValueMemoryStore._r16) instead of being derived per call:
they must always equal what xmem_affinity_rows16 makes of the store's current (key, shrinkage) rows - after add, replace_at,
sieve_by_range, remove_obsolete_features and arena growth - and a readout with caller-kept rows must equal one without, bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fresh_rows(store):
    from xmem2_amd import ops
    n = store.size
    out = torch.empty((n, ops.ROWS16_FLOATS), dtype=torch.float32, device='cuda')
    return ops.affinity_rows16(store.key_rows().contiguous(), store.shrinkage_rows().contiguous() if store.shrinkage_rows() is not None else None, out)


def _same(store, tag):
    got, want = store.rows16(), _fresh_rows(store)
    torch.cuda.synchronize()
    assert got.shape == want.shape and torch.equal(got.view(torch.int32), want.view(torch.int32)), f'{tag}: kept operand rows are stale'


def test_store_keeps_operand_rows_in_step():
    from xmem2_amd.kv_memory_store import KeyValueMemoryStore
    g = torch.Generator(device='cuda').manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device='cuda')
    st = KeyValueMemoryStore(count_usage=True)
    hw = 300
    for f in range(5):                                              # appends (the arena grows: 4096 rows at first)
        st.add(rnd(hw, 64), rnd(2, hw, 512), torch.rand(hw, generator=g, device='cuda') * 3 + 1, torch.rand(hw, 64, generator=g, device='cuda'), [1, 2])
        _same(st, f'add {f}')
    for f in range(12):                                             # beyond the first capacity -> reallocation
        st.add(rnd(hw, 64), rnd(2, hw, 512), torch.rand(hw, generator=g, device='cuda') * 3 + 1, torch.rand(hw, 64, generator=g, device='cuda'), [1, 2])
    assert st.size == 17 * hw
    _same(st, 'after growth')
    st.replace_at(3, rnd(hw, 64), rnd(2, hw, 512), torch.rand(hw, generator=g, device='cuda') + 1, torch.rand(hw, 64, generator=g, device='cuda'))
    _same(st, 'replace_at')
    st.sieve_by_range(0, -5 * hw, min_size=5 * hw + hw)             # consolidation-style sieve: keep the last 5 frames
    assert st.size == 5 * hw
    _same(st, 'sieve (suffix kept)')
    st.remove_at(hw, hw)                                            # a block from the middle
    _same(st, 'remove_at')
    st.update_usage_from(torch.rand(50, 30, device='cuda'), torch.randint(0, st.size, (50, 30), device='cuda').int(), 0)
    one = KeyValueMemoryStore(count_usage=True)                     # single group: eviction allowed
    for f in range(4):
        one.add(rnd(hw, 64), rnd(1, hw, 512), torch.rand(hw, generator=g, device='cuda') + 1, None, [1])
    one.update_usage_from(torch.rand(64, 30, device='cuda'), torch.randint(0, one.size, (64, 30), device='cuda').int(), 0)
    one.remove_obsolete_features(2 * hw)
    assert 0 < one.size <= 2 * hw + hw
    _same(one, 'remove_obsolete_features')
    no_s = KeyValueMemoryStore(count_usage=False)                   # a store without shrinkage (treated as 1)
    no_s.add(rnd(hw, 64), rnd(1, hw, 512), None, None, [1])
    _same(no_s, 'no shrinkage')


@pytest.mark.parametrize('n,hw', [(20000, 300), (51840, 1620)])
def test_readout_with_kept_rows_equals_readout_without(n, hw):
    from xmem2_amd import ops
    g = torch.Generator(device='cuda').manual_seed(n)
    mk = torch.randn(n, 64, generator=g, device='cuda') * 0.9
    ms = torch.rand(n, generator=g, device='cuda') * 3 + 1
    qk = torch.randn(hw, 64, generator=g, device='cuda') * 0.9
    qe = torch.rand(hw, 64, generator=g, device='cuda') * 0.9 + 0.05
    cuts = [0, n // 3 + 7, n // 3 + 7, n]                          # three slots, the middle one empty, ragged sizes
    segs = [(mk[a:b], ms[a:b]) if b > a else (None, None) for a, b in zip(cuts[:-1], cuts[1:])]
    sizes = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    w0, i0, s0 = ops.affinity_topk(segs, qk, qe, 30, want_sim=True)                           # un-hinted reference
    rows = [ops.affinity_rows16(k.contiguous(), s.contiguous(), torch.empty((k.shape[0], ops.ROWS16_FLOATS), device='cuda')) if k is not None else None
            for k, s in segs]
    with_rows = [(k, s, r) for (k, s), r in zip(segs, rows)]
    mixed = [with_rows[0], segs[1], segs[2]]                        # only one segment brings its rows: the others are derived in the call
    for name, sg in (('kept rows', with_rows), ('mixed', mixed), ('derived', segs)):
        for hname, h in (('perfect', (i0, sizes, 20)), ('garbage', (torch.zeros_like(i0), sizes, 0))):
            w, i, s = ops.affinity_topk(sg, qk, qe, 30, want_sim=True, hint=h)
            torch.cuda.synchronize()
            assert torch.equal(s, s0) and torch.equal(i, i0) and torch.equal(w, w0), f'{name} / {hname} hint: differs from the un-hinted call'
