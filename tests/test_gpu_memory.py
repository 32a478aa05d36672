"""GPU parity (memory level): the scripted MemoryManager sequences recorded from the imported reference
(tests/golden/mem_*.npz) replayed through the arena stores + fused kernels."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden
from test_oracle_goldens import _feed, _query
from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rows(t):            # [1,C,h,w] -> [hw, C] cuda
    return t[0].flatten(1).t().contiguous().cuda()


class _SimTap:
    """Records the similarity matrices the oracle hands to its top-k softmax (one per object group and match call)."""

    def __enter__(self):
        self.sims, self._orig = [], R.do_softmax

        def tap(similarity, *a, **kw):
            self.sims.append(similarity.clone())
            return self._orig(similarity, *a, **kw)
        R.do_softmax = tap
        return self

    def __exit__(self, *exc):
        R.do_softmax = self._orig


@pytest.mark.parametrize('tag', ['single_group', 'two_groups', 'lt_eviction', 'perm_edit'])
def test_memory_scripts_hip(tag):
    """The HIP memory against the reference-recorded readouts.  A query may deviate from the recording only through a
    top-k fork: its k-th and (k+1)-th similarities (taken from the oracle run next to it) are a near-tie, so the other
    element may legitimately be picked.  Forks are budgeted at 0.5 % of the queries of a match (at least one) and every one of them is
    checked to BE such a near-tie; everything else must agree to fp32 round-off."""
    from xmem2_amd.memory_manager import MemoryManager
    g = load_golden('mem_' + tag)
    script = ast.literal_eval(str(g['script']))
    cfg = ast.literal_eval(str(g['config']))
    h, w = (int(x) for x in g['hw'])
    mm, orc = MemoryManager(cfg), R.RefMemory(cfg)
    k = cfg['top_k']
    n_fork = n_q = 0
    for step, op in enumerate(script):
        if op[0] in ('perm', 'temp'):
            objects, ti = op[1], (op[2] if len(op) > 2 else None)
            key, shr, val, sel = _feed(step, len(objects), (h, w))
            value = val[0].flatten(2).transpose(1, 2).contiguous().cuda()          # [K, HW, Cv]
            mm.add_memory(rows(key), shr.view(-1).cuda(), value, list(objects), selection=rows(sel),
                          permanent=(op[0] == 'perm'), ti=ti, hw_shape=(h, w))
            orc.add_memory(key, shr, val, list(objects), selection=sel, permanent=(op[0] == 'perm'), ti=ti)
        elif op[0] == 'replace':
            key, shr, val, sel = _feed(step, op[2], (h, w))
            value = val[0].flatten(2).transpose(1, 2).contiguous().cuda()
            mm.update_permanent_memory(op[1], rows(key), shr.view(-1).cuda(), value, selection=rows(sel))
            orc.update_permanent_memory(op[1], key, shr, val, selection=sel)
        elif op[0] == 'remove':
            mm.remove_from_permanent_memory(op[1])
            orc.remove_from_permanent_memory(op[1])
            assert sorted(mm.frame_id_to_permanent_mem_idx.items()) == [tuple(r) for r in g[f'perm_index_{step}'].tolist()]
        else:
            qk, qe = _query(step, (h, w))
            out = mm.match_memory(qk.cuda(), qe.cuda())
            torch.cuda.synchronize()
            with _SimTap() as tap:
                orc.match_memory(qk, qe)
            ref = T(g[f'readout_{step}'])
            err = (out.cpu() - ref).abs()
            scale = float(ref.abs().max())
            per_q = err.amax(dim=(0, 1)).flatten()
            forked = per_q > 5e-5 * scale
            n_fork += int(forked.sum()); n_q += forked.numel()
            assert int(forked.sum()) <= max(1, round(0.005 * forked.numel())) and float(per_q.max()) < 0.1 * scale, \
                f'{tag} step {step}: {int(forked.sum())}/{forked.numel()} queries deviate, max err {float(per_q.max()):.3e} (scale {scale:.3e})'
            for q in torch.nonzero(forked).flatten().tolist():       # every fork must be a k-th / (k+1)-th near-tie in some group
                gaps = []
                for sim in tap.sims:
                    if sim.shape[1] > k:
                        v = torch.topk(sim[0, :, q], k + 1).values
                        gaps.append(float(v[k - 1] - v[k]) / max(1.0, abs(float(v[k - 1]))))
                assert gaps and min(gaps) <= 2e-5, f'{tag} step {step} query {q}: deviates without a top-k near-tie (gaps {gaps})'
            if f'tmp_use_{step}' in g.files and mm.temporary_work_mem.size > 0:
                u = mm.temporary_work_mem.use_count.cpu().numpy()
                if u.shape == g[f'tmp_use_{step}'].shape:
                    du = np.abs(u - g[f'tmp_use_{step}'])
                    assert (du > 1e-4).mean() <= 0.01 and du.max() < 0.1, f'{tag} step {step}: usage deviates {du.max():.3e}'
                    np.testing.assert_allclose(mm.temporary_work_mem.life_count.cpu().numpy(), g[f'tmp_life_{step}'], rtol=1e-6)
        sizes = (mm.temporary_work_mem.size, mm.permanent_work_mem.size, mm.long_mem.size)
        assert sizes == tuple(g[f'sizes_{step}']), f'{tag} step {step}: sizes {sizes} vs {tuple(g[f"sizes_{step}"])}'
        vs = g[f'vsizes_{step}']
        for si, st in enumerate((mm.temporary_work_mem, mm.permanent_work_mem, mm.long_mem)):
            for gi in range(2):
                assert (st.get_v_size(gi) if gi < st.num_groups else -1) == int(vs[si, gi])
    print(f'{tag}: {n_fork} forked queries of {n_q}')
    if 'lt_key' in g.files and tag != 'lt_eviction':
        # prototypes themselves: same keys (exact gather) when the usage ranking did not fork
        lk = mm.long_mem.key.cpu()
        same = float((lk == T(g['lt_key'])).all(1).float().mean())
        assert same > 0.9, f'only {same:.2f} of prototype keys identical to the reference'
        ref_v = T(g['lt_value_0'])
        err = (mm.long_mem.value[0].cpu() - ref_v).abs()
        assert float(err.mean()) < 5e-3 * float(ref_v.abs().mean() + 1e-9)


def test_clear_memory_keep_permanent(hip_net):
    from conftest import base_config
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    core = InferenceCore(hip_net, base_config(mem_every=2))
    fr = T(synthetic_frames(4, 96, 128)).cuda(); mk = T(synthetic_masks(4, 1, 96, 128)).cuda()
    core.set_all_labels([1])
    assert core.put_to_permanent_memory(fr[0], mk[0], ti=0) is False
    assert core.put_to_permanent_memory(fr[1], mk[1], ti=0) is True           # same ti -> replace
    for t in range(4):
        core.step(fr[t], None, None)
    assert core.memory.temporary_work_mem.size > 0 and core.permanent_memory_frames == [0]
    core.clear_memory(keep_permanent=True)
    assert core.memory.temporary_work_mem.size == 0 and core.memory.permanent_work_mem.size == 6 * 8
    p = core.step(fr[2], None, None)
    assert p.shape == (2, 96, 128) and bool(torch.isfinite(p).all())


def test_read_memory_and_memory_util_dense_forms(hip_net):
    """the training-time read (`XMem.read_memory`, full softmax + dense readout) and the
    reference-shaped memory_util wrappers against the oracle."""
    from xmem2_amd import memory_util as MU
    g = torch.Generator().manual_seed(5)
    B, K, CK, CV, T, h, w = 2, 2, 64, 512, 3, 6, 8
    qk = torch.randn(B, CK, h, w, generator=g) * 0.8
    qe = torch.rand(B, CK, h, w, generator=g)
    mk = torch.randn(B, CK, T, h, w, generator=g) * 0.8
    ms = torch.rand(B, 1, T, h, w, generator=g) * 2 + 1
    mv = torch.randn(B, K, CV, T, h, w, generator=g)
    sim_ref = R.get_similarity(mk, ms, qk, qe)
    aff_ref = R.do_softmax(sim_ref)
    want = torch.bmm(mv.flatten(1, 2).flatten(start_dim=2), aff_ref).view(B, K, CV, h, w)
    got = hip_net.read_memory(qk.cuda(), qe.cuda(), mk.cuda(), ms.cuda(), mv.cuda())
    assert got.shape == want.shape
    torch.testing.assert_close(got.cpu(), want, rtol=2e-3, atol=2e-4)
    sim = MU.get_similarity(mk.cuda(), ms.cuda(), qk.cuda(), qe.cuda())
    torch.testing.assert_close(sim.cpu(), sim_ref, rtol=1e-4, atol=1e-4)
    sim2 = MU.get_similarity(mk.cuda(), None, qk.cuda(), None)
    torch.testing.assert_close(sim2.cpu(), R.get_similarity(mk, None, qk, None), rtol=1e-4, atol=1e-4)
    aff, usage = MU.do_softmax(sim, top_k=10, return_usage=True)
    aff_r, usage_r = R.do_softmax(sim_ref.clone(), top_k=10, return_usage=True)
    torch.testing.assert_close(aff.cpu(), aff_r, rtol=2e-3, atol=1e-6)
    torch.testing.assert_close(usage.cpu(), usage_r, rtol=2e-3, atol=1e-6)


def test_do_softmax_topk_rows_kernel_vs_oracle():
    """do_softmax(top_k) on a materialised similarity (memory_util.py:41-54) through xmem_softmax_rows_topk: random rows against
    the oracle, k = 1, k = n, and exact ties at the k-th value (a stable sort's choice: the lowest indices)."""
    from xmem2_amd import memory_util as MU, ops
    g = torch.Generator().manual_seed(11)
    sim = torch.randn(2, 5000, 37, generator=g) * 3                      # B x N x HW
    for k in (1, 30, 64):
        aff, usage = MU.do_softmax(sim.cuda(), top_k=k, return_usage=True)
        aff_r, usage_r = R.do_softmax(sim.clone(), top_k=k, return_usage=True)
        assert torch.equal(aff.cpu() != 0, aff_r != 0), f'k={k}: different elements kept'
        torch.testing.assert_close(aff.cpu(), aff_r, rtol=1e-5, atol=1e-8)
        torch.testing.assert_close(usage.cpu(), usage_r, rtol=1e-5, atol=1e-8)
        assert int((aff != 0).sum(1).min()) == int((aff != 0).sum(1).max()) == k
    small = torch.randn(1, 7, 5, generator=g)
    full = MU.do_softmax(small.cuda(), top_k=7)                          # k = n: the un-shifted softmax over everything
    torch.testing.assert_close(full.cpu(), small.exp() / small.exp().sum(1, keepdim=True), rtol=1e-5, atol=1e-8)
    rows = torch.full((3, 300), -2.0)
    rows[:, 17] = 4.0; rows[:, 250] = 3.0                                # two clear winners, 298 elements tied for the rest
    rows[1, 100:140] = 0.5                                               # row 1: 40 ties at the k-th value, 8 of them kept
    out = ops.softmax_rows_topk(rows.cuda().clone(), 10).cpu()
    for r in range(3):
        kept = torch.nonzero(out[r]).flatten().tolist()
        want = [17, 250] + (list(range(100, 108)) if r == 1 else [i for i in range(300) if i not in (17, 250)][:8])
        assert sorted(kept) == sorted(want), f'row {r}: kept {kept}'
        e = rows[r, kept].exp()
        torch.testing.assert_close(out[r, kept], e / e.sum(), rtol=1e-5, atol=1e-8)
    with pytest.raises(RuntimeError):
        ops.softmax_rows_topk(rows.cuda(), 301)
