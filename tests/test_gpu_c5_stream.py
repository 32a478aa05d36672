"""In-stream test of BASELINE configs[4] as stated: synthetic 1080p (pads to 1088 x 1920, HW = 8 160), 5 objects, 512 permanent memory
frames (N = 4 177 920; keys 1.07 GB, values 42.8 GB of the 288 GB) through the pipeline that serves `bench.py --workload c5`
(batched key hints, hinted fp16 filter + exact refine over the whole memory, sparse readout of 5 objects, usage counting, memory frames).
Reference: inference/memory_manager.py:61-190 (match_memory; :99-120 the SUFFIX alignment of object groups), :212-281 (add_memory).

The five objects form TWO object groups, as when objects appear later in a video: the first 256 permanent frames are annotated with
objects 1-3, the last 256 with objects 1-5, so group 1 (objects 4, 5) owns the LAST 256 frames' keys only - its affinity runs over the
suffix of the store, its readout over its own value arena (memory_manager.py:99-120).

The oracle cannot run this stream (N x HW = 136 GB per frame), so, as tests/test_gpu_c4_stream.py does:
  * every step's OWN match_memory calls (one per group: segments, queries, hint as the stream passed them) are re-derived for a random
    sample of queries with the oracle's get_similarity + top-k over ALL the group's elements: index sets equal or proven k-th ties,
    similarities, weights, and the readout row of EVERY object of the group (sum_k w_k v[idx_k], fp64 on the host);
  * the usage the step adds to the temporary store equals the first group's affinity mass on it;
  * group structure and memory sizes follow the oracle's RefCore on the same schedule at a small resolution;
  * the stream again without hints, and again with every readout enqueued a frame ahead, gives bit-identical masks and memories."""
import gc

import numpy as np
import pytest
import torch

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy
K_ALL, K_FIRST = 5, 3


def _cfg():
    from conftest import base_config
    return base_config(mem_every=2)                      # memory frames at steps 2 and 4 of 6: the temporary store grows inside the stream


def _preload(core, frames, masks, P, base, dev=True):
    """P permanent frames = `base` frames shifted by distinct offsets (as bench.py run_gpu); the first half annotated with objects
    1..3, the second half with 1..5."""
    for j in range(P):
        sh = (3 * (j // base), 5 * (j // base))
        f, m = torch.roll(frames[j % base], sh, (1, 2)), torch.roll(masks[j % base], sh, (1, 2))
        if j == 0:
            core.set_all_labels(list(range(1, K_FIRST + 1)))
        if j == P // 2:
            core.set_all_labels(list(range(1, K_ALL + 1)))
        core.put_to_permanent_memory(f, m[:K_FIRST] if j < P // 2 else m, ti=j)


def _oracle_structure(ref_net, steps):
    """Group structure + sizes (in frames) per step of the oracle on the same schedule at 128 x 176, 8 permanent frames."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, P, base = (128, 176), 8, 4
    fr, mk = T(synthetic_frames(base + steps, *hw)), T(synthetic_masks(base + steps, K_ALL, *hw))
    ref = R.RefCore(ref_net, _cfg())
    _preload(ref, fr, mk, P, base)
    n = (hw[0] // 16) * (hw[1] // 16)
    pm = ref.memory.permanent_work_mem
    groups = [list(g) for g in pm.obj_groups]
    perm_v = [pm.get_v_size(g) // n for g in range(pm.num_groups)]
    out = []
    for i in range(steps):
        ref.step(fr[base + i], None, None, end=(i == steps - 1))
        tm = ref.memory.temporary_work_mem
        out.append((tm.size // n, [tm.get_v_size(g) // n for g in range(tm.num_groups)], ref.memory.long_mem.size))
    return groups, perm_v, P, out


def _run_stream(hip_net, frames, masks, P, base, steps, hinted, check=None, early=False):
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd import ops
    core = InferenceCore(hip_net, _cfg())
    core.memory.use_affinity_hint = hinted
    core.early_readout = early
    _preload(core, frames, masks, P, base)
    dev = [frames[base + i] for i in range(steps)]
    out = []
    for i in range(steps):
        if i % 4 == 0:
            core.prefetch_keys(dev[i:i + 4], inputs_complete=True)
        p = core.step(dev[i], None, None, end=(i == steps - 1))
        assert p.shape[0] == K_ALL + 1
        out.append(ops.argmax_u8(p).cpu().numpy())
        if check is not None:
            check(i, core)
    m = core.memory
    state = dict(tmp=m.temporary_work_mem.size, lt=m.long_mem.size, use=m.temporary_work_mem.use_count.clone(),
                 tmp_v=[m.temporary_work_mem.get_v_size(g) for g in range(m.temporary_work_mem.num_groups)],
                 hidden=m.get_hidden().clone())
    del core, m
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()       # 44 GB of arenas go back before the next stream is preloaded
    return out, state


def test_c5_stream_1080p_5_objects_512_permanent_frames_two_groups(hip_net, ref_net):
    from xmem2_amd import ops
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, P, base, steps, top_k = (1080, 1920), 512, 8, 6, 30
    n_hw = (1088 // 16) * (1920 // 16)
    assert n_hw == 8160
    frames = T(synthetic_frames(base + steps, *hw)).cuda()
    masks = T(synthetic_masks(base + steps, K_ALL, *hw)).cuda()
    groups, perm_v, P_small, traj = _oracle_structure(ref_net, steps)
    assert groups == [[0, 1, 2], [3, 4]] and perm_v == [P_small, P_small // 2], (groups, perm_v)

    calls, readouts, usage_before = [], [], {}
    orig_aff, orig_ro = ops.affinity_topk, ops.readout_sparse

    def spy_aff(segs, qk, qe, top_k, want_sim=False, hint=None):
        w, idx, sim = orig_aff(segs, qk, qe, top_k, want_sim=True, hint=hint)
        calls.append(dict(segs=segs, qk=qk, qe=qe, w=w, idx=idx, sim=sim, hinted=hint is not None))
        return w, idx, (sim if want_sim else None)

    def spy_ro(vsegs, w, idx, cv, out, out_ld, obj_stride, out_off=0):
        r = orig_ro(vsegs, w, idx, cv, out, out_ld, obj_stride, out_off=out_off)
        readouts.append(dict(vsegs=vsegs, out=out, out_ld=out_ld, obj_stride=obj_stride, out_off=out_off, cv=cv))
        return r

    gen = torch.Generator().manual_seed(9)
    stats = dict(same_sets=0, sampled=0)

    def check(i, core):
        m = core.memory
        pm, tm = m.permanent_work_mem, m.temporary_work_mem
        # structure as the oracle's on the same schedule (sizes in frames; suffix groups: memory_manager.py:99-120)
        assert [list(g) for g in pm.obj_groups] == groups and [pm.get_v_size(g) for g in range(2)] == [P * n_hw, P // 2 * n_hw]
        assert (tm.size // n_hw, [tm.get_v_size(g) // n_hw for g in range(tm.num_groups)], m.long_mem.size) == traj[i], f'step {i}'
        assert len(calls) == 2 * (i + 1) and len(readouts) == 2 * (i + 1), 'one affinity call and one readout per object group'
        for gi in (0, 1):
            c, r = calls[-2 + gi], readouts[-2 + gi]
            n_obj = len(groups[gi])
            assert len(r['vsegs']) == n_obj
            n_seg = [(sg[0].shape[0] if sg[0] is not None else 0) for sg in c['segs']]
            N = sum(n_seg)
            # group 1 sees only the SUFFIX of every store: the last 256 permanent frames (+ the temporary frames, all of which hold it)
            assert n_seg[-1] == (P if gi == 0 else P // 2) * n_hw, (gi, n_seg)
            segs = [sg for sg in c['segs'] if sg[0] is not None and sg[0].shape[0] > 0]
            pick = torch.randperm(n_hw, generator=gen)[:16]
            mk = torch.cat([sg[0] for sg in segs], 0).cpu()
            ms = torch.cat([sg[1] for sg in segs], 0).cpu()
            ref = R.get_similarity(mk.t().unsqueeze(0), ms.view(1, 1, -1), c['qk'].cpu()[pick].t().unsqueeze(0),
                                   c['qe'].cpu()[pick].t().unsqueeze(0))[0]                   # [N, n_pick]
            del mk, ms
            rv, ri = torch.topk(ref, top_k, dim=0)
            gidx, gv, gw = c['idx'].cpu().long()[pick], c['sim'].cpu()[pick], c['w'].cpu()[pick]
            assert int(gidx.min()) >= 0 and int(gidx.max()) < N
            same = (torch.sort(gidx, 1)[0] == torch.sort(ri.t(), 1)[0]).all(1)
            own = torch.gather(ref.t(), 1, gidx)
            kth = rv[-1].unsqueeze(1)
            # a pick may fall short of the oracle's k-th value only by fp32 evaluation noise (two fp32 evaluations of a difference of
            # sums of ~C_k terms of magnitude |mk^2 qe| ~ 1e1: ~1e-5; at N = 4.2 M the k-th and (k+1)-th values are that close).  A
            # MISSED candidate falls short by the spacing of the ranks (1e-3 ... 1e-2).
            short = float((kth - own).clamp(min=0).max())
            stats['max_shortfall'] = max(stats.get('max_shortfall', 0.0), short)
            if short >= 1e-4:
                w2, i2, s2 = orig_aff(c['segs'], c['qk'], c['qe'], top_k, want_sim=True, hint=None)
                own2 = torch.gather(ref.t(), 1, i2.cpu().long()[pick])
                raise AssertionError(f'step {i} group {gi}: a pick falls {short:.3e} short of the oracle\'s k-th similarity (hinted={c["hinted"]}); the '
                                     f'un-hinted select on the same operands: {float((kth - own2).clamp(min=0).max()):.3e}; per query '
                                     f'{[round(float(v), 6) for v in (kth - own).clamp(min=0).max(1).values]}')
            assert float((torch.sort(gv, 1, descending=True)[0] - rv.t()).abs().max()) < 2e-4, f'step {i} group {gi}: top-k similarities differ'
            rw = torch.softmax(rv.t().double(), 1)
            srt = torch.sort(gv, 1, descending=True)
            assert float((torch.gather(gw, 1, srt[1]).double() - rw).abs().max()) < 2e-5, f'step {i} group {gi}: affinity weights differ'
            stats['same_sets'] += int(same.sum()); stats['sampled'] += len(pick)
            # the readout rows of EVERY object of the group for the sampled queries, fp64 on the host
            bounds = np.cumsum([0] + n_seg)
            for o in range(n_obj):
                acc = torch.zeros((len(pick), r['cv']), dtype=torch.float64)
                for s, v in enumerate(r['vsegs'][o]):
                    if v is None or n_seg[s] == 0:
                        continue
                    assert v.shape[0] == n_seg[s], 'value rows of a group are aligned with the key suffix it sees'
                    sel = (gidx >= bounds[s]) & (gidx < bounds[s + 1])
                    if not bool(sel.any()):
                        continue
                    rows = v[(gidx[sel] - bounds[s]).to(v.device)].cpu().double()
                    acc.index_add_(0, torch.nonzero(sel)[:, 0], rows * gw[sel].double().unsqueeze(1))
                flat = r['out'].reshape(-1)
                base_off = r['out_off'] + o * r['obj_stride']
                rows_idx = (pick.to(flat.device) * r['out_ld']).unsqueeze(1) + base_off + torch.arange(r['cv'], device=flat.device).unsqueeze(0)
                got = flat[rows_idx].cpu().double()
                assert float((got - acc).abs().max()) < 1e-4 * max(1.0, float(acc.abs().max())), f'step {i} group {gi} object {o}: readout rows differ'
        # usage (memory_manager.py:93-97,133-141): from the FIRST group only, onto the temporary store only
        c0 = calls[-2]
        if i in usage_before and usage_before[i][0] > 0:
            n_tmp_b, use_b = usage_before[i]
            idx, w = c0['idx'].long(), c0['w']
            on_tmp = idx < n_tmp_b                                   # segments are [long (empty) | temporary | permanent]
            mass = float(w[on_tmp].double().sum())
            if tm.size >= n_tmp_b:
                gained = float(tm.use_count.double().sum()) - use_b
                assert abs(gained - mass) <= 1e-3 * max(1.0, mass), f'step {i}: usage gained {gained} vs affinity mass on the temporary store {mass}'
        usage_before[i + 1] = (tm.size, float(tm.use_count.double().sum()) if tm.size > 0 else 0.0)

    ops.affinity_topk, ops.readout_sparse = spy_aff, spy_ro
    try:
        hinted_masks, st1 = _run_stream(hip_net, frames, masks, P, base, steps, True, check)
    finally:
        ops.affinity_topk, ops.readout_sparse = orig_aff, orig_ro
    assert all(c['hinted'] for c in calls[2:]), 'every call after the first frame must carry a hint'
    assert traj[-1][0] == 2, 'the schedule must write two memory frames inside the stream'
    assert st1['tmp'] == traj[-1][0] * n_hw and st1['tmp_v'] == [v * n_hw for v in traj[-1][1]]
    present = [float(np.mean([(m == c).mean() for m in hinted_masks])) for c in range(1, K_ALL + 1)]
    assert sum(p > 1e-4 for p in present) >= 3, f'degenerate masks: object fractions {present}'
    del calls[:], readouts[:]
    gc.collect(); torch.cuda.empty_cache()
    plain_masks, st2 = _run_stream(hip_net, frames, masks, P, base, steps, False)
    diff = sum(int((a != b).sum()) for a, b in zip(hinted_masks, plain_masks))
    assert diff == 0, f'the hinted stream differs from the un-hinted stream on {diff} pixels'
    early_masks, st3 = _run_stream(hip_net, frames, masks, P, base, steps, True, early=True)
    diff = sum(int((a != b).sum()) for a, b in zip(hinted_masks, early_masks))
    assert diff == 0, f'the early-readout stream differs from the in-step stream on {diff} pixels'
    for st in (st2, st3):
        assert (st['tmp'], st['lt'], st['tmp_v']) == (st1['tmp'], st1['lt'], st1['tmp_v'])
        assert torch.equal(st['use'], st1['use']) and torch.equal(st['hidden'], st1['hidden'])
    print(f'C5 stream: {steps} steps, N = {P * n_hw} (+{st1["tmp"]}), groups {groups}; sampled queries with the oracle\'s exact index set '
          f'{stats["same_sets"]}/{stats["sampled"]}; largest shortfall of a pick below the oracle\'s k-th similarity {stats["max_shortfall"]:.2e}; '
          f'object fractions {[round(p, 4) for p in present]}')
