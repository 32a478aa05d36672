"""In-stream test of BASELINE configs[3] as stated: 720p, 1 object, 256 permanent memory frames (N = 921 600) + long-term
consolidation, through the pipeline that serves `bench.py --workload c4` (batched key hints, hinted fp16 filter + exact refine,
sparse readout, usage counting, compress_features).  Reference: inference/memory_manager.py:61-190 (match_memory),
:272-281 (the trigger), :316-390 (compress_features + consolidation).

The oracle cannot run this stream (it materialises N x HW = 13.3 GB per frame and would preload 256 frames at 720p), so every
step is pinned three ways instead:
  * the stream's OWN match_memory call (segments, queries, hint as the stream passed them) is re-derived for a random sample of
    queries with the oracle's get_similarity + top-k over ALL N elements: index sets equal (or proven k-th/(k+1)-th ties),
    weights and the readout row (sum_k w_k v[idx_k], fp64 on the host) equal;
  * the usage the step adds to the temporary store equals the affinity mass that landed on it (never on the permanent store);
  * the memory sizes follow, frame by frame, the trajectory of the oracle's RefCore run on the SAME schedule at a small
    resolution (sizes are multiples of HW; the prototype count is absolute).
Size-independent properties on top: the whole stream again WITHOUT hints (the un-hinted fp32 select on every frame), and again with
every readout enqueued a frame ahead on the readout stream (early readout), gives bit-identical masks and memories."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _cfg():
    from conftest import base_config
    # memory frames at steps 2, 4, 6, 8; the third fills the temporary store (T_max = 3) -> compress_features at step 6:
    # 2 frames of candidates -> 128 prototypes, 1 frame stays; three more hinted frames follow
    return base_config(mem_every=2, max_mid_term_frames=3, min_mid_term_frames=1, num_prototypes=128)


def _oracle_size_trajectory(ref_net, steps):
    """(temporary / HW, permanent / HW, long) per step of the oracle on the same schedule at 128 x 176 (HW = 88)."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw = (128, 176)
    fr, mk = T(synthetic_frames(2 + steps, *hw)), T(synthetic_masks(2 + steps, 1, *hw))
    ref = R.RefCore(ref_net, _cfg())
    ref.set_all_labels([1])
    ref.put_to_permanent_memory(fr[0], mk[0])
    n = (hw[0] // 16) * (hw[1] // 16)
    out = []
    for i in range(steps):
        ref.step(fr[2 + i], None, None, end=(i == steps - 1))
        m = ref.memory
        out.append((m.temporary_work_mem.size // n, m.long_mem.size))
    return out


def _run_stream(hip_net, frames, masks, P, base, steps, hinted, check=None, early=False):
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd import ops
    core = InferenceCore(hip_net, _cfg())
    core.set_all_labels([1])
    core.memory.use_affinity_hint = hinted
    core.early_readout = early        # (the per-step check reads "the last affinity call": it needs every readout inside its own step)
    for j in range(P):                                     # 8 base frames shifted by distinct offsets (as bench.py make_clip / run_gpu)
        sh = (3 * (j // base), 5 * (j // base))
        core.put_to_permanent_memory(torch.roll(frames[j % base], sh, (1, 2)), torch.roll(masks[j % base], sh, (1, 2)), ti=j)
    dev = [frames[base + i] for i in range(steps)]
    out = []
    for i in range(steps):
        if i % 4 == 0:                                     # the batched key pass in BOTH runs: same convolution plans, same features
            core.prefetch_keys(dev[i:i + 4])
        p = core.step(dev[i], None, None, end=(i == steps - 1))
        out.append(ops.argmax_u8(p).cpu().numpy())
        if check is not None:
            check(i, core)
    return out, core


def test_c4_stream_720p_256_permanent_frames_with_consolidation(hip_net, ref_net):
    from xmem2_amd import ops
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, P, base, steps, top_k = (720, 1280), 256, 8, 10, 30
    n_hw = (720 // 16) * (1280 // 16)
    frames = T(synthetic_frames(base + steps, *hw)).cuda()
    masks = T(synthetic_masks(base + steps, 1, *hw)).cuda()
    traj = _oracle_size_trajectory(ref_net, steps)
    assert any(lt > 0 for _, lt in traj[:-2]), 'the schedule must contain a consolidation with frames after it'

    calls, readouts, usage_before = [], [], {}
    orig_aff, orig_ro = ops.affinity_topk, ops.readout_sparse

    def spy_aff(segs, qk, qe, top_k, want_sim=False, hint=None):
        w, idx, sim = orig_aff(segs, qk, qe, top_k, want_sim=True, hint=hint)
        calls.append(dict(segs=segs, qk=qk, qe=qe, w=w, idx=idx, sim=sim, hinted=hint is not None))
        return w, idx, (sim if want_sim else None)

    def spy_ro(vsegs, w, idx, cv, out, out_ld, obj_stride, out_off=0):
        r = orig_ro(vsegs, w, idx, cv, out, out_ld, obj_stride, out_off=out_off)
        readouts.append(dict(vsegs=vsegs, out=out, out_ld=out_ld, out_off=out_off, cv=cv))
        return r

    gen = torch.Generator().manual_seed(5)
    stats = dict(same_sets=0, sampled=0, flagged_hinted=0)

    def check(i, core):
        """the stream's own readout of step i against the oracle on a random sample of its queries"""
        c, r = calls[-1], readouts[-1]
        m = core.memory
        sizes = (m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size)
        assert sizes == (traj[i][0] * n_hw, P * n_hw, traj[i][1]), f'step {i}: memory sizes {sizes} vs the oracle trajectory {traj[i]}'
        segs = [sg for sg in c['segs'] if sg[0] is not None and sg[0].shape[0] > 0]
        n_seg = [(sg[0].shape[0] if sg[0] is not None else 0) for sg in c['segs']]
        N = sum(n_seg)
        assert N >= P * n_hw
        pick = torch.randperm(n_hw, generator=gen)[:24]
        mk = torch.cat([sg[0] for sg in segs], 0).cpu()
        ms = torch.cat([sg[1] for sg in segs], 0).cpu()
        ref = R.get_similarity(mk.t().unsqueeze(0), ms.view(1, 1, -1), c['qk'].cpu()[pick].t().unsqueeze(0),
                               c['qe'].cpu()[pick].t().unsqueeze(0))[0]                      # [N, n_pick]
        rv, ri = torch.topk(ref, top_k, dim=0)
        gi, gv, gw = c['idx'].cpu().long()[pick], c['sim'].cpu()[pick], c['w'].cpu()[pick]
        same = (torch.sort(gi, 1)[0] == torch.sort(ri.t(), 1)[0]).all(1)
        own = torch.gather(ref.t(), 1, gi)                                                   # the oracle's similarity at OUR indices
        kth = rv[-1].unsqueeze(1)
        # where the sets differ the picks must be (near-)ties of the k-th similarity (shifted copies hold exact duplicates)
        assert bool((own >= kth - 2e-5 * kth.abs().clamp(min=1.0)).all()), f'step {i}: a pick does not reach the oracle\'s k-th similarity'
        assert float((torch.sort(gv, 1, descending=True)[0] - rv.t()).abs().max()) < 2e-4, f'step {i}: top-k similarities differ'
        rw = torch.softmax(rv.t().double(), 1)                  # exp(v) / sum exp(v) (memory_util.py:47-49) - the shift cancels
        srt = torch.sort(gv, 1, descending=True)
        assert float((torch.gather(gw, 1, srt[1]).double() - rw).abs().max()) < 2e-5, f'step {i}: affinity weights differ'
        assert float((gw.sum(1) - 1).abs().max()) < 1e-5
        stats['same_sets'] += int(same.sum()); stats['sampled'] += len(pick)
        # the readout rows of the sampled queries: sum_k w_k * value[idx_k] in fp64 on the host
        vs = r['vsegs'][0]
        bounds = np.cumsum([0] + n_seg)
        acc = torch.zeros((len(pick), r['cv']), dtype=torch.float64)
        for s, v in enumerate(vs):
            if v is None or n_seg[s] == 0:
                continue
            sel = (gi >= bounds[s]) & (gi < bounds[s + 1])
            if not bool(sel.any()):
                continue
            rows = v[(gi[sel] - bounds[s]).to(v.device)].cpu().double()
            qn = torch.nonzero(sel)[:, 0]
            acc.index_add_(0, qn, rows * gw[sel].double().unsqueeze(1))
        got = r['out'].view(-1, r['out_ld'])[pick.to(r['out'].device), r['out_off']:r['out_off'] + r['cv']].cpu().double()
        assert float((got - acc).abs().max()) < 1e-4 * max(1.0, float(acc.abs().max())), f'step {i}: readout rows differ'
        # usage (memory_manager.py:133-141): the temporary store gained exactly the affinity mass that landed on it
        tmp = m.temporary_work_mem
        if i in usage_before and usage_before[i][0] > 0:
            n_long_b, n_tmp_b, use_b = usage_before[i][1], usage_before[i][0], usage_before[i][2]
            idx, w = c['idx'].long(), c['w']
            on_tmp = (idx >= n_long_b) & (idx < n_long_b + n_tmp_b)
            mass = float(w[on_tmp].double().sum())
            # (the consolidation of THIS step sieves the store after the readout: compare only when it did not)
            if tmp.size >= n_tmp_b and tmp.size > 0:             # (rows a memory frame appended start at usage 0; a sieve removes rows)
                gained = float(tmp.use_count.double().sum()) - use_b
                assert abs(gained - mass) <= 1e-3 * max(1.0, mass), f'step {i}: usage gained {gained} vs affinity mass on the temporary store {mass}'
        usage_before[i + 1] = (tmp.size, m.long_mem.size, float(tmp.use_count.double().sum()) if tmp.size > 0 else 0.0)

    ops.affinity_topk, ops.readout_sparse = spy_aff, spy_ro
    try:
        hinted_masks, core = _run_stream(hip_net, frames, masks, P, base, steps, True, check)
    finally:
        ops.affinity_topk, ops.readout_sparse = orig_aff, orig_ro
    assert len(calls) == steps and all(c['hinted'] for c in calls[1:]), 'every call after the first must carry a hint'
    assert max(sum((sg[0].shape[0] if sg[0] is not None else 0) for sg in c['segs']) for c in calls) > P * n_hw + n_hw
    assert stats['same_sets'] >= 0.5 * stats['sampled'], stats          # duplicates aside, most index sets are the oracle's
    obj = np.mean([(m == 1).mean() for m in hinted_masks])
    assert 0.01 < obj < 0.6, f'degenerate masks (object fraction {obj:.3f})'
    del calls[:], readouts[:]
    # size-independent property: no hints (the un-hinted fp32 select on every frame) -> the same masks bit for bit
    plain_masks, core2 = _run_stream(hip_net, frames, masks, P, base, steps, False)
    diff = sum(int((a != b).sum()) for a, b in zip(hinted_masks, plain_masks))
    assert diff == 0, f'the hinted stream differs from the un-hinted stream on {diff} pixels'
    m1, m2 = core.memory, core2.memory
    assert (m1.temporary_work_mem.size, m1.long_mem.size) == (m2.temporary_work_mem.size, m2.long_mem.size)
    # ... and with every readout enqueued a frame ahead, under the previous frame's decoder (InferenceCore early readout): same
    # kernels on the same operands, usage applied when the frame is stepped -> the same masks, memory and usage bit for bit
    early_masks, core3 = _run_stream(hip_net, frames, masks, P, base, steps, True, early=True)
    diff = sum(int((a != b).sum()) for a, b in zip(hinted_masks, early_masks))
    assert diff == 0, f'the early-readout stream differs from the in-step stream on {diff} pixels'
    m3 = core3.memory
    assert (m1.temporary_work_mem.size, m1.long_mem.size) == (m3.temporary_work_mem.size, m3.long_mem.size)
    assert torch.equal(m1.temporary_work_mem.use_count, m3.temporary_work_mem.use_count)
    assert torch.equal(m1.long_mem.key_rows(), m3.long_mem.key_rows())
    print(f'C4 stream: {steps} steps, N up to {P * n_hw + 3 * n_hw}; sampled queries with the oracle\'s exact index set '
          f'{stats["same_sets"]}/{stats["sampled"]}; sizes {[t for t in traj]}; object fraction {obj:.3f}')
