"""CPU suite: the oracle (oracle/cpu_ref.py) against the golden fixtures generated from the imported reference
(tests/golden/make_goldens.py).  Generation asserted bit-equality in this container; here a tight tolerance is used
so the suite also passes on a host whose MKL/oneDNN dispatch differs (the GPU box)."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cpu_ref as R

T = torch.from_numpy
TOL = dict(rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('tag', ['small', 'mid'])
def test_similarity_softmax_readout(tag):
    g = load_golden('op_' + tag)
    mk, ms, qk, qe, mv = (T(g[k]) for k in ('mk', 'ms', 'qk', 'qe', 'mv'))
    sim = R.get_similarity(mk, ms, qk, qe)
    ref = T(g['sim_se'])
    np.testing.assert_allclose((sim if tag == 'small' else sim[:, ::7, ::5]).numpy(), ref.numpy(), **TOL)
    np.testing.assert_allclose(R.get_similarity(mk, ms, qk, None)[:, ::7, ::5].numpy(), g['sim_s_sample'], **TOL)
    np.testing.assert_allclose(R.get_similarity(mk, None, qk, qe)[:, ::7, ::5].numpy(), g['sim_e_sample'], **TOL)
    np.testing.assert_allclose(R.get_similarity(mk, None, qk, None)[:, ::7, ::5].numpy(), g['sim_none_sample'], **TOL)
    aff, usage = R.do_softmax(sim.clone(), top_k=30, inplace=True, return_usage=True)
    np.testing.assert_allclose(usage.numpy(), g['usage'], **TOL)
    np.testing.assert_allclose((mv @ aff).numpy(), g['readout'], rtol=1e-4, atol=1e-5)
    w, idx = R.topk_softmax_sparse(sim, 30)
    np.testing.assert_allclose(np.sort(w.numpy(), 1), np.sort(g['topk_w'], 1), **TOL)
    full = R.do_softmax(sim.clone())
    np.testing.assert_allclose(full.sum(1).numpy(), g['full_softmax_colsum'], **TOL)
    np.testing.assert_allclose(full[:, ::7, ::5].numpy(), g['full_softmax_sample'], **TOL)


def test_topk_needs_enough_elements():
    sim = torch.randn(1, 10, 5)
    with pytest.raises(RuntimeError):
        R.do_softmax(sim, top_k=30)


def test_misc_pad_mapper_iou():
    g = load_golden('misc')
    p, pad = R.pad_divide_by(T(g['pad_in']), 16)
    assert tuple(pad) == tuple(g['pad'])
    np.testing.assert_array_equal(p.numpy(), g['pad_out'])
    np.testing.assert_array_equal(R.unpad(p, pad).numpy(), g['pad_in'])
    m = R.RefMaskMapper()
    a1, l1 = m.convert_mask(g['mask_in'], exhaustive=True)
    np.testing.assert_array_equal(a1.numpy(), g['onehot1']); assert list(l1) == list(g['labels1'])
    a2, l2 = m.convert_mask(g['mask_in2'], exhaustive=True)
    np.testing.assert_array_equal(a2.numpy(), g['onehot2']); assert list(l2) == list(g['labels2'])
    assert list(m.remappings.keys()) == list(g['remap_keys']) and list(m.remappings.values()) == list(g['remap_vals'])
    np.testing.assert_array_equal(m.remap_index_mask(g['remap_in']), g['remap_out'])
    assert abs(R.compute_array_iou(g['iou_seg'], g['iou_gt']) - float(g['iou'])) < 1e-7


def _feed(step, n_obj, hw):
    from xmem2_amd.synth import hash_normal, hash_uniform
    h, w = hw
    rnd = lambda shape, s, sc=1.0: T(hash_normal(int(np.prod(shape)), s).reshape(shape) * np.float32(sc))
    uni = lambda shape, s, lo, hi: T(hash_uniform(int(np.prod(shape)), s, lo, hi).reshape(shape))
    return (rnd((1, 64, h, w), 5000 + step * 10 + 1, 0.9), uni((1, 1, h, w), 5000 + step * 10 + 2, 1.0, 4.0),
            rnd((1, n_obj, 128, h, w), 5000 + step * 10 + 4), uni((1, 64, h, w), 5000 + step * 10 + 3, 0.05, 0.95))


def _query(step, hw):
    from xmem2_amd.synth import hash_normal, hash_uniform
    h, w = hw
    qk = T(hash_normal(64 * h * w, 9000 + step * 10 + 1).reshape(1, 64, h, w) * np.float32(0.9))
    qe = T(hash_uniform(64 * h * w, 9000 + step * 10 + 2, 0.05, 0.95).reshape(1, 64, h, w))
    return qk, qe


@pytest.mark.parametrize('tag', ['single_group', 'two_groups', 'lt_eviction', 'perm_edit'])
def test_memory_scripts(tag):
    g = load_golden('mem_' + tag)
    script = ast.literal_eval(str(g['script']))
    cfg = ast.literal_eval(str(g['config']))
    hw = tuple(int(x) for x in g['hw'])
    mm = R.RefMemory(cfg)
    for step, op in enumerate(script):
        if op[0] in ('perm', 'temp'):
            objects, ti = op[1], (op[2] if len(op) > 2 else None)
            key, shr, val, sel = _feed(step, len(objects), hw)
            mm.add_memory(key, shr, val, list(objects), selection=sel, permanent=(op[0] == 'perm'), ti=ti)
        elif op[0] == 'replace':
            key, shr, val, sel = _feed(step, op[2], hw)
            mm.update_permanent_memory(op[1], key, shr, val, selection=sel)
        elif op[0] == 'remove':
            mm.remove_from_permanent_memory(op[1])
            assert sorted(mm.frame_id_to_permanent_mem_idx.items()) == [tuple(r) for r in g[f'perm_index_{step}'].tolist()]
        else:
            qk, qe = _query(step, hw)
            r = mm.match_memory(qk, qe)
            np.testing.assert_allclose(r.numpy(), g[f'readout_{step}'], rtol=2e-4, atol=2e-5)
        sizes = (mm.temporary_work_mem.size, mm.permanent_work_mem.size, mm.long_mem.size)
        assert sizes == tuple(g[f'sizes_{step}'])


def test_network_level(ref_net):
    g = load_golden('net_96x128')
    frame, masks, hidden0, readout = (T(g[k]) for k in ('frame', 'masks', 'hidden0', 'readout'))
    key, shr, sel, f16, f8, f4 = ref_net.encode_key(frame)
    for a, n in zip((key, shr, sel, f16, f8, f4), ('key', 'shrinkage', 'selection', 'f16', 'f8', 'f4')):
        np.testing.assert_allclose(a.numpy(), g[n], rtol=2e-4, atol=2e-5, err_msg=n)
    prob = R.aggregate(masks[0], dim=0)
    val, hid = ref_net.encode_value(frame, f16, hidden0, prob[1:].unsqueeze(0), is_deep_update=True)
    np.testing.assert_allclose(val.numpy(), g['value'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(hid.numpy(), g['hidden_value'], rtol=2e-4, atol=2e-5)
    hs, logits, pr = ref_net.segment((f16, f8, f4), readout, hidden0, h_out=True, strip_bg=False)
    np.testing.assert_allclose(hs.numpy(), g['hidden_seg'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(pr.numpy(), g['prob'], rtol=2e-4, atol=2e-5)


def test_e2e_480p_short(ref_net):
    """5 frames at the benchmark geometry through RefCore vs the reference's recorded argmax / prob sums."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    g = load_golden('e2e_480p_1obj')
    cfg = ast.literal_eval(str(g['config']))
    t = int(g['shape'][0])
    frames, masks = synthetic_frames(t, 480, 854), synthetic_masks(t, 1, 480, 854)
    core = R.RefCore(ref_net, cfg)
    core.set_all_labels([1])
    core.put_to_permanent_memory(T(frames[0]), T(masks[0]))
    for ti in range(t):
        mk = T(masks[ti]) if ti == 0 else None
        p = core.step(T(frames[ti]), mk, [1] if mk is not None else None, end=(ti == t - 1),
                      do_not_add_mask_to_memory=(mk is not None))
        am = torch.argmax(p, 0).numpy().astype(np.uint8)
        assert (am != g['argmax'][ti]).mean() < 2e-4
        np.testing.assert_allclose(p.double().sum((1, 2)).numpy(), g['prob_sum'][ti], rtol=1e-4)
        # the reference's own thread-count noise reaches ~1e-2 at isolated pixels (goldens: 1 thread)
        d = np.abs(p[:, 4::8, 4::8].numpy() - g['prob_ds8'][ti])
        assert d.mean() < 2e-4 and d.max() < 5e-2, (d.mean(), d.max())
    m = core.memory
    assert [m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size] == list(g['sizes'][-1])


def _selector_cases():
    import ast as _ast
    g = load_golden('selector')
    for name in [str(n) for n in g['names']]:
        masks = [T(m) for m in g[f'{name}/masks']]
        yield name, T(g[f'{name}/keys']), T(g[f'{name}/shr']), T(g[f'{name}/sel']), masks, _ast.literal_eval(str(g[f'{name}/kwargs'])), \
            [int(v) for v in g[f'{name}/chosen']], g[f'{name}/oracle_scores']


def test_selector_oracle_reproduces_the_reference_recorded_choices():
    """tests/golden/selector.npz: choices recorded from the IMPORTED reference function (frame_selection.py:99-244, behind
    arithmetic-free import placeholders, masks at key resolution - tests/golden/make_selector_goldens.py).  The oracle's restatement
    must make the same choices, with the recorded score trace."""
    n = 0
    for name, keys, shr, sel, masks, kw, chosen, scores in _selector_cases():
        got = R.select_next_candidates(keys, shr, sel, masks, **kw)
        assert list(got) == chosen, name
        np.testing.assert_allclose(np.stack(R.select_next_candidates.last_scores), scores, rtol=1e-5, atol=1e-7, err_msg=name)
        n += 1
    assert n == 6


def test_autocast_restatement_types_and_size_of_its_deviation(synth_sd):
    """oracle.cpu_ref.RefNetAutocast (CUDA autocast's operator policy restated; PARITY UNPINNED): the dtype each output has under
    the policy (fp16 convolution outputs, fp32 shrinkage - `pow` is an fp32 operator -, fp32 GRU state, fp32 probabilities) and
    a deviation from the fp32 network of the size eleven mantissa bits give."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    n32, n16 = R.RefNet(synth_sd), R.RefNetAutocast(synth_sd)
    fr, mk = T(synthetic_frames(1, 96, 128)), T(synthetic_masks(1, 2, 96, 128))
    a, b = n32.encode_key(fr), n16.encode_key(fr)
    want = [torch.float16, torch.float32, torch.float16, torch.float16, torch.float16, torch.float16]
    for x, y, dt, nm in zip(a, b, want, 'key shrinkage selection f16 f8 f4'.split()):
        assert y.dtype == dt, nm
        rel = float((x - y.float()).abs().max() / x.abs().max())
        assert 1e-5 < rel < 1e-2, (nm, rel)
    h = torch.zeros(1, 2, 64, 6, 8)
    va, vb = n32.encode_value(fr, a[3], h, mk), n16.encode_value(fr, b[3], h, mk)
    assert vb[0].dtype == torch.float16 and vb[1].dtype == torch.float32
    ro = torch.randn(1, 2, 512, 6, 8, generator=torch.Generator().manual_seed(3)) * 0.1
    sa, sb = n32.segment(a[3:], ro, va[1]), n16.segment(b[3:], ro.half(), vb[1])
    assert sb[0].dtype == torch.float32 and sb[2].dtype == torch.float32
    assert float((sa[2] - sb[2]).abs().max()) < 5e-2 and float((sa[0] - sb[0]).abs().max()) < 1e-2
    # the memory's GEMMs under the policy: fp16 results of a_sq / two_ab -> similarity errors of the size of an fp16 ulp of |a_sq|
    g = torch.Generator().manual_seed(5)
    mkk, ms = torch.randn(1, 64, 300, generator=g) * 0.5, 1 + torch.rand(1, 1, 300, generator=g)
    qk, qe = torch.randn(1, 64, 40, generator=g) * 0.5, torch.rand(1, 64, 40, generator=g)
    s32, s16 = R.get_similarity(mkk, ms, qk, qe), R.get_similarity_autocast(mkk, ms, qk.half(), qe.half())
    assert s16.dtype == torch.float32
    err = float((s32 - s16).abs().max())
    assert 1e-4 < err < 5e-2, err
