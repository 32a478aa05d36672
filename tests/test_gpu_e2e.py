"""GPU parity (end to end): InferenceCore.step over the synthetic clips recorded from the imported reference.

Acceptance: IoU >= 0.999 per clip against the reference's argmax masks and argmax identical wherever
the reference's own top-2 probability margin exceeds its thread-noise floor."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import BOTH_FP32_CLASS_MODES, load_golden
from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _run_clip(hip_net, tag, hw, n_obj):
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd import ops
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    g = load_golden('e2e_' + tag)
    cfg = ast.literal_eval(str(g['config']))
    t = int(g['shape'][0])
    frames = T(synthetic_frames(t, *hw)).cuda(); masks = T(synthetic_masks(t, n_obj, *hw)).cuda()
    labels = [int(x) for x in g['labels']]
    core = InferenceCore(hip_net, cfg)
    core.set_all_labels(labels)
    for j in g['perm_frames']:
        core.put_to_permanent_memory(frames[int(j)], masks[int(j)])
    mask_frames = set(int(x) for x in g['mask_frames'])
    out, sizes, probs = [], [], []
    for ti in range(t):
        mk = masks[ti] if ti in mask_frames else None
        p = core.step(frames[ti], mk, labels if mk is not None else None, end=(ti == t - 1),
                      do_not_add_mask_to_memory=(mk is not None))
        assert p.shape == (n_obj + 1,) + tuple(hw)
        assert float((p.sum(0) - 1).abs().max()) < 5e-6, 'class probabilities must sum to one (softmax of the aggregated logits)'
        out.append(ops.argmax_u8(p).cpu().numpy())
        probs.append(p[:, 4::8, 4::8].cpu().numpy())
        m = core.memory
        sizes.append([m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size])
    return np.stack(out), np.array(sizes), np.stack(probs), g


@BOTH_FP32_CLASS_MODES
@pytest.mark.parametrize('tag,hw,n_obj', [('480p_1obj', (480, 854), 1), ('240p_2obj', (240, 427), 2)])
def test_e2e_clip(hip_net, tag, hw, n_obj):
    got, sizes, probs, g = _run_clip(hip_net, tag, hw, n_obj)
    ref = g['argmax']
    np.testing.assert_array_equal(sizes, g['sizes'])
    ious = [R.compute_array_iou(got[i], ref[i]) for i in range(len(ref))]
    mism = float((got != ref).mean())
    d = np.abs(probs - g['prob_ds8'])
    print(f'{tag}: min IoU {min(ious):.5f}, argmax mismatch {mism:.2e}, prob max err {d.max():.3e} mean {d.mean():.3e}')
    # clip-level IoU per object (the reference's own 8-thread vs 1-thread runs reach only 0.998 on single frames of
    # this clip because object 1 is ~1800 px: a handful of boundary pixels; see DESIGN.md "noise floor")
    labels = [int(x) for x in g['labels']]
    clip_iou = [((got == c) & (ref == c)).sum() / max(((got == c) | (ref == c)).sum(), 1) for c in labels]
    print(f'{tag}: clip IoU per object {clip_iou}')
    assert min(clip_iou) >= 0.999, f'{tag}: clip-level IoU {clip_iou} < 0.999'
    assert min(ious) >= 0.99, f'{tag}: per-frame IoU {min(ious):.5f} < 0.99 (frame {int(np.argmin(ious))})'
    assert mism < 1e-4, f'{tag}: argmax mismatch fraction {mism:.2e}'
    # argmax identical wherever the reference's top-2 margin is clear (sampled grid where probabilities are stored)
    pr = g['prob_ds8']
    srt = np.sort(pr, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    clear = margin > 2e-2          # 2x the reference's own 8-vs-1-thread noise (1.1e-2)
    assert np.array_equal(probs.argmax(1)[clear], pr.argmax(1)[clear]), 'argmax differs where the reference margin is clear'
    assert d.mean() < 5e-4, f'mean prob error {d.mean():.3e}'


def test_step_flags_and_key_outputs(hip_net):
    """return_key_and_stuff shapes, disable_memory_updates, valid_labels merge path, 2 objects."""
    from conftest import base_config
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    fr = T(synthetic_frames(3, 96, 130)).cuda(); mk = T(synthetic_masks(3, 2, 96, 130)).cuda()
    core = InferenceCore(hip_net, base_config(mem_every=1))
    core.set_all_labels([1, 2])
    core.put_to_permanent_memory(fr[0], mk[0])
    ti0 = core.curr_ti
    p, key, shr, sel = core.step(fr[1], None, None, disable_memory_updates=True, return_key_and_stuff=True)
    assert core.curr_ti == ti0 and core.memory.temporary_work_mem.size == 0
    assert p.shape == (3, 96, 130) and key.shape == (1, 64, 6, 9) and shr.shape == (1, 1, 6, 9) and sel.shape == (1, 64, 6, 9)
    k2, s2, e2 = core.encode_frame_key(fr[1])
    assert torch.allclose(k2, key) and torch.allclose(s2, shr) and torch.allclose(e2, sel)
    # a mask for object 1 only: object 2 keeps the prediction outside the given region
    p2 = core.step(fr[2], mk[2], [1])
    assert p2.shape == (3, 96, 130) and bool(torch.isfinite(p2).all())
    assert float((p2.sum(0) - 1).abs().max()) < 1e-5
    assert core.memory.temporary_work_mem.size == 6 * 9


def test_run_on_video_surface(tmp_path):
    """The reference's run_on_video call shape end to end on files: stats DataFrame, mask PNGs, IoU column."""
    from PIL import Image
    from xmem2_amd.run_on_video import run_on_video
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    imgs, msks, out = tmp_path / 'JPEGImages', tmp_path / 'Annotations', tmp_path / 'out'
    imgs.mkdir(); msks.mkdir()
    t, hw = 7, (96, 128)
    frames = synthetic_frames(t, *hw); masks = synthetic_masks(t, 2, *hw)
    palette = [0, 0, 0, 200, 0, 0, 0, 200, 0] + [0] * (256 * 3 - 9)
    for i in range(t):
        rgb = np.clip((frames[i].transpose(1, 2, 0) * 0.229 + 0.45) * 255, 0, 255).astype(np.uint8)
        Image.fromarray(rgb).save(imgs / f'frame_{i:06d}.png')
        idx = (masks[i, 0] * 1 + masks[i, 1] * 2).astype(np.uint8)
        im = Image.fromarray(idx, mode='P'); im.putpalette(palette); im.save(msks / f'frame_{i:06d}.png')
    stats = run_on_video(str(imgs), str(msks), str(out), frames_with_masks=[0, 4], compute_iou=True, print_progress=False,
                         overwrite_config={'model': None, 'size': -1, 'mem_every': 2}, save_overlay=True)
    assert list(stats['frame']) == [f'frame_{i:06d}.png' for i in range(t)]
    assert list(stats['mask_provided']) == [i in (0, 4) for i in range(t)]
    assert all(stats['iou'][i] == -1 for i in (0, 4)) and all(0.0 <= stats['iou'][i] <= 1.0 for i in (1, 2, 3, 5, 6))
    written = sorted(os.listdir(out / 'masks'))
    assert written == [f'frame_{i:06d}.png' for i in range(t)] and len(os.listdir(out / 'overlay')) == t
    # a frame whose mask was given comes back as that mask
    got = np.array(Image.open(out / 'masks' / 'frame_000000.png').convert('RGB'))
    want = np.array(Image.open(msks / 'frame_000000.png').convert('RGB'))
    assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        run_on_video(str(imgs), str(msks), str(out), frames_with_masks=[], overwrite_config={'model': None, 'size': -1})


def test_e2e_objects_appearing_later_vs_oracle(hip_net, ref_net):
    """Three objects, the third annotated only at frame 6 (a second object group, YouTubeVOS style), default-like
    schedule with consolidation: HIP path vs the oracle on the same frames."""
    from conftest import base_config
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd import ops
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    t, hw = 16, (128, 192)
    cfg = base_config(mem_every=2, max_mid_term_frames=4, min_mid_term_frames=2, num_prototypes=32)
    frames = T(synthetic_frames(t, *hw)); masks = T(synthetic_masks(t, 3, *hw))
    core, ref = InferenceCore(hip_net, cfg), R.RefCore(ref_net, cfg)
    mism, n_pix, worst = 0, 0, 0.0
    for ti in range(t):
        if ti == 0:
            labels, mk = [1, 2], masks[ti, :2]
        elif ti == 6:
            labels, mk = [1, 2, 3], masks[ti]
        else:
            labels, mk = None, None
        if labels is not None:
            core.set_all_labels(labels); ref.set_all_labels(labels)
        kw = dict(end=(ti == t - 1))
        # only the NEW object carries a valid label at frame 6: the others keep their prediction (merge path)
        vl = None if mk is None else ([1, 2] if ti == 0 else [3])
        p = core.step(frames[ti].cuda(), mk.cuda() if mk is not None else None, vl, **kw)
        q = ref.step(frames[ti], mk.clone() if mk is not None else None, vl, **kw)
        assert p.shape == q.shape
        a, b = ops.argmax_u8(p).cpu().numpy(), torch.argmax(q, 0).numpy().astype(np.uint8)
        mism += int((a != b).sum()); n_pix += a.size
        worst = max(worst, float((p.cpu() - q).abs().mean()))
        m, rm = core.memory, ref.memory
        assert (m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size) == \
               (rm.temporary_work_mem.size, rm.permanent_work_mem.size, rm.long_mem.size), f'frame {ti}'
        assert m.temporary_work_mem.num_groups == rm.temporary_work_mem.num_groups
    print(f'objects-appearing-later: argmax mismatch {mism}/{n_pix}, worst mean |dp| {worst:.2e}')
    assert mism / n_pix < 5e-4 and worst < 5e-4
    assert core.memory.temporary_work_mem.num_groups == 2 and core.memory.long_mem.size > 0


@pytest.mark.parametrize('cfg_over,curated', [(dict(deep_update_every=3, mem_every=2), False),
                                              (dict(deep_update_every=-1, mem_every=1000), True)])
def test_update_schedules_vs_oracle(hip_net, ref_net, cfg_over, curated):
    """Non-synchronised deep updates (deep_update_every >= 0) and manually_curated_masks (memory frames = annotated
    frames only), frame by frame against the oracle; also update_config at run time."""
    from conftest import base_config
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    t, hw = 10, (96, 144)
    cfg = base_config(**cfg_over)
    frames = T(synthetic_frames(t, *hw)); masks = T(synthetic_masks(t, 1, *hw))
    core, ref = InferenceCore(hip_net, cfg), R.RefCore(ref_net, cfg)
    for c in (core, ref):
        c.set_all_labels([1])
    core.put_to_permanent_memory(frames[0].cuda(), masks[0].cuda())
    ref.put_to_permanent_memory(frames[0], masks[0])
    for ti in range(t):
        mk = masks[ti] if ti in (0, 5) else None
        kw = dict(end=(ti == t - 1), manually_curated_masks=curated, do_not_add_mask_to_memory=(ti == 0))
        if ti == 7:
            cfg2 = dict(cfg, mem_every=1, top_k=20)
            core.update_config(cfg2); ref.update_config(cfg2)
        p = core.step(frames[ti].cuda(), mk.cuda() if mk is not None else None, [1] if mk is not None else None, **kw)
        q = ref.step(frames[ti], mk.clone() if mk is not None else None, [1] if mk is not None else None, **kw)
        d = (p.cpu() - q).abs()
        assert float(d.mean()) < 3e-4 and float((p.cpu().argmax(0) != q.argmax(0)).float().mean()) < 1e-3, f'frame {ti}'
        m, rm = core.memory, ref.memory
        assert (m.temporary_work_mem.size, m.permanent_work_mem.size) == (rm.temporary_work_mem.size, rm.permanent_work_mem.size)
        hd = (m.get_hidden().permute(0, 3, 1, 2).cpu() - rm.get_hidden()[0]).abs()
        assert float(hd.max()) < 5e-3, f'hidden state diverged at frame {ti}: {float(hd.max()):.2e}'
    with pytest.raises(AssertionError):
        core.update_config(dict(cfg, enable_long_term=False))


def test_prefetch_keys_is_only_a_hint(hip_net):
    """Batched key-encoder hints (prefetch_keys) must not change what step() computes: a hinted stream (batches of 4,
    a batch of 3, single hints, a stale hint that gets dropped, memory frames + a consolidation, 2 objects) against an
    un-hinted stream on the same frames.  Batched convolutions may pick other tile plans, so equality is to fp32
    round-off, not bitwise."""
    from conftest import base_config
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd import ops
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    t, hw = 20, (96, 128)
    fr = T(synthetic_frames(t, *hw)).cuda(); mk = T(synthetic_masks(t, 2, *hw)).cuda()
    cfg = base_config(mem_every=2, max_mid_term_frames=4, min_mid_term_frames=2, num_prototypes=16)
    cores = [InferenceCore(hip_net, dict(cfg)) for _ in range(2)]
    for c in cores:
        c.set_all_labels([1, 2])
        c.put_to_permanent_memory(fr[0], mk[0])
    plain = [cores[0].step(fr[i], None, None).clone() for i in range(1, t)]
    c = cores[1]
    hinted = []
    schedule = {1: 4, 5: 4, 9: 3, 12: 1, 13: 1, 14: 4}               # frame index -> hint size issued before it
    i = 1
    while i < t:
        if i in schedule:
            devs = c.prefetch_keys([fr[j] for j in range(i, min(i + schedule[i], t))])
            assert all(d.data_ptr() == fr[i + k].data_ptr() for k, d in enumerate(devs))
        if i == 16:                                                  # hinted frames 14..17 pending: jump the queue
            c.prefetch_keys([fr[18], fr[19]])                         # more hints on top, then an out-of-order frame
            hinted.append(c.step(fr[16].clone(), None, None).clone())
            assert len(c._pfq) == 0                                  # stale hints were dropped
        else:
            hinted.append(c.step(fr[i], None, None).clone())
        i += 1
    assert cores[0].memory.long_mem.size == c.memory.long_mem.size > 0
    assert cores[0].memory.temporary_work_mem.size == c.memory.temporary_work_mem.size
    for a, b in zip(plain, hinted):
        assert float((a - b).abs().max()) < 2e-3
        diff = ops.argmax_u8(a) != ops.argmax_u8(b)
        assert float(diff.float().mean()) < 5e-4               # a handful of zero-margin pixels of 12288 may flip
        top2 = torch.topk(a, 2, dim=0).values
        assert not bool((diff & ((top2[0] - top2[1]) > 4e-3)).any()), 'argmax differs where the margin is clear'
    with pytest.raises(ValueError):
        c.prefetch_keys([fr[0], fr[1][:, :64]])


@pytest.mark.parametrize('hw,n_obj,perm,steps', [((720, 1280), 1, 3, 3), ((1080, 1920), 2, 1, 2), ((1080, 1920), 5, 1, 1)],
                         ids=['720p_1obj', '1080p_2obj', '1080p_5obj_config5'])
@BOTH_FP32_CLASS_MODES
def test_e2e_large_frames_vs_oracle(hip_net, ref_net, hw, n_obj, perm, steps):
    """BASELINE configs 4 / 5 frame geometry (720p, 1080p -> padded 1088 x 1920) end to end against the oracle on a few
    frames: permanent preload, batched key hints, one memory frame, 1-2 objects.  Same acceptance as the 480p clips."""
    from conftest import base_config
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd import ops
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    t = perm + steps
    frames = T(synthetic_frames(t, *hw)); masks = T(synthetic_masks(t, n_obj, *hw))
    cfg = base_config(mem_every=2)
    core, ref = InferenceCore(hip_net, cfg), R.RefCore(ref_net, cfg)
    labels = list(range(1, n_obj + 1))
    for c in (core, ref):
        c.set_all_labels(labels)
    for j in range(perm):
        core.put_to_permanent_memory(frames[j].cuda(), masks[j].cuda())
        ref.put_to_permanent_memory(frames[j], masks[j])
    dev = [frames[perm + i].cuda() for i in range(steps)]
    core.prefetch_keys(dev)                                    # one batched key-encoder hint for all query frames
    inter = uni = mism = clear_mism = 0
    for i in range(steps):
        p = core.step(dev[i], None, None)
        q = ref.step(frames[perm + i], None, None)
        assert p.shape == q.shape == (n_obj + 1,) + tuple(hw)
        a, b = ops.argmax_u8(p).cpu().numpy(), torch.argmax(q, 0).numpy().astype(np.uint8)
        mism += int((a != b).sum())
        top2 = torch.topk(q, 2, dim=0).values
        clear_mism += int(((a != b) & ((top2[0] - top2[1]).numpy() > 2e-2)).sum())
        inter += int(((a > 0) & (b > 0) & (a == b)).sum()); uni += int(((a > 0) | (b > 0)).sum())
        assert float((p.cpu() - q).abs().mean()) < 5e-4
        m, rm = core.memory, ref.memory
        assert (m.temporary_work_mem.size, m.permanent_work_mem.size) == (rm.temporary_work_mem.size, rm.permanent_work_mem.size)
    iou = inter / max(uni, 1)
    print(f'{hw}: IoU {iou:.5f}, argmax mismatch {mism}/{steps * hw[0] * hw[1]}, of which at a clear oracle margin: {clear_mism}')
    # more objects -> more boundary pixels at zero margin: the mismatch budget scales with the object count; where the
    # oracle's own top-2 margin is clear (2x its thread-noise floor) the argmax must be identical
    assert iou >= 0.999 and mism / (steps * hw[0] * hw[1]) < 1e-4 * max(1, 2 * n_obj) and clear_mism == 0


def test_run_on_video_with_augmented_permanent_memory(tmp_path, hip_net):
    """augment_images_with_masks (run_on_video.py:231-242): each annotated frame enters the permanent memory 1 + 11 times
    ('best_all'); the harness call runs end to end and the augmented frames are valid memory frames."""
    from PIL import Image
    from conftest import base_config
    from xmem2_amd.augmentations import get_determenistic_augmentations
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.run_on_video import run_on_video, VideoReader
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    imgs, msks, out = tmp_path / 'JPEGImages', tmp_path / 'Annotations', tmp_path / 'out'
    imgs.mkdir(); msks.mkdir()
    t, hw = 5, (96, 128)
    frames = synthetic_frames(t, *hw); masks = synthetic_masks(t, 1, *hw)
    palette = [0, 0, 0, 255, 255, 255] + [0] * (256 * 3 - 6)
    for i in range(t):
        rgb = np.clip((frames[i].transpose(1, 2, 0) * 0.229 + 0.45) * 255, 0, 255).astype(np.uint8)
        Image.fromarray(rgb).save(imgs / f'frame_{i:06d}.png')
        im = Image.fromarray(masks[i, 0].astype(np.uint8), mode='P'); im.putpalette(palette); im.save(msks / f'frame_{i:06d}.png')
    stats = run_on_video(str(imgs), str(msks), str(out), frames_with_masks=[0], print_progress=False,
                         augment_images_with_masks=True, overwrite_config={'model': None, 'size': -1}, save_overlay=False)
    assert len(stats) == t and len(os.listdir(out / 'masks')) == t
    # the preload itself, step by step: 12 permanent frames of h*w elements for one annotation
    reader = VideoReader('', str(imgs), str(msks), size=-1, use_all_masks=True)
    smp = reader[0]
    core = InferenceCore(hip_net, base_config())
    core.set_all_labels([1])
    msk = T(masks[0])
    core.put_to_permanent_memory(smp.rgb_u8.cuda(), msk.cuda())
    for img_aug, mask_aug in get_determenistic_augmentations((3,) + hw, msk, subset='best_all'):
        core.put_to_permanent_memory(reader.frame_u8(img_aug(smp.raw_image_pil)).cuda(), mask_aug(msk).cuda())
    assert core.memory.permanent_work_mem.size == 12 * (hw[0] // 16) * (hw[1] // 16)
    p = core.step(T(frames[1]).cuda(), None, None)
    assert bool(torch.isfinite(p).all()) and float((p.sum(0) - 1).abs().max()) < 1e-4


def _noise_floor_gate(name, gpu, floor, frac_cap, floor_factor=1.5, floor_deficit=None):
    """The GPU path against oracle(1 thread) may deviate by what north_star allows (IoU >= 0.999 per object, argmax mismatch
    `frac_cap`) or, where the reference's OWN thread-count noise on the same frames is larger, by `floor_factor` (1.5) x that
    measured floor - never more, and never below IoU 0.997.  `floor` = oracle(8 threads) vs oracle(1 thread), `gpu` = HIP path vs
    oracle(1 thread), both from clip_util.compare.  The yardstick for every object is the reference's WORST object on these frames
    (`floor_deficit` overrides it, e.g. with the worst phase of the clip): two fp32 implementations of a feedback loop - predicted
    masks re-enter the memory - are two samples of one round-off-driven divergence, and which object a sample hits hardest is
    chance; the aggregate argmax mismatch is gated against the floor's own aggregate with the same factor."""
    fd = max(1.0 - f for f in floor['iou']) if floor_deficit is None else floor_deficit
    allowed = min(3e-3, max(1e-3, floor_factor * fd))
    for k, (g, f) in enumerate(zip(gpu['iou'], floor['iou'])):
        assert 1.0 - g <= allowed, (f'{name}: object {k + 1} IoU {g:.5f} vs oracle(1 thread); the oracle\'s own 8-vs-1-thread IoU on the '
                                    f'same frames is {f:.5f} (worst object / phase deficit {fd:.2e}): allowed deficit {allowed:.2e}')
    cap = max(frac_cap, floor_factor * floor['mismatch'] / max(floor['pixels'], 1))
    assert gpu['mismatch'] / max(gpu['pixels'], 1) <= cap, f"{name}: argmax mismatch {gpu['mismatch']}/{gpu['pixels']} > {cap:.2e}"


def test_e2e_480p_three_objects_consolidation_vs_oracle(hip_net_mo, ref_net_mo):
    """BASELINE config 3 at its stated size: 480p, 3 objects, a long-term consolidation inside the clip (mem_every=2,
    T_max=4 -> compress_features fires at the 5th temporary frame), batched key hints - against the oracle frame by frame,
    WITH the oracle's own thread-count noise measured on the same frames (8 threads vs 1 thread).
    Round 5: the clip runs on the multi-object conditioning of the synthetic checkpoint, on which it DISCRIMINATES (1.2 % of the
    pixels near a tie instead of 45 %; the reference agrees with itself at IoU 0.9994-0.9996), so north_star's IoU >= 0.999 per
    object is the gate that applies.  Measured on MI355X (profiles/r05_c3_parity_by_plan.txt): with the textbook F(4x4) points the
    HIP path missed it (IoU 0.9974-0.9979, 73 px before the consolidation against the reference's own 20); with the points
    {0, +-3/4, +-3/2, inf} it is 3 px, IoU 0.99975-1.0 - closer to oracle(1 thread) than oracle(8 threads) is."""
    import clip_util as U
    hip_net, ref_net = hip_net_mo, ref_net_mo
    clip = U.c3_clip()
    o1, p1, s1 = U.run_oracle(ref_net, clip, 1)
    o8, p8, s8 = U.run_oracle(ref_net, clip, 8)
    a, p, s = U.run_gpu(hip_net, clip)
    assert s == s1, 'memory sizes differ from the oracle'
    # the clip must DISCRIMINATE (round 5: multi-object conditioning of the checkpoint): few pixels near a tie, every object present,
    # and the reference agreeing with itself across thread counts above north_star's 0.999
    near_tie = float(np.mean([float(((torch.topk(q, 2, dim=0).values[0] - torch.topk(q, 2, dim=0).values[1]) < 1e-2).float().mean()) for q in p1]))
    self_cmp = U.compare(o8, o1, clip.labels)
    sizes_px = [int((np.stack(o1) == c).sum()) // len(o1) for c in clip.labels]
    print(f'480p x 3 objects: oracle pixels within 0.01 of a tie {near_tie:.4f}; object sizes (px / frame) {sizes_px}; oracle(8) vs oracle(1): {U.fmt(self_cmp)}')
    assert near_tie < 0.05 and min(sizes_px) > 500, (near_tie, sizes_px)
    first_lt = next((i for i, z in enumerate(s) if z[2] > 0), None)
    assert first_lt is not None and first_lt < len(a) - 2, 'the clip must include a consolidation with frames after it'
    for i in range(len(a)):
        assert float((p[i] - p1[i]).abs().mean()) < (5e-4 if i < first_lt else 2e-3), f'frame {i + 1}'
    gpu, floor = U.compare(a, o1, clip.labels), U.compare(o8, o1, clip.labels)
    print(f'480p x 3 objects, whole clip ({len(a)} frames, consolidation at frame {first_lt + 1}):\n   HIP    vs oracle(1 thr): {U.fmt(gpu)}\n'
          f'   oracle(8 thr) vs (1 thr): {U.fmt(floor)}')
    _noise_floor_gate('480p x 3 objects, whole clip', gpu, floor, frac_cap=1e-4)
    # per phase: every second frame is written back to the memory with its PREDICTED masks (mem_every=2), so round-off feeds
    # back, and the consolidation picks prototypes by a top-k over accumulated usage (memory_manager.py:355): 1-ulp differences
    # can fork that discrete choice - for the 8-thread oracle exactly as for the GPU.  Each phase of the GPU path is
    # bounded by the reference's own WORST deficit anywhere on this clip.
    phases = [U.compare(o8, o1, clip.labels, lo, hi) for lo, hi in ((0, first_lt), (first_lt, len(a)))]
    worst = max(1.0 - v for ph in phases for v in ph['iou'])
    for ph, (lo, hi) in enumerate([(0, first_lt), (first_lt, len(a))]):
        g = U.compare(a, o1, clip.labels, lo, hi)
        tag = f'480p x 3 objects, {"before" if ph == 0 else "after"} the consolidation'
        print(f'{tag}:\n   HIP    vs oracle(1 thr): {U.fmt(g)}\n   oracle(8 thr) vs (1 thr): {U.fmt(phases[ph])}')
        _noise_floor_gate(tag, g, phases[ph], frac_cap=max(1e-4, 1.5 * floor['mismatch'] / floor['pixels']), floor_deficit=worst)


def test_e2e_480p_three_objects_plain_checkpoint_noise_floor(hip_net, ref_net):
    """The config-3 clip on the PLAIN synthetic checkpoint (when the clip moved to the 'multi_object' conditioning, the plain
    checkpoint with 3 objects was no longer gated at all).  On it 45 % of the pixels are ties between objects and the reference's own
    argmax is noise (its IoU against itself across thread counts ~0.998), so the gate is the round-4 one: the HIP path may deviate from
    oracle(1 thread) by 1.5x what oracle(8 threads) does on the same frames, never below IoU 0.997."""
    import clip_util as U
    clip = U.c3_clip()
    clip.name += '_plain_checkpoint'                     # (the oracle cache is keyed by clip name: other weights, other trajectory)
    o1, _, s1 = U.run_oracle(ref_net, clip, 1)
    o8, _, _ = U.run_oracle(ref_net, clip, 8)
    a, _, s = U.run_gpu(hip_net, clip)
    assert s == s1, 'memory sizes differ from the oracle'
    gpu, floor = U.compare(a, o1, clip.labels), U.compare(o8, o1, clip.labels)
    print(f'480p x 3 objects, plain checkpoint:\n   HIP    vs oracle(1 thr): {U.fmt(gpu)}\n   oracle(8 thr) vs (1 thr): {U.fmt(floor)}')
    _noise_floor_gate('480p x 3 objects, plain checkpoint', gpu, floor, frac_cap=1e-4)


def test_e2e_480p_three_objects_bench_c3_stream_vs_float64_reference(hip_net_mo):
    """The stream `bench.py --workload c3` itself (25 frames, one permanent frame, mem_every=5: five memory frames carry PREDICTED masks
    back into the memory) against the REFERENCE EVALUATED IN FLOAT64 (tests/golden/make_c3_bench_goldens.py: the imported reference's own
    code with every tensor a double - the exact answer of its algorithm on these frames).

    Why float64: round 5's bench print on this clip (HIP vs the fp32 oracle: 119 px, IoU 0.99848; oracle at 8 threads vs 1 thread: 32 px)
    read as a parity gap 3.7x outside "the reference's own noise".  It is the fp32 REFERENCE that forks there: against float64 the
    reference's fp32 path is off by 48 px at frame 7 alone (max |dp| 1.5e-2; 76 px / IoU 0.99864 over the clip in the build container,
    102 px / 0.99811 on an MI355X host), the HIP path by 0 at that frame and 47 over the clip (IoU >= 0.99902), whatever the convolution
    form (profiles/r06_c3_bench_stream_margins.txt, r06_c3_parity_by_plan.txt).  The oracle at 8 vs 1 thread share most of their
    arithmetic and fork together, so that pair under-states the noise of an fp32 evaluation of this feedback loop.
    Gate: IoU >= 0.999 per object against float64 (north_star), and no further from the exact answer than 1.5x the reference's own fp32
    path is - in argmax pixels overall and at a float64 top-2 margin > 2e-3."""
    import clip_util as U
    g = load_golden('c3_bench_stream')
    clip = U.c3_bench_clip(int(g['steps']))
    assert ast.literal_eval(str(g['config'])) == clip.cfg and tuple(g['shape']) == tuple(clip.frames.shape)
    a, p, s = U.run_gpu(hip_net_mo, clip)
    assert np.array_equal(np.array(s), g['sizes']), 'memory sizes differ from the reference'
    A, R64, R32 = np.stack(a), g['argmax_f64'], g['argmax_f32_1thr']
    clear = np.unpackbits(g['clear_2e3'])[:A.size].reshape(A.shape).astype(bool)
    gpu, ref32 = U.compare(list(A), list(R64), clip.labels), U.compare(list(R32), list(R64), clip.labels)
    g_m, r_m = int(((A != R64) & clear).sum()), int(((R32 != R64) & clear).sum())
    pd = np.abs(np.stack([q[:, 4::8, 4::8].numpy() for q in p]) - g['prob_f64_ds8'])
    print(f'bench c3 stream ({len(a)} frames) against the reference in float64:\n   HIP path               : {U.fmt(gpu)}; at a margin > 2e-3: {g_m} px; max |dp| {pd.max():.2e}\n'
          f'   reference fp32 (1 thr) : {U.fmt(ref32)}; at a margin > 2e-3: {r_m} px; max |dp| {float(g["max_abs_dp_f32_vs_f64"].max()):.2e}\n'
          f'   HIP vs reference fp32  : {U.fmt(U.compare(list(A), list(R32), clip.labels))}\n'
          f'   per frame HIP != f64: {[(int((A[i] != R64[i]).sum())) for i in range(len(a))]}\n   per frame f32 != f64: {[(int((R32[i] != R64[i]).sum())) for i in range(len(a))]}')
    assert min(gpu['iou']) >= 0.999, f"north_star: IoU >= 0.999 per object against the exact answer, got {gpu['iou']}"
    assert gpu['mismatch'] <= 1.5 * ref32['mismatch'] + 8, f"argmax mismatch vs float64: HIP {gpu['mismatch']} px, the reference's fp32 path {ref32['mismatch']}"
    assert g_m <= 1.5 * r_m + 8, f'argmax mismatch vs float64 at the 2e-3 margin: HIP {g_m} px, the reference\'s fp32 path {r_m}'
    assert pd.max() <= 1.5 * float(g['max_abs_dp_f32_vs_f64'].max()) + 1e-3


def test_e2e_240p_two_objects_noise_floor(hip_net, ref_net):
    """The 240p two-object golden clip (object 1 is ~1800 px) with the reference's own noise floor next to it: oracle at 8
    threads vs the 1-thread goldens on the same frames, then the HIP path under the same gate as config 3."""
    import clip_util as U
    clip = U.golden_clip('240p_2obj', (240, 427), 2)
    g = load_golden('e2e_240p_2obj')
    o1, _, s1 = U.run_oracle(ref_net, clip, 1)
    # the goldens were recorded by the imported reference at 1 thread IN THE BUILD CONTAINER; the same oracle at 1 thread on this
    # box's CPU is bit-equal only if MKL / oneDNN pick the same kernels there - a third sample of the reference's own noise
    container = U.compare(o1, list(g['argmax']), clip.labels)
    print(f'240p x 2 objects: oracle(1 thr) on this host vs the goldens recorded in the build container: {U.fmt(container)}')
    o8, _, _ = U.run_oracle(ref_net, clip, 8)
    a, _, s = U.run_gpu(hip_net, clip)
    assert s == s1
    gpu, floor = U.compare(a, o1, clip.labels), U.compare(o8, o1, clip.labels)
    print(f'240p x 2 objects:\n   HIP    vs oracle(1 thr): {U.fmt(gpu)}\n   oracle(8 thr) vs (1 thr): {U.fmt(floor)}')
    _noise_floor_gate('240p x 2 objects', gpu, floor, frac_cap=1e-4)

