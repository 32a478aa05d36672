"""GPU parity of the harness (BASELINE config 1): `xmem2_amd.run_on_video.run_on_video` on FILES against

* masks recorded from the imported reference (`tests/golden/chair_*.npz`, made by `make_goldens.py::gen_chair` which drives the
  reference's `InferenceCore` over the decoded chair frames exactly as `inference/run_on_video.py:59-66,94-137,165-173` does);
* the oracle's `RefCore` + `post_process` on the same decoded arrays for the resized (`size=240`) path.

Also: loading a checkpoint through `config['model']` / `XMem(config, model_path)` and the single-object -> multi-object stem
surgery of `model/network.py:184-198`."""
import ast
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu
T = torch.from_numpy
CHAIR = os.path.join(GOLDEN, 'chair')


@pytest.fixture(scope='module')
def checkpoint(synth_sd, tmp_path_factory):
    path = tmp_path_factory.mktemp('ckpt') / 'XMem_synth.pth'
    torch.save(synth_sd, path)
    return str(path)


def _read_written_masks(out_dir, names, ref_png):
    """The harness writes RGB PNGs in the annotation's palette colours: map them back to label ids."""
    from PIL import Image
    pal = np.array(Image.open(ref_png).convert('P').getpalette()[:3 * 256], np.int64).reshape(-1, 3)
    lut = {tuple(c): i for i, c in reversed(list(enumerate(pal.tolist())))}       # lowest index wins
    out = []
    for n in names:
        rgb = np.array(Image.open(os.path.join(out_dir, 'masks', n[:-4] + '.png')).convert('RGB'), np.int64)
        ids = np.full(rgb.shape[:2], 255, np.uint8)
        for c in np.unique(rgb.reshape(-1, 3), axis=0):
            ids[(rgb == c).all(-1)] = lut[tuple(c.tolist())]
        out.append(ids)
    return np.stack(out)


def _compare(got, want, what):
    ious = [R.compute_array_iou(got[i], want[i]) for i in range(len(want))]
    mism = float((got != want).mean())
    clip = ((got > 0) & (want > 0)).sum() / max(((got > 0) | (want > 0)).sum(), 1)
    print(f'{what}: clip IoU {clip:.5f}, min frame IoU {min(ious):.5f}, argmax mismatch {mism:.2e}')
    assert clip >= 0.999 and min(ious) >= 0.995, f'{what}: IoU {clip:.5f} / {min(ious):.5f}'
    assert mism < 1e-4, f'{what}: mismatch {mism:.2e}'


@pytest.mark.parametrize('tag', ['fm0', 'fm0_5'])
def test_run_on_video_chair_vs_reference_goldens(tag, checkpoint, tmp_path):
    """Config 1: the chair frames from disk through run_on_video (default config: size=480, mem_every=10, long-term on),
    weights loaded from a checkpoint FILE via config['model']; the written PNGs against the reference-recorded masks."""
    from xmem2_amd.run_on_video import run_on_video
    g = load_golden('chair_' + tag)
    over = ast.literal_eval(str(g['overwrite_config']))
    over['model'] = checkpoint
    fm = [int(x) for x in g['frames_with_masks']]
    stats = run_on_video(os.path.join(CHAIR, 'JPEGImages'), os.path.join(CHAIR, 'Annotations'), str(tmp_path / 'out'),
                         frames_with_masks=fm, compute_iou=True, print_progress=False, overwrite_config=over)
    names = [str(n) for n in g['names']]
    assert list(stats['frame']) == names
    assert list(stats['mask_provided']) == [bool(b) for b in g['mask_provided']]
    np.testing.assert_allclose(np.array(stats['iou'], np.float64), g['iou'], atol=2e-3)
    got = _read_written_masks(str(tmp_path / 'out'), names, os.path.join(CHAIR, 'Annotations', names[0][:-4] + '.png'))
    assert got.shape == g['argmax'].shape == (10, 480, 720)
    _compare(got, g['argmax'], f'chair {tag}')
    assert len(os.listdir(tmp_path / 'out' / 'overlay')) == len(names)


def test_run_on_video_resized_vs_oracle(checkpoint, synth_sd, ref_net, tmp_path):
    """`size=240` (frames resized 720x480 -> 360x240 on the way in, probabilities resized back before the argmax,
    run_on_video.py:165-170) and `size=-1` on the same files, against RefCore + post_process fed with the same decoded and
    PIL-resized arrays (torchvision's Resize on a PIL image IS PIL's resize; decode parity itself stays unpinned)."""
    from PIL import Image
    from xmem2_amd.configuration import VIDEO_INFERENCE_CONFIG
    from xmem2_amd.run_on_video import run_on_video
    names = sorted(os.listdir(os.path.join(CHAIR, 'JPEGImages')))[:6]
    imgs, msks = tmp_path / 'JPEGImages', tmp_path / 'Annotations'
    imgs.mkdir(); msks.mkdir()
    for n in names:
        os.symlink(os.path.join(CHAIR, 'JPEGImages', n), imgs / n)
        os.symlink(os.path.join(CHAIR, 'Annotations', n[:-4] + '.png'), msks / (n[:-4] + '.png'))
    mean = np.array([0.485, 0.456, 0.406], np.float32); std = np.array([0.229, 0.224, 0.225], np.float32)
    for size, work_hw in ((240, (240, 360)), (-1, (480, 720))):
        over = {'model': checkpoint, 'size': size, 'mem_every': 2}
        out = tmp_path / f'out{size}'
        run_on_video(str(imgs), str(msks), str(out), frames_with_masks=[0], print_progress=False, overwrite_config=dict(over))
        got = _read_written_masks(str(out), names, os.path.join(CHAIR, 'Annotations', names[0][:-4] + '.png'))
        # the oracle on the same arrays
        cfg = dict(VIDEO_INFERENCE_CONFIG); cfg.update(over)
        cfg['enable_long_term_count_usage'] = False          # 6 frames: the derived flag of run_on_video.py:190-196
        core, mapper = R.RefCore(ref_net, cfg), R.RefMaskMapper()
        rgbs = []
        for n in names:
            im = Image.open(os.path.join(CHAIR, 'JPEGImages', n)).convert('RGB')
            if size > 0:
                im = im.resize((work_hw[1], work_hw[0]), Image.BILINEAR)
            x = T(np.array(im, np.uint8)).permute(2, 0, 1).to(torch.float32).div(255)
            rgbs.append(((x - T(mean)[:, None, None]) / T(std)[:, None, None]).contiguous())
        gt0 = np.array(Image.open(os.path.join(CHAIR, 'Annotations', names[0][:-4] + '.png')).convert('P'), np.uint8)
        msk, labels = mapper.convert_mask(gt0, exhaustive=True)
        msk = torch.Tensor(msk)
        if size > 0:                                          # video_reader.py:148-153: nearest
            msk = torch.nn.functional.interpolate(msk.unsqueeze(0), work_hw, mode='nearest')[0]
        core.set_all_labels(list(mapper.remappings.values()))
        core.put_to_permanent_memory(rgbs[0], msk)
        want = []
        for ti in range(len(names)):
            m = msk if ti == 0 else None
            p = core.step(rgbs[ti], m, labels if m is not None else None, end=(ti == len(names) - 1),
                          do_not_add_mask_to_memory=(m is not None))
            want.append(mapper.remap_index_mask(R.post_process(p, (480, 720) if size > 0 else None)))
        _compare(got, np.stack(want), f'chair size={size}')


def test_checkpoint_path_and_single_object_stem(synth_sd, checkpoint, ref_net):
    """`XMem(config, model_path)` reads C_k / C_v / C_h from the file (model/network.py:134-182) and a single-object
    checkpoint (4-channel value stem) is padded to the 5-channel multi-object stem (model/network.py:184-198)."""
    from xmem2_amd.network import XMem
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    cfg = {}
    net = XMem(cfg, checkpoint, map_location='cpu').to('cuda').eval()
    assert (cfg['key_dim'], cfg['value_dim'], cfg['hidden_dim']) == (64, 512, 64)
    frame = T(synthetic_frames(1, 96, 128, seed=3)[0])[None]
    masks = T(synthetic_masks(1, 2, 96, 128)[0])[None]
    key, shr, sel, f16, f8, f4 = net.encode_key(frame.cuda())
    rk, rs, re, rf16, _, _ = ref_net.encode_key(frame)
    torch.testing.assert_close(key.cpu(), rk, rtol=2e-3, atol=2e-4)
    torch.testing.assert_close(shr.cpu(), rs, rtol=2e-3, atol=2e-4)
    # single-object checkpoint: drop the `other masks` input channel, load with zero padding -> identical to a
    # multi-object checkpoint whose 5th input channel is zero
    k = 'value_encoder.conv1.weight'
    sd4 = dict(synth_sd); sd4[k] = synth_sd[k][:, :4].clone()
    sd5 = dict(synth_sd); sd5[k] = torch.cat([sd4[k], torch.zeros(64, 1, 7, 7)], 1)
    net_so = XMem({}, None).to('cuda').eval()
    net_so.load_weights(sd4, init_as_zero_if_needed=True)
    assert tuple(net_so.state_dict()[k].shape) == (64, 5, 7, 7) and float(net_so.state_dict()[k][:, 4].abs().max()) == 0.0
    hidden0 = torch.zeros(1, 2, 64, 6, 8)
    prob = R.aggregate(masks[0], dim=0)
    val, _ = net_so.encode_value(frame.cuda(), f16, hidden0.cuda(), prob[1:].unsqueeze(0).cuda(), is_deep_update=False)
    want, _ = R.RefNet(sd5).encode_value(frame, rf16, hidden0, prob[1:].unsqueeze(0), is_deep_update=False)
    torch.testing.assert_close(val.cpu(), want, rtol=2e-3, atol=2e-4)
    # random (orthogonal) padding when not asked for zeros: right shape, non-zero, original 4 channels untouched
    net_rand = XMem({}, None).to('cuda').eval()
    net_rand.load_weights(dict(sd4), init_as_zero_if_needed=False)
    w = net_rand.state_dict()[k]
    assert tuple(w.shape) == (64, 5, 7, 7) and float(w[:, 4].abs().max()) > 0 and torch.equal(w[:, :4], sd4[k])
    # a checkpoint with a wrong tensor shape is refused like nn.Module.load_state_dict does
    bad = dict(synth_sd); bad['key_proj.key_proj.weight'] = torch.zeros(32, 1024, 3, 3)
    with pytest.raises(RuntimeError):
        XMem({}, None).to('cuda').load_weights(bad)


def test_remove_from_permanent_memory_core_vs_oracle(hip_net, ref_net):
    """InferenceCore.remove_from_permanent_memory through the full step path, incl. a frame whose saved position is > 0
    (the reference hands the POSITION to remove_at as an element offset, kv_memory_store.py:120-123 - kept)."""
    from conftest import base_config
    from xmem2_amd import ops
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    t, hw = 8, (96, 128)
    cfg = base_config(mem_every=3)
    frames, masks = T(synthetic_frames(t, *hw)), T(synthetic_masks(t, 2, *hw))
    core, ref = InferenceCore(hip_net, cfg), R.RefCore(ref_net, cfg)
    for c in (core, ref):
        c.set_all_labels([1, 2])
    for ti in (0, 3, 6):
        assert core.put_to_permanent_memory(frames[ti].cuda(), masks[ti].cuda(), ti=ti) is False
        ref.put_to_permanent_memory(frames[ti], masks[ti], ti=ti)
    assert core.permanent_memory_frames == ref.permanent_memory_frames == [0, 3, 6]

    def check(tag):
        for ti in (1, 2):
            p = core.step(frames[ti].cuda(), None, None)
            q = ref.step(frames[ti], None, None)
            a, b = ops.argmax_u8(p).cpu().numpy(), torch.argmax(q, 0).numpy().astype(np.uint8)
            assert (a != b).mean() < 5e-4, f'{tag}: mismatch {(a != b).mean():.2e}'
            assert float((p.cpu() - q).abs().mean()) < 5e-4
        assert core.memory.permanent_work_mem.size == ref.memory.permanent_work_mem.size
        assert core.permanent_memory_frames == ref.permanent_memory_frames

    check('before')
    for idx in (3, 0):
        core.remove_from_permanent_memory(idx); ref.remove_from_permanent_memory(idx)
        check(f'after removing {idx}')
    with pytest.raises(KeyError):
        core.remove_from_permanent_memory(3)


def test_launcher_runs_videos_on_this_gpu(checkpoint, tmp_path):
    """xmem2_amd.launch end to end on the box's GPU(s): two chair sub-clips dealt to `--gpus 1` worker(s), real run_on_video,
    masks written per video, summary merged; the longer video goes first.  (World size 2 and the refusal paths run on CPU in
    tests/test_multi_gpu.py.)"""
    import json
    import subprocess
    import sys
    names = sorted(os.listdir(os.path.join(CHAIR, 'JPEGImages')))
    for vid, sel in (('short', names[:4]), ('long', names[:7])):
        for sub, ext in (('JPEGImages', '.jpg'), ('Annotations', '.png')):
            d = tmp_path / sub / vid
            d.mkdir(parents=True)
            for n in sel:
                os.symlink(os.path.join(CHAIR, sub, n[:-4] + ext), d / (n[:-4] + ext))
    out = tmp_path / 'out'
    root = os.path.dirname(os.path.dirname(GOLDEN))
    cmd = [sys.executable, '-m', 'xmem2_amd.launch', '--gpus', '1', '--videos', str(tmp_path / 'JPEGImages'),
           '--masks', str(tmp_path / 'Annotations'), '--out', str(out), '--frames-with-masks', '0', '--compute-iou',
           '--config', json.dumps({'model': checkpoint, 'size': -1})]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    summ = json.load(open(out / 'summary.json'))
    assert summ['n_gpus'] == 1 and summ['total_frames'] == 11 and not summ['ranks_missing']
    assert [v['name'] for v in summ['videos']] == ['long', 'short'] and all(v['rank'] == 0 for v in summ['videos'])
    assert len(os.listdir(out / 'long' / 'masks')) == 7 and len(os.listdir(out / 'short' / 'masks')) == 4
    # same masks as a direct run_on_video call on the same files
    from xmem2_amd.run_on_video import run_on_video
    run_on_video(str(tmp_path / 'JPEGImages' / 'short'), str(tmp_path / 'Annotations' / 'short'), str(tmp_path / 'direct'),
                 frames_with_masks=[0], print_progress=False, overwrite_config={'model': checkpoint, 'size': -1})
    ref_png = os.path.join(CHAIR, 'Annotations', names[0][:-4] + '.png')
    a = _read_written_masks(str(out / 'short'), names[:4], ref_png)
    b = _read_written_masks(str(tmp_path / 'direct'), names[:4], ref_png)
    assert np.array_equal(a, b)
