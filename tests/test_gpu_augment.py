"""SURVEY 8(f) rank 3 on the device: the deterministic permanent-memory augmentations (csrc/augment.hip) against the host path
(xmem2_amd/augmentations.py = the PIL / torch code paths torchvision dispatches to, frame_selection_utils.py:50-218), pixel by
pixel, and the batched preload against frame-by-frame put_to_permanent_memory."""
import os
import time

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _images():
    from PIL import Image
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (96, 130, 3), dtype=np.uint8)              # every sampling / rounding difference shows
    chair = np.array(Image.open(os.path.join(GOLDEN, 'chair', 'JPEGImages', 'frame_000000.jpg')).convert('RGB'))
    return {'noise 96x130': noise, f'chair {chair.shape[0]}x{chair.shape[1]}': chair}


@pytest.mark.parametrize('subset', ['all', 'best_all', 'best_3_with_symmetrical'])
def test_device_augmentations_match_the_host_path(subset):
    from PIL import Image
    from xmem2_amd import augmentations as A
    for tag, arr in _images().items():
        H, W = arr.shape[:2]
        rng = np.random.default_rng(H)
        msk = (rng.random((2, H, W)) > 0.55).astype(np.float32)
        msk[1] *= (1 - msk[0])
        pil = Image.fromarray(arr)
        host = A.get_determenistic_augmentations((3, H, W), torch.from_numpy(msk), subset=subset)
        specs = A.augmentation_specs((3, H, W), subset)
        dimg, dmsk = A.augment_on_device(torch.from_numpy(arr).cuda(), torch.from_numpy(msk).cuda(), subset=subset)
        torch.cuda.synchronize()
        assert dimg.shape == (len(host), H, W, 3) and len(dmsk) == len(host)
        for i, ((img_aug, mask_aug), (name, kind, _)) in enumerate(zip(host, specs)):
            want = np.array(img_aug(pil).convert('RGB'))
            got = dimg[i].cpu().numpy()
            diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
            wm = mask_aug(torch.from_numpy(msk)).numpy()
            gm = dmsk[i].cpu().numpy()
            if kind in ('brightness', 'posterize', 'gray', 'sharpness'):
                assert diff.max() == 0, f'{tag} / {name}: {int((diff > 0).sum())} values differ (max {int(diff.max())})'
            elif kind == 'blur':
                assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, f'{tag} / {name}: max {int(diff.max())} LSB, {(diff > 0).mean():.2e} of the values'
            else:                                                   # nearest sampling: identical source pixel up to float ties
                frac = float((diff.max(-1) > 0).mean())
                assert frac < 2e-4, f'{tag} / {name}: {frac:.2e} of the pixels sample another source pixel'
            assert gm.shape == wm.shape
            assert float((gm != wm).mean()) < 2e-4, f'{tag} / {name}: mask differs at {float((gm != wm).mean()):.2e} of the pixels'
    assert A.augment_on_device(torch.from_numpy(arr).cuda(), None, subset='original_only') is None
    with pytest.raises(RuntimeError):
        A.augment_on_device(torch.from_numpy(arr), None, subset='best_all')            # host tensor: no CPU path


def test_batched_preload_equals_frame_by_frame(hip_net):
    """put_many_to_permanent_memory (one batch-12 key pass + one batch-12 value pass) against 12 put_to_permanent_memory calls:
    same memory (to the round-off of batched convolution plans), same first prediction; and it is what run_on_video uses."""
    from conftest import base_config
    from xmem2_amd import augmentations as A, ops
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    H, W, K = 128, 176, 2
    fr = synthetic_frames(3, H, W); mk = torch.from_numpy(synthetic_masks(3, K, H, W)).cuda()
    rgb = torch.from_numpy(np.clip((fr[0].transpose(1, 2, 0) * 0.229 + 0.45) * 255, 0, 255).astype(np.uint8)).cuda()
    aug_rgb, aug_msk = A.augment_on_device(rgb, mk[0], subset='best_all')
    images = [rgb] + [aug_rgb[i] for i in range(aug_rgb.shape[0])]
    masks = [mk[0]] + aug_msk
    cores = [InferenceCore(hip_net, base_config()) for _ in range(2)]
    for c in cores:
        c.set_all_labels([1, 2])
    t0 = time.perf_counter()
    for im, m in zip(images, masks):
        cores[0].put_to_permanent_memory(im, m)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    assert cores[1].put_many_to_permanent_memory(images, masks) == 12
    torch.cuda.synchronize(); t2 = time.perf_counter()
    a, b = cores[0].memory.permanent_work_mem, cores[1].memory.permanent_work_mem
    hw = (H // 16) * (W // 16)
    assert a.size == b.size == 12 * hw and a.num_groups == b.num_groups == 1
    for name, x, y in (('key', a.key_rows(), b.key_rows()), ('shrinkage', a.shrinkage_rows(), b.shrinkage_rows()),
                       ('selection', a.selection_rows(), b.selection_rows()), ('value', a.value_rows(0), b.value_rows(0))):
        err = float((x - y).abs().max()) / max(float(x.abs().max()), 1e-9)
        assert err < 2e-4, f'{name}: batched preload differs from the sequential one by {err:.2e} of its scale'
    # the kept fp16 operand rows of the batched preload: one per element, and exactly what the rows kernel derives from the
    # batched store's own (key, shrinkage) - a stale or misaligned row block after put_many_to_permanent_memory would differ
    for st in (a, b):
        r16 = st.rows16()
        assert r16.shape[0] == st.size
        fresh = ops.affinity_rows16(st.key_rows().contiguous(), st.shrinkage_rows().contiguous(), torch.empty_like(r16))
        assert torch.equal(r16.view(torch.int32), fresh.view(torch.int32))
    q = torch.from_numpy(fr[1]).cuda()
    p0, p1 = cores[0].step(q, None, None), cores[1].step(q, None, None)
    assert float((p0 - p1).abs().mean()) < 2e-4 and float((ops.argmax_u8(p0) != ops.argmax_u8(p1)).float().mean()) < 1e-3
    print(f'preload of 12 frames at {H}x{W}: sequential {1e3 * (t1 - t0):.1f} ms, batched {1e3 * (t2 - t1):.1f} ms')
    with pytest.raises(ValueError):
        cores[1].put_many_to_permanent_memory(images[:2], masks[:1])


def test_run_on_video_augmented_preload_device_vs_host(tmp_path, hip_net):
    """run_on_video(augment_images_with_masks=True): the device path (opt-in: augment_on_device=True) and the host path (the default) give
    the same masks; the annotated frame enters the memory 12 times either way."""
    from PIL import Image
    from xmem2_amd.run_on_video import run_on_video
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    imgs, msks = tmp_path / 'JPEGImages', tmp_path / 'Annotations'
    imgs.mkdir(); msks.mkdir()
    t, hw = 5, (96, 128)
    frames = synthetic_frames(t, *hw); masks = synthetic_masks(t, 1, *hw)
    palette = [0, 0, 0, 255, 255, 255] + [0] * (256 * 3 - 6)
    for i in range(t):
        rgb = np.clip((frames[i].transpose(1, 2, 0) * 0.229 + 0.45) * 255, 0, 255).astype(np.uint8)
        Image.fromarray(rgb).save(imgs / f'frame_{i:06d}.png')
        im = Image.fromarray(masks[i, 0].astype(np.uint8), mode='P'); im.putpalette(palette); im.save(msks / f'frame_{i:06d}.png')
    outs, times = {}, {}
    for mode, flag in (('device', True), ('host', False)):
        out = tmp_path / f'out_{mode}'
        t0 = time.perf_counter()
        stats = run_on_video(str(imgs), str(msks), str(out), frames_with_masks=[0], print_progress=False, augment_images_with_masks=True,
                             overwrite_config={'model': None, 'size': -1, 'augment_on_device': flag}, save_overlay=False)
        times[mode] = time.perf_counter() - t0
        assert len(stats) == t
        outs[mode] = [np.array(Image.open(out / 'masks' / f'frame_{i:06d}.png')) for i in range(t)]
    same = np.mean([float((a == b).mean()) for a, b in zip(outs['device'], outs['host'])])
    print(f'augmented preload: device-path masks equal the host-path masks at {same:.5f} of the pixels; run times {times}')
    assert same > 0.999
