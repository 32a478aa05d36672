"""Video harness with the reference's ``run_on_video`` surface (inference/run_on_video.py:31-282).

Frame decoding, mask reading and PNG writing are host I/O outside the timed region of the reference
(run_on_video.py:106-113) and outside this tier's hot path; this module supplies a minimal reader/writer
(PIL only - torchvision / cv2 are not available) so that the drop-in call works end to end:

    stats = run_on_video(imgs_in_path, masks_in_path, masks_out_path, frames_with_masks=[0, 10])

What is reproduced from the reference: config merge and the derived ``enable_long_term_count_usage``
(:190-196), preload of all annotated frames into permanent memory before the loop (:59-66), the per-frame
``step`` call and its flags (:98-108), resize + argmax post-processing (:165-173, on the GPU), the stats rows
(:115-123) and the output layout ``<out>/masks/<frame>.png`` (+ ``overlay/<frame>.jpg``).
Not reproduced: mp4 extraction (needs cv2).  `augment_images_with_masks` uses xmem2_amd/augmentations.py (unpinned restatement).
"""
import collections
import os
import queue
import threading
from dataclasses import dataclass
from time import perf_counter
from typing import Iterable, Optional
from warnings import warn

import numpy as np
import torch

from . import ops
from .configuration import VIDEO_INFERENCE_CONFIG
from .inference_core import InferenceCore
from .mask_mapper import MaskMapper
from .network import XMem
from .tensor_util import compute_array_iou

IM_MEAN = np.array([0.485, 0.456, 0.406], np.float32)     # dataset/range_transform.py:5-8
IM_STD = np.array([0.229, 0.224, 0.225], np.float32)


@dataclass
class Sample:
    """inference/data/video_reader.py:20-28.  `rgb_u8` is the decoded (and, if asked, resized) H x W x 3 uint8 frame;
    `rgb` - the reference's normalised 3 x H x W float tensor - is derived from it on first use (the harness itself
    hands `rgb_u8` to the device, where ToTensor + Normalize + padding are one kernel)."""
    rgb_u8: torch.Tensor
    raw_image_pil: object
    frame: str
    save: bool
    shape: tuple
    need_resize: bool
    mask: Optional[np.ndarray] = None
    _rgb: Optional[torch.Tensor] = None

    @property
    def rgb(self):
        if self._rgb is None:
            arr = (self.rgb_u8.numpy().astype(np.float32) / 255.0 - IM_MEAN) / IM_STD
            self._rgb = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        return self._rgb


class VideoReader:
    """Directory-of-frames reader (inference/data/video_reader.py:31-118 without cv2 / torchvision)."""

    def __init__(self, vid_name, video_path, mask_dir, size=-1, use_all_masks=False):
        from PIL import Image
        self._Image = Image
        if os.path.isfile(video_path):
            raise NotImplementedError('video files need cv2 for frame extraction; pass a directory of frames')
        self.vid_name, self.image_dir, self.mask_dir = vid_name, video_path, mask_dir
        self.size, self.use_all_masks = size, use_all_masks
        self.frames = sorted(os.listdir(self.image_dir))
        masks = sorted(os.listdir(mask_dir))
        self.first_gt_path = os.path.join(mask_dir, masks[0])
        self.reference_mask = Image.open(self.first_gt_path).convert('P')
        self.reference_mask.load()                   # decoded once here: the writer threads only read it afterwards

    def __len__(self):
        return len(self.frames)

    def _target_hw(self, h, w):
        if self.size < 0:
            return h, w
        s = self.size / min(h, w)                  # Resize(size): shorter side -> size
        return (self.size, int(w * s)) if h <= w else (int(h * s), self.size)

    def frame_u8(self, img):
        """PIL RGB image -> the working-size H x W x 3 uint8 tensor (the Resize of im_transform, video_reader.py:61-65;
        ToTensor + Normalize happen on the device)."""
        Image = self._Image
        shape = (img.size[1], img.size[0])
        th, tw = self._target_hw(*shape)
        work = img if (th, tw) == shape else img.resize((tw, th), Image.BILINEAR)
        return torch.from_numpy(np.array(work, dtype=np.uint8))                   # owns its memory

    def __getitem__(self, idx) -> Sample:
        Image = self._Image
        name = self.frames[idx]
        img = Image.open(os.path.join(self.image_dir, name)).convert('RGB')
        shape = (img.size[1], img.size[0])
        rgb_u8 = self.frame_u8(img)
        gt_path = os.path.join(self.mask_dir, name[:-4] + '.png')
        if not os.path.exists(gt_path):
            gt_path = os.path.join(self.mask_dir, name[:-4] + '.PNG')
        mask = None
        if (self.use_all_masks or gt_path == self.first_gt_path) and os.path.exists(gt_path):
            mask = np.array(Image.open(gt_path).convert('P'), dtype=np.uint8)
        return Sample(rgb_u8=rgb_u8, raw_image_pil=img, frame=name, save=True, shape=shape,
                      need_resize=not (self.size < 0), mask=mask)

    def resize_mask(self, onehot):
        """nearest resize of a [K,H,W] one-hot mask to the working size (video_reader.py:148-153)."""
        h, w = onehot.shape[-2:]
        m = min(h, w)
        th, tw = int(h / m * self.size), int(w / m * self.size)
        if (th, tw) == (h, w):
            return onehot
        ys = (np.arange(th) * (h / th)).astype(np.int64)
        xs = (np.arange(tw) * (w / tw)).astype(np.int64)
        return onehot[:, torch.from_numpy(ys)][:, :, torch.from_numpy(xs)]

    def map_the_colors_back(self, pred_mask):
        Image = self._Image
        return pred_mask.quantize(palette=self.reference_mask, dither=Image.Dither.NONE).convert('RGB')


class _AsyncSaver:
    """Background threads that colour-map and write masks / overlays (the reference uses two writer processes,
    util/image_saver.py:240-345).  PIL releases the GIL inside quantize / PNG / JPEG encoding, so threads scale."""

    def __init__(self, out_dir, vid_name, max_queue=200, workers=4):
        self.root = os.path.join(out_dir, vid_name)
        self.q = queue.Queue(max_queue)
        self.err = None
        self.threads = [threading.Thread(target=self._run, daemon=True) for _ in range(max(1, workers))]
        for t in self.threads:
            t.start()

    def _run(self):
        while True:
            job = self.q.get()
            if job is None:
                return
            try:
                for img, sub, name in job():
                    d = os.path.join(self.root, sub)
                    os.makedirs(d, exist_ok=True)
                    img.save(os.path.join(d, name), **({'compress_level': 1} if name.endswith('.png') else {}))
            except Exception as e:                                   # surfaced by close()
                self.err = e

    def submit(self, job):
        """job() -> iterable of (PIL image, sub-directory, file name); runs on a writer thread."""
        self.q.put(job)

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err is not None:
            raise self.err


class FramePrefetcher:
    """Decode ahead on worker threads (the reference's DataLoader worker, inference/run_on_video.py:80-92): frames are
    requested in order and arrive as Samples whose uint8 frame is already in pinned host memory."""

    def __init__(self, reader, depth=16, workers=8):
        from concurrent.futures import ThreadPoolExecutor
        self.reader, self.depth = reader, max(1, depth)
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix='xmem-decode')
        self.futures = collections.deque()
        self.next_submit = 0

    def _load(self, idx):
        smp = self.reader[idx]
        smp.rgb_u8 = smp.rgb_u8.pin_memory()
        return smp

    def _top_up(self):
        while self.next_submit < len(self.reader) and len(self.futures) < self.depth:
            self.futures.append(self.pool.submit(self._load, self.next_submit))
            self.next_submit += 1

    def get(self, n):
        """The next n frames, in order (blocks on the decoder only if it fell behind)."""
        self._top_up()
        out = []
        for _ in range(n):
            out.append(self.futures.popleft().result())
            self._top_up()
        return out

    def close(self):
        self.pool.shutdown(wait=False, cancel_futures=True)


def _overlay(img, mask_rgb, alpha=0.5):
    from PIL import Image
    m = np.asarray(mask_rgb.resize(img.size, Image.NEAREST) if mask_rgb.size != img.size else mask_rgb)
    fg = m.sum(-1) > 0
    a = np.full(m.shape[:2], 255, np.uint8)
    a[fg] = int(alpha * 255)
    return Image.composite(img, Image.fromarray(m), Image.fromarray(a, mode='L'))


class AsyncMaskFetcher:
    """Device->host delivery of the uint8 index masks one frame behind the GPU: the copy of frame t goes to pinned
    memory on the compute stream and is waited for only after frame t+1 has been enqueued, so the GPU never idles
    on the host round trip (the reference blocks on `.cpu()` every frame, run_on_video.py:170-172)."""

    def __init__(self, depth=3):
        self.depth = depth
        self.slots = [None] * depth
        self.pending = []          # (tag, host tensor, event)
        self._n = 0

    def submit(self, tag, mask_gpu):
        i = self._n % self.depth
        self._n += 1
        buf = self.slots[i]
        if buf is None or buf.shape != mask_gpu.shape:
            buf = torch.empty(mask_gpu.shape, dtype=torch.uint8, pin_memory=True)
            self.slots[i] = buf
        buf.copy_(mask_gpu, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((tag, buf, ev))
        ready = []
        while len(self.pending) >= self.depth:      # keep at most depth-1 frames in flight
            ready.append(self._pop())
        return ready

    def _pop(self):
        tag, buf, ev = self.pending.pop(0)
        ev.synchronize()
        return tag, buf.numpy().copy()

    def drain(self):
        out = []
        while self.pending:
            out.append(self._pop())
        return out


def _post_process_gpu(sample, prob):
    """run_on_video.py:165-170 on the device: resize to the original shape if needed, argmax over classes."""
    if sample.need_resize and tuple(prob.shape[-2:]) != tuple(sample.shape):
        prob = ops.resize_bilinear(prob, sample.shape)
    return ops.argmax_u8(prob)


def _post_process(sample, prob):
    """run_on_video.py:165-173: resize to the original shape if needed, argmax over classes, uint8 on the host."""
    if sample.need_resize and tuple(prob.shape[-2:]) != tuple(sample.shape):
        prob = ops.resize_bilinear(prob, sample.shape)
    return ops.argmax_u8(prob).cpu().numpy()


def _load_main_objects(imgs_in_path, masks_in_path, config, device):
    model_path = config['model']
    network = XMem(config, model_path, pretrained_key_encoder=False, pretrained_value_encoder=False).to(device).eval()
    if model_path is None:
        warn('No model weights were loaded, as config["model"] was not specified.')
    vid_reader = VideoReader('', imgs_in_path, masks_in_path, size=config['size'], use_all_masks=True)
    vid_length = len(vid_reader)
    config['enable_long_term_count_usage'] = (                       # run_on_video.py:190-196
        config['enable_long_term'] and
        (vid_length / (config['max_mid_term_frames'] - config['min_mid_term_frames']) * config['num_prototypes'])
        >= config['max_long_term_elements'])
    return MaskMapper(), InferenceCore(network, config=config), vid_reader


def _inference_on_video(frames_with_masks, imgs_in_path, masks_in_path, masks_out_path, original_memory_mechanism=False,
                        compute_iou=False, manually_curated_masks=False, print_progress=True,
                        augment_images_with_masks=False, overwrite_config: dict = None, save_overlay=True,
                        object_color_if_single_object=(255, 255, 255), print_fps=False, image_saving_max_queue_size=200):
    import pandas as pd
    from PIL import Image
    if not torch.cuda.is_available():
        raise RuntimeError('xmem2_amd.run_on_video needs an MI355X (HIP) device - there is no CPU path')
    device = torch.device('cuda', torch.cuda.current_device())
    torch.autograd.set_grad_enabled(False)
    frames_with_masks = set(frames_with_masks)
    config = VIDEO_INFERENCE_CONFIG.copy()
    overwrite_config = {} if overwrite_config is None else overwrite_config
    overwrite_config['masks_out_path'] = masks_out_path
    config.update(overwrite_config)
    mapper, processor, vid_reader = _load_main_objects(imgs_in_path, masks_in_path, config, device)
    vid_length = len(vid_reader)

    to_permanent = [0] if original_memory_mechanism else sorted(frames_with_masks)
    loaded, preload_time = False, 0.0
    for j in to_permanent:                                           # _preload_permanent_memory, :201-244
        sample = vid_reader[j]
        if sample.mask is None:
            raise FileNotFoundError(f"Couldn't find mask {j}! Check that the filename is the same as for frame {j}.")
        msk, _ = mapper.convert_mask(sample.mask, exhaustive=True)
        if min(msk.shape) == 0:
            warn(f'Skipping adding frame {j} to permanent memory, as the mask is empty')
            continue
        if sample.need_resize:
            msk = vid_reader.resize_mask(msk)
        processor.set_all_labels(list(mapper.remappings.values()))
        a = perf_counter()
        # Opt-in (config['augment_on_device'] = True; default False): the device path is pinned to this repo's host restatement only
        # to float ties (blur within 1 LSB, < 2e-4 of the pixels of the nearest-sampling transforms, batched conv plans ~2e-4), and the
        # default output is the fp32 parity contract - a caller chooses the 20x faster preload knowingly.
        on_device = augment_images_with_masks and not sample.need_resize and config.get('augment_on_device', False)
        if on_device:
            # the annotated frame and its 11 'best_all' augmentations: made on the device in one launch, preloaded through ONE
            # batched key + value pass (the reference runs 12 sequential passes over host-side PIL transforms, :231-242).
            # With a working-size resize the reference augments BEFORE resizing: that case keeps the host path below.
            from .augmentations import augment_on_device
            rgb_dev, msk_dev = sample.rgb_u8.to(device), msk.to(device)
            aug_rgb, aug_msk = augment_on_device(rgb_dev, msk_dev, subset='best_all')
            processor.put_many_to_permanent_memory([rgb_dev] + [aug_rgb[i] for i in range(aug_rgb.shape[0])], [msk_dev] + aug_msk)
        else:
            processor.put_to_permanent_memory(sample.rgb_u8.to(device), msk.to(device))
        torch.cuda.synchronize()
        preload_time += perf_counter() - a
        loaded = True
        if augment_images_with_masks and not on_device:              # run_on_video.py:231-242, subset 'best_all' (host path)
            from .augmentations import get_determenistic_augmentations
            h, w = sample.rgb_u8.shape[:2]
            for img_aug, mask_aug in get_determenistic_augmentations((3, h, w), msk, subset='best_all'):
                rgb_aug = vid_reader.frame_u8(img_aug(sample.raw_image_pil))
                processor.put_to_permanent_memory(rgb_aug.to(device), mask_aug(msk).to(device))
    if not loaded:
        raise ValueError('No valid masks provided!')

    stats, total_time = [], 0.0
    saver = _AsyncSaver(config['masks_out_path'], vid_reader.vid_name, image_saving_max_queue_size) if config['save_masks'] else None
    fetcher = AsyncMaskFetcher()

    def finish(tag, out_mask):                                       # host side of a frame whose mask has arrived
        sample, had_mask = tag
        stat = {'frame': sample.frame, 'mask_provided': had_mask}
        if compute_iou:
            gt = sample.mask
            stat['iou'] = float(compute_array_iou(out_mask, gt)) if (gt is not None and not had_mask) else -1
        stats.append(stat)
        if saver is not None:
            ids = mapper.remap_index_mask(out_mask)                  # label LUT as of this frame (cheap); the rest is off-thread

            def job(ids=ids, sample=sample):
                out_img = vid_reader.map_the_colors_back(Image.fromarray(ids))
                yield out_img, 'masks', sample.frame[:-4] + '.png'
                if save_overlay:
                    yield _overlay(sample.raw_image_pil, out_img), 'overlay', sample.frame[:-4] + '.jpg'
            saver.submit(job)

    key_batch = max(1, int(config.get('key_batch', 4)))                # frames per batched key-encoder hint
    decoder = FramePrefetcher(vid_reader, depth=4 * key_batch, workers=int(config.get('decode_workers', 8)))
    pending, next_idx = collections.deque(), 0                       # decoded + hinted frames, in frame order

    def refill():
        nonlocal next_idx
        remaining = vid_length - next_idx
        if remaining <= 0:
            return
        n = key_batch if remaining >= key_batch else 1               # the tail goes frame by frame (no new graph shapes)
        samples = decoder.get(n)
        devs = processor.prefetch_keys([smp.rgb_u8 for smp in samples])   # uint8 H2D + normalise + key encoder, side stream
        pending.extend(zip(samples, devs))
        next_idx += n

    loop_t0 = perf_counter()
    try:
        for ti in range(vid_length):
            if len(pending) < key_batch:                             # frames come from the decode threads (DataLoader role:
                refill()                                             # outside the timed region, run_on_video.py:80-113)
            sample, rgb = pending.popleft()
            msk = labels = None
            if ti in frames_with_masks and sample.mask is not None:
                msk, labels = mapper.convert_mask(sample.mask, exhaustive=True)
                if sample.need_resize:
                    msk = vid_reader.resize_mask(msk)
                msk = msk.to(device)
                processor.set_all_labels(list(mapper.remappings.values()))
            skip_add = (ti == 0) if original_memory_mechanism else (msk is not None)
            a = perf_counter()
            prob = processor.step(rgb, msk, labels, end=(ti == vid_length - 1),
                                  manually_curated_masks=manually_curated_masks, do_not_add_mask_to_memory=skip_add)
            done = fetcher.submit((sample, msk is not None), _post_process_gpu(sample, prob))
            total_time += perf_counter() - a
            for tag, out_mask in done:
                finish(tag, out_mask)
        a = perf_counter()
        done = fetcher.drain()
        total_time += perf_counter() - a
        for tag, out_mask in done:
            finish(tag, out_mask)
    finally:
        decoder.close()
        loop_wall = perf_counter() - loop_t0
        if saver is not None:
            saver.close()
        total_wall = perf_counter() - loop_t0
    if print_fps:
        print(f'TOTAL PRELOADING TIME: {preload_time:.4f}s')
        print(f'TOTAL PROCESSING TIME: {total_time:.4f}s')
        print(f'TOTAL PROCESSING FPS: {vid_length / total_time:.4f}')
        print(f'TOTAL FPS (excluding image saving): {vid_length / (preload_time + total_time):.4f}')
        print(f'WALL-CLOCK FPS of the frame loop incl. decode: {vid_length / loop_wall:.4f}; incl. writing every mask: '
              f'{vid_length / total_wall:.4f}')
    return pd.DataFrame(stats)


def run_on_video(imgs_in_path, masks_in_path, masks_out_path, frames_with_masks: Iterable[int] = (0,),
                 compute_iou=False, print_progress=True, **kwargs):
    """Same signature / return as inference/run_on_video.py:247-282: per-frame stats DataFrame
    (frame, mask_provided[, iou]); predicted masks are written under ``masks_out_path/masks``."""
    return _inference_on_video(imgs_in_path=imgs_in_path, masks_in_path=masks_in_path, masks_out_path=masks_out_path,
                               frames_with_masks=frames_with_masks, compute_iou=compute_iou,
                               print_progress=print_progress, **kwargs)


def _pil_to_tensor01(pic):
    """What torchvision's ToTensor yields for the PNG modes the harness writes/reads (run_on_video.py:334,362):
    uint8 planes scaled by 1/255, C x H x W; palette images contribute their raw INDEX plane (so object id 1 becomes
    1/255 - the reference feeds exactly that to the selector)."""
    arr = np.array(pic, copy=True)
    if pic.mode == '1':
        arr = arr.astype(np.uint8) * 255
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)


def select_k_next_best_annotation_candidates(imgs_in_path, masks_in_path, masks_out_path=None, k: int = 5,
                                             print_progress=True, previously_chosen_candidates=[0],
                                             use_previously_predicted_masks=True, alpha=0.5,
                                             min_mask_presence_percent=0.25, **kwargs):
    """inference/run_on_video.py:285-370, same arguments and return (list of new frame indices)."""
    import tempfile
    from pathlib import Path
    from PIL import Image
    from .frame_selection import extract_keys, select_next_candidates

    if not torch.cuda.is_available():
        raise RuntimeError('xmem2_amd needs an MI355X (HIP) device - there is no CPU path')
    device = torch.device('cuda', torch.cuda.current_device())
    config = dict(VIDEO_INFERENCE_CONFIG)
    config.update(kwargs.get('overwrite_config') or {})     # the reference extracts keys with the default config
    _, processor, vid_reader = _load_main_objects(imgs_in_path, masks_in_path, config, device)
    frame_keys, shrinkages, selections, *_ = extract_keys(vid_reader, processor, print_progress=print_progress,
                                                          flatten=False, keep_on_device=True)
    tmp = None
    p_masks_out = Path(masks_out_path) if masks_out_path is not None else None
    if use_previously_predicted_masks:
        assert masks_out_path is not None, \
            'When `use_existing_masks=True`, you need to put the path to previously predicted masks in `masks_out_path`'
    else:
        if p_masks_out is None:
            tmp = tempfile.TemporaryDirectory()
            p_masks_out = Path(tmp.name)
        run_on_video(imgs_in_path=imgs_in_path, masks_in_path=masks_in_path, masks_out_path=p_masks_out,
                     frames_with_masks=previously_chosen_candidates, compute_iou=False, print_progress=print_progress,
                     **kwargs)
    try:
        masks = [_pil_to_tensor01(Image.open(p)) for p in sorted((p_masks_out / 'masks').iterdir())]
    except Exception:
        warn('Loading previously predicting masks failed for `select_k_next_best_annotation_candidates`.')
        raise
    if len(masks) != len(frame_keys):
        raise FileNotFoundError(f'Not enough masks ({len(masks)}) for {len(frame_keys)} frames provided when using '
                                f'`use_previously_predicted_masks=True`!')
    chosen = select_next_candidates(torch.cat(frame_keys), shrinkages=torch.cat(shrinkages), selections=torch.cat(selections),
                                    masks=masks, num_next_candidates=k,
                                    previously_chosen_candidates=previously_chosen_candidates, print_progress=print_progress,
                                    alpha=alpha, only_new_candidates=True,
                                    min_mask_presence_percent=min_mask_presence_percent, device=device)
    if tmp is not None:
        tmp.cleanup()
    return chosen
