"""Annotation-candidate selection with the reference's surface.

    inference/frame_selection/frame_selection.py:99-244      select_next_candidates
    inference/frame_selection/frame_selection_utils.py:11-44 extract_keys
    inference/frame_selection/frame_selection.py:18-27       first_frame_only / uniformly_selected_frames

The reference scores every (candidate, chosen) pair by materialising two HW x HW similarity matrices on the host
loop, and re-scores all pairs in every greedy iteration (O(k^2 N) pairs).  Here the per-frame operands live on the
GPU ([F, HW, 2 C_k] fp32, 0.83 MB per 480p frame), one launch of `xmem_cycle_dissimilarity` scores all F frames
against one chosen frame without materialising anything, and the running minimum over chosen frames is kept
between iterations, so k picks cost |previous| + k - 1 launches.  The result (greedy argmax of the minimum) is the
same function of the same scores; only the fp32 summation order inside a score differs.
"""
from typing import List

import numpy as np
import torch

from . import ops


def first_frame_only(*args, **kwargs) -> List[int]:
    return [0]


def uniformly_selected_frames(dataloader, *args, how_many_frames=10, **kwargs) -> List[int]:
    return np.linspace(0, len(dataloader) - 1, how_many_frames).astype(int).tolist()


def _rgb_of(item):
    if torch.is_tensor(item):
        return item
    if isinstance(item, dict):
        return item['rgb']
    u8 = getattr(item, 'rgb_u8', None)           # this repo's reader: decoded uint8 frame, normalised on the device
    return u8 if u8 is not None else item.rgb


def extract_keys(dataloder, processor, print_progress=False, flatten=True, **kwargs):
    """frame_selection_utils.py:11-44: run the key encoder over every frame.  Returns
    (frame_keys, shrinkages, selections, device, num_frames, key_sum) with the per-frame tensors on the CPU, as the
    reference does; pass ``keep_on_device=True`` to skip the host copies (select_next_candidates accepts both)."""
    keep = bool(kwargs.get('keep_on_device', False))
    frame_keys, shrinkages, selections = [], [], []
    device, key_sum, ti = None, None, -1
    dev = processor.network.device
    for ti, data in enumerate(dataloder):
        rgb = _rgb_of(data)
        if rgb.dim() == 4:                       # a DataLoader adds a batch dimension
            rgb = rgb[0]
        key, shrinkage, selection = processor.encode_frame_key(rgb.to(dev, non_blocking=True))
        if key_sum is None:
            device = key.device
            key_sum = torch.zeros(key.shape, device=device, dtype=torch.float64)
        key_sum += key.to(torch.float64)
        if flatten:
            key, shrinkage, selection = (t.flatten(start_dim=2) for t in (key, shrinkage, selection))
        if not keep:
            key, shrinkage, selection = key.cpu(), shrinkage.cpu(), selection.cpu()
        frame_keys.append(key); shrinkages.append(shrinkage); selections.append(selection)
    return frame_keys, shrinkages, selections, device, ti + 1, key_sum


def _rows(x, device):
    """F x C x h x w (any memory format, any device) -> contiguous F x (h*w) x C rows on `device`."""
    x = x.to(device=device, dtype=torch.float32)
    nhwc = x.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():                 # true NCHW input: one transpose kernel
        nhwc = ops.nchw_to_nhwc(x)
    return nhwc.reshape(x.shape[0], -1, x.shape[1])


class SelectorState:
    """Device-resident operands of one video: built once, scored many times."""

    def __init__(self, keys, shrinkages, selections, masks, alpha, epsilon, device):
        if keys.dim() == 5:
            keys, shrinkages, selections = keys.squeeze(1), shrinkages.squeeze(1), selections.squeeze(1)
        F_, ck, h, w = keys.shape
        self.n, self.h, self.w, self.ck = F_, h, w, ck
        key_rows = _rows(keys, device)
        sel_rows = _rows(selections, device)
        self.shr = shrinkages.to(device=device, dtype=torch.float32).reshape(F_, h * w).contiguous()
        self.Mexp = torch.empty((F_, h * w, 2 * ck), dtype=torch.float32, device=device)
        self.Qexp = torch.empty_like(self.Mexp)
        self.bsq = torch.empty((F_, h * w), dtype=torch.float32, device=device)
        self.presence = torch.zeros((F_,), dtype=torch.int32, device=device)
        self.mask_numel = []
        for i, mask in enumerate(masks):
            m = mask if mask.dim() == 3 else mask.unsqueeze(0)
            m = m.to(device=device, dtype=torch.float32).contiguous()
            self.mask_numel.append(m.shape[1] * m.shape[2])
            ops.selector_prepare(key_rows[i], sel_rows[i], m, h, w, alpha, epsilon,
                                 self.Mexp[i], self.Qexp[i], self.bsq[i], self.presence[i:i + 1])

    def scores_against(self, chosen, valid_dev=None):
        """float32 [F] on the device: cycle dissimilarity of every frame vs frame `chosen`."""
        return ops.cycle_dissimilarity(self.Mexp, self.Qexp, self.bsq, self.shr, chosen, valid_dev).to(torch.float32)


def select_next_candidates(keys: torch.Tensor, shrinkages, selections, masks: List[torch.Tensor], num_next_candidates: int,
                           previously_chosen_candidates: List[int] = (0,), print_progress=False, alpha=0.5,
                           min_mask_presence_percent=0.25, device: torch.device = 'cuda:0', progress_callback=None,
                           only_new_candidates=True, epsilon=0.5):
    """frame_selection.py:99-244, same arguments and return value.

    keys F x C_k x h x w (as ``torch.cat(frame_keys)`` from ``extract_keys(flatten=False)``), shrinkages F x 1 x h x w,
    selections F x C_k x h x w, masks: one C x H x W (or H x W) tensor per frame."""
    assert len(keys) == len(masks)
    assert len(keys) > 0
    assert num_next_candidates > 0
    assert len(previously_chosen_candidates) > 0
    assert 0.0 <= alpha <= 1.0
    assert min_mask_presence_percent >= 0
    assert len(previously_chosen_candidates) < len(keys)

    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('select_next_candidates: xmem2_amd has no CPU path (device must be a HIP device)')
    with torch.no_grad(), torch.cuda.device(device):
        state = SelectorState(keys, shrinkages, selections, masks, alpha, epsilon, device)
        n = state.n
        # mask presence test (frame_selection.py:161-176): the percentage is formed in fp32 as in the reference
        counts = state.presence.cpu()
        numel = torch.tensor(state.mask_numel, dtype=torch.int64)
        percent = counts.to(torch.int64) / numel * 100
        valid = (percent >= min_mask_presence_percent).numpy().copy()
        for i in previously_chosen_candidates:
            valid[i] = True
        print(f"Frames with invalid (empty or too small) masks: {int((~valid).sum())} / {len(masks)}")
        valid_dev = torch.from_numpy(valid.astype(np.uint8)).to(device)

        chosen = list(previously_chosen_candidates)
        running = None
        trace = []
        for m in chosen:
            d = state.scores_against(m, valid_dev)
            running = d if running is None else torch.minimum(running, d)
        for i in range(num_next_candidates):
            scores = running.cpu()               # invalid frames are already 0 (frame_selection.py:201-203)
            trace.append(scores.numpy().copy())
            new = int(torch.argmax(scores))      # host argmax on the fp32 scores: same tie rule as the reference
            chosen.append(new)
            if i + 1 < num_next_candidates:
                running = torch.minimum(running, state.scores_against(new, valid_dev))
            if progress_callback is not None:
                progress_callback.emit(i + 1)
        select_next_candidates.last_scores = trace   # per-iteration candidate scores (diagnostics / tests)
        if only_new_candidates:
            chosen = chosen[len(previously_chosen_candidates):]
        return chosen
