"""Padding helpers and the IoU definition (util/tensor_util.py:6-77) - host-side integer logic only."""
import numpy as np


def pad_amounts(h, w, d=16):
    """(lw, uw, lh, uh) exactly as pad_divide_by computes them (util/tensor_util.py:47-59)."""
    new_h = h + d - h % d if h % d > 0 else h
    new_w = w + d - w % d if w % d > 0 else w
    lh, uh = int((new_h - h) / 2), int(new_h - h) - int((new_h - h) / 2)
    lw, uw = int((new_w - w) / 2), int(new_w - w) - int((new_w - w) / 2)
    return int(lw), int(uw), int(lh), int(uh)


def pad_divide_by(in_img, d):
    """Zero-pad the last two dims to a multiple of d (device memory ops only)."""
    import torch
    h, w = in_img.shape[-2:]
    lw, uw, lh, uh = pad_amounts(h, w, d)
    if lw + uw + lh + uh == 0:
        return in_img, (lw, uw, lh, uh)
    out = torch.zeros(tuple(in_img.shape[:-2]) + (h + lh + uh, w + lw + uw), dtype=in_img.dtype, device=in_img.device)
    out[..., lh:lh + h, lw:lw + w].copy_(in_img)
    return out, (lw, uw, lh, uh)


def unpad(img, pad):
    """util/tensor_util.py:63-77."""
    if img.dim() not in (3, 4):
        raise NotImplementedError
    if pad[2] + pad[3] > 0:
        img = img[..., pad[2]:img.shape[-2] - pad[3], :]
    if pad[0] + pad[1] > 0:
        img = img[..., pad[0]:img.shape[-1] - pad[1]]
    return img


def compute_array_iou(seg, gt):
    """Mean per-object IoU of two index masks (util/tensor_util.py:18-44)."""
    seg, gt = np.squeeze(np.asarray(seg)), np.squeeze(np.asarray(gt))

    def iou(a, b):
        inter = float(np.logical_and(a, b).sum())
        union = float(np.logical_or(a, b).sum())
        return (inter + 1e-6) / (union + 1e-6)

    ious = [iou(seg == c, gt == c) for c in np.unique(seg) if c != 0]
    if not ious:
        ious = [iou(seg == 0, gt == 0)]
    return sum(ious) / len(ious)
