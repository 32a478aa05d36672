"""Deterministic synthetic inputs: conditioned weights, frames and masks.

Trained weights (saves/XMem.pth, scripts/download_models.sh:1) cannot be fetched
offline, and PyTorch's default init makes the reference's top-k softmax underflow
to NaN (model/memory_util.py:48-49).  This module produces a
*conditioned* state_dict with the reference's 412 tensor names from a counter
based integer hash, so the same bytes come out on any machine / torch version.
It also generates the seeded synthetic clips SURVEY.md 8(d) describes.

Only integer arithmetic, int->float conversion, adds and multiplies are used, so
the values are bit-reproducible (no libm calls).
"""
import math

import numpy as np

from .arch import state_dict_spec

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on a uint64 array."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return x ^ (x >> np.uint64(31))


def _uniform(n, stream):
    """n floats in [0,1) with 24 random bits each; `stream` selects the sequence."""
    with np.errstate(over='ignore'):
        idx = np.arange(n, dtype=np.uint64)
        h = _mix(idx * np.uint64(0x2545F4914F6CDD1D) + _mix(np.uint64(stream) + np.zeros(1, np.uint64)))
    return ((h >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24)))


def hash_normal(n, stream):
    """Approximately N(0,1): centred Irwin-Hall sum of four uniforms (exactly reproducible)."""
    acc = np.zeros(n, np.float64)
    for j in range(4):
        acc += _uniform(n, stream * 4 + j + 1000003)
    return ((acc - 2.0) * math.sqrt(3.0)).astype(np.float32)


def hash_uniform(n, stream, lo=0.0, hi=1.0):
    return (lo + (hi - lo) * _uniform(n, stream * 4 + 7000001)).astype(np.float32)


def synthetic_state_dict(seed=0, key_dim=64, value_dim=512, hidden_dim=64, as_torch=True, conditioning=None):
    """Conditioned synthetic checkpoint with the reference's tensor names.

    Convolutions: He-normal (as model/resnet.py:132-135) ; BatchNorm: non-trivial
    affine and running statistics so that the folded scale/shift path is exercised;
    the last BN of every residual branch is damped (x0.35) so activations do not
    explode through the 13 bottlenecks; key / shrinkage / selection projections are
    scaled down so that similarities stay in exp()'s non-underflowing range.

    conditioning='multi_object' (the K >= 2 benchmark / parity clips, round 5): with the plain conditioning every object's logit
    map follows the SHARED image features (skip connections, the f16 half of the decoder's fuser), so several objects saturate on
    the same pixels and 45 % of a 3-object frame's pixels are exact or near ties between two objects - a clip on which the
    reference's own argmax is noise.  This variant makes the object-specific path dominate: the value encoder's mask / other-mask
    input channels x8 / x-8, the decoder fuser's f16 half x0.1, the decoder's skip convolutions x0.3 (applied to the SAME hash
    values, so it is as reproducible as the base checkpoint).  Measured on the config-3 clip (480p, 3 objects, consolidation):
    pixels within 0.01 of a tie 45 % -> 1 %, the reference's 8-vs-1-thread IoU 0.9977-0.9993 -> 0.9992-0.9995.
    """
    if conditioning not in (None, 'multi_object'):
        raise ValueError(f'unknown conditioning {conditioning!r}')
    spec = state_dict_spec(key_dim, value_dim, hidden_dim)
    out = {}
    for ti, (name, shape) in enumerate(spec.items()):
        stream = seed * 100003 + ti
        n = int(np.prod(shape)) if len(shape) else 1
        if name.endswith('num_batches_tracked'):
            arr = np.array(0, dtype=np.int64)
        elif name.endswith('running_mean'):
            arr = 0.05 * hash_normal(n, stream)
        elif name.endswith('running_var'):
            arr = hash_uniform(n, stream, 0.8, 1.25)
        elif len(shape) == 1 and name.endswith('.weight'):      # BN gamma
            arr = hash_uniform(n, stream, 0.85, 1.15)
            leaf = name.split('.')[-2]
            is_last_bn = leaf == 'bn3' or (leaf == 'bn2' and 'value_encoder.layer' in name)
            if is_last_bn:
                arr = arr * np.float32(0.35)
        elif name == 'decoder.pred.bias':
            arr = np.full(n, -7.0, np.float32)
        elif len(shape) == 1:                                     # conv / linear / BN bias
            arr = 0.02 * hash_normal(n, stream)
        elif len(shape) == 2:                                     # CBAM MLP
            arr = hash_normal(n, stream) * np.float32(math.sqrt(1.0 / shape[1]))
        else:                                                     # conv weight
            cout, cin, kh, kw = shape
            if name.startswith('key_encoder') or name.startswith('value_encoder.conv1') \
                    or name.startswith('value_encoder.layer'):
                std = math.sqrt(2.0 / (kh * kw * cout))           # resnet.py:133-134
            else:
                std = math.sqrt(1.0 / (kh * kw * cin))
            arr = hash_normal(n, stream) * np.float32(std)
            if name == 'key_proj.key_proj.weight':
                arr = arr * np.float32(5.0)
            elif name == 'key_proj.d_proj.weight':
                arr = arr * np.float32(6.0)
            elif name == 'key_proj.e_proj.weight':
                arr = arr * np.float32(5.0)
            elif name == 'decoder.pred.weight':
                arr = arr * np.float32(18.0)
        arr = np.asarray(arr).reshape(shape)
        if conditioning == 'multi_object':
            if name == 'value_encoder.conv1.weight' and shape[1] >= 5:
                arr = arr.copy()
                arr[:, 3] *= np.float32(8.0)
                arr[:, 4] *= np.float32(-8.0)
            elif name in ('decoder.fuser.block1.conv1.weight', 'decoder.fuser.block1.downsample.weight'):
                arr = arr.copy()
                arr[:, :1024] *= np.float32(0.1)
            elif name in ('decoder.up_16_8.skip_conv.weight', 'decoder.up_8_4.skip_conv.weight'):
                arr = arr * np.float32(0.3)
        out[name] = arr
    if as_torch:
        import torch
        return {k: (torch.from_numpy(np.ascontiguousarray(v)) if v.ndim else torch.tensor(int(v), dtype=torch.int64))
                for k, v in out.items()}
    return out


# ----------------------------------------------------------------------------------------------
# synthetic clips
# ----------------------------------------------------------------------------------------------

def _smooth_field(h, w, stream, passes=3, radius=8):
    """Low-pass Gaussian-like field: white noise box-filtered `passes` times, unit variance."""
    f = hash_normal(h * w, stream).reshape(h, w).astype(np.float64)
    k = 2 * radius + 1
    for _ in range(passes):
        for axis in (0, 1):
            c = np.cumsum(np.concatenate([np.zeros_like(np.take(f, [0], axis)), f], axis), axis)
            n = f.shape[axis]
            idx_hi = np.minimum(np.arange(n) + radius + 1, n)
            idx_lo = np.maximum(np.arange(n) - radius, 0)
            f = (np.take(c, idx_hi, axis) - np.take(c, idx_lo, axis)) / k
    f = f - f.mean()
    return (f / (f.std() + 1e-12)).astype(np.float32)


def synthetic_frames(num_frames, height=480, width=854, seed=1234):
    """[T,3,H,W] float32 frames, ImageNet-normalised scale (~N(0,1)).

    A smooth random texture translating 1 px / frame plus 0.1*N(0,1) pixel noise.
    """
    canvas = np.stack([_smooth_field(height, width + num_frames, seed * 10 + c) for c in range(3)])
    frames = np.empty((num_frames, 3, height, width), np.float32)
    for t in range(num_frames):
        noise = hash_normal(3 * height * width, seed * 1000 + 17 + t).reshape(3, height, width)
        frames[t] = canvas[:, :, t:t + width] + np.float32(0.1) * noise
    return frames


def synthetic_masks(num_frames, num_objects=1, height=480, width=854):
    """[T,K,H,W] float32 one-hot object masks: object 1 = ellipse drifting 1 px/frame,
    further objects = disjoint rectangles."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    masks = np.zeros((num_frames, num_objects, height, width), np.float32)
    ay, ax = height * 100.0 / 480.0, width * 150.0 / 854.0
    for t in range(num_frames):
        cy, cx = height / 2.0, width / 2.0 - t
        masks[t, 0] = (((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1.0)
        for k in range(1, num_objects):
            y0 = int(height * 0.05) + (k - 1) * int(height * 0.12)
            x0 = int(width * 0.04)
            rect = (yy >= y0) & (yy < y0 + int(height * 0.09)) & (xx >= x0) & (xx < x0 + int(width * 0.12))
            masks[t, k] = rect & (masks[t, 0] == 0)
    return masks
