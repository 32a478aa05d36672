"""Deterministic permanent-memory augmentations.

The reference multiplies every annotated frame by a fixed list of (image transform, mask transform) pairs before
putting it into the permanent memory (`inference/frame_selection/frame_selection_utils.py:49-218`, used by
`inference/run_on_video.py:231-242` with subset 'best_all').  Its transforms are torchvision objects applied to the
decoded PIL image (image side) and to the K x H x W float mask tensor (mask side).  torchvision is not available in the
build environment, so this module restates exactly those code paths - PIL's ImageEnhance / ImageOps / Image.transform for
the image side, the affine-grid + nearest `grid_sample` of torchvision's tensor backend for the mask side - and is
therefore NOT pinned against a run of the reference (host-side, load-time work: a dozen frames per annotation).

Every entry is a pair of callables with a `.name`, as in the reference: `img_aug(PIL.Image) -> PIL.Image`,
`mask_aug(Tensor K x H x W) -> Tensor`.
"""
import math

import torch
import torch.nn.functional as F


def _inverse_affine_matrix(center, angle, translate, scale, shear):
    """torchvision.transforms.functional._get_inverse_affine_matrix (inverted=True): output pixel -> input pixel."""
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def _shear2(shear):
    return [float(shear), 0.0] if isinstance(shear, (int, float)) else [float(s) for s in shear]


def affine_pil(img, angle=0.0, translate=(0, 0), scale=1.0, shear=0.0):
    """F.affine on a PIL image: nearest resampling, zero fill, centre = image centre (functional.py affine, PIL branch)."""
    from PIL import Image
    w, h = img.size
    m = _inverse_affine_matrix([w * 0.5, h * 0.5], angle, [float(t) for t in translate], scale, _shear2(shear))
    return img.transform((w, h), Image.AFFINE, m, resample=Image.NEAREST, fillcolor=0 if img.mode in ('L', 'P', '1') else (0,) * len(img.getbands()))


def affine_tensor(x, angle=0.0, translate=(0, 0), scale=1.0, shear=0.0):
    """F.affine on a `... x H x W` tensor: the tensor backend's centred affine grid + grid_sample(nearest, zeros,
    align_corners=False) (functional_tensor.py _gen_affine_grid / _apply_grid_transform)."""
    h, w = x.shape[-2:]
    m = _inverse_affine_matrix([0.0, 0.0], angle, [float(t) for t in translate], scale, _shear2(shear))
    theta = torch.tensor(m, dtype=torch.float32).reshape(1, 2, 3)
    d = 0.5
    base = torch.empty(1, h, w, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w))
    base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).unsqueeze_(-1))
    base[..., 2].fill_(1)
    rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
    grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
    lead = x.shape[:-2]
    img = x.reshape(1, -1, h, w).to(dtype=torch.float32, device='cpu')
    out = F.grid_sample(img, grid, mode='nearest', padding_mode='zeros', align_corners=False)
    return out.reshape(*lead, h, w).to(dtype=x.dtype if x.is_floating_point() else torch.float32, device=x.device)


def gaussian_blur_pil(img, kernel_size=7):
    """FT.gaussian_blur on a PIL image: pil_to_tensor -> float conv with a reflect-padded separable Gaussian
    (sigma = 0.15 k + 0.35) -> round -> uint8 -> PIL (functional.py gaussian_blur, functional_tensor.py gaussian_blur)."""
    import numpy as np
    from PIL import Image
    k = int(kernel_size)
    sigma = k * 0.15 + 0.35
    t = torch.from_numpy(np.array(img, dtype=np.uint8, copy=True))
    t = t[None] if t.dim() == 2 else t.permute(2, 0, 1)
    half = (k - 1) * 0.5
    xs = torch.linspace(-half, half, steps=k)
    pdf = torch.exp(-0.5 * (xs / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :])
    c = t.shape[0]
    x = F.pad(t[None].to(torch.float32), [k // 2] * 4, mode='reflect')
    y = F.conv2d(x, k2.expand(c, 1, k, k), groups=c)[0]
    y = torch.round(y).to(torch.uint8)
    arr = y[0].numpy() if c == 1 else y.permute(1, 2, 0).contiguous().numpy()
    return Image.fromarray(arr, mode=img.mode)


class _Named:
    def __init__(self, name, fn):
        self.name, self._fn = name, fn

    def __call__(self, x):
        return self._fn(x)


class _Both(_Named):
    """A geometric augmentation: PIL branch for images, tensor branch for masks (as torchvision dispatches)."""

    def __init__(self, name, **kw):
        self.name, self._kw = name, kw

    def __call__(self, x):
        return affine_tensor(x, **self._kw) if torch.is_tensor(x) else affine_pil(x, **self._kw)


def augmentation_specs(img_size=None, subset: str = None):
    """The reference's fixed list (frame_selection_utils.py:49-218) as data: [(name, kind, params)], kind in {'brightness', 'gray',
    'posterize', 'sharpness', 'blur', 'affine'}.  Both the host builders below and the device path (`augment_on_device`) read it."""
    assert subset in {'best_3', 'best_3_with_symmetrical', 'best_all', 'original_only', 'all'}
    translate_distance = (img_size[-1] // 5) if img_size is not None else 200
    S = {
        'bright': ('brightness', dict(factor=1.5)), 'dark': ('brightness', dict(factor=0.5)), 'gray': ('gray', {}),
        'reduce_bits': ('posterize', dict(bits=3)), 'sharp': ('sharpness', dict(factor=16.0)), 'blur': ('blur', dict(kernel_size=7)),
        'rotate_right': ('affine', dict(angle=30.0)), 'rotate_left': ('affine', dict(angle=-30.0)),
        'translate_right': ('affine', dict(translate=(translate_distance, 0))),
        'zoom_out': ('affine', dict(scale=0.5)), 'zoom_in': ('affine', dict(scale=1.5)),
        'shear_right': ('affine', dict(shear=20)), 'shear_left': ('affine', dict(shear=-20)),
    }
    order = {
        'best_3': ['blur', 'zoom_in', 'shear_right'],
        'best_3_with_symmetrical': ['blur', 'zoom_in', 'shear_right', 'shear_left'],
        'best_all': ['bright', 'dark', 'reduce_bits', 'sharp', 'blur', 'rotate_right', 'rotate_left', 'zoom_out', 'zoom_in',
                     'shear_right', 'shear_left'],
        'original_only': None,
        'all': ['bright', 'dark', 'gray', 'reduce_bits', 'sharp', 'blur', 'rotate_right', 'rotate_left', 'translate_right', 'zoom_out',
                'zoom_in', 'shear_right', 'shear_left'],
    }[subset]
    return None if order is None else [(n,) + S[n] for n in order]


def augment_on_device(rgb_u8, mask, subset='best_all'):
    """All augmented frames and masks of one annotated frame in ONE kernel launch (csrc/augment.hip).

    rgb_u8: the decoded frame at working size, uint8 [H, W, 3] on the device; mask: float [K, H, W] on the device (or None).
    Returns (frames uint8 [n, H, W, 3], masks: list of n tensors [K, H, W] - views of one buffer, the input mask itself for the
    colour augmentations) in the reference's list order, or None for subset 'original_only'."""
    import ctypes as C
    from . import ops
    from ._lib import AugDesc, check, load, ptr, stream_ptr
    if not rgb_u8.is_cuda or rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 3 or rgb_u8.shape[2] != 3:
        raise RuntimeError('augment_on_device: expected a device uint8 frame [H, W, 3]')
    H, W = int(rgb_u8.shape[0]), int(rgb_u8.shape[1])
    specs = augmentation_specs((3, H, W), subset)
    if specs is None:
        return None
    n = len(specs)
    descs = (AugDesc * n)()
    TYPES = dict(brightness=0, posterize=1, gray=2, sharpness=3, blur=4, affine=5)
    geometric = []
    for i, (name, kind, kw) in enumerate(specs):
        descs[i].type = TYPES[kind]
        descs[i].factor = float(kw.get('factor', kw.get('bits', 0.0)))
        geometric.append(kind == 'affine')
        if kind == 'affine':
            a = dict(angle=0.0, translate=(0, 0), scale=1.0, shear=0.0); a.update(kw)
            tr = [float(t) for t in a['translate']]
            m_img = _inverse_affine_matrix([W * 0.5, H * 0.5], a['angle'], tr, a['scale'], _shear2(a['shear']))
            m_msk = _inverse_affine_matrix([0.0, 0.0], a['angle'], tr, a['scale'], _shear2(a['shear']))
            theta = torch.tensor(m_msk, dtype=torch.float32).reshape(2, 3)
            resc = theta.t() / torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32)          # [3, 2] as affine_tensor computes it
            for j in range(6):
                descs[i].image_matrix[j] = m_img[j]
            for j, v in enumerate([resc[0, 0], resc[1, 0], resc[2, 0], resc[0, 1], resc[1, 1], resc[2, 1]]):
                descs[i].mask_grid[j] = float(v)
    img = rgb_u8.contiguous()
    out_img = torch.empty((n, H, W, 3), dtype=torch.uint8, device=img.device)
    K = 0 if mask is None else int(mask.shape[0])
    msk = mask.to(dtype=torch.float32).contiguous() if mask is not None else None
    out_mask = torch.empty((n, K, H, W), dtype=torch.float32, device=img.device) if msk is not None else None
    lib = load()
    need = lib.xmem_augment_workspace_bytes(n, H, W)
    ws = ops.workspace(need, img.device, 'augment')
    check(lib.xmem_augment_frames(ptr(img), ptr(msk), H, W, K, descs, n, ptr(out_img), ptr(out_mask), ptr(ws), need, stream_ptr()))
    masks = [(out_mask[i] if geometric[i] else msk) for i in range(n)] if msk is not None else [None] * n
    return out_img, masks


def get_determenistic_augmentations(img_size=None, mask=None, subset: str = None):
    """frame_selection_utils.py:49-218, same name (sic), arguments and list order; returns [(img_aug, mask_aug), ...]."""
    assert subset in {'best_3', 'best_3_with_symmetrical', 'best_all', 'original_only', 'all'}
    from PIL import Image, ImageEnhance, ImageOps
    bright = _Named('bright', lambda im: ImageEnhance.Brightness(im).enhance(1.5))      # ColorJitter(brightness=(1.5, 1.5))
    dark = _Named('dark', lambda im: ImageEnhance.Brightness(im).enhance(0.5))
    gray = _Named('gray', lambda im: Image.merge('RGB', [im.convert('L')] * 3))          # Grayscale(num_output_channels=3)
    reduce_bits = _Named('reduce_bits', lambda im: ImageOps.posterize(im, 3))            # RandomPosterize(bits=3, p=1)
    sharp = _Named('sharp', lambda im: ImageEnhance.Sharpness(im).enhance(16))           # RandomAdjustSharpness(16, p=1)
    blur = _Named('blur', lambda im: gaussian_blur_pil(im, 7))                           # FT.gaussian_blur(kernel_size=7)
    translate_distance = (img_size[-1] // 5) if img_size is not None else 200
    rotate_right = _Both('rotate_right', angle=30.0)                                      # RandomAffine(degrees=(30, 30))
    rotate_left = _Both('rotate_left', angle=-30.0)
    translate_right = _Both('translate_right', translate=(translate_distance, 0))
    zoom_out = _Both('zoom_out', scale=0.5)
    zoom_in = _Both('zoom_in', scale=1.5)
    shear_right = _Both('shear_right', shear=20)
    shear_left = _Both('shear_left', shear=-20)
    identity = _Named('identity', lambda x: x)

    if subset == 'best_3':
        return [(blur, identity), (zoom_in, zoom_in), (shear_right, shear_right)]
    if subset == 'best_3_with_symmetrical':
        return [(blur, identity), (zoom_in, zoom_in), (shear_right, shear_right), (shear_left, shear_left)]
    if subset == 'best_all':
        return [(bright, identity), (dark, identity), (reduce_bits, identity), (sharp, identity), (blur, identity),
                (rotate_right, rotate_right), (rotate_left, rotate_left), (zoom_out, zoom_out), (zoom_in, zoom_in),
                (shear_right, shear_right), (shear_left, shear_left)]
    if subset == 'original_only':
        return None                      # the reference builds a list here and falls off the end of the function (:179-195)
    return [(bright, identity), (dark, identity), (gray, identity), (reduce_bits, identity), (sharp, identity),
            (blur, identity), (rotate_right, rotate_right), (rotate_left, rotate_left), (translate_right, translate_right),
            (zoom_out, zoom_out), (zoom_in, zoom_in), (shear_right, shear_right), (shear_left, shear_left)]
