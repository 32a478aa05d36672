#!/usr/bin/env python
"""Pins the oracle of the annotation-candidate selector against the REFERENCE's own function.

    python tests/golden/make_selector_goldens.py          # writes tests/golden/selector.npz  (needs /root/reference)

`inference/frame_selection/frame_selection.py` cannot be imported as it stands in the build container: its module header
imports cv2 and torchvision (and, through frame_selection_utils -> inference.data.video_reader, progressbar), none of which is
installed.  `select_next_candidates` itself (:99-244) needs exactly one thing from them: `torchvision.transforms.Resize(
(h, w), NEAREST)` applied to each mask.  This script therefore installs IMPORT PLACEHOLDERS THAT CONTAIN NO ARITHMETIC:

  * `cv2`, `progressbar`, `torchvision` (+ `.transforms`, `.transforms.functional`) are empty modules whose attributes are inert
    placeholder classes - constructible (module-level `transforms.Normalize(...)` in dataset/range_transform.py), any CALL raises;
  * `Resize` is the one exception: called with a tensor that already HAS the requested size it returns that tensor unchanged, any
    other size raises.  Every scenario below hands over masks at key resolution, so the reference's own code runs end to end and
    nothing written here takes part in its arithmetic.

What this pins: the loop, the validity test, the composite keys, the cycle dissimilarity and the greedy argmax of the reference's
function, bit for bit, against `oracle.cpu_ref.select_next_candidates` (asserted below: identical choices, and the oracle's score
trace is recorded for the GPU test).  What stays unpinned by necessity: torchvision's NEAREST resize of full-resolution masks
(the oracle restates it as F.interpolate(mode='nearest'), which is what torchvision dispatches tensors to) and the
cv2 / torchvision augmentations of frame_selection_utils.py."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
torch.set_grad_enabled(False)


class _Inert:
    """placeholder for a class of a library the image lacks: can be constructed, cannot be used"""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError('import placeholder: this library is not installed in the build container')


class _Placeholder(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        return type(name, (_Inert,), {})


class _IdentityResize:
    """torchvision.transforms.Resize stand-in WITHOUT arithmetic: the identity on tensors that already have the requested size"""
    def __init__(self, size, interpolation=None, **k):
        self.size = tuple(size)

    def __call__(self, t):
        if tuple(t.shape[-2:]) != self.size:
            raise RuntimeError(f'import placeholder: a real resize {tuple(t.shape[-2:])} -> {self.size} was requested; torchvision is not installed')
        return t


def install_placeholders():
    for name in ('cv2', 'progressbar', 'torchvision', 'torchvision.transforms', 'torchvision.transforms.functional'):
        sys.modules[name] = _Placeholder(name)
    tv, tr = sys.modules['torchvision'], sys.modules['torchvision.transforms']
    tv.transforms = tr
    tr.functional = sys.modules['torchvision.transforms.functional']
    tr.Resize = _IdentityResize
    tr.InterpolationMode = types.SimpleNamespace(NEAREST='nearest', BILINEAR='bilinear', BICUBIC='bicubic')     # names only


def scenarios():
    """(name, F, h, w, n_obj, empty frames, call kwargs)"""
    return [('alpha_half', 7, 6, 8, 2, (), dict(num_next_candidates=3, previously_chosen_candidates=(0,), alpha=0.5)),
            ('alpha_zero', 7, 6, 8, 2, (), dict(num_next_candidates=3, previously_chosen_candidates=(0,), alpha=0.0)),
            ('alpha_one_two_previous_all', 6, 11, 13, 1, (), dict(num_next_candidates=2, previously_chosen_candidates=(0, 4), alpha=1.0,
                                                                 only_new_candidates=False)),
            ('tiny_masks_ignored', 8, 6, 8, 2, (2, 5, 0), dict(num_next_candidates=3, previously_chosen_candidates=(0,), alpha=0.5,
                                                             min_mask_presence_percent=5.0)),
            ('all_invalid', 5, 6, 8, 1, (0, 1, 2, 3, 4), dict(num_next_candidates=1, previously_chosen_candidates=(1,), alpha=0.5)),
            ('epsilon_low_240p_grid', 5, 15, 27, 1, (), dict(num_next_candidates=2, previously_chosen_candidates=(0,), alpha=0.3, epsilon=0.2))]


def inputs(F, h, w, n_obj, empty, seed):
    """keys of three scene clusters + noise, masks AT KEY RESOLUTION (values in [0.6, 1] inside an ellipse)"""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(3, 64, h, w, generator=g) * 0.5
    scale = 0.05 + 0.4 * torch.rand(F, generator=g)
    keys = base[torch.arange(F) % 3] + torch.randn(F, 64, h, w, generator=g) * scale.view(F, 1, 1, 1)
    shr = 1 + torch.rand(F, 1, h, w, generator=g) ** 2 * 3
    sel = torch.rand(F, 64, h, w, generator=g)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    masks = []
    for f in range(F):
        m = torch.zeros(n_obj, h, w)
        if f not in empty:
            for o in range(n_obj):
                cy, cx = h * (0.3 + 0.4 * o) + 0.15 * f, w * (0.3 + 0.3 * o) + 0.2 * f
                m[o] = (((yy - cy) / (h * 0.3)) ** 2 + ((xx - cx) / (w * 0.25)) ** 2 < 1).float() * (0.6 + 0.4 * torch.rand(h, w, generator=g))
        masks.append(m)
    return keys, shr, sel, masks


def main():
    install_placeholders()
    sys.path.insert(0, REF)
    from inference.frame_selection.frame_selection import select_next_candidates as ref_select
    sys.path.insert(0, ROOT)
    from oracle import cpu_ref
    torch.set_num_threads(1)
    out = {}
    for i, (name, F, h, w, n_obj, empty, kw) in enumerate(scenarios()):
        keys, shr, sel, masks = inputs(F, h, w, n_obj, empty, seed=100 + i)
        want = ref_select(keys, shr, sel, [m.clone() for m in masks], device='cpu', **kw)
        got = cpu_ref.select_next_candidates(keys, shr, sel, masks, **kw)
        assert list(got) == list(want), (name, got, want)
        trace = np.stack(cpu_ref.select_next_candidates.last_scores)
        print(f'{name}: reference chose {list(want)} - the oracle agrees (best scores {[float(t.max()) for t in trace]})')
        out.update({f'{name}/keys': keys.numpy(), f'{name}/shr': shr.numpy(), f'{name}/sel': sel.numpy(),
                    f'{name}/masks': torch.stack(masks).numpy(), f'{name}/chosen': np.array(list(want), dtype=np.int64),
                    f'{name}/oracle_scores': trace, f'{name}/kwargs': np.array(repr(kw))})
    # extract_keys (frame_selection_utils.py:11-44): the reference function over a three-frame "dataloader" with the IMPORTED reference
    # network behind a processor that exposes encode_frame_key exactly as inference_core.py:53-61 does (InferenceCore.__init__ itself
    # touches cuda:0 at :26, SURVEY 8c work-around 1)
    from inference.frame_selection.frame_selection_utils import extract_keys as ref_extract
    from inference.inference_core import InferenceCore as RefInferenceCore
    from model.network import XMem as RefXMem
    from xmem2_amd.synth import synthetic_frames, synthetic_state_dict
    sd = synthetic_state_dict(0)
    cfg = dict(mem_every=10, deep_update_every=-1, enable_long_term=True, enable_long_term_count_usage=True, hidden_dim=64, key_dim=64,
               value_dim=512, top_k=30, max_mid_term_frames=10, min_mid_term_frames=5, num_prototypes=128, max_long_term_elements=10000,
               model=None)
    net = RefXMem(cfg, None, pretrained_key_encoder=False, pretrained_value_encoder=False).eval()
    net.load_state_dict(sd)

    class Core(RefInferenceCore):
        def __init__(self, network, config):                      # inference_core.py:13-23 without the cuda:0 warm-up
            self.config = config; self.network = network
            self.mem_every = config['mem_every']; self.deep_update_every = config['deep_update_every']
            self.enable_long_term = config['enable_long_term']; self.deep_update_sync = (self.deep_update_every < 0)
            self.clear_memory(); self.all_labels = None

    frames = synthetic_frames(3, 96, 128, seed=21)
    loader = [types.SimpleNamespace(rgb=torch.from_numpy(f)) for f in frames]
    for flatten in (True, False):
        fk, fs, fe, dev, n, key_sum = ref_extract(loader, Core(net, cfg), flatten=flatten)
        assert n == 3 and str(dev) == 'cpu'
        tag = 'extract_flat' if flatten else 'extract_grid'
        out.update({f'{tag}/keys': torch.stack(fk).numpy(), f'{tag}/shr': torch.stack(fs).numpy(), f'{tag}/sel': torch.stack(fe).numpy(),
                    f'{tag}/key_sum': key_sum.numpy()})
        print(f'extract_keys(flatten={flatten}): reference returned {n} frames, key {tuple(fk[0].shape)}, key_sum dtype {key_sum.dtype}')
    out['extract/frames'] = frames
    out['names'] = np.array([s[0] for s in scenarios()])
    out['meta'] = np.array('choices recorded from the imported reference function (inference/frame_selection/frame_selection.py:99-244) behind '
                           'arithmetic-free import placeholders, masks at key resolution; torch ' + torch.__version__ + ', 1 thread')
    np.savez_compressed(os.path.join(HERE, 'selector.npz'), **out)
    print('wrote', os.path.join(HERE, 'selector.npz'))


if __name__ == '__main__':
    main()
