"""Shared clip runners of the multi-object parity tests and of tests/parity_by_plan.py (test infrastructure: imports the oracle).

A clip is described once (frames, masks, schedule); `run_oracle` drives oracle.cpu_ref.RefCore over it at a stated torch
thread count, `run_gpu` drives the HIP InferenceCore over the same frames.  Both return per-frame argmax masks, full
probabilities and the memory sizes, so that callers can compare GPU vs oracle(1 thread) AND oracle(8 threads) vs oracle(1
thread) - the reference's own thread-count noise - on the same frames."""
import ast

import numpy as np
import torch

from conftest import base_config, load_golden
from oracle import cpu_ref as R

T = torch.from_numpy


class Clip:
    def __init__(self, name, cfg, frames, masks, labels, perm_frames, mask_frames, first_step, key_batch=0, end_flag=True):
        self.name, self.cfg, self.frames, self.masks, self.labels = name, cfg, frames, masks, labels
        self.perm_frames, self.mask_frames, self.first_step, self.key_batch = perm_frames, mask_frames, first_step, key_batch
        self.end_flag = end_flag              # False: the last frame is stepped like any other (a stream that goes on, as bench.py's)
        self.t = frames.shape[0]
        self.hw = tuple(frames.shape[-2:])


C3_CONDITIONING = 'multi_object'       # the checkpoint variant the config-3 clip is run with (xmem2_amd.synth)


def c3_clip():
    """BASELINE config 3 at its stated size: 480p, 3 objects, mem_every=2, T_max=4 -> a consolidation at the 5th temporary frame.
    Run it with the 'multi_object' conditioning of the synthetic checkpoint (fixtures hip_net_mo / ref_net_mo): on the plain one 45 % of
    the pixels are ties between objects and the reference's own argmax is noise."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    t, hw, K = 13, (480, 854), 3
    cfg = base_config(mem_every=2, max_mid_term_frames=4, min_mid_term_frames=2, num_prototypes=64)
    return Clip('480p_3obj_consolidation', cfg, T(synthetic_frames(t, *hw)), T(synthetic_masks(t, K, *hw)), [1, 2, 3], [0], set(), 1, key_batch=4)


def c3_bench_clip(n_steps=25):
    """The stream `bench.py --workload c3` times and compares (WORKLOADS['c3'] + workload_config): 480p, 3 objects, ONE permanent
    frame, mem_every=5 (five memory frames written back with their PREDICTED masks inside 25 steps, no consolidation), key hints in
    batches of 4, no `end` flag, 'multi_object' conditioning.  Round 5's bench print on this clip (119 px against the oracle's own 32)
    was outside the reference's noise; this is that clip as a test."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, K = (480, 854), 3
    cfg = base_config(mem_every=5)            # == bench.workload_config(WORKLOADS['c3'])
    return Clip('480p_3obj_bench_c3', cfg, T(synthetic_frames(1 + 32, *hw)[:1 + n_steps]), T(synthetic_masks(1 + 32, K, *hw)[:1 + n_steps]),
                [1, 2, 3], [0], set(), 1, key_batch=4, end_flag=False)


def golden_clip(tag, hw, n_obj):
    """The clips of tests/golden/e2e_*.npz (recorded from the imported reference at 1 thread)."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    g = load_golden('e2e_' + tag)
    cfg = ast.literal_eval(str(g['config']))
    t = int(g['shape'][0])
    return Clip(tag, cfg, T(synthetic_frames(t, *hw)), T(synthetic_masks(t, n_obj, *hw)), [int(x) for x in g['labels']],
                [int(j) for j in g['perm_frames']], set(int(x) for x in g['mask_frames']), 0)


def _step_args(clip, ti):
    mk = clip.masks[ti] if ti in clip.mask_frames else None
    return mk, (clip.labels if mk is not None else None), dict(end=(clip.end_flag and ti == clip.t - 1), do_not_add_mask_to_memory=(mk is not None))


def _sizes(m):
    return (m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size)


_ORACLE_CACHE = {}


def run_oracle(ref_net, clip, threads):
    """(argmax masks, probabilities, memory sizes) per frame of the oracle at `threads` torch threads; cached per (clip, threads)
    within a test session (the oracle is deterministic at a fixed thread count)."""
    ck = (clip.name, threads)
    if ck not in _ORACLE_CACHE:
        _ORACLE_CACHE[ck] = _run_oracle(ref_net, clip, threads)
    return _ORACLE_CACHE[ck]


def _run_oracle(ref_net, clip, threads):
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        ref = R.RefCore(ref_net, dict(clip.cfg))
        ref.set_all_labels(clip.labels)
        for j in clip.perm_frames:
            ref.put_to_permanent_memory(clip.frames[j], clip.masks[j])
        arg, prob, sizes = [], [], []
        for ti in range(clip.first_step, clip.t):
            mk, vl, kw = _step_args(clip, ti)
            q = ref.step(clip.frames[ti], mk.clone() if mk is not None else None, vl, **kw)
            arg.append(torch.argmax(q, 0).numpy().astype(np.uint8)); prob.append(q.clone()); sizes.append(_sizes(ref.memory))
        return arg, prob, sizes
    finally:
        torch.set_num_threads(prev)


def run_gpu(hip_net, clip):
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd import ops
    core = InferenceCore(hip_net, dict(clip.cfg))
    core.set_all_labels(clip.labels)
    dev = [clip.frames[i].cuda() for i in range(clip.t)]
    for j in clip.perm_frames:
        core.put_to_permanent_memory(dev[j], clip.masks[j].cuda())
    arg, prob, sizes = [], [], []
    for ti in range(clip.first_step, clip.t):
        if clip.key_batch and (ti - clip.first_step) % clip.key_batch == 0:
            core.prefetch_keys(dev[ti:ti + clip.key_batch])
        mk, vl, kw = _step_args(clip, ti)
        p = core.step(dev[ti], mk.cuda() if mk is not None else None, vl, **kw)
        arg.append(ops.argmax_u8(p).cpu().numpy()); prob.append(p.cpu()); sizes.append(_sizes(core.memory))
    return arg, prob, sizes


def compare(a, b, labels, lo=0, hi=None):
    """Clip-level IoU per object and argmax mismatch of two argmax-mask lists over frames [lo, hi)."""
    hi = len(a) if hi is None else hi
    if hi <= lo:
        return dict(iou=[1.0] * len(labels), mismatch=0, pixels=0, min_frame_iou=1.0)
    A, B = np.stack(a[lo:hi]), np.stack(b[lo:hi])
    iou = [float(((A == c) & (B == c)).sum() / max(((A == c) | (B == c)).sum(), 1)) for c in labels]
    frame_iou = min(R.compute_array_iou(A[i], B[i]) for i in range(len(A)))
    return dict(iou=iou, mismatch=int((A != B).sum()), pixels=int(A.size), min_frame_iou=float(frame_iou))


def fmt(c):
    return 'IoU ' + '/'.join(f'{v:.5f}' for v in c['iou']) + f", mismatch {c['mismatch']}/{c['pixels']} ({c['mismatch'] / max(c['pixels'], 1):.2e}), min frame IoU {c['min_frame_iou']:.5f}"
