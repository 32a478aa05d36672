#!/usr/bin/env python
"""Fixtures of the reference's GPU MODE as restated on the CPU (oracle.cpu_ref.RefNetAutocast / get_similarity_autocast):
torch.cuda.amp.autocast around the frame loop (inference/run_on_video.py:76), fp32 preload (:59-66).

    python tests/golden/make_autocast_goldens.py          # writes tests/golden/e2e_<clip>_autocast.npz

PARITY UNPINNED: CUDA autocast cannot run in the build container, so these are outputs of a RESTATEMENT of autocast's operator
policy, not of the reference itself (unlike every other fixture in this directory, which make_goldens.py records from the imported
reference).  What they are for: the two golden clips run through (a) the full autocast policy - fp16 convolutions AND the memory's
fp16 similarity / readout GEMMs - and (b) the policy on the network alone with the memory's matmuls left in fp32 (what the fp16
loop of this repository does).  Their deviation from the fp32 goldens of the same clips is the floor `tests/test_gpu_fp16_loop.py`
gates the HIP fp16 loop against: the loop must be at least as close to the reference's fp32 path as the reference's own GPU mode
is by this restatement."""
import ast
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
torch.set_grad_enabled(False)

from oracle import cpu_ref as R  # noqa: E402
from xmem2_amd.synth import synthetic_frames, synthetic_masks, synthetic_state_dict  # noqa: E402

CLIPS = [('480p_1obj', (480, 854), 1), ('240p_2obj', (240, 427), 2)]


def run(sd, tag, hw, n_obj, memory_autocast):
    g = np.load(os.path.join(HERE, f'e2e_{tag}.npz'), allow_pickle=False)
    cfg = ast.literal_eval(str(g['config']))
    t = int(g['shape'][0])
    frames, masks = torch.from_numpy(synthetic_frames(t, *hw)), torch.from_numpy(synthetic_masks(t, n_obj, *hw))
    labels = [int(x) for x in g['labels']]
    core = R.RefCore(R.RefNet(sd), cfg, autocast_network=R.RefNetAutocast(sd), autocast_memory=memory_autocast)
    core.set_all_labels(labels)
    for j in g['perm_frames']:
        core.put_to_permanent_memory(frames[int(j)], masks[int(j)])
    mask_frames = set(int(x) for x in g['mask_frames'])
    arg, pds, sizes = [], [], []
    for ti in range(t):
        mk = masks[ti] if ti in mask_frames else None
        p = core.step(frames[ti], mk, labels if mk is not None else None, end=(ti == t - 1), do_not_add_mask_to_memory=(mk is not None))
        assert p.dtype == torch.float32 and bool(torch.isfinite(p).all())
        arg.append(torch.argmax(p, 0).numpy().astype(np.uint8)); pds.append(p[:, 4::8, 4::8].numpy().copy())
        m = core.memory
        sizes.append([m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size])
    return np.stack(arg), np.stack(pds), np.array(sizes), g


def stats(a, ref, labels):
    iou = [float(((a == c) & (ref == c)).sum() / max(((a == c) | (ref == c)).sum(), 1)) for c in labels]
    return iou, float((a != ref).mean())


def main():
    torch.set_num_threads(1)                     # as the fp32 goldens
    sd = synthetic_state_dict(0)
    for tag, hw, n_obj in CLIPS:
        t0 = time.time()
        full = run(sd, tag, hw, n_obj, True)
        net_only = run(sd, tag, hw, n_obj, False)
        g = full[3]
        labels = [int(x) for x in g['labels']]
        ref = g['argmax']
        np.testing.assert_array_equal(full[2], g['sizes'])
        np.testing.assert_array_equal(net_only[2], g['sizes'])
        s_full, s_net = stats(full[0], ref, labels), stats(net_only[0], ref, labels)
        perr = lambda x: float(np.abs(x - g['prob_ds8']).mean(axis=(1, 2, 3)).max())
        print(f'{tag}: autocast policy (network + memory GEMMs) vs fp32 goldens: IoU {s_full[0]}, mismatch {s_full[1]:.2e}, worst mean |dp| {perr(full[1]):.2e}')
        print(f'{tag}: autocast policy on the network, fp32 memory matmuls  : IoU {s_net[0]}, mismatch {s_net[1]:.2e}, worst mean |dp| {perr(net_only[1]):.2e}   ({time.time() - t0:.0f} s)')
        np.savez_compressed(os.path.join(HERE, f'e2e_{tag}_autocast.npz'),
                            argmax_full=full[0], prob_ds8_full=full[1], argmax_net_only=net_only[0], prob_ds8_net_only=net_only[1],
                            iou_full_vs_fp32=np.array(s_full[0]), mismatch_full_vs_fp32=np.array(s_full[1]),
                            iou_net_only_vs_fp32=np.array(s_net[0]), mismatch_net_only_vs_fp32=np.array(s_net[1]),
                            worst_mean_abs_dp_full=np.array(perr(full[1])), worst_mean_abs_dp_net_only=np.array(perr(net_only[1])),
                            meta=np.array('oracle.cpu_ref.RefCore(autocast_network=RefNetAutocast) at 1 thread; PARITY UNPINNED: a restatement of CUDA '
                                          'autocast\'s operator policy, not an output of the reference; torch ' + torch.__version__))


if __name__ == '__main__':
    main()
