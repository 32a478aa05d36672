"""Stream-ordering contract of the frame pipeline (round 6).

Round 5's 'unexplained wrong stream'  was a caller-side race that the library made easy: `prefetch_keys` read DEVICE
inputs on its side stream without waiting for the stream that produced them.  bench.py's parity leg cloned its frames on the main stream
and hinted them at once; after seconds of host work (GPU idle, the queues waking up together) the batched key pass packed frames whose
clone had not landed yet - stale allocator memory, deterministic for a given process history, wrong masks from frame 0 on.  Whether it
showed depended on which hardware queue the side stream was mapped to (an extra stream - the early-readout one - shifted the mapping),
which is why it looked like a property of the early readout.

* `prefetch_keys` now orders the side stream behind the caller's current stream (opt-out: `inputs_complete=True`);
* the streams live on the network, cores on one network are ordered against each other, a dying core waits for its side work;
* an early readout that the next step will not consume is waited for before that step (or a memory edit) touches what it reads."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _cfg(**over):
    from conftest import base_config
    return base_config(**over)


def _stream(hip_net, cfg, perm, clip, n_obj, make_inputs, early=True, inputs_complete=False, kb=4):
    """One core on `hip_net`: preload, then the clip through batched hints.  `make_inputs(frames)` produces the tensors handed to
    prefetch_keys / step for one batch - the place where a caller may create them on the main stream right before the hint."""
    from xmem2_amd.inference_core import InferenceCore
    core = InferenceCore(hip_net, cfg)
    core.early_readout = early
    core.set_all_labels(list(range(1, n_obj + 1)))
    for f, m in perm:
        core.put_to_permanent_memory(f, m)
    out = []
    for a in range(0, len(clip), kb):
        batch = make_inputs(clip[a:a + kb])
        core.prefetch_keys(batch, inputs_complete=inputs_complete)
        for d in batch:
            out.append(core.step(d, None, None).clone())
    core.cancel_prefetch()
    return out, core


def _busy(device, ms=60):
    """~ms of work on the current stream (test scaffolding: plain torch arithmetic)."""
    a = torch.randn(4096, 4096, device=device)
    for _ in range(max(1, ms // 2)):
        a = (a @ a).clamp_(-1, 1)
    return a


def test_prefetch_keys_waits_for_inputs_produced_on_the_callers_stream(hip_net):
    """Inputs cloned on the main stream BEHIND a backlog of main-stream work, into allocator blocks that hold NaN, and hinted at once:
    the batched key pass must see the finished clones (bit-identical to the same stream on long-finished inputs)."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, t = (240, 432), 9
    frames = T(synthetic_frames(t, *hw)).cuda(); masks = T(synthetic_masks(t, 1, *hw)).cuda()
    cfg = _cfg(mem_every=10 ** 9)
    perm = [(frames[0], masks[0])]
    clip = [frames[i] for i in range(1, t)]
    torch.cuda.synchronize()
    truth, _ = _stream(hip_net, cfg, perm, clip, 1, lambda fs: list(fs), inputs_complete=True)

    def racy(fs):
        poison = [torch.full_like(f, float('nan')) for f in fs]         # blocks of exactly the clones' size, full of NaN ...
        torch.cuda.synchronize()
        del poison                                                       # ... back in the allocator
        keep = _busy(fs[0].device)                                       # the clones queue behind this on the main stream
        out = [f.clone() for f in fs]
        del keep
        return out

    got, _ = _stream(hip_net, cfg, perm, clip, 1, racy)
    for i, (a, b) in enumerate(zip(truth, got)):
        assert bool(torch.isfinite(b).all()), f'frame {i + 1}: the key pass read inputs that were not written yet'
        assert torch.equal(a, b), f'frame {i + 1}: differs from the stream on finished inputs (max {float((a - b).abs().max()):.2e})'
    # informational: what the SAME call sequence gives when the caller wrongly claims its inputs are complete (a race by construction;
    # whether it shows depends on the stream -> hardware-queue mapping of this process, so it is printed, not asserted)
    unsafe, _ = _stream(hip_net, cfg, perm, clip, 1, racy, inputs_complete=True)
    bad = sum(int(not torch.equal(a, b)) for a, b in zip(truth, unsafe))
    print(f'inputs_complete=True on unfinished inputs: {bad} of {len(truth)} frames differ (the race the default closes)')


@pytest.mark.parametrize('n_obj', [1, 2])
def test_second_core_after_an_early_readout_core_replays_the_process_history(hip_net, n_obj):
    """The history of bench.py's parity leg: core A streams with early readout on a shared network; the host then idles (the oracle's
    place) while core B is preloaded; core B's inputs are cloned on the main stream and hinted at once.  Core B's masks must be the
    ones a third core computes from long-finished inputs without early readout - bit for bit."""
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, t, P = (240, 432), 17, 4
    frames = T(synthetic_frames(t, *hw)).cuda(); masks = T(synthetic_masks(t, n_obj, *hw)).cuda()
    cfg = _cfg(mem_every=10 ** 9)
    perm = [(frames[j], masks[j]) for j in range(P)]
    clip = [frames[i] for i in range(P, t)]
    torch.cuda.synchronize()
    a_out, core_a = _stream(hip_net, cfg, perm, clip, n_obj, lambda fs: list(fs), early=True, inputs_complete=True)
    assert core_a.early_readout

    def slow_perm():
        for f, m in perm:
            time.sleep(0.15)                                              # GPU idle between the preload calls
            yield f.clone(), m.clone()

    b_out, core_b = _stream(hip_net, cfg, slow_perm(), clip, n_obj, lambda fs: [f.clone() for f in fs], early=True)
    c_out, _ = _stream(hip_net, cfg, perm, clip, n_obj, lambda fs: list(fs), early=False, inputs_complete=True)
    for i, (b, c) in enumerate(zip(b_out, c_out)):
        assert torch.equal(b, c), f'core B frame {i}: {int((b.argmax(0) != c.argmax(0)).sum())} argmax pixels differ'
    for i, (a, c) in enumerate(zip(a_out, c_out)):
        assert torch.equal(a, c), f'core A frame {i} differs'
    del core_a, core_b


def test_early_readout_is_retired_before_steps_and_edits_that_do_not_consume_it(hip_net):
    """a readout enqueued ahead must be waited for by (a) a step that does not segment (need_segment=False), (b)
    remove_from_permanent_memory, (c) update_config, (d) set_all_labels - none of which consumes it.  Same results as with early readout
    off, and nothing left pending after each of them."""
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, t = (240, 432), 22
    frames = T(synthetic_frames(t, *hw)).cuda(); masks = T(synthetic_masks(t, 1, *hw)).cuda()
    cfg = _cfg(mem_every=4, max_mid_term_frames=3, min_mid_term_frames=1, num_prototypes=32)

    def run(early):
        core = InferenceCore(hip_net, cfg)
        core.early_readout = early
        core.set_all_labels([1])
        for j in range(3):
            core.put_to_permanent_memory(frames[j], masks[j], ti=j)
        probs, pending_seen = [], 0
        i = 3
        while i < t:
            if (i - 3) % 4 == 0:
                core.prefetch_keys(list(frames[i:i + 4]), inputs_complete=True)
            pending_seen += int(core._early is not None)
            if i == 12:                                       # (a) every label given: the step only writes the mask to the memory
                p = core.step(frames[i], masks[i], [1])
                assert core._early is None or core._early['pf'] is not None
            elif i == 15:                                     # (b)
                core.remove_from_permanent_memory(1)
                assert core._early is None
                p = core.step(frames[i], None, None)
            elif i == 17:                                     # (c)
                core.update_config(dict(cfg, top_k=20))
                assert core._early is None
                p = core.step(frames[i], None, None)
            elif i == 19:                                     # (d)
                core.set_all_labels([1])
                assert core._early is None
                p = core.step(frames[i], None, None)
            else:
                p = core.step(frames[i], None, None, end=(i == t - 1))
            probs.append(p.clone())
            i += 1
        m = core.memory
        return probs, (m.temporary_work_mem.size, m.permanent_work_mem.size, m.long_mem.size), m.get_hidden().clone(), pending_seen

    p0, s0, h0, n0 = run(False)
    p1, s1, h1, n1 = run(True)
    assert n0 == 0 and n1 >= 4, (n0, n1)
    assert s0 == s1 and torch.equal(h0, h1)
    for i, (a, b) in enumerate(zip(p0, p1)):
        assert torch.equal(a, b), f'step {i + 3}: probabilities differ (max {float((a - b).abs().max()):.2e})'


def test_two_live_cores_hinting_on_one_network_keep_their_own_frames(hip_net):
    """Two cores alive at once on one network (two videos interleaved by one host thread), both hinting batches of the same size:
    the key-stage buffer groups belong to a core's owner token, so core B's hint never lands in the group that still holds core A's
    unconsumed frames.  Each core's masks equal the ones it computes alone - bit for bit - and neither had to drop a hint."""
    from xmem2_amd.inference_core import InferenceCore
    from xmem2_amd.synth import synthetic_frames, synthetic_masks
    hw, t, kb = (240, 432), 13, 4
    fa = T(synthetic_frames(t, *hw, seed=3)).cuda(); ma = T(synthetic_masks(t, 1, *hw)).cuda()
    fb = T(synthetic_frames(t, *hw, seed=4)).cuda(); mb = ma.flip(-1).contiguous()
    cfg = _cfg(mem_every=10 ** 9)
    torch.cuda.synchronize()
    solo_a, _ = _stream(hip_net, cfg, [(fa[0], ma[0])], [fa[i] for i in range(1, t)], 1, lambda fs: list(fs), inputs_complete=True)
    solo_b, _ = _stream(hip_net, cfg, [(fb[0], mb[0])], [fb[i] for i in range(1, t)], 1, lambda fs: list(fs), inputs_complete=True)

    cores = []
    for f, m in ((fa, ma), (fb, mb)):
        c = InferenceCore(hip_net, cfg)
        c.set_all_labels([1])
        c.put_to_permanent_memory(f[0], m[0])
        cores.append(c)
    assert cores[0]._uid != cores[1]._uid
    outs = ([], [])
    for a in range(1, t, kb):
        for c, f in zip(cores, (fa, fb)):                                # both hint before either consumes
            c.prefetch_keys([f[i] for i in range(a, min(a + kb, t))], inputs_complete=True)
        for i in range(a, min(a + kb, t)):                               # ... and consume frame by frame, alternating
            for k, (c, f) in enumerate(zip(cores, (fa, fb))):
                assert any(e['ptr'] == f[i].data_ptr() for e in c._pfq), f'core {k} lost its hint of frame {i}'
                outs[k].append(c.step(f[i], None, None).clone())
    for k, solo in enumerate((solo_a, solo_b)):
        for i, (x, y) in enumerate(zip(solo, outs[k])):
            assert torch.equal(x, y), f'core {k} frame {i + 1}: {int((x.argmax(0) != y.argmax(0)).sum())} argmax pixels differ from its solo stream'
    for c in cores:
        c.cancel_prefetch()
