"""bench.py's timed region must be self-contained (schema 6): with frames pipelined across step() calls - key batches hinted ahead on a
side stream, the next frame's memory readout enqueued ahead on a third - nothing that belongs to a timed frame may be computed before the
opening barrier, and the region must hold exactly the key passes its frames need (profiles/r06_timed_region_ab.txt: a 20-step region
that starts with its first key batch already encoded runs FASTER than the 200-step steady state).  The test drives bench.run_gpu itself with
spies on the core's entry points and on bench.barrier (called right before the clock starts and right after it stops)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_timed_region_holds_all_the_work_of_its_frames(monkeypatch):
    # a warm-up that is not a multiple of the key batch (the driver's --warmup 5) and a last batch that straddles the end of the region
    steps, warmup = 10, 5
    import bench
    from xmem2_amd import InferenceCore

    log, cores = [], []
    real_init, real_hint, real_step = InferenceCore.__init__, InferenceCore.prefetch_keys, InferenceCore.step

    def init(self, *a, **k):
        real_init(self, *a, **k)
        cores.append(self)

    def hint(self, images, **k):
        images = list(images)
        log.append(('hint', len(images)))
        return real_hint(self, images, **k)

    def step(self, image, *a, **k):
        head = self._pfq[0] if self._pfq else None
        hinted = head is not None and image.is_cuda and head['ptr'] == image.data_ptr()
        early = self._early is not None and hinted and self._early['pf'] is head
        log.append(('step', hinted, early))
        return real_step(self, image, *a, **k)

    def barrier(device):
        c = cores[-1]
        log.append(('barrier', len(c._pfq), c._early is not None))

    monkeypatch.setattr(InferenceCore, '__init__', init)
    monkeypatch.setattr(InferenceCore, 'prefetch_keys', hint)
    monkeypatch.setattr(InferenceCore, 'step', step)
    monkeypatch.setattr(bench, 'barrier', barrier)

    args = bench.parse_args(['--scale-only', '--steps', str(steps), '--warmup', str(warmup)])
    KB = max(1, args.key_batch)
    res = bench.run_gpu(args, torch.device('cuda', 0), 0, 1)
    assert len(res['masks']) == steps

    marks = [i for i, e in enumerate(log) if e[0] == 'barrier']
    assert len(marks) == 2, log
    opening, closing = log[marks[0]], log[marks[1]]
    # nothing pending when the clock starts: no hinted frame, no readout enqueued ahead
    assert opening == ('barrier', 0, False)
    region = log[marks[0] + 1:marks[1]]
    # the region opens with its own first key batch, then steps exactly `steps` frames
    assert region[0] == ('hint', KB)
    frames = [e for e in region if e[0] == 'step']
    assert len(frames) == steps
    # every timed frame consumes a key encoded INSIDE the region (hinted), the first one reads the memory inside its own step()
    assert all(f[1] for f in frames)
    assert frames[0][2] is False
    if steps > 1:
        assert any(f[2] for f in frames[1:])                     # the early readout is in use from the second frame on
    # key passes inside the region: one per started batch of its frames - never fewer (work left outside), at most the whole batches
    passes = [e for e in region if e[0] == 'hint']
    need = (steps + KB - 1) // KB
    assert len(passes) == need and all(p[1] == KB for p in passes)
    # frames hinted past the end (a straddling last batch) are extra work INSIDE the region - their key pass, and the readout the last step
    # enqueued ahead for the first of them; nothing else is left pending
    left = need * KB - steps
    assert closing[1] == left and closing[2] is (left > 0)
