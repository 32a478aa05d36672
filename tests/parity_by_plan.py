"""Attribution of the multi-object parity margin (GPU box):  python tests/parity_by_plan.py > profiles/rNN_c3_parity_by_plan.txt

Runs the BASELINE config-3 clip (480p, 3 objects, consolidation) and the 240p two-object golden clip through the HIP path with
(a) the shipped convolution plans (F(4x4) / F(2x2) Winograd + direct), (b) F(4x4) replaced by F(2x2), (c) every convolution in
the direct implicit-GEMM form, and prints each variant's IoU / argmax mismatch against oracle(1 thread) next to the oracle's own
8-thread-vs-1-thread figures on the same frames.  Test infrastructure (it imports the oracle); not part of the product path."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402

torch.set_grad_enabled(False)

import clip_util as U  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from xmem2_amd import ops  # noqa: E402
from xmem2_amd.network import XMem  # noqa: E402
from xmem2_amd.synth import synthetic_state_dict  # noqa: E402


def main():
    sds = {None: synthetic_state_dict(0), 'multi_object': synthetic_state_dict(0, conditioning='multi_object')}
    clips = [U.c3_clip(), U.golden_clip('240p_2obj', (240, 427), 2), U.c3_bench_clip()]
    if len(sys.argv) > 1:                                        # e.g. `parity_by_plan.py 240p_2obj`: only the named clips
        clips = [c for c in clips if c.name in sys.argv[1:]]
    print(f'device {torch.cuda.get_device_name(0)}; oracle threads: 1 (the goldens\' count) and 8; host cores {os.cpu_count()}')
    for clip in clips:
        t0 = time.time()
        sd = sds[U.C3_CONDITIONING if clip.name.startswith('480p_3obj') else None]       # the config-3 clip runs on the multi-object conditioning
        ref_net = cpu_ref.RefNet(sd)
        o1, _, s1 = U.run_oracle(ref_net, clip, 1)
        o8, _, _ = U.run_oracle(ref_net, clip, 8)
        first_lt = next((i for i, z in enumerate(s1) if z[2] > 0), None)
        spans = [('whole clip', 0, None)]
        if first_lt is not None:
            spans = [('before the consolidation', 0, first_lt), ('after the consolidation', first_lt, None)]
        print(f'\n== {clip.name}: {clip.t} frames {clip.hw}, objects {clip.labels}, consolidation at step {first_lt} (oracle runs {time.time() - t0:.0f} s)')
        for name, lo, hi in spans:
            print(f'   {name:26s} oracle(8 thr) vs oracle(1 thr): {U.fmt(U.compare(o8, o1, clip.labels, lo, hi))}')
        for form, label in ((None, 'shipped plans (F(4x4)+F(2x2)+direct)'), ('f2', 'F(4x4) -> F(2x2)'), ('direct', 'direct form everywhere'),
                            ('direct_sk2', 'direct form, every contraction summed in 2 slabs'),
                            ('direct_sk3', 'direct form, every contraction summed in 3 slabs')):
            ops.CONV_FORM = form
            net = XMem({'key_dim': 64, 'value_dim': 512, 'hidden_dim': 64}, None).to('cuda').eval()
            net.load_weights(sd)
            a, _, s = U.run_gpu(net, clip)
            assert s == s1
            for name, lo, hi in spans:
                print(f'   {name:26s} HIP [{label}] vs oracle(1 thr): {U.fmt(U.compare(a, o1, clip.labels, lo, hi))}')
            del net
        ops.CONV_FORM = None


if __name__ == '__main__':
    main()
