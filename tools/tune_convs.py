"""Run the B32 workload (and the test geometries) once with the autotuner on and dump the measured conv plans."""
import os, sys
os.environ.setdefault('XMEM_CONV_AUTOTUNE', '1')          # this tool IS the autotuner (off by default in the product)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_grad_enabled(False)
from xmem2_amd import ops, XMem, InferenceCore
from xmem2_amd.synth import synthetic_state_dict, synthetic_frames, synthetic_masks
import bench

PREC = os.environ.get('XMEM_PRECISION', 'fp32')               # fp32x: measure the split-operand kernels -> conv_plans_fp32x.json
if PREC == 'fp32x':
    ops._load_plans(True)
    if os.environ.get('XMEM_RETUNE_ALL'):
        ops._plans_x = {}
else:
    ops._plans = {} if os.environ.get('XMEM_RETUNE_ALL') else ops._load_plans()   # default: keep shipped plans, add new shapes
net = XMem(dict(bench.b32_config(), precision=PREC), None).to('cuda').eval(); net.load_weights(synthetic_state_dict(0))
net.use_graphs = False
GEOMS = [(480, 854, 1), (480, 854, 2), (480, 854, 3), (720, 1280, 1), (240, 427, 1), (240, 427, 2), (1080, 1920, 1)]
if os.environ.get('XMEM_TUNE_GEOMS'):            # e.g. "1080x1920x5,1080x1920x3,720x1280x2": more (resolution, objects) pairs - config 5 is 1080p x 5 objects
    GEOMS = [tuple(int(v) for v in g.split('x')) for g in os.environ['XMEM_TUNE_GEOMS'].split(',')]
for (H, W, K) in GEOMS:
    cfg = bench.b32_config(); cfg['mem_every'] = 2
    fr = torch.from_numpy(synthetic_frames(4, H, W)).cuda(); mk = torch.from_numpy(synthetic_masks(4, K, H, W)).cuda()
    core = InferenceCore(net, cfg); core.set_all_labels(list(range(1, K + 1)))
    core.put_to_permanent_memory(fr[0], mk[0])
    for t in range(1, 4):
        core.step(fr[t], None, None)
    if K == 1:                                   # batched key-encoder hints (prefetch_keys): batch shapes incl. skip convs
        for B in (2, 4, 8):
            img = torch.zeros(B, (H + 15) // 16 * 16, (W + 15) // 16 * 16, 4, device='cuda')
            with ops.precision(PREC):
                net._encode_key_eager(img, True, True, False, True)
    # the decoder fuser with its frame-only half pre-convolved in the key pass (prefetched frames): the per-object half's shapes
    h16, w16 = (H + 15) // 16, (W + 15) // 16
    with ops.precision(PREC):
        xf = torch.randn(1, h16, w16, 1024, device='cuda').to(ops.act_dtype())
        pre = (ops.conv2d(xf, net._w['decoder.fuser.block1.conv1@x'], relu_in=True), ops.conv2d(xf, net._w['decoder.fuser.block1.downsample@x']))
        net._fusion(torch.randn(K, h16, w16, 1024 + 512 + 64, device='cuda').to(ops.act_dtype()), 'decoder.fuser', x=xf, pre=pre)
    torch.cuda.synchronize()
    print(H, W, K, 'plans so far', len(ops._tuned_now_x if PREC == 'fp32x' else ops._tuned_now))
if PREC == 'fp16':                           # the fp16 loop's half kernels: their own table (conv_plans_fp16.json)
    n = ops.dump_tuned_plans_half(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/conv_plans_fp16.json')
    print('dumped', n)
    for k, v in sorted(ops._tuned_now_h.items()):
        print(k, v)
    sys.exit(0)
n = ops.dump_tuned_plans(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/conv_plans.json', split=(PREC == 'fp32x'))
print('dumped', n)
for k, v in sorted((ops._tuned_now_x if PREC == 'fp32x' else ops._tuned_now).items()):
    print(k, v)
