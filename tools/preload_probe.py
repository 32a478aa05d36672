"""The augmented permanent-memory preload of ONE annotated 480p frame (inference/run_on_video.py:231-242, subset 'best_all'):
  host path   : 11 PIL / torch-CPU augmentations + 12 sequential put_to_permanent_memory calls (what the reference does)
  device path : xmem_augment_frames (one launch) + put_many_to_permanent_memory (one batch-12 key pass + one batch-12 value pass)
Wall-clock milliseconds, device synchronised around each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
import bench
from xmem2_amd import InferenceCore, XMem
from xmem2_amd import augmentations as A
from xmem2_amd.synth import synthetic_frames, synthetic_masks, synthetic_state_dict
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
H, W, K = 480, 854, 1
cfg = bench.b32_config()
net = XMem(dict(cfg), None).to(dev).eval(); net.load_weights(synthetic_state_dict(0))
fr = synthetic_frames(2, H, W); mk = torch.from_numpy(synthetic_masks(2, K, H, W))
rgb = np.clip((fr[0].transpose(1, 2, 0) * 0.229 + 0.45) * 255, 0, 255).astype(np.uint8)
pil = Image.fromarray(rgb)
u8 = lambda im: torch.from_numpy(np.array(im, dtype=np.uint8))
res = {}
for rep in range(3):                                           # the first repetition pays graph captures / workspace growth
    core = InferenceCore(net, cfg); core.set_all_labels([1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    core.put_to_permanent_memory(u8(pil).to(dev), mk[0].to(dev))
    ta = time.perf_counter()
    augs = [(u8(ia(pil)), ma(mk[0])) for ia, ma in A.get_determenistic_augmentations((3, H, W), mk[0], subset='best_all')]
    tb = time.perf_counter()
    for im, m in augs:
        core.put_to_permanent_memory(im.to(dev), m.to(dev))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    res['host'] = (1e3 * (t1 - t0), 1e3 * (tb - ta))
    core2 = InferenceCore(net, cfg); core2.set_all_labels([1])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rgb_dev, m_dev = u8(pil).to(dev), mk[0].to(dev)
    aug_rgb, aug_msk = A.augment_on_device(rgb_dev, m_dev, subset='best_all')
    torch.cuda.synchronize(); ta = time.perf_counter()
    core2.put_many_to_permanent_memory([rgb_dev] + [aug_rgb[i] for i in range(aug_rgb.shape[0])], [m_dev] + aug_msk)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    res['device'] = (1e3 * (t1 - t0), 1e3 * (ta - t0))
    assert core.memory.permanent_work_mem.size == core2.memory.permanent_work_mem.size == 12 * 30 * 54
print(f'480p, 1 object, 12 permanent frames per annotation (last of 3 repetitions):')
print(f'  host path   {res["host"][0]:8.1f} ms  (of which the 11 PIL / torch-CPU augmentations {res["host"][1]:.1f} ms)')
print(f'  device path {res["device"][0]:8.1f} ms  (of which the augmentation launch incl. upload {res["device"][1]:.1f} ms)')
print(f'  speed-up    {res["host"][0] / res["device"][0]:8.1f} x')
