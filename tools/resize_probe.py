import os, sys, tempfile
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from PIL import Image
from xmem2_amd.run_on_video import run_on_video
from xmem2_amd.synth import synthetic_frames, synthetic_masks
root = tempfile.mkdtemp(dir='/tmp')
imgs, msks, out = (os.path.join(root, d) for d in ('JPEGImages', 'Annotations', 'out'))
os.makedirs(imgs); os.makedirs(msks)
t, hw = 9, (120, 200)
fr = synthetic_frames(t, *hw); mk = synthetic_masks(t, 2, *hw)
pal = [0, 0, 0, 200, 0, 0, 0, 200, 0] + [0] * (768 - 9)
for i in range(t):
    rgb = np.clip((fr[i].transpose(1, 2, 0) * 0.229 + 0.45) * 255, 0, 255).astype(np.uint8)
    Image.fromarray(rgb).save(os.path.join(imgs, f'{i:05d}.jpg'))
    idx = (mk[i, 0] * 1 + mk[i, 1] * 2).astype(np.uint8)
    im = Image.fromarray(idx, mode='P'); im.putpalette(pal); im.save(os.path.join(msks, f'{i:05d}.png'))
stats = run_on_video(imgs, msks, out, frames_with_masks=[0, 5], compute_iou=True, print_progress=False,
                     overwrite_config={'model': None, 'size': 80, 'mem_every': 2}, save_overlay=True)
print(stats)
m = np.array(Image.open(os.path.join(out, 'masks', '00003.png')))
print('mask shape', m.shape, 'overlay', Image.open(os.path.join(out, 'overlay', '00003.jpg')).size)
assert m.shape[:2] == hw
