"""Stand-in for run_on_video used by the launcher tests (CPU, no GPU): writes one marker file per 'frame'."""
import os

import pandas as pd


def run(imgs_in_path, masks_in_path, masks_out_path, frames_with_masks=(0,), compute_iou=False, print_progress=True,
        overwrite_config=None, **kwargs):
    names = sorted(os.listdir(imgs_in_path))
    os.makedirs(os.path.join(masks_out_path, 'masks'), exist_ok=True)
    for n in names:
        with open(os.path.join(masks_out_path, 'masks', n[:-4] + '.png'), 'w') as f:
            f.write(f'rank {os.environ.get("RANK")} local {os.environ.get("LOCAL_RANK")} cfg {sorted((overwrite_config or {}).items())}')
    rows = [{'frame': n, 'mask_provided': i in set(frames_with_masks), 'iou': (-1 if i in set(frames_with_masks) else 0.5)}
            for i, n in enumerate(names)]
    return pd.DataFrame(rows)
