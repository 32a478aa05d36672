"""Index mask -> one-hot with label remapping (host-side, inference/data/mask_mapper.py:7-63)."""
import numpy as np
import torch


class MaskMapper:
    """Converts an indexed mask to one-hot, remapping non-contiguous labels to 1..K.
    Default mode: only NEW labels may appear; exhaustive mode: label 0 is background and every pixel is labelled."""

    def __init__(self):
        self.labels = []
        self.remappings = {}
        self.coherent = True     # no remapping needed while labels arrive as 1, 2, 3 ...

    def convert_mask(self, mask, exhaustive=False):
        present = np.unique(mask).astype(np.uint8)
        present = present[present != 0].tolist()
        fresh = list(set(present) - set(self.labels))
        if not exhaustive:
            assert len(fresh) == len(present), 'Old labels found in non-exhaustive mode'
        for i, lab in enumerate(fresh):
            mapped = i + len(self.labels) + 1
            self.remappings[lab] = mapped
            if self.coherent and mapped != lab:
                self.coherent = False
        if exhaustive:
            new_mapped = range(1, len(self.labels) + len(fresh) + 1)
        elif self.coherent:
            new_mapped = fresh
        else:
            new_mapped = range(len(self.labels) + 1, len(self.labels) + len(fresh) + 1)
        self.labels.extend(fresh)
        onehot = np.zeros((len(self.labels),) + tuple(mask.shape), np.uint8)
        for k, lab in enumerate(self.labels):
            onehot[k] = (mask == lab)
        return torch.from_numpy(onehot).float(), new_mapped

    def remap_index_mask(self, mask):
        if self.coherent:
            return mask
        out = np.zeros_like(mask)
        for lab, i in self.remappings.items():
            out[mask == i] = lab
        return out
