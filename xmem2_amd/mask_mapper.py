"""Host-side label bookkeeping for annotation masks (behaviour of inference/data/mask_mapper.py:7-63).

An annotation arrives as an index image whose object ids may be sparse (e.g. palette entries 5 and 9).
The network works with dense ids 1..K in order of first appearance.  `MaskMapper` keeps that mapping, hands out
one-hot tensors `[K, H, W]` and maps predicted index masks back to the original ids.
"""
import numpy as np
import torch


class MaskMapper:
    def __init__(self):
        self.labels = []          # original ids in order of first appearance
        self.remappings = {}      # original id -> dense id (1-based)
        self.coherent = True      # stays True while original ids already are 1, 2, 3, ... in that order

    def _register(self, ids):
        """Assign dense ids to not-yet-seen original ids; returns the newly seen ones."""
        unseen = list(set(ids) - set(self.labels))
        base = len(self.labels)
        for offset, original in enumerate(unseen, start=1):
            dense = base + offset
            self.remappings[original] = dense
            self.coherent = self.coherent and dense == original
        return unseen

    def convert_mask(self, mask, exhaustive=False):
        """mask: H x W index image.  Returns (one-hot float tensor [K, H, W], dense ids this call contributes).

        exhaustive=False: every id in `mask` must be new (id 0 = "don't care", YouTubeVOS style).
        exhaustive=True : id 0 is background, every pixel is labelled; the returned ids are 1..K."""
        ids = np.unique(mask).astype(np.uint8)
        ids = [int(v) for v in ids if v != 0]
        known_before = len(self.labels)
        unseen = self._register(ids)
        if not exhaustive and len(unseen) != len(ids):
            raise AssertionError('Old labels found in non-exhaustive mode')
        self.labels.extend(unseen)
        if exhaustive:
            contributed = range(1, len(self.labels) + 1)
        elif self.coherent:
            contributed = unseen
        else:
            contributed = range(known_before + 1, len(self.labels) + 1)
        if self.labels:
            wanted = np.asarray(self.labels, dtype=mask.dtype).reshape(-1, *([1] * mask.ndim))
            onehot = (mask[None] == wanted).astype(np.uint8)
        else:
            onehot = np.zeros((0,) + tuple(mask.shape), np.uint8)
        return torch.from_numpy(onehot).float(), contributed

    def remap_index_mask(self, mask):
        """Dense ids of a predicted index mask -> original ids."""
        if self.coherent:
            return mask
        lut = np.zeros(256, dtype=mask.dtype)
        for original, dense in self.remappings.items():
            lut[dense] = original
        return lut[mask]
