"""Reference-shaped wrappers of the memory read for callers that hold plain `B x C x N` tensors
(model/memory_util.py:7-80): `get_similarity`, `do_softmax`, `get_affinity`, `readout`.

The inference path never materialises the N x HW affinity (MemoryManager.match_memory_rows -> xmem_affinity_topk +
xmem_readout_sparse); these functions are the dense, full-softmax form the reference uses at training time
(`XMem.read_memory`, model/network.py:89-105) and in the long-term consolidation, built from the same C entry points
(xmem_similarity_dense, xmem_softmax_rows_suffix, xmem_softmax_rows_topk, xmem_weighted_rows).  They do materialise B x N x HW.
"""
import torch

from . import ops


def _rows(t):
    """B x C x ... -> list over the batch of contiguous [N, C] row matrices."""
    t = t.flatten(start_dim=2)
    return [t[b].t().contiguous() for b in range(t.shape[0])]


def get_similarity(mk, ms, qk, qe):
    """memory_util.py:7-39.  mk B x CK x N.., ms B x 1 x N.. | None, qk / qe B x CK x HW.. (qe may be None) -> B x N x HW."""
    mkr, qkr = _rows(mk), _rows(qk)
    qer = _rows(qe) if qe is not None else [None] * len(qkr)
    msr = [ms.flatten(start_dim=1)[b].contiguous() for b in range(ms.shape[0])] if ms is not None else [None] * len(mkr)
    out = [ops.similarity_dense(mkr[b], msr[b], qkr[b], qer[b]).t() for b in range(len(mkr))]     # [P, N] -> N x HW
    return torch.stack(out, 0)


def do_softmax(similarity, top_k=None, inplace=False, return_usage=False):
    """memory_util.py:41-65 on a materialised B x N x HW similarity: top-k softmax without max shift, or the stable full
    softmax over the memory axis.  (`inplace` is accepted for signature compatibility; a new tensor is returned.)"""
    B, N, HW = similarity.shape
    rows = similarity.transpose(1, 2).contiguous()                                       # [B, HW, N]: softmax over each row
    for b in range(B):
        if top_k is not None:
            ops.softmax_rows_topk(rows[b], top_k)                                        # xmem_softmax_rows_topk
        else:
            ops.softmax_rows_suffix(rows[b], N)
    affinity = rows.transpose(1, 2)
    if return_usage:
        return affinity, affinity.sum(dim=2)
    return affinity


def get_affinity(mk, ms, qk, qe):
    """memory_util.py:67-71 (training-time shorthand, no top-k)."""
    return do_softmax(get_similarity(mk, ms, qk, qe))


def readout(affinity, mv):
    """memory_util.py:73-80: mv B x CV x T x H x W, affinity B x THW x HW -> B x CV x H x W."""
    B, CV, T, H, W = mv.shape
    out = []
    for b in range(B):
        aff_rows = affinity[b].t().contiguous()                                          # [HW, N]
        vals = mv[b].reshape(CV, T * H * W).t().contiguous()                             # [N, CV]
        out.append(ops.weighted_rows(aff_rows, T * H * W, vals).t())                     # [HW, CV] -> CV x HW
    return torch.stack(out, 0).reshape(B, CV, H, W)
