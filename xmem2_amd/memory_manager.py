"""Temporary / permanent / long-term memory and the fused readout (replaces inference/memory_manager.py:8-425).

Same attributes and method names as the reference ``MemoryManager`` (``match_memory``, ``add_memory``,
``compress_features`` ..., ``temporary_work_mem`` / ``permanent_work_mem`` / ``long_mem`` with ``.size``,
``max_work_elements`` ... which the GUI reads), but:

* the three stores are arenas (kv_memory_store.py) passed to the kernels as (pointer, length) segments in
  the reference's order ``[long | temporary | permanent]`` - the per-frame ``torch.cat`` of keys, shrinkage
  and values (memory_manager.py:82-83,126,143-144,181) is gone;
* ``get_similarity`` + ``do_softmax(top_k)`` + ``v @ affinity`` is one fused MFMA / streaming-top-k kernel plus
  a sparse gather of k value rows per query (csrc/affinity.hip): the N x HW affinity is never materialised;
* usage is accumulated order-independently from the sparse affinity.

Tensors handed in by ``InferenceCore`` are row-major: key ``[HW, C_k]``, shrinkage ``[HW]``, selection
``[HW, C_k]``, value ``[K, HW, C_v]``; hidden state is NHWC ``[K, h, w, C_h]``.
"""
import os
import warnings

import torch

from . import ops
from .kv_memory_store import KeyValueMemoryStore, _Arena


class MemoryManager:
    """Manages all three memory stores and the transition between working/long-term memory."""

    def __init__(self, config):
        self.config = config
        self.hidden_dim = config['hidden_dim']
        self.top_k = config['top_k']
        self.enable_long_term = config['enable_long_term']
        self.enable_long_term_usage = config['enable_long_term_count_usage']
        if self.enable_long_term:
            self._read_lt(config)
        self.CK = self.CV = None
        self.H = self.W = None
        self.hidden = None               # NHWC [K, h, w, C_h]
        self.temporary_work_mem = KeyValueMemoryStore(count_usage=self.enable_long_term)
        self.permanent_work_mem = KeyValueMemoryStore(count_usage=False)
        self.frame_id_to_permanent_mem_idx = dict()
        if self.enable_long_term:
            self.long_mem = KeyValueMemoryStore(count_usage=self.enable_long_term_usage)
        self.reset_config = True
        # per object group: (top-k indices, segment sizes, grid width) of the previous match_memory call - handed to the next
        # call as a bound hint (XMEM_AFFINITY_HINT=0 disables; outputs are identical either way)
        self._aff_hint = {}
        self.use_affinity_hint = os.environ.get('XMEM_AFFINITY_HINT', '1') != '0'
        # bumped by everything that changes what a readout sees (elements added / replaced / removed / consolidated, top_k): a readout
        # enqueued ahead of its frame (InferenceCore: early readout) is only consumed if the memory still has the version it read
        self.version = 0

    def _read_lt(self, config):
        self.max_mt_frames = config['max_mid_term_frames']
        self.min_mt_frames = config['min_mid_term_frames']
        self.num_prototypes = config['num_prototypes']
        self.max_long_elements = config['max_long_term_elements']

    def update_config(self, config):
        """memory_manager.py:42-55."""
        self.reset_config = True
        self.version += 1
        self.hidden_dim = config['hidden_dim']
        self.top_k = config['top_k']
        assert self.enable_long_term == config['enable_long_term'], 'cannot update this'
        assert self.enable_long_term_usage == config['enable_long_term_count_usage'], 'cannot update this'
        self.enable_long_term_usage = config['enable_long_term_count_usage']
        if self.enable_long_term:
            self._read_lt(config)

    # ---- readout ---------------------------------------------------------------------------------
    def match_memory_rows(self, qk, qe, out, out_ld, obj_stride, out_off=0, disable_usage_updates=False, defer_usage=None):
        """memory_manager.py:61-190 on row-major operands.

        qk / qe: [HW, C_k] (qe may be None).  Writes the readout of object o (objects ordered by group, as the
        reference's torch.cat over groups) to out[o][q][out_off : out_off + C_v] (pixel stride out_ld floats,
        object stride obj_stride floats).  Returns the number of objects written.
        defer_usage: a list - the usage updates of this readout (memory_manager.py:133-141) are appended to it as
        (store, w, idx, first) instead of being applied; `apply_usage` applies them later (a readout enqueued ahead of its frame does
        not know yet whether the frame's step() will ask for disable_memory_updates)."""
        tmp, perm = self.temporary_work_mem, self.permanent_work_mem
        num_groups = max(tmp.num_groups, perm.num_groups)
        use_long = self.enable_long_term and self.long_mem.engaged()
        lt = self.long_mem if use_long else None
        obj_base = 0
        for gi in range(num_groups):
            # the stores that take part for this group, each contributing its LAST v_size(gi) elements
            # (in long-term mode the list always has the three slots [long | temporary | permanent], the long one empty until
            # the first consolidation / for groups it does not hold: the hint of the previous frame then keeps its layout when
            # the long-term store comes into play - the frame after a consolidation is a hinted call like any other)
            stores = []
            if self.enable_long_term:
                stores.append(lt if (lt is not None and gi < lt.num_groups) else None)
            stores += [tmp, perm]
            segs, vsegs = [], None
            for st in stores:
                if st is None:
                    segs.append((None, None, None))
                    continue
                vs = st.get_v_size(gi)
                start = st.size - vs
                segs.append((st.key_rows(start), st.shrinkage_rows(start), st.rows16(start)))
            # the previous frame's top-k indices of this group bound the k-th similarity of this frame from below (a hint only:
            # the result does not depend on it, see xmem_affinity_topk_hinted)
            sizes = [(sg[0].shape[0] if sg[0] is not None else 0) for sg in segs]
            hint = self._aff_hint.get(gi) if self.use_affinity_hint else None
            if hint is not None and (hint[0].shape[0] != qk.shape[0] or len(hint[1]) != len(sizes)):
                hint = None
            if hint is not None and hint[0].is_cuda:
                # the previous call may have run (and allocated its outputs) on another stream - the readout stream of a core, or the
                # main one: tell the allocator this stream reads the indices, so that their block is not handed out while it does
                hint[0].record_stream(torch.cuda.current_stream())
            w, idx, _ = ops.affinity_topk(segs, qk, qe, self.top_k, hint=hint)
            self._aff_hint[gi] = (idx, sizes, self.W if self.W else 0)
            if gi == 0 and self.enable_long_term and not disable_usage_updates:
                # usage from the first group only (it sees every key), memory_manager.py:93-97,133-141,150-155
                first = 0
                pending = []
                if lt is not None:
                    if self.enable_long_term_usage:
                        pending.append((lt, w, idx, 0))          # usage[:, :long_mem_size]
                    first = lt.size
                pending.append((tmp, w, idx, first))            # usage[:, long : long + temp]; never permanent
                if defer_usage is not None:
                    defer_usage.extend(pending)
                else:
                    self.apply_usage(pending)
            n_obj = stores[-1].value_rows(gi).shape[0]
            vsegs = [[(st.value_rows(gi)[o] if st is not None else None) for st in stores] for o in range(n_obj)]
            ops.readout_sparse(vsegs, w, idx, self.CV, out, out_ld, obj_stride,
                               out_off=out_off + obj_base * obj_stride)
            obj_base += n_obj
        return obj_base

    @staticmethod
    def apply_usage(pending):
        for store, w, idx, first in pending:
            store.update_usage_from(w, idx, first)

    def match_memory(self, query_key, selection, disable_usage_updates=False):
        """Reference-shaped entry: query_key / selection [1,C_k,h,w] -> [K, C_v, h, w] (NHWC-backed view)."""
        h, w = query_key.shape[-2:]
        to_rows = lambda t: (t[0].permute(1, 2, 0).reshape(h * w, -1) if t[0].permute(1, 2, 0).is_contiguous()
                             else ops.nchw_to_nhwc(t).view(h * w, -1))
        qk = to_rows(query_key)
        qe = to_rows(selection) if selection is not None else None
        K = sum(a.shape[0] for a in (self.permanent_work_mem.value_rows(g) for g in range(self.permanent_work_mem.num_groups)))
        out = torch.empty((K, h, w, self.CV), dtype=torch.float32, device=qk.device)
        self.match_memory_rows(qk, qe, out, self.CV, h * w * self.CV, 0, disable_usage_updates)
        return out.permute(0, 3, 1, 2)

    # ---- permanent memory ------------------------------------------------------------------------
    def update_permanent_memory(self, frame_idx, key, shrinkage, value, selection=None):
        """memory_manager.py:192-202."""
        pos = self.frame_id_to_permanent_mem_idx[frame_idx]
        self.version += 1
        self.permanent_work_mem.replace_at(pos, key, value, shrinkage, selection)

    def remove_from_permanent_memory(self, frame_idx):
        """memory_manager.py:204-210.  NOTE (reference quirk, Appendix B): the saved *frame position* is passed
        as an *element offset* and later frames are not renumbered; kept as is for GUI compatibility."""
        pos = self.frame_id_to_permanent_mem_idx[frame_idx]
        self.version += 1
        self.permanent_work_mem.remove_at(pos, self.HW)
        del self.frame_id_to_permanent_mem_idx[frame_idx]

    # ---- insertion -------------------------------------------------------------------------------
    def add_memory(self, key, shrinkage, value, objects, selection=None, permanent=False, ignore=False, ti=None,
                   hw_shape=None):
        """memory_manager.py:212-281 with row-major operands: key [HW,C_k], shrinkage [HW], value [K,HW,C_v],
        selection [HW,C_k] | None.  hw_shape = (h, w) of the stride-16 grid."""
        self.version += 1
        if self.H is None or self.reset_config:
            self.reset_config = False
            if hw_shape is not None:
                self.H, self.W = hw_shape
            else:
                self.H, self.W = key.shape[0], 1
            self.HW = self.H * self.W
            if self.enable_long_term:
                self.min_work_elements = self.min_mt_frames * self.HW
                self.max_work_elements = self.max_mt_frames * self.HW
        self.CK = key.shape[1]
        self.CV = value.shape[2]
        if selection is not None and not self.enable_long_term:
            warnings.warn('the selection factor is only needed in long-term mode', UserWarning)
        if ignore:
            pass        # annotated frames are already in the permanent memory (inference_core.py:80)
        elif permanent:
            pos = self.permanent_work_mem.add(key, value, shrinkage, selection, objects)
            if ti is not None:
                self.frame_id_to_permanent_mem_idx[ti] = pos
        else:
            self.temporary_work_mem.add(key, value, shrinkage, selection, objects)
        nt, npm = self.temporary_work_mem.num_groups, self.permanent_work_mem.num_groups
        if not self.temporary_work_mem.engaged() or (nt != npm):
            # keep both stores engaged with the same object groups (memory_manager.py:253-267)
            if selection is None:
                raise TypeError("'NoneType' object is not subscriptable")   # reference crashes here without long-term (Appendix B)
            empty = (key[0:0], value[:, 0:0], shrinkage[0:0], selection[0:0], objects)
            (self.temporary_work_mem if npm > nt else self.permanent_work_mem).add(*empty)
        if self.enable_long_term and self.temporary_work_mem.size >= self.max_work_elements:
            if self.long_mem.size >= (self.max_long_elements - self.num_prototypes):
                self.long_mem.remove_obsolete_features(self.max_long_elements - self.num_prototypes)
            self.compress_features()

    # ---- hidden state ----------------------------------------------------------------------------
    def create_hidden_state(self, n, sample_key=None, hw_shape=None, device=None):
        """memory_manager.py:283-294; the hidden state is NHWC [n, h, w, C_h]."""
        if hw_shape is None:
            hw_shape = tuple(sample_key.shape[-2:])
            device = sample_key.device
        h, w = hw_shape
        if self.hidden is None:
            self.hidden = torch.zeros((n, h, w, self.hidden_dim), dtype=torch.float32, device=device)
        elif self.hidden.shape[0] != n:
            grown = torch.zeros((n, h, w, self.hidden_dim), dtype=torch.float32, device=device)
            grown[:self.hidden.shape[0]].copy_(self.hidden)
            self.hidden = grown
        assert self.hidden.shape[0] == n

    def set_hidden(self, hidden):
        self.hidden = hidden

    def get_hidden(self):
        return self.hidden

    def frame_already_saved(self, ti):
        return ti in self.frame_id_to_permanent_mem_idx

    # ---- long-term consolidation -----------------------------------------------------------------
    def compress_features(self):
        """memory_manager.py:316-347."""
        tmp = self.temporary_work_mem
        total = tmp.size
        n_c = total - self.min_work_elements
        # candidate value rows per group: the group's rows that fall before the last min_work_elements
        cand_counts = []
        for gi in range(tmp.num_groups):
            n_g = tmp.get_v_size(gi)
            if n_g == total:
                cand_counts.append(n_g - self.min_work_elements)
            else:
                assert self.HW <= n_g < total
                cand_counts.append(n_g - self.min_work_elements if n_g > self.min_work_elements else None)
        proto_key, proto_value, proto_shrinkage = self.consolidation(n_c, cand_counts)
        tmp.sieve_by_range(0, -self.min_work_elements, min_size=self.min_work_elements + self.HW)
        self.long_mem.add(proto_key, proto_value, proto_shrinkage, selection=None, objects=None)

    def consolidation(self, n_c, cand_counts):
        """Prototype selection + potentiation over the first n_c temporary elements, memory_manager.py:349-390."""
        tmp = self.temporary_work_mem
        P = self.num_prototypes
        cand_k = tmp.key_rows()[:n_c]
        cand_s = tmp.shrinkage_rows()
        cand_s = cand_s[:n_c] if cand_s is not None else None
        cand_e = tmp.selection_rows()
        cand_e = cand_e[:n_c] if cand_e is not None else None
        usage = tmp.get_usage_rows()[:n_c].contiguous()
        _, proto_idx = ops.topk_1d(usage, P, largest=True)          # torch.topk(usage, k=P, sorted=True)
        proto_key = ops.gather_rows(cand_k, proto_idx)
        proto_sel = ops.gather_rows(cand_e, proto_idx) if cand_e is not None else None
        sim = ops.similarity_dense(cand_k, cand_s, proto_key, proto_sel)      # [P, n_c]
        proto_value, proto_shrinkage = [], None
        idx_host = None
        for gi, cnt in enumerate(cand_counts):
            if cnt is None:
                proto_value.append(None)
                continue
            aff = ops.softmax_rows_suffix(sim.clone() if (gi + 1 < len(cand_counts) or cnt != n_c) else sim, cnt)
            if cnt == n_c:
                valid = None                                         # every prototype is valid for a full group
            else:
                if idx_host is None:
                    idx_host = proto_idx.cpu()
                valid = torch.nonzero(idx_host >= (n_c - cnt)).flatten().to(torch.int32)
                if valid.numel() == 0:
                    proto_value.append(None)
                    continue
                valid = valid.to(sim.device)
            gv = tmp.value_rows(gi)                                  # [n_obj, n_g, Cv]; candidates are rows [0, cnt)
            outs = []
            for o in range(gv.shape[0]):
                pv = ops.weighted_rows(aff, cnt, gv[o, :cnt])        # [P, Cv]
                outs.append(ops.gather_rows(pv, valid) if valid is not None else pv)
            proto_value.append(torch.stack(outs, 0))
            if gi == 0 and cand_s is not None:
                proto_shrinkage = ops.weighted_rows(aff, cnt, cand_s[n_c - cnt:]).view(-1)
        return proto_key, proto_value, proto_shrinkage

    def copy_perm_mem_only(self):
        """memory_manager.py:392-425: a fresh manager that keeps only the permanent store."""
        new = MemoryManager(config=self.config)
        perm = self.permanent_work_mem
        if not perm.engaged() or perm.size == 0:
            return new
        new.permanent_work_mem = perm
        new.frame_id_to_permanent_mem_idx = self.frame_id_to_permanent_mem_idx
        s0, e0 = perm.shrinkage_rows(), perm.selection_rows()
        # the reference engages the temporary store with empty tensors and then re-uses the old group lists
        tmp = new.temporary_work_mem
        tmp._init(perm.key_rows().shape[1], perm.device, s0 is not None, e0 is not None)
        tmp.obj_groups = self.temporary_work_mem.obj_groups
        tmp.all_objects = self.temporary_work_mem.all_objects
        tmp._v = [_Arena([len(g)], perm.value_rows(0).shape[2], perm.device) for g in tmp.obj_groups]
        new.CK, new.CV, new.H, new.W, new.HW = self.CK, self.CV, self.H, self.W, self.HW
        if self.enable_long_term:
            new.min_work_elements, new.max_work_elements = self.min_work_elements, self.max_work_elements
        new.reset_config = False
        new.create_hidden_state(len(perm.all_objects), hw_shape=(self.H, self.W), device=perm.device)
        return new
