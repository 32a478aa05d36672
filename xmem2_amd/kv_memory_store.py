"""Key/value memory store as preallocated HBM arenas (replaces inference/kv_memory_store.py:4-240).

The reference grows every tensor with ``torch.cat`` on each memory frame and keeps keys as ``1 x C x N``.
Here each store owns row-major arenas sized for the device (keys ``[cap, C_k]``: one 256-byte row per
memory element, values ``[n_obj, cap, C_v]``: one 2 KiB row per element and object) that grow
geometrically, so an append is a device copy into the tail and the affinity / readout kernels address
a store as a (pointer, length) segment - no concatenation ever happens.

Object groups keep the reference's suffix alignment (memory_manager.py:99-120): group g's values belong
to the LAST ``get_v_size(g)`` keys of the store.

Row-major views are exposed for the kernels (``key_rows`` ...); the reference-shaped properties
(``key`` ``1 x C x N`` ...) are zero-copy transposed views for callers such as the GUI gauges.
"""
import os

import torch

from . import ops

KEEP_ROWS16 = os.environ.get('XMEM_KEEP_ROWS16', '1') != '0'     # 0: the readout derives its fp16 operand rows per call (A/B runs)


class _Arena:
    """Growable [*, cap, C] device buffer with a fill count along dim -2 (or -1 for 1-D rows)."""

    def __init__(self, lead, width, device, cap=0):
        self.lead, self.width, self.device = lead, width, device
        self.n = 0
        self.buf = None
        if cap:
            self._alloc(cap)

    def _shape(self, cap):
        s = list(self.lead) + [cap]
        if self.width:
            s.append(self.width)
        return s

    def _alloc(self, cap):
        new = torch.empty(self._shape(cap), dtype=torch.float32, device=self.device)
        if self.buf is not None and self.n:
            self._rows(new, 0, self.n).copy_(self._rows(self.buf, 0, self.n))
        self.buf = new

    def _rows(self, buf, a, b):
        return buf[..., a:b, :] if self.width else buf[..., a:b]

    @property
    def cap(self):
        return 0 if self.buf is None else self.buf.shape[len(self.lead)]

    def rows(self, a=0, b=None):
        b = self.n if b is None else b
        if self.buf is None:
            return torch.empty(self._shape(0), dtype=torch.float32, device=self.device)
        return self._rows(self.buf, a, b)

    def append(self, rows, count):
        """Append `count` rows (a tensor, or a python float to fill with)."""
        if count == 0:
            return
        need = self.n + count
        if need > self.cap:
            self._alloc(max(need, 2 * self.cap, 4096))
        dst = self._rows(self.buf, self.n, need)
        if isinstance(rows, float):
            dst.fill_(rows)
        else:
            dst.copy_(rows)
        self.n = need

    def reserve(self, count):
        """Make room for `count` more rows and count them in; the caller writes them (no fill kernel)."""
        need = self.n + count
        if need > self.cap:
            self._alloc(max(need, 2 * self.cap, 4096))
        self.n = need

    def keep(self, ranges):
        """Keep only the given (a, b) row ranges, in order (compaction in place)."""
        pos = 0
        for a, b in ranges:
            if b <= a:
                continue
            if a != pos:
                src = self._rows(self.buf, a, b)
                if a < pos + (b - a):      # overlapping move: go through a temporary
                    src = src.clone()
                self._rows(self.buf, pos, pos + (b - a)).copy_(src)
            pos += b - a
        self.n = pos

    def take(self, index):
        """Keep rows at int32 device `index` (ascending)."""
        m = index.numel()
        if self.lead:
            for i in range(self.lead[0]):
                self.buf[i, :m].copy_(ops.gather_rows(self.buf[i, :self.n], index))
        else:
            self.buf[:m].copy_(ops.gather_rows(self.buf[:self.n], index))
        self.n = m


class KeyValueMemoryStore:
    """Works for key/value pairs type storage, e.g. working and long-term memory (kv_memory_store.py:4-34)."""

    def __init__(self, count_usage: bool, device=None):
        self.count_usage = count_usage
        self.device = device
        self._k = self._s = self._e = None
        self._use = self._life = None
        self._r16 = None             # fp16 filter operand rows of the keys (288 B per element, csrc/affinity_common.hpp), kept in step with _k / _s
        self._v = []                 # one _Arena [n_obj_in_group, cap, Cv] per object group
        self.obj_groups = []
        self.all_objects = []
        self._engaged = False
        self._has_s = self._has_e = False

    # ---- kernel-facing views -------------------------------------------------------------------
    def key_rows(self, start=0):
        return self._k.rows(start)

    def shrinkage_rows(self, start=0):
        return self._s.rows(start) if self._has_s else None

    def selection_rows(self, start=0):
        return self._e.rows(start) if self._has_e else None

    def rows16(self, start=0):
        """Operand rows of the readout's fp16 filter for elements [start, size): derived from (key, shrinkage) when elements are
        added / replaced, moved with them when the store is sieved - never per frame."""
        return self._r16.rows(start) if self._r16 is not None else None

    def _refresh_rows16(self, a, b):
        if self._r16 is not None and b > a:
            ops.affinity_rows16(self._k.rows(a, b), self._s.rows(a, b) if self._has_s else None, self._r16.rows(a, b))

    def value_rows(self, gi):
        """[n_obj_in_group, v_size, Cv]"""
        return self._v[gi].rows()

    # ---- reference-shaped read-only views --------------------------------------------------------
    @property
    def key(self):
        return self._k.rows().t().unsqueeze(0) if self._engaged else None

    @property
    def shrinkage(self):
        return self._s.rows().view(1, 1, -1) if self._engaged and self._has_s else None

    @property
    def selection(self):
        return self._e.rows().t().unsqueeze(0) if self._engaged and self._has_e else None

    @property
    def value(self):
        return [a.rows().transpose(1, 2) for a in self._v]

    @property
    def use_count(self):
        return self._use.rows().view(1, 1, -1)

    @property
    def life_count(self):
        return self._life.rows().view(1, 1, -1)

    @property
    def size(self):
        return self._k.n if self._engaged else 0

    @property
    def num_groups(self):
        return len(self._v)

    def get_v_size(self, ni: int):
        return self._v[ni].n

    def engaged(self):
        return self._engaged

    # ---- mutation --------------------------------------------------------------------------------
    def _init(self, ck, device, has_s, has_e):
        self.device = device
        self._k = _Arena([], ck, device)
        self._s = _Arena([], 0, device)
        self._e = _Arena([], ck, device)
        self._r16 = _Arena([], ops.ROWS16_FLOATS, device) if (ck == 64 and KEEP_ROWS16) else None
        self._has_s, self._has_e = has_s, has_e
        if self.count_usage:
            self._use = _Arena([], 0, device)
            self._life = _Arena([], 0, device)
        self._engaged = True

    def add(self, key, value, shrinkage, selection, objects):
        """kv_memory_store.py:36-94 with row-major inputs: key [n,Ck], shrinkage [n]|None, selection [n,Ck]|None,
        value [K,n,Cv] tensor (objects given) or list of per-group [n_obj,n_g,Cv]|None (long-term).  Returns the
        frame position of the added block."""
        n = key.shape[0]
        if not self._engaged:
            self._init(key.shape[1], key.device, shrinkage is not None, selection is not None)
        self._k.append(key, n)
        if shrinkage is not None:
            self._s.append(shrinkage, n)
        if selection is not None:
            self._e.append(selection, n)
        if self._r16 is not None and n:
            self._r16.reserve(n)                       # then derive from the rows just stored
            self._refresh_rows16(self._k.n - n, self._k.n)
        if self.count_usage:
            self._use.append(0.0, n)
            self._life.append(1e-7, n)        # kv_memory_store.py:37-38
        if objects is not None:
            assert isinstance(value, torch.Tensor)
            remaining = [o - 1 for o in objects]       # background is not part of value
            for gi, group in enumerate(self.obj_groups):
                for o in group:
                    remaining.remove(o)                # raises if object groups overlap
                self._v[gi].append(self._pick(value, group), n)
            if remaining:
                group = list(remaining)
                arena = _Arena([len(group)], value.shape[2], value.device)
                arena.append(self._pick(value, group), n)
                self._v.append(arena)
                self.obj_groups.append(group)
                self.all_objects.extend(group)
                assert sorted(self.all_objects) == self.all_objects, 'Objects MUST be inserted in sorted order '
        else:
            assert isinstance(value, list)
            for gi, gv in enumerate(value):
                if gv is None:
                    continue
                if gi < self.num_groups:
                    self._v[gi].append(gv, gv.shape[1])
                else:
                    arena = _Arena([gv.shape[0]], gv.shape[2], gv.device)
                    arena.append(gv, gv.shape[1])
                    self._v.append(arena)
        return int((self._k.n + 1e-9) // (n + 1e-9)) - 1 if n else -1

    @staticmethod
    def _pick(value, group):
        """value[group] without a gather when the group is all objects in order."""
        if value.shape[1] == 0 or group == list(range(value.shape[0])):
            return value
        return value[group]

    def update_usage_from(self, w, idx, first):
        """use_count += usage[first:first+size]; life_count += 1 (kv_memory_store.py:96-103), usage taken from
        the sparse affinity (w, idx) of the readout."""
        if not self.count_usage or self.size == 0:
            return
        ops.usage_update(w, idx, first, self.size, self._use.rows(), self._life.rows())

    def replace_at(self, start_pos: int, key, value, shrinkage=None, selection=None):
        """kv_memory_store.py:105-118; value [K,n,Cv] row-major."""
        n = key.shape[0]
        a, b = start_pos * n, (start_pos + 1) * n
        self._k.rows(a, b).copy_(key)
        for gi in range(self.num_groups):
            # the reference assigns value[gi] (object gi, broadcast over the group's objects) - kept as is
            self._v[gi].rows(a, b).copy_(value[gi])
        if self._has_s and shrinkage is not None:
            self._s.rows(a, b).copy_(shrinkage)
        if self._has_e and selection is not None:
            self._e.rows(a, b).copy_(selection)
        self._refresh_rows16(a, b)

    def remove_at(self, start: int, elem_size: int):
        """kv_memory_store.py:120-123."""
        self.sieve_by_range(start, start + elem_size, min_size=0)

    @staticmethod
    def _resolve(n, start, end):
        s = start if start >= 0 else max(n + start, 0)
        s = min(s, n)
        if end == 0:
            return s, n            # "negative zero": sieve everything from start (kv_memory_store.py:131-133)
        e = end if end >= 0 else max(n + end, 0)
        return s, min(max(e, s), n)

    def sieve_by_range(self, start: int, end: int, min_size: int):
        """Keep elements OUTSIDE [start, end) (python slice semantics per tensor), kv_memory_store.py:125-158."""
        def sieve(arena):
            s, e = self._resolve(arena.n, start, end)
            arena.keep([(0, s), (e, arena.n)])
        sieve(self._k)
        if self._r16 is not None:
            sieve(self._r16)
        if self.count_usage:
            sieve(self._use); sieve(self._life)
        if self._has_s:
            sieve(self._s)
        if self._has_e:
            sieve(self._e)
        for gi in range(self.num_groups):
            if self._v[gi].n >= min_size:
                sieve(self._v[gi])

    def remove_obsolete_features(self, max_size: int):
        """Least-used eviction, kv_memory_store.py:160-181: drop every element whose usage is <= the
        (size-max_size)-th smallest usage."""
        if self.num_groups > 1:
            raise NotImplementedError('The current data structure does not support feature removal with '
                                      'multiple object groups (e.g., some objects start to appear later in the video)')
        k = self.size - max_size
        usage = self.get_usage_rows()
        vals, _ = ops.topk_1d(usage, k, largest=False)
        index, count = ops.select_greater(usage, vals[k - 1:k])
        m = int(count.item())
        index = index[:m]
        for arena in [self._k, self._use, self._life] + ([self._s] if self._has_s else []) + \
                     ([self._e] if self._has_e else []) + ([self._r16] if self._r16 is not None else []) + list(self._v):
            arena.take(index)

    def get_usage_rows(self):
        if not self.count_usage:
            raise RuntimeError('I did not count usage!')
        return ops.usage_ratio(self._use.rows(), self._life.rows())

    def get_usage(self):
        return self.get_usage_rows().view(1, 1, -1)
