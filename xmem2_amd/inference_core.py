"""Per-frame state machine: the reference's ``InferenceCore`` surface (inference/inference_core.py:11-185)
driving the HIP kernels.  ``step()``, ``put_to_permanent_memory()``, ``clear_memory()``, ``update_config()``,
``set_all_labels()``, ``encode_frame_key()``, ``remove_from_permanent_memory()`` and
``permanent_memory_frames`` keep the reference's signatures and semantics; tensors enter and leave in the
reference's conventions (image ``3 x H x W`` float32 normalised, mask ``K x H x W``, prob ``(K+1) x H x W``).
"""
import collections
import functools
import os
import weakref

import torch

from . import ops
from .memory_manager import MemoryManager
from .tensor_util import pad_amounts


def _on_network_device(fn):
    """Run the method with the network's GPU as torch's current device: the kernels are launched on torch's CURRENT stream,
    so a core whose network lives on cuda:1 must not enqueue on cuda:0's stream when the caller never called set_device."""
    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        # ONE host thread drives the kernels of a process at a time (un-scoped scratch and the precision mode are per process,
        # ops.workspace): the whole call holds the driver lock - captured stages, graph replays and the eager readout between them alike -
        # so that threads sharing a process take turns call by call instead of interleaving inside one.
        with ops._DRIVER:
            dev = getattr(self.network, 'device', None)
            if dev is None or dev.type != 'cuda' or torch.cuda.current_device() == (dev.index if dev.index is not None else torch.cuda.current_device()):
                return fn(self, *args, **kwargs)
            with torch.cuda.device(dev):
                return fn(self, *args, **kwargs)
    return wrapper


def _release_core(network, uid, inflight):
    """Finalizer of an InferenceCore: work it enqueued on the side / readout streams (a batched key pass into the network's shared
    key-stage buffers, a readout into a decoder stage's static input) must be over before its owner token is recycled."""
    for ev in inflight.values():
        if ev is not None:
            try:
                ev.synchronize()
            except Exception:
                pass
    network.release_owner(uid)


def _net_stream(net, name, make):
    """Side streams live on the NETWORK, not on the core: the key-stage buffers of a slot and the decoder stages' static inputs belong
    to the network, so successive (or concurrent) cores on one network are stream-ordered against each other on them."""
    d = net.__dict__.setdefault('_core_streams', {})
    if name not in d:
        d[name] = make()
    return d[name]


class InferenceCore:
    def __init__(self, network, config):
        self.config = config
        self.network = network
        self._read_config(config)
        self.clear_memory()
        self.all_labels = None
        # frame pipelining: prefetch_key() runs the key encoder of the NEXT frame on a side stream (own graph slot and
        # scratch) while the current frame's readout / decoder occupy the main stream
        self._side = None
        self._pfq = collections.deque()      # prefetched frames in the order step() will consume them
        self._group_free = {}                # buffer group -> event after which the side stream may overwrite it
        self._group_parity = {}
        # early readout: the memory readout of the NEXT prefetched frame (it needs that frame's key and the memory, not this frame's
        # decoder) is enqueued on a third stream as soon as this frame's own readout / memory insertion is done, and runs under this
        # frame's decoder.  Same kernels on the same operands; consumed by the next step() only if the memory is unchanged.
        # DEFAULT since round 6 (XMEM_EARLY_READOUT=0 or core.early_readout = False turns it off): bit-identical to the in-step order in every
        # test (tests/test_gpu_stream_b32.py, test_gpu_c4_stream.py, test_gpu_c5_stream.py, test_gpu_prefetch_sync.py), +4 % at B32.  Round 5
        # kept it opt-in because `bench.py`'s parity leg showed one wrong stream with it; that was a caller-side race of the leg itself -
        # device inputs hinted before the stream that cloned them had finished (see prefetch_keys) - which the extra stream only made visible
        # (profiles/r06_early_readout_root_cause.txt).
        self._ro_stream = None
        self._early = None
        self.early_readout = os.environ.get('XMEM_EARLY_READOUT', '1') != '0'
        # latest events of this core's work on the side / readout streams: the finalizer waits for them before the owner token (and with
        # it the captured decoder stages and their static buffers) goes to the next core
        self._inflight = {'side': None, 'early': None}
        # owner token of this core's captured decoder stages (they update ITS hidden state in place); recycled when the core dies
        if hasattr(network, 'acquire_owner'):
            self._uid = network.acquire_owner()
            weakref.finalize(self, _release_core, network, self._uid, self._inflight)
        else:
            self._uid = id(self)
        # warm-up on the network's own device (the reference hard-codes cuda:0, inference_core.py:26)
        if getattr(network, 'device', None) is not None and network.device.type == 'cuda':
            with torch.cuda.device(network.device):
                self.network._need_weights()
                self.network._encode_key_eager(torch.zeros((1, 64, 64, 4), device=network.device), True, True)

    def _read_config(self, config):
        self.mem_every = config['mem_every']
        self.deep_update_every = config['deep_update_every']
        self.enable_long_term = config['enable_long_term']
        self.deep_update_sync = (self.deep_update_every < 0)   # < 0: deep update synchronised with memory frames

    def clear_memory(self, keep_permanent=False):
        """inference_core.py:28-38."""
        self.curr_ti = -1
        self.last_mem_ti = 0
        if not self.deep_update_sync:
            self.last_deep_update_ti = -self.deep_update_every
        self._retire_early()                          # a readout enqueued ahead reads the stores that are about to go
        self.memory = self.memory.copy_perm_mem_only() if keep_permanent else MemoryManager(config=self.config)

    def update_config(self, config):
        """inference_core.py:40-47."""
        self._retire_early()
        self._read_config(config)
        self.memory.update_config(config)

    def set_all_labels(self, all_labels):
        self._retire_early()
        self.all_labels = all_labels

    # ---- helpers ---------------------------------------------------------------------------------
    def _pack(self, image, out=None):
        """3 x H x W float (the reference's normalised frame, inference_core.py:73-74) or - ingest on the device,
        SURVEY 8f rank 2 - the decoded H x W x 3 uint8 frame, normalised here as video_reader.py:61-76 does."""
        u8 = image.dtype == torch.uint8 and image.dim() == 3 and image.shape[2] == 3
        if not u8 and (image.dim() != 3 or image.shape[0] != 3):
            raise NotImplementedError('image must be 3 x H x W (float) or H x W x 3 (uint8)')
        H, W = (image.shape[0], image.shape[1]) if u8 else image.shape[-2:]
        lw, uw, lh, uh = pad_amounts(H, W, 16)
        self.pad = (lw, uw, lh, uh)
        if u8:
            return ops.pack_image_u8(image, H + lh + uh, W + lw + uw, lh, lw, out=out), (H, W), (H + lh + uh, W + lw + uw)
        if image.dtype != torch.float32:
            image = image.float()
        return ops.pack_image(image, H + lh + uh, W + lw + uw, lh, lw, out=out), (H, W), (H + lh + uh, W + lw + uw)

    def _pad_mask(self, mask, hw, hw_p):
        mask = mask.to(dtype=torch.float32)
        if hw == hw_p:
            return mask.contiguous()
        out = torch.zeros((mask.shape[0],) + tuple(hw_p), dtype=torch.float32, device=mask.device)
        lw, uw, lh, uh = self.pad
        out[:, lh:lh + hw[0], lw:lw + hw[1]].copy_(mask)
        return out

    def _key_views(self, key, shrinkage, selection, h, w):
        ck = key.shape[1]
        k = key.view(1, h, w, ck).permute(0, 3, 1, 2)
        s = shrinkage.view(1, 1, h, w) if shrinkage is not None else None
        e = selection.view(1, h, w, ck).permute(0, 3, 1, 2) if selection is not None else None
        return k, s, e

    @_on_network_device
    def encode_frame_key(self, image):
        """inference_core.py:53-61."""
        image4, _, _ = self._pack(image)
        key, shr, sel, f16, _, _ = self.network.encode_key_nhwc(image4, need_sk=True, need_ek=True)
        return tuple(v.clone() for v in self._key_views(key, shr, sel, f16.shape[1], f16.shape[2]))

    def prefetch_key(self, image, inputs_complete=False):
        """Enqueue the key encoder for the NEXT frame (see `prefetch_keys`); returns the device tensor to hand to
        the next `step()`."""
        return self.prefetch_keys([image], inputs_complete=inputs_complete)[0]

    @_on_network_device
    def prefetch_keys(self, images, inputs_complete=False):
        """Enqueue ONE batched key-encoder pass for the next `len(images)` frames on a side stream.

        The key encoder depends on nothing but the image, so a streaming caller that already holds the coming frames
        (the reference's DataLoader does, inference/run_on_video.py:80-92) can hint them: the pass runs in its own HIP
        graph and scratch while the current frames are still being read out / decoded on the main stream, and a batch
        of frames amortises the launch-bound 1/16-resolution layers (1.16 -> 0.82 ms per frame at batch 4).  `images`:
        3 x H x W float32 (or decoded H x W x 3 uint8) tensors of one shape, pinned-CPU or device tensors.  Returns the
        device tensors to pass to the following `step()` calls IN ORDER; a `step()` on any other tensor simply drops
        the pending hints.  Not part of the reference surface: `step()` computes the same function without it.

        DEVICE inputs are read on the side stream.  By default the side stream first waits for everything enqueued so far on the
        CALLER's current stream, so a tensor the caller has just produced there (a `.clone()`, a resize, an H2D copy) is complete when
        the pass reads it.  `inputs_complete=True` skips that wait for callers whose inputs were finished long ago (resident clips):
        the pass may then start under the frame that is still being decoded.  (Round 6: bench.py's parity leg handed over clones that
        were still being written on the main stream - the 'unexplained wrong stream' of round 5)"""
        net = self.network
        images = list(images)
        if not images:
            return []
        if not (getattr(net, 'use_graphs', False) and net.device.type == 'cuda') or ops.eager_only():
            return [im.to(net.device) if not im.is_cuda else im for im in images]
        if any(tuple(im.shape) != tuple(images[0].shape) for im in images):
            raise ValueError('prefetch_keys: all frames of a batch must have the same shape')
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = _net_stream(net, 'side', lambda: ops.side_stream(net.device))
        if not inputs_complete and any(im.is_cuda for im in images):
            self._side.wait_stream(main)                         # producers of the inputs on the caller's stream
        B = len(images)
        par = self._group_parity.get(B, 1) ^ 1                   # two buffer groups per batch size, used alternately
        # (the group - a key-stage graph and its static buffers - belongs to THIS core's owner token: two live cores hinting on one
        # network never overwrite each other's unconsumed frames; a recycled token takes over the dead core's captured stages)
        gid = ('g', B, par, self._uid)
        if any(e['gid'] == gid for e in self._pfq):              # unconsumed frames still live there: give them up
            self._drop_prefetch()
        self._group_parity[B] = par
        if self._group_free.get(gid) is not None:                # main-stream readers of the group's buffers are done
            self._side.wait_event(self._group_free[gid])
        saved_pad = getattr(self, 'pad', None)
        with torch.cuda.stream(self._side):
            devs = [im.to(net.device, non_blocking=True) if not im.is_cuda else im for im in images]
            u8 = devs[0].dtype == torch.uint8
            H0, W0 = (devs[0].shape[0], devs[0].shape[1]) if u8 else devs[0].shape[-2:]
            lw_, uw_, lh_, uh_ = pad_amounts(H0, W0, 16)
            image4 = torch.empty((B, H0 + lh_ + uh_, W0 + lw_ + uw_, 4), dtype=torch.float32, device=net.device)
            packed = [self._pack(d, out=image4[i:i + 1]) for i, d in enumerate(devs)]     # straight into the batched buffer
            pad = self.pad
            outs = net.encode_key_nhwc(image4, need_sk=True, need_ek=True, with_skips=True, slot=gid, inline_skips=True)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._inflight['side'] = ev
        if saved_pad is not None:
            self.pad = saved_pad
        for t in [image4] + devs:
            t.record_stream(main)
        key, shr, sel, f16, f8, f4 = outs[:6]
        extras = outs[6]                                        # skip8, skip4 [, the decoder fuser's f16 half: conv1@x, downsample@x]
        n = f16.shape[1] * f16.shape[2]
        _, hw, hw_p = packed[0]
        for i, d in enumerate(devs):
            self._pfq.append(dict(ptr=d.data_ptr(), shape=tuple(d.shape), image4=image4[i:i + 1], hw=hw, hw_p=hw_p, pad=pad,
                                  outs=(key[i * n:(i + 1) * n], shr[i * n:(i + 1) * n], sel[i * n:(i + 1) * n],
                                        f16[i:i + 1], f8[i:i + 1], f4[i:i + 1], tuple(t[i:i + 1] for t in extras)),
                                  event=ev, slot=(gid, i), gid=gid, keep=d))
        return devs

    # ---- the per-frame step ----------------------------------------------------------------------
    @_on_network_device
    def step(self, image, mask=None, valid_labels=None, end=False, manually_curated_masks=False,
             disable_memory_updates=False, do_not_add_mask_to_memory=False, return_key_and_stuff=False):
        """inference_core.py:62-152.  image: 3*H*W, mask: num_objects*H*W or None -> prob (K+1)*H*W."""
        self.curr_ti += 1
        pf = None
        if self._pfq:
            head = self._pfq[0]
            if image.is_cuda and head['ptr'] == image.data_ptr() and head['shape'] == tuple(image.shape):
                pf = self._pfq.popleft()
            else:
                self._drop_prefetch()                               # stale hints: give them up
        if pf is not None:
            image4, hw, hw_p = pf['image4'], pf['hw'], pf['hw_p']
            self.pad = pf['pad']
            slot = pf['slot']
        else:
            image4, hw, hw_p = self._pack(image)
            slot = 0                                                # main-stream buffers: stream order protects them
        if manually_curated_masks:
            is_mem_frame = (mask is not None) and (not end)
        else:
            is_mem_frame = ((self.curr_ti - self.last_mem_ti >= self.mem_every) or (mask is not None)) and (not end)
        is_ignore = do_not_add_mask_to_memory
        need_segment = (valid_labels is None) or (len(self.all_labels) != len(valid_labels))
        if self._early is not None and not (need_segment and pf is not None and self._early['pf'] is pf):
            self._retire_early()          # not this step's to consume: wait for it before this step touches the memory / key buffers
        is_deep_update = ((self.deep_update_sync and is_mem_frame) or
                          (not self.deep_update_sync and self.curr_ti - self.last_deep_update_ti >= self.deep_update_every)
                          ) and (not end)
        is_normal_update = (not self.deep_update_sync or not is_deep_update) and (not end)

        net, mem = self.network, self.memory
        if pf is not None:
            torch.cuda.current_stream().wait_event(pf['event'])     # the side stream finished this frame's key encoder
            enc = pf['outs']
        else:
            enc = net.encode_key_nhwc(image4, need_sk=True, need_ek=(self.enable_long_term or need_segment),
                                      with_skips=need_segment, slot=slot)
        key, shrinkage, selection, f16, f8, f4 = enc[:6]
        skips = enc[6] if len(enc) > 6 else None
        h, w = f16.shape[1], f16.shape[2]

        if disable_memory_updates:
            is_normal_update = is_deep_update = is_mem_frame = False
            self.curr_ti -= 1

        prob = prob_padded = None
        ro_done = None
        if need_segment:
            hidden = mem.get_hidden()
            K = hidden.shape[0]
            cat16 = net.new_decoder_input(K, h, w, f16.device, slot=slot, owner=self._uid, h_out=is_normal_update,
                                          has_skips=skips is not None, out_hw=hw, pad_tl=(self.pad[2], self.pad[0]))
            ld = cat16.shape[3]
            early = self._take_early(pf, mem, K, cat16)
            if early is not None:
                # this frame's readout ran ahead, under the previous frame's decoder: only its usage updates are still due
                if early['cat16'].data_ptr() != cat16.data_ptr():        # another decoder variant than predicted (last frame, ...)
                    ops.copy_channels(early['cat16'], cat16, 1024, c=net.value_dim, src_off=1024)
                if not disable_memory_updates:
                    mem.apply_usage(early['pending'])
            else:
                mem.match_memory_rows(key, selection, cat16, ld, h * w * ld, out_off=1024,
                                      disable_usage_updates=disable_memory_updates)
            ro_done = torch.cuda.Event()
            ro_done.record()
            new_hidden, prob, prob_padded = net.segment_nhwc(f16, f8, f4, cat16, hidden, hw, (self.pad[2], self.pad[0]),
                                                             h_out=is_normal_update, skips=skips, slot=slot, owner=self._uid)
            if is_normal_update:
                mem.set_hidden(new_hidden)

        if mask is not None:
            mask_p = self._pad_mask(mask, hw, hw_p)
            if prob_padded is not None:
                bits = (1 << mask_p.shape[0]) - 1
                if valid_labels is not None:
                    bits = 0
                    for i in range(mask_p.shape[0]):
                        if (i + 1) in valid_labels:
                            bits |= (1 << i)
                mask_p = ops.merge_masks(prob_padded[1:], mask_p, bits)
            prob_padded = ops.aggregate_masks(mask_p)
            lw, _, lh, _ = self.pad
            prob = prob_padded[:, lh:lh + hw[0], lw:lw + hw[1]]
            if not disable_memory_updates:
                mem.create_hidden_state(len(self.all_labels), hw_shape=(h, w), device=f16.device)

        if is_mem_frame:
            value, hidden = net.encode_value_nhwc(image4, f16, mem.get_hidden(), prob_padded[1:],
                                                  is_deep_update=is_deep_update, slot=slot)
            if value.dtype != torch.float32:          # fp16 loop: the memory keeps fp32 values (as torch.cat promotes the reference's
                v32 = torch.empty(value.shape, dtype=torch.float32, device=value.device)   # fp16 values onto its fp32-preloaded
                value = ops.copy_channels(value, v32, 0)                                     # stores, kv_memory_store.py:36-94)
            mem.add_memory(key, shrinkage, value.view(value.shape[0], h * w, value.shape[3]), self.all_labels,
                           selection=selection if self.enable_long_term else None, ignore=is_ignore, hw_shape=(h, w))
            self.last_mem_ti = self.curr_ti
            if is_deep_update:
                mem.set_hidden(hidden)
                self.last_deep_update_ti = self.curr_ti

        if pf is not None:
            done = torch.cuda.Event()
            done.record()
            self._group_free[pf['gid']] = done        # latest main-stream reader of that group's key-encoder buffers
        if self.early_readout and need_segment and self._pfq and mask is None and not end:
            # (a memory frame's value encoder and insertion are enqueued by now: the next readout must see them)
            self._enqueue_early_readout(ro_done if (ro_done is not None and not is_mem_frame) else None)
        if return_key_and_stuff:
            views = self._key_views(key, shrinkage, selection, h, w)
            return (prob,) + tuple(v.clone() if v is not None else None for v in views)   # caller-owned copies
        return prob

    # ---- early readout -----------------------------------------------------------------------------
    def _enqueue_early_readout(self, after):
        """match_memory of the next prefetched frame on the readout stream.  `after`: main-stream event the readout depends on
        (this frame's readout: the hint, the usage order) - None = everything enqueued on the main stream so far."""
        net, mem = self.network, self.memory
        nxt = self._pfq[0]
        hidden = mem.get_hidden()
        if hidden is None or not (getattr(net, 'use_graphs', False) and net.device.type == 'cuda') or ops.eager_only():
            return
        key, _, selection, f16 = nxt['outs'][0], nxt['outs'][1], nxt['outs'][2], nxt['outs'][3]
        h, w, K = f16.shape[1], f16.shape[2], hidden.shape[0]
        main = torch.cuda.current_stream()
        if self._ro_stream is None:
            self._ro_stream = _net_stream(net, 'readout', lambda: ops.readout_stream(net.device))
        if after is None:
            after = torch.cuda.Event()
            after.record(main)
        R = self._ro_stream
        # the decoder variant the next step() will most likely run (no mask, not the last frame): its static input buffer
        nxt_mem = (self.curr_ti + 1 - self.last_mem_ti >= self.mem_every)
        nxt_deep = (self.deep_update_sync and nxt_mem) or \
                   (not self.deep_update_sync and self.curr_ti + 1 - self.last_deep_update_ti >= self.deep_update_every)
        h_out = (not self.deep_update_sync) or (not nxt_deep)
        pending = []
        # only into the STATIC input buffer of a decoder stage that is already captured for that slot: while an owner's stages are
        # still being captured (its first frames) the readout stays in its step - no scratch copies of the decoder input on this stream
        cat16 = net.new_decoder_input(K, h, w, f16.device, slot=nxt['slot'], owner=self._uid, h_out=h_out,
                                      has_skips=len(nxt['outs']) > 6 and nxt['outs'][6] is not None, static_only=True,
                                      out_hw=nxt['hw'], pad_tl=(nxt['pad'][2], nxt['pad'][0]))
        if cat16 is None:
            return
        R.wait_event(nxt['event'])                    # the side stream finished that frame's key encoder
        R.wait_event(after)
        with torch.cuda.stream(R), ops.ws_scope(f'@early#{self._uid}#'):
            ld = cat16.shape[3]
            mem.match_memory_rows(key, selection, cat16, ld, h * w * ld, out_off=1024, defer_usage=pending)
            done = torch.cuda.Event()
            done.record(R)
        self._inflight['early'] = done
        self._early = dict(pf=nxt, mem=mem, version=mem.version, K=K, cat16=cat16, pending=pending, done=done)

    def _take_early(self, pf, mem, K, cat16):
        """The readout enqueued ahead for this frame, or None (none was, or the frame / memory / object count is not what it read).
        Either way the main stream first waits for it: it uses scratch and buffers the main stream is about to touch."""
        e, self._early = self._early, None
        if e is None:
            return None
        main = torch.cuda.current_stream()
        main.wait_event(e['done'])
        ok = pf is not None and e['pf'] is pf and e['mem'] is mem and e['version'] == mem.version and e['K'] == K \
            and tuple(e['cat16'].shape) == tuple(cat16.shape) and e['cat16'].dtype == cat16.dtype
        for _, w_, idx_, _ in e['pending']:           # allocated on the readout stream, consumed (usage update) on this one
            w_.record_stream(main); idx_.record_stream(main)
        return e if ok else None

    def _retire_early(self):
        e = getattr(self, '_early', None)
        if e is not None:
            torch.cuda.current_stream().wait_event(e['done'])
            self._early = None

    def cancel_prefetch(self):
        """Forget pending `prefetch_keys` hints (the next `step()` runs its own key encoder)."""
        self._drop_prefetch()

    def _drop_prefetch(self):
        self._retire_early()
        while self._pfq:
            e = self._pfq.popleft()
            torch.cuda.current_stream().wait_event(e['event'])

    @_on_network_device
    def put_to_permanent_memory(self, image, mask, ti=None):
        """inference_core.py:154-179.  Always full fp32, also when the network runs the frame loop in its reduced-precision mode
        (the reference preloads outside its autocast region, inference/run_on_video.py:59-66 vs :76)."""
        self._drop_prefetch()
        net, mem = self.network, self.memory
        prev, net._call_precision = getattr(net, '_call_precision', None), 'fp32'
        try:
            return self._put_to_permanent_memory(image, mask, ti)
        finally:
            net._call_precision = prev

    def _put_to_permanent_memory(self, image, mask, ti):
        image4, hw, hw_p = self._pack(image)
        net, mem = self.network, self.memory
        key, shrinkage, selection, f16, _, _ = net.encode_key_nhwc(image4, need_sk=True, need_ek=True)
        h, w = f16.shape[1], f16.shape[2]
        prob_padded = ops.aggregate_masks(self._pad_mask(mask, hw, hw_p))
        mem.create_hidden_state(len(self.all_labels), hw_shape=(h, w), device=f16.device)
        value, _ = net.encode_value_nhwc(image4, f16, mem.get_hidden(), prob_padded[1:], is_deep_update=False)
        value = value.view(value.shape[0], h * w, value.shape[3])
        is_update = mem.frame_already_saved(ti)
        sel = selection if self.enable_long_term else None
        if is_update:
            mem.update_permanent_memory(ti, key, shrinkage, value, selection=sel)
        else:
            mem.add_memory(key, shrinkage, value, self.all_labels, selection=sel, permanent=True, ti=ti, hw_shape=(h, w))
        return is_update

    @_on_network_device
    def put_many_to_permanent_memory(self, images, masks):
        """Several annotated (or augmented) frames into the permanent memory through ONE batched key-encoder pass and ONE batched
        value-encoder pass - the preload of inference/run_on_video.py:59-66 + :231-242 (12 frames per annotation with the
        'best_all' augmentations) without 12 sequential passes.  Same result as `put_to_permanent_memory(image, mask)` frame by
        frame in list order (to the round-off of batched convolution plans); always full fp32; frames of one shape, uint8
        H x W x 3 or float 3 x H x W, masks K x H x W.  Not part of the reference surface."""
        images, masks = list(images), list(masks)
        if not images or len(images) != len(masks):
            raise ValueError('put_many_to_permanent_memory: one mask per image')
        if any(tuple(im.shape) != tuple(images[0].shape) for im in images) or any(tuple(m.shape) != tuple(masks[0].shape) for m in masks):
            raise ValueError('put_many_to_permanent_memory: all frames (and all masks) must have one shape')
        self._drop_prefetch()
        net, mem = self.network, self.memory
        dev = net.device
        images = [im.to(dev) for im in images]
        B = len(images)
        u8 = images[0].dtype == torch.uint8
        H0, W0 = (images[0].shape[0], images[0].shape[1]) if u8 else images[0].shape[-2:]
        lw, uw, lh, uh = pad_amounts(H0, W0, 16)
        image4 = torch.empty((B, H0 + lh + uh, W0 + lw + uw, 4), dtype=torch.float32, device=dev)
        for i, im in enumerate(images):
            _, hw, hw_p = self._pack(im, out=image4[i:i + 1])
        net._need_weights()
        with ops.precision('fp32'):
            key, shrinkage, selection, f16, _, _ = net._encode_key_eager(image4, True, True)
        h, w = f16.shape[1], f16.shape[2]
        n = h * w
        probs = [ops.aggregate_masks(self._pad_mask(m.to(dev), hw, hw_p))[1:] for m in masks]
        value = net.encode_value_frames_nhwc(image4, f16, probs)
        K = probs[0].shape[0]
        for b in range(B):
            mem.create_hidden_state(len(self.all_labels), hw_shape=(h, w), device=dev)
            v = value[b * K:(b + 1) * K].reshape(K, n, value.shape[3])
            sel = selection[b * n:(b + 1) * n] if self.enable_long_term else None
            mem.add_memory(key[b * n:(b + 1) * n], shrinkage[b * n:(b + 1) * n], v, self.all_labels, selection=sel, permanent=True,
                           ti=None, hw_shape=(h, w))
        return B

    def remove_from_permanent_memory(self, frame_idx):
        self._retire_early()                          # a readout enqueued ahead is reading the store that is about to shrink
        self.memory.remove_from_permanent_memory(frame_idx)

    @property
    def permanent_memory_frames(self):
        return list(self.memory.frame_id_to_permanent_mem_idx.keys())
