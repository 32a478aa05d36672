"""XMem network on MI355X: the reference's ``XMem`` surface (model/network.py:17-198) over HIP kernels.

``XMem(config, model_path, map_location, ...)`` keeps the constructor, ``encode_key`` / ``encode_value`` /
``segment`` / ``load_weights`` and the 412-name state_dict of the reference, so an upstream ``XMem.pth``
loads unchanged.  Internally nothing is an nn.Module: at load time every convolution is re-laid-out to
``[Cout][KH][KW][Cin]`` with its eval-mode BatchNorm folded to a per-channel (scale, shift) epilogue
(ATen's own alpha/beta form), and every activation lives in NHWC so that the implicit-GEMM kernel
(csrc/conv_mfma.hip) reads K-contiguous operands.

Public methods take / return NCHW-shaped tensors like the reference (zero-copy permuted views of the
NHWC buffers); the ``*_nhwc`` methods are the hot path used by ``InferenceCore``.
"""
import collections
import os
import warnings
import weakref

import torch

from . import ops
from .arch import BN_EPS, infer_dims, state_dict_spec
from .ops import ConvWeights


def _pad4(n):
    return (n + 3) // 4 * 4


def _padc(n):
    """channel padding of a buffer the network allocates: 4 floats or (fp16 loop) 8 halfs = one 16-byte operand chunk"""
    return (n + 7) // 8 * 8 if ops.act_dtype() == torch.float16 else _pad4(n)


class XMem:
    def __init__(self, config, model_path=None, map_location=None, pretrained_key_encoder=True, pretrained_value_encoder=True):
        """Same signature as model/network.py:18.  `pretrained_*` are accepted for compatibility; torchvision
        ImageNet weights cannot be fetched offline, so a checkpoint (or load_weights) is the only weight source."""
        self.single_object = config.get('single_object', False)
        self.device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
        self._sd = None
        self._w = {}
        self._cbam = {}
        # launch-bound stages (61 convolutions + ~40 small kernels per frame) are replayed as HIP graphs
        self.use_graphs = os.environ.get('XMEM_HIP_GRAPHS', '1') != '0'
        # 'fp32' (default: the parity contract) | 'fp16' (opt-in: the fp16 loop - half activations in HBM, half-operand convolutions
        # on the fp16 MFMA with fp32 accumulation: the counterpart of the reference's autocast loop, run_on_video.py:76; the
        # permanent-memory preload stays fp32 as in run_on_video.py:59-66) | 'fp16w' / 'fp32x' (experiments, ops.PRECISIONS)
        self.precision = config.get('precision', os.environ.get('XMEM_PRECISION', 'fp32'))
        if self.precision not in ops.PRECISIONS:
            raise ValueError(f"config['precision'] must be one of {ops.PRECISIONS}, got {self.precision!r}")
        self._stages = collections.OrderedDict()     # captured stages, least recently replayed first
        self._zeros = {}
        self._owner_free, self._owner_next = [], 0   # owner tokens of the cores driving this network (see acquire_owner)
        self._call_precision = None      # per-call override (InferenceCore preloads permanent memory in fp32)
        self.max_stages = int(os.environ.get('XMEM_MAX_STAGES', '160'))   # captured HIP-graph stages kept before the cache is dropped
        # the decoder's skip convolutions depend only on f8 / f4: inside the captured key-encoder graph they run on a
        # forked stream next to the small-grid layer2 / layer3 kernels.  Measured neutral on MI355X (A/B on one box:
        # 272 vs 273 fps), so it is off by default (XMEM_OVERLAP=1 enables it).
        self.overlap_skips = os.environ.get('XMEM_OVERLAP', '0') != '0'
        self.share_x = os.environ.get('XMEM_SHARE_X', '1') != '0'     # several objects: convolve the shared f16 half of the fusers once
        self.prefuse_x = os.environ.get('XMEM_PREFUSE_X', '1') != '0'  # prefetched frames: the decoder fuser's f16 half in the batched key pass
        self._side = None
        # GroupResBlocks with a downsample branch (fuser block1, up_16_8.out_conv): conv1 and the downsample convolution read the same
        # tensor and are independent - inside a captured stage the downsample CAN run on a forked stream beside conv1 (same kernels,
        # bit-identical results, tests/test_gpu_network.py).  Measured on MI355X (round 5, profiles/r05_branch_overlap_ab.txt): it LOSES -
        # B32 614 -> 561 frames/s, C3 283 -> 276, C4 209 -> 199: with the key pass and the early readout on their own streams the chip
        # has no idle share for a fourth queue, and the fork / join edges serialise the captured graph.  Off (XMEM_BRANCH_OVERLAP=1 enables).
        self.branch_overlap = os.environ.get('XMEM_BRANCH_OVERLAP', '0') != '0'
        self.fuse_hidden_update = os.environ.get('XMEM_FUSE_HIDDEN_UPDATE', '1') != '0'   # the three pointwise convolutions of HiddenUpdater as one (A/B knob)
        self.gather_hidden_input = os.environ.get('XMEM_GATHER_HIDDEN_INPUT', '1') != '0'  # ... and their concatenated input built by one launch (A/B knob)
        self._branch = None
        # scratch of the side-stream key-encoder stages is scoped to this instance and released with it
        self._scope = ops.new_scope()
        weakref.finalize(self, ops.release_scope, self._scope)
        weights = self.init_hyperparameters(config, model_path, map_location)
        if weights is not None:
            self.load_weights(weights, init_as_zero_if_needed=True)

    # ---- reference surface --------------------------------------------------------------------
    def init_hyperparameters(self, config, model_path=None, map_location=None):
        """model/network.py:134-182: C_k/C_v/C_h from the checkpoint if given, else config/defaults; writes them back."""
        if model_path is not None:
            weights = torch.load(model_path, map_location=map_location or 'cpu', weights_only=True)
            self.key_dim, self.value_dim, self.hidden_dim = infer_dims(weights)
            self.disable_hidden = self.hidden_dim == 0
        else:
            weights = None
            self.key_dim = config.get('key_dim', 64)
            self.value_dim = config.get('value_dim', 512)
            self.hidden_dim = config.get('hidden_dim', 64)
            self.disable_hidden = self.hidden_dim <= 0
        config['key_dim'], config['value_dim'], config['hidden_dim'] = self.key_dim, self.value_dim, self.hidden_dim
        return weights

    def load_weights(self, src_dict, init_as_zero_if_needed=False):
        """model/network.py:184-198: pads a single-object 4-channel value stem to the 5-channel multi-object one."""
        src_dict = dict(src_dict)
        k = 'value_encoder.conv1.weight'
        if k in src_dict and src_dict[k].shape[1] == 4 and not self.single_object:
            pads = torch.zeros((64, 1, 7, 7), device=src_dict[k].device)
            if not init_as_zero_if_needed:
                torch.nn.init.orthogonal_(pads)
            src_dict[k] = torch.cat([src_dict[k], pads], 1)
        self.load_state_dict(src_dict)

    def load_state_dict(self, sd, strict=True):
        spec = state_dict_spec(self.key_dim, self.value_dim, self.hidden_dim, self.single_object)
        if strict:
            missing = [k for k in spec if k not in sd and not k.endswith('num_batches_tracked')]
            unexpected = [k for k in sd if k not in spec]
            if missing or unexpected:
                raise RuntimeError(f'Error(s) in loading state_dict for XMem: missing {missing[:5]}..., unexpected {unexpected[:5]}...')
            for k, shape in spec.items():
                if k in sd and tuple(sd[k].shape) != tuple(shape) and not (len(shape) == 0 and sd[k].numel() == 1):
                    raise RuntimeError(f'size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {tuple(shape)}')
        self._sd = {k: v.detach().to('cpu') for k, v in sd.items()}
        if self.device.type == 'cuda':
            self._upload()

    def state_dict(self):
        return dict(self._sd) if self._sd is not None else {}

    def to(self, device):
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = device
        if device.type == 'cuda' and self._sd is not None:
            self._upload()
        return self

    def cuda(self, index=None):
        return self.to(torch.device('cuda', index if index is not None else torch.cuda.current_device()))

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError('xmem2_amd.XMem is inference-only (training is out of scope, SURVEY.md section 2 #19)')
        return self

    # ---- weight preparation ---------------------------------------------------------------------
    def _conv_w(self, name, bn=None, stride=1, pad=0):
        sd = self._sd
        w = sd[name + '.weight'].float()
        cout, cin = w.shape[0], w.shape[1]
        wk = w.permute(0, 2, 3, 1)
        if cin % 4:
            wk = torch.nn.functional.pad(wk, (0, _pad4(cin) - cin))
        if bn is not None:
            invstd = 1.0 / torch.sqrt(sd[bn + '.running_var'].float() + BN_EPS)
            scale = invstd * sd[bn + '.weight'].float()
            shift = sd[bn + '.bias'].float() - sd[bn + '.running_mean'].float() * scale
        else:
            scale = torch.ones(cout)
            shift = sd[name + '.bias'].float() if (name + '.bias') in sd else torch.zeros(cout)
        dev = self.device
        return ConvWeights(wk.contiguous().to(dev), scale.contiguous().to(dev), shift.contiguous().to(dev), stride, pad, cin_true=cin)

    def _upload(self):
        sd, W = self._sd, {}
        W['key_encoder.conv1'] = self._conv_w('key_encoder.conv1', 'key_encoder.bn1', 2, 3)

        def bottleneck_stage(prefix, blocks, stride):
            for b in range(blocks):
                p, s = f'{prefix}.{b}', (stride if b == 0 else 1)
                W[p + '.conv1'] = self._conv_w(p + '.conv1', p + '.bn1', 1, 0)
                W[p + '.conv2'] = self._conv_w(p + '.conv2', p + '.bn2', s, 1)
                W[p + '.conv3'] = self._conv_w(p + '.conv3', p + '.bn3', 1, 0)
                if (p + '.downsample.0.weight') in sd:
                    W[p + '.downsample'] = self._conv_w(p + '.downsample.0', p + '.downsample.1', s, 0)

        def basic_stage(prefix, blocks, stride):
            for b in range(blocks):
                p, s = f'{prefix}.{b}', (stride if b == 0 else 1)
                W[p + '.conv1'] = self._conv_w(p + '.conv1', p + '.bn1', s, 1)
                W[p + '.conv2'] = self._conv_w(p + '.conv2', p + '.bn2', 1, 1)
                if (p + '.downsample.0.weight') in sd:
                    W[p + '.downsample'] = self._conv_w(p + '.downsample.0', p + '.downsample.1', s, 0)

        bottleneck_stage('key_encoder.res2', 3, 1)
        bottleneck_stage('key_encoder.layer2', 4, 2)
        bottleneck_stage('key_encoder.layer3', 6, 2)
        W['value_encoder.conv1'] = self._conv_w('value_encoder.conv1', 'value_encoder.bn1', 2, 3)
        if W['value_encoder.conv1'].cin != 8:       # 5 (or 4) input channels live in an 8-channel packed tensor
            cw = W['value_encoder.conv1']
            wk = torch.nn.functional.pad(cw.w, (0, 8 - cw.cin))
            W['value_encoder.conv1'] = ConvWeights(wk.contiguous(), cw.scale, cw.shift, 2, 3, cin_true=cw.cin_true)
        basic_stage('value_encoder.layer1', 2, 1)
        basic_stage('value_encoder.layer2', 2, 2)
        basic_stage('value_encoder.layer3', 2, 2)

        def group_res(p):
            if (p + '.downsample.weight') in sd:
                W[p + '.downsample'] = self._conv_w(p + '.downsample', None, 1, 1)
            W[p + '.conv1'] = self._conv_w(p + '.conv1', None, 1, 1)
            W[p + '.conv2'] = self._conv_w(p + '.conv2', None, 1, 1)

        def fusion(p):
            group_res(p + '.block1')
            group_res(p + '.block2')
            a = p + '.attention'
            dev = self.device
            self._cbam[a] = dict(
                w1=sd[a + '.ChannelGate.mlp.1.weight'].float().contiguous().to(dev),
                b1=sd[a + '.ChannelGate.mlp.1.bias'].float().contiguous().to(dev),
                w2=sd[a + '.ChannelGate.mlp.3.weight'].float().contiguous().to(dev),
                b2=sd[a + '.ChannelGate.mlp.3.bias'].float().contiguous().to(dev),
                sw=sd[a + '.SpatialGate.spatial.conv.weight'].float().reshape(2, 7, 7).contiguous().to(dev),
                sb=sd[a + '.SpatialGate.spatial.conv.bias'].float().contiguous().to(dev))

        fusion('value_encoder.fuser')
        fusion('decoder.fuser')

        def split_shared(name, x_dim=1024):
            """FeatureFusionBlock sees cat([x, g]) where x (the f16 feature) is the same for every object: W*cat = W[:, :x]*x +
            W[:, x:]*g, so with several objects the x half is convolved once and enters the per-object half as a
            broadcast residual (its scale is 1, the bias stays with the per-object half)."""
            cw = W[name]
            ones, zeros = torch.ones_like(cw.scale), torch.zeros_like(cw.shift)
            W[name + '@x'] = ConvWeights(cw.w[..., :x_dim].contiguous(), ones, zeros, cw.stride, cw.pad)
            W[name + '@g'] = ConvWeights(cw.w[..., x_dim:].contiguous(), cw.scale, cw.shift, cw.stride, cw.pad)

        for blk in ('value_encoder.fuser.block1', 'decoder.fuser.block1'):
            for c in ('.conv1', '.downsample'):
                if (blk + c) in W and W[blk + c].cin > 1024 and (W[blk + c].cin - 1024) % 32 == 0:
                    split_shared(blk + c)
        if self.hidden_dim > 0:
            W['value_encoder.hidden_reinforce.transform'] = self._conv_w('value_encoder.hidden_reinforce.transform', None, 1, 1)
            W['decoder.hidden_update.g16_conv'] = self._conv_w('decoder.hidden_update.g16_conv', None, 1, 0)
            W['decoder.hidden_update.g8_conv'] = self._conv_w('decoder.hidden_update.g8_conv', None, 1, 0)
            W['decoder.hidden_update.g4_conv'] = self._conv_w('decoder.hidden_update.g4_conv', None, 1, 0)
            W['decoder.hidden_update.transform'] = self._conv_w('decoder.hidden_update.transform', None, 1, 1)
            # HiddenUpdater (model/modules.py:84-110) sums three pointwise convolutions, g16_conv(g16) + g8_conv(area(g8)) + g4_conv(area(g4)):
            # ONE pointwise convolution over the concatenated input [g16 | g8 | g4] with the filters concatenated along Cin and the
            # biases summed - one GEMM (K = 512 + 256 + 260) instead of three short ones chained through residual reads with two
            # split-K reductions (44 -> ~20 us per frame at 480p).  Same products; the sum runs over one chain instead of three.
            parts = [W['decoder.hidden_update.' + n] for n in ('g16_conv', 'g8_conv', 'g4_conv')]
            if all(float((p.scale - 1).abs().max()) == 0.0 for p in parts):
                wk = torch.cat([p.w for p in parts], 3).contiguous()
                W['decoder.hidden_update.g_fused'] = ConvWeights(wk, parts[0].scale, (parts[0].shift + parts[1].shift + parts[2].shift).contiguous(),
                                                                 1, 0, cin_true=sum(p.cin_true for p in parts))
        W['decoder.up_16_8.skip_conv'] = self._conv_w('decoder.up_16_8.skip_conv', None, 1, 1)
        group_res('decoder.up_16_8.out_conv')
        W['decoder.up_8_4.skip_conv'] = self._conv_w('decoder.up_8_4.skip_conv', None, 1, 1)
        group_res('decoder.up_8_4.out_conv')
        W['decoder.pred'] = self._conv_w('decoder.pred', None, 1, 1)
        # KeyProjection: key | shrinkage | selection share one implicit GEMM (Cout = 2*C_k + 1 = 129), padded with zero filters to the
        # row stride of its output (132): a channel count that is a multiple of 4 is what lets the layer take the Winograd plans
        # (K = 9 * 1024: 228 -> 85 us at batch 4); the padding columns of `proj` receive zeros, key_post never reads them
        parts = [self._conv_w('key_proj.' + n, None, 1, 1) for n in ('key_proj', 'd_proj', 'e_proj')]
        wk = torch.cat([p.w for p in parts], 0)
        sc, sh = torch.cat([p.scale for p in parts], 0), torch.cat([p.shift for p in parts], 0)
        extra = _pad4(wk.shape[0]) - wk.shape[0]
        if extra:
            wk = torch.cat([wk, wk.new_zeros((extra,) + tuple(wk.shape[1:]))], 0)
            sc, sh = torch.cat([sc, sc.new_ones(extra)], 0), torch.cat([sh, sh.new_zeros(extra)], 0)
        W['key_proj'] = ConvWeights(wk.contiguous(), sc.contiguous(), sh.contiguous(), 1, 1, cin_true=parts[0].cin_true)
        self._w = W

    def _need_weights(self):
        if not self._w:
            if self._sd is None:
                warnings.warn('XMem: no checkpoint loaded - using the deterministic synthetic weights (xmem2_amd.synth)')
                from .synth import synthetic_state_dict
                self._sd = synthetic_state_dict(0, self.key_dim, self.value_dim, self.hidden_dim)
            if self.device.type != 'cuda':
                raise RuntimeError('xmem2_amd.XMem runs on an MI355X (HIP) device only; call .to("cuda") - there is no CPU path')
            self._upload()

    # ---- HIP-graph staging ------------------------------------------------------------------------
    def _run_stage(self, name, key, inputs, fn, alias=(), mutates=()):
        """Run `fn(*inputs)` eagerly, or capture it once per (name, shapes, flags) into a HIP graph with static
        input / output buffers and replay it.  Kernels are launched through ctypes on torch's current stream, which
        is the capturing stream inside torch.cuda.graph, so they are captured like any other launch."""
        prec = self._call_precision or self.precision
        only = os.environ.get('XMEM_PRECISION_ONLY')          # tools: restrict a non-default precision to one stage kind
        if only and prec == 'fp16':
            # the fp16 loop keeps HALF activations across stage boundaries (key -> segment -> value): one stage alone cannot run in
            # another storage type.  The knob is for the operand-only modes (fp16w / fp32x), whose tensors stay fp32.
            raise RuntimeError("XMEM_PRECISION_ONLY does not compose with precision='fp16' (half activations cross the stage boundaries); "
                               "use it with 'fp16w' or 'fp32x'")
        if only and prec != 'fp32' and name != only:
            prec = 'fp32'
        if not self.use_graphs or ops.eager_only() or name in os.environ.get('XMEM_EAGER_STAGES', '').split(','):
            with ops.precision(prec):
                return fn(*inputs)
        full_key = (name, key, prec) + tuple(tuple(t.shape) if t is not None else None for t in inputs)
        st = self._stages.get(full_key)
        if st is None:
            # inputs listed in `alias` are themselves stable buffers (outputs of another stage): use them in place
            static_in = [(t if i in alias else t.clone()) if t is not None else None for i, t in enumerate(inputs)]
            # warm-up: sizes every workspace before the capture.  Inputs the stage updates in place (`mutates`: the hidden
            # state) are cloned for it, otherwise warm-up + first replay would advance the state twice.
            with ops.precision(prec):
                self._in_stage = True            # warm-up and capture take the same (forked) launch sequence: same workspaces
                try:
                    fn(*[(t.clone() if (i in mutates and t is not None) else t) for i, t in enumerate(static_in)])
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        static_out = fn(*static_in)
                finally:
                    self._in_stage = False
            st = (graph, static_in, static_out)
            while len(self._stages) >= self.max_stages:
                self._evict_lru()
            self._stages[full_key] = st
        else:
            self._stages.move_to_end(full_key)
        graph, static_in, static_out = st
        for dst, src in zip(static_in, inputs):
            if dst is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        graph.replay()
        return static_out

    def _evict_lru(self):
        """Bound the cache (every resolution / object count / slot / owner adds graphs with private pools): drop the least
        recently replayed stage.  Decoder and value stages read a key-encoder stage's outputs in place; when a key stage goes,
        the stages aliasing its buffers go with it (they would otherwise pay a copy per frame into buffers nobody produces
        into any more) and are re-captured on demand against the new key stage."""
        key, st = self._stages.popitem(last=False)
        if key[0] != 'key':
            return
        spans = [(o.data_ptr(), o.data_ptr() + o.numel() * o.element_size()) for o in st[2] if isinstance(o, torch.Tensor)]
        for o in st[2]:
            if isinstance(o, tuple):
                spans += [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in o if isinstance(t, torch.Tensor)]
        inside = lambda t: isinstance(t, torch.Tensor) and any(a <= t.data_ptr() < b for a, b in spans)
        for k in [k for k, v in self._stages.items() if k[0] != 'key' and any(inside(t) for t in v[1])]:
            del self._stages[k]

    def acquire_owner(self):
        """Owner token of an InferenceCore: its decoder stages advance ITS hidden state in place, so stages are keyed by
        owner.  Tokens are small integers recycled when a core dies (`release_owner`): the next core (eval.py builds one per
        video) takes over the dead core's captured stages - its fresh hidden state is copied into the stage's state buffer
        once and the core then continues on that buffer - instead of re-capturing every decoder graph per video."""
        if self._owner_free:
            return self._owner_free.pop()
        self._owner_next += 1
        return self._owner_next

    def release_owner(self, token):
        self._owner_free.append(token)

    def _is_stage_output(self, t):
        """True when `t` lives inside a static output buffer of a key-encoder stage (a whole output, or the slice of one
        frame of a batched pass): such tensors keep their address, so later stages may capture them in place."""
        a = t.data_ptr()
        for k, st in self._stages.items():
            if k[0] != 'key':
                continue
            for o in st[2]:
                if isinstance(o, torch.Tensor) and o.data_ptr() <= a < o.data_ptr() + o.numel() * o.element_size():
                    return True
        return False

    # ---- building blocks (NHWC) -----------------------------------------------------------------
    def _bottleneck(self, x, p):
        W = self._w
        o = ops.conv2d(x, W[p + '.conv1'], relu_out=True)
        o = ops.conv2d(o, W[p + '.conv2'], relu_out=True)
        res = ops.conv2d(x, W[p + '.downsample']) if (p + '.downsample') in W else x
        return ops.conv2d(o, W[p + '.conv3'], res=res, relu_out=True)

    def _basic(self, x, p):
        W = self._w
        o = ops.conv2d(x, W[p + '.conv1'], relu_out=True)
        res = ops.conv2d(x, W[p + '.downsample']) if (p + '.downsample') in W else x
        return ops.conv2d(o, W[p + '.conv2'], res=res, relu_out=True)

    def _stage(self, x, prefix, blocks, fn):
        for b in range(blocks):
            x = fn(x, f'{prefix}.{b}')
        return x

    def _group_res(self, g, p, out=None, out_ld=None):
        """GroupResBlock, model/group_modules.py:44-52: conv2(relu(conv1(relu(g)))) + (downsample(g) | g)."""
        W = self._w
        if (p + '.downsample') in W and self._fork_ok():
            res = self._forked(lambda: ops.conv2d(g, W[p + '.downsample']))
            o = ops.conv2d(g, W[p + '.conv1'], relu_in=True, relu_out=True)
            self._join()
            return ops.conv2d(o, W[p + '.conv2'], res=res, out=out, out_ld=out_ld)
        o = ops.conv2d(g, W[p + '.conv1'], relu_in=True, relu_out=True)
        res = ops.conv2d(g, W[p + '.downsample']) if (p + '.downsample') in W else g
        return ops.conv2d(o, W[p + '.conv2'], res=res, out=out, out_ld=out_ld)

    # ---- forked branch inside a stage ------------------------------------------------------------
    def _fork_ok(self):
        """Fork only where the launches end up in a HIP graph (a stage being warmed up or captured): eager launches would pay two
        cross-stream waits per block for nothing."""
        return self.branch_overlap and self.use_graphs and not ops.eager_only() and getattr(self, '_in_stage', False)

    def _forked(self, fn):
        """Run fn() on the branch stream (own scratch scope: the convolution workspaces are per stream), forked from the current one."""
        main = torch.cuda.current_stream()
        if self._branch is None:
            self._branch = torch.cuda.Stream(device=self.device)
        self._branch.wait_stream(main)
        with torch.cuda.stream(self._branch), ops.ws_scope(getattr(ops._tls, 'suffix', '') + f'@branch#{self._scope}#'):
            out = fn()
        return out

    def _join(self):
        torch.cuda.current_stream().wait_stream(self._branch)

    def _shares_x(self, p, n_obj):
        """True when `_fusion(cat, p, x)` convolves the shared x half once (several objects, split weights uploaded)."""
        W = self._w
        return n_obj > 1 and self.share_x and (p + '.block1.conv1@x') in W and (p + '.block1.downsample@x') in W

    def _fusion(self, cat, p, x=None, pre=None):
        """FeatureFusionBlock, model/modules.py:31-41, on the already concatenated [x | g] tensor.  With several objects
        and `x` given ([1,h,w,1024], the shared f16 half of `cat`), block1's two 3x3 convolutions run the x half once
        and only the g half per object (saves 64 % / 80 % of block1's FLOPs for every object but the first)."""
        W = self._w
        b1 = p + '.block1'
        if x is not None and (pre is not None or self._shares_x(p, cat.shape[0])):
            xd = x.shape[3]
            gpart = cat[..., xd:]
            ld, cg = cat.shape[3], cat.shape[3] - xd
            # pre = (conv1@x(relu(x)), downsample@x(x)) already made in the batched key pass (prefetched frames)
            sx = pre[0] if pre is not None else ops.conv2d(x, W[b1 + '.conv1@x'], relu_in=True)
            dx = pre[1] if pre is not None else ops.conv2d(x, W[b1 + '.downsample@x'])
            if self._fork_ok():
                res = self._forked(lambda: ops.conv2d(gpart, W[b1 + '.downsample@g'], res=dx, res_broadcast=True, in_ld=ld, cin=cg))
                o = ops.conv2d(gpart, W[b1 + '.conv1@g'], relu_in=True, relu_out=True, res=sx, res_broadcast=True, in_ld=ld, cin=cg)
                self._join()
            else:
                o = ops.conv2d(gpart, W[b1 + '.conv1@g'], relu_in=True, relu_out=True, res=sx, res_broadcast=True, in_ld=ld, cin=cg)
                res = ops.conv2d(gpart, W[b1 + '.downsample@g'], res=dx, res_broadcast=True, in_ld=ld, cin=cg)
            g = ops.conv2d(o, W[b1 + '.conv2'], res=res)
        else:
            g = self._group_res(cat, b1)
        g = ops.cbam_residual(g, self._cbam[p + '.attention'])
        return self._group_res(g, p + '.block2')

    def read_memory(self, query_key, query_selection, memory_key, memory_shrinkage, memory_value):
        """model/network.py:89-105 (training-time read: full softmax over all memory elements, dense readout).
        query_* B x CK x H x W, memory_key B x CK x T x H x W, memory_shrinkage B x 1 x T x H x W,
        memory_value B x num_objects x CV x T x H x W -> B x num_objects x CV x H x W."""
        from .memory_util import get_affinity, readout
        batch_size, num_objects = memory_value.shape[:2]
        mv = memory_value.flatten(start_dim=1, end_dim=2)
        affinity = get_affinity(memory_key, memory_shrinkage, query_key, query_selection)
        memory = readout(affinity, mv)
        return memory.view(batch_size, num_objects, self.value_dim, *memory.shape[-2:])

    # ---- hot path (NHWC) ------------------------------------------------------------------------
    def encode_key_nhwc(self, image4, need_sk=True, need_ek=True, with_skips=False, slot=0, inline_skips=False):
        """image4 [B,Hp,Wp,4] -> key [B*h*w,Ck], shrinkage [B*h*w]|None, selection|None, f16, f8, f4 (NHWC)
        [+ (skip8, skip4), the decoder's skip convolutions of f8 / f4, when with_skips and the graph path is active].
        With graphs on, the returned tensors are the stage's static buffers: valid until the next call
        with the same `slot` (two slots let the key encoder of frame t+1 run while frame t is still being decoded)."""
        self._need_weights()
        overlap = bool(with_skips and not inline_skips and self.overlap_skips and self.use_graphs and not ops.eager_only()
                       and image4.shape[0] == 1)
        inline = bool(with_skips and inline_skips)      # a prefetched pass also runs the decoder's skip convolutions
        # key-encoder graphs may run on a side stream: never share scratch with the decoder, nor with another network instance
        self._key_ws = f'@key{slot}#{self._scope}#'
        with ops.ws_scope(self._key_ws):
            out = self._run_stage('key', (need_sk, need_ek, overlap, inline, slot), [image4],
                                  lambda im: self._encode_key_eager(im, need_sk, need_ek, overlap, inline))
        if with_skips:
            return out if (overlap or inline) else tuple(out) + (None,)
        return out[:6]

    def _encode_key_eager(self, image4, need_sk, need_ek, overlap=False, inline_skips=False):
        W = self._w
        x = ops.conv2d(image4, W['key_encoder.conv1'], relu_out=True)       # the stem reads the fp32 image in every mode
        x = ops.maxpool3x3s2(x, out_dtype=ops.act_dtype())                # fp16 loop: activations become halfs here
        f4 = self._stage(x, 'key_encoder.res2', 3, self._bottleneck)
        skip4 = skip8 = None
        main = torch.cuda.current_stream()
        if overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(device=image4.device)
            self._side.wait_stream(main)                       # fork: f4 is ready
            with torch.cuda.stream(self._side), ops.ws_scope(f'@side4#{self._scope}#'):
                skip4 = ops.conv2d(f4, W['decoder.up_8_4.skip_conv'])
        f8 = self._stage(f4, 'key_encoder.layer2', 4, self._bottleneck)
        if overlap:
            self._side.wait_stream(main)                       # f8 is ready
            with torch.cuda.stream(self._side), ops.ws_scope(f'@side8#{self._scope}#'):
                skip8 = ops.conv2d(f8, W['decoder.up_16_8.skip_conv'])
        f16 = self._stage(f8, 'key_encoder.layer3', 6, self._bottleneck)
        B, h, w, _ = f16.shape
        ld = _pad4(2 * self.key_dim + 1)
        proj = torch.empty((B, h, w, ld), dtype=torch.float32, device=f16.device)
        ops.conv2d(f16, W['key_proj'], out=proj, out_ld=ld, out_dtype=torch.float32)     # keys / shrinkage / selection are fp32 in every mode
        key, shr, sel = ops.key_post(proj, self.key_dim, need_sk, need_ek)
        if inline_skips:                      # same stream: f8 / f4 only depend on the image (model/modules.py:186,231-232)
            extras = (ops.conv2d(f8, W['decoder.up_16_8.skip_conv']), ops.conv2d(f4, W['decoder.up_8_4.skip_conv']))
            if self.prefuse_x and ('decoder.fuser.block1.conv1@x') in W and ('decoder.fuser.block1.downsample@x') in W:
                # FeatureFusionBlock convolves cat([f16, readout, hidden]) (model/modules.py:31-41): W * cat = W_x * f16 + W_g * [readout |
                # hidden], and the f16 half (1024 of 1600 input channels of block1's two 3x3 convolutions) depends on the frame only -
                # so it is convolved HERE, in the batched pass on the side stream, and enters the decoder as a residual
                b1 = 'decoder.fuser.block1'
                extras += (ops.conv2d(f16, W[b1 + '.conv1@x'], relu_in=True), ops.conv2d(f16, W[b1 + '.downsample@x']))
            return key, shr, sel, f16, f8, f4, extras
        if overlap:
            main.wait_stream(self._side)                       # join before the stage (and its graph capture) ends
            return key, shr, sel, f16, f8, f4, (skip8, skip4)
        return key, shr, sel, f16, f8, f4

    def encode_value_nhwc(self, image4, f16, hidden, masks, is_deep_update=True, slot=0):
        """image4 [1,Hp,Wp,4], f16 [1,h,w,1024], hidden [K,h,w,Ch], masks [K,Hp,Wp] -> value [K,h,w,Cv], hidden."""
        self._need_weights()
        value, new_hidden = self._run_stage('value', (bool(is_deep_update), slot), [image4, f16, hidden, masks],
                                            lambda a, b, c, d: self._encode_value_eager(a, b, c, d, is_deep_update),
                                            alias=(1,) if self._is_stage_output(f16) else ())
        if self.use_graphs and new_hidden is not None and new_hidden is not hidden:
            new_hidden = new_hidden.clone()                 # the hidden state outlives the stage's static buffer
        return value, new_hidden

    def _encode_value_eager(self, image4, f16, hidden, masks, is_deep_update):
        W = self._w
        x = ops.pack_value_input(image4, masks)
        act = ops.act_dtype()
        g = ops.conv2d(x, W['value_encoder.conv1'], relu_out=True)     # relu and max-pool commute (modules.py:137-138)
        g = ops.maxpool3x3s2(g, out_dtype=act)
        g = self._stage(g, 'value_encoder.layer1', 2, self._basic)
        g = self._stage(g, 'value_encoder.layer2', 2, self._basic)
        g = self._stage(g, 'value_encoder.layer3', 2, self._basic)
        K, h, w, cg = g.shape
        cat = torch.empty((K, h, w, f16.shape[3] + cg), dtype=act, device=g.device)
        if not self._shares_x('value_encoder.fuser', K):
            ops.copy_channels(f16, cat, 0)
        ops.copy_channels(g, cat, f16.shape[3])
        value = self._fusion(cat, 'value_encoder.fuser', x=f16)
        if is_deep_update and self.hidden_dim > 0:
            cat2 = torch.empty((K, h, w, self.value_dim + self.hidden_dim), dtype=act, device=g.device)
            ops.copy_channels(value, cat2, 0)
            ops.copy_channels(hidden, cat2, self.value_dim)
            values = ops.conv2d(cat2, W['value_encoder.hidden_reinforce.transform'])
            hidden = ops.gru_gate(values, hidden)
        return value, hidden

    def encode_value_frames_nhwc(self, image4, f16, masks):
        """Value encoder over SEVERAL frames at once (the batched permanent-memory preload: an annotated frame and its
        augmentations, inference/run_on_video.py:59-66,231-242): image4 [B,Hp,Wp,4], f16 [B,h,w,1024], masks = list of B tensors
        [K,Hp,Wp] -> value [B*K,h,w,Cv] (frame-major).  No deep update (put_to_permanent_memory never makes one), eager,
        in the network's full-precision arithmetic."""
        self._need_weights()
        W = self._w
        B = image4.shape[0]
        K = masks[0].shape[0]
        with ops.precision('fp32'):
            x = torch.empty((B * K,) + tuple(image4.shape[1:3]) + (8,), dtype=torch.float32, device=image4.device)
            for b in range(B):
                x[b * K:(b + 1) * K].copy_(ops.pack_value_input(image4[b:b + 1], masks[b]))
            g = ops.conv2d(x, W['value_encoder.conv1'], relu_out=True)
            g = ops.maxpool3x3s2(g)
            g = self._stage(g, 'value_encoder.layer1', 2, self._basic)
            g = self._stage(g, 'value_encoder.layer2', 2, self._basic)
            g = self._stage(g, 'value_encoder.layer3', 2, self._basic)
            _, h, w, cg = g.shape
            cat = torch.empty((B * K, h, w, f16.shape[3] + cg), dtype=torch.float32, device=g.device)
            for b in range(B):                        # each frame's own f16 in front of its objects' features
                ops.copy_channels(f16[b:b + 1], cat[b * K:(b + 1) * K], 0)
            ops.copy_channels(g, cat, f16.shape[3])
            return self._fusion(cat, 'value_encoder.fuser', x=None)

    def new_decoder_input(self, K, h, w, device, slot=0, owner=0, h_out=None, has_skips=None, static_only=False, out_hw=None, pad_tl=None):
        """[K,h,w, 1024+Cv+Ch] buffer; the readout kernel writes channels [1024, 1024+Cv) in place.
        Once the matching decoder stage is captured this is its static input buffer (no copy before the replay).
        static_only: None instead of a fresh buffer when no captured stage matches.  out_hw / pad_tl: the remaining components of the
        stage key (two resolutions that pad to one h x w have stages of their own)."""
        shape = (K, h, w, 1024 + self.value_dim + self.hidden_dim)
        prec = self._call_precision or self.precision
        if self.use_graphs and not ops.eager_only():
            for k, st in self._stages.items():
                if k[0] != 'segment' or k[2] != (self._call_precision or self.precision) or k[1][-2] != slot or k[1][-1] != owner \
                        or tuple(st[1][3].shape) != shape:
                    continue
                if (h_out is not None and k[1][2] != bool(h_out)) or (has_skips is not None and k[1][3] != bool(has_skips)):
                    continue
                if (out_hw is not None and k[1][0] != tuple(out_hw)) or (pad_tl is not None and k[1][1] != tuple(pad_tl)):
                    continue
                return st[1][3]
        if static_only:
            return None
        return torch.empty(shape, dtype=torch.float16 if prec == 'fp16' else torch.float32, device=device)

    def _zero_scratch(self, shape, device, dtype=torch.float32):
        """Persistent zero-initialised buffer (allocated outside any capture): kernels overwrite only its data channels, the
        padding channels stay zero - no per-frame fill kernel."""
        key = (tuple(shape), str(device), dtype)
        buf = self._zeros.get(key)
        if buf is None:
            buf = torch.zeros(shape, dtype=dtype, device=device)
            self._zeros[key] = buf
        return buf

    def segment_nhwc(self, f16, f8, f4, cat16, hidden, out_hw, pad_tl, h_out=True, skips=None, slot=0, owner=0):
        """Decoder + soft aggregation.  cat16 holds the memory readout at channels [1024,1024+Cv).
        Returns new_hidden|None, prob [K+1,H,W] (unpadded), prob_padded [K+1,Hp,Wp].
        The hidden state is updated IN PLACE (`new_hidden is hidden` when h_out): the captured stage reads and writes the
        caller's state tensor (`owner` keys the stage so that two cores sharing one network never share a state buffer)."""
        self._need_weights()
        K, h, w, _ = cat16.shape
        g4d = None
        if h_out and self.hidden_dim > 0:
            c4 = self._w['decoder.pred'].cin
            half = (self._call_precision or self.precision) == 'fp16'
            gf = self._w.get('decoder.hidden_update.g_fused')
            if gf is not None and not half and self.fuse_hidden_update:
                # the concatenated input of the fused hidden-update convolution [g16 | area(g8) | area(g4), area(logits), zero padding]
                g4d = self._zero_scratch((K, h, w, gf.cin), cat16.device, torch.float32)
            else:
                g4d = self._zero_scratch((K, h, w, (c4 + 1 + 7) // 8 * 8 if half else _pad4(c4 + 1)), cat16.device,
                                         torch.float16 if half else torch.float32)
        if skips is not None and len(skips) >= 4:
            out = self._run_stage('segment', (tuple(out_hw), tuple(pad_tl), bool(h_out), True, slot, owner),
                                  [f16, f8, f4, cat16, hidden, skips[0], skips[1], skips[2], skips[3]],
                                  lambda a, b, c, d, e, s8, s4, sx, dx: self._segment_eager(a, b, c, d, e, h_out, (s8, s4, sx, dx), g4d),
                                  alias=(0, 1, 2, 4, 5, 6, 7, 8) if self._is_stage_output(f16) else (4,), mutates=(4,))
        elif skips is not None:
            out = self._run_stage('segment', (tuple(out_hw), tuple(pad_tl), bool(h_out), True, slot, owner),
                                  [f16, f8, f4, cat16, hidden, skips[0], skips[1]],
                                  lambda a, b, c, d, e, s8, s4: self._segment_eager(a, b, c, d, e, h_out, (s8, s4), g4d),
                                  alias=(0, 1, 2, 4, 5, 6) if self._is_stage_output(f16) else (4,), mutates=(4,))
        else:
            out = self._run_stage('segment', (tuple(out_hw), tuple(pad_tl), bool(h_out), False, slot, owner),
                                  [f16, f8, f4, cat16, hidden],
                                  lambda a, b, c, d, e: self._segment_eager(a, b, c, d, e, h_out, None, g4d),
                                  alias=(0, 1, 2, 4) if self._is_stage_output(f16) else (4,), mutates=(4,))
        new_hidden, logits = out
        # soft aggregation outside the captured stage: prob / prob_padded are fresh tensors the caller may keep (no clone)
        H, Wd = out_hw
        prob, prob_padded = ops.logits_to_prob(logits.view(K, 4 * h, 4 * w), H, Wd, pad_tl[0], pad_tl[1])
        return new_hidden, prob, prob_padded

    def _segment_eager(self, f16, f8, f4, cat16, hidden, h_out, skips=None, g4d=None):
        W = self._w
        K, h, w, _ = cat16.shape
        hd = self.hidden_dim
        pre = (skips[2], skips[3]) if (skips is not None and len(skips) >= 4) else None
        if pre is None and not self._shares_x('decoder.fuser', K):
            ops.copy_channels(f16, cat16, 0)               # with several objects the f16 half is convolved once (see _fusion)
        if hd > 0:
            ops.copy_channels(hidden, cat16, 1024 + self.value_dim)
        g16 = self._fusion(cat16, 'decoder.fuser', x=f16, pre=pre)
        skip8 = skips[0] if skips is not None else ops.conv2d(f8, W['decoder.up_16_8.skip_conv'])
        g8 = self._group_res(ops.upsample2x_add(g16, skip8), 'decoder.up_16_8.out_conv')
        skip4 = skips[1] if skips is not None else ops.conv2d(f4, W['decoder.up_8_4.skip_conv'])
        g4 = self._group_res(ops.upsample2x_add(g8, skip4), 'decoder.up_8_4.out_conv')
        logits = ops.conv2d(g4, W['decoder.pred'], relu_in=True, out_dtype=torch.float32)          # [K,4h,4w,1], fp32 in every mode
        new_hidden = None
        if h_out and hd > 0:
            c4 = g4.shape[3]
            gf = W.get('decoder.hidden_update.g_fused')
            if g4d is None:
                g4d = self._zero_scratch((K, h, w, _padc(c4 + 1)), g4.device, ops.act_dtype())
            if gf is not None and g4d.shape[3] == gf.cin and g4d.dtype == torch.float32:
                # one pointwise convolution over [g16 | area(g8) | area(g4), area(logits)] (see _upload)
                c16, c8, ld = g16.shape[3], g8.shape[3], g4d.shape[3]
                if self.gather_hidden_input and all(t.is_contiguous() for t in (g16, g8, g4, logits)):
                    ops.hidden_update_gather(g16, g8, g4, logits, g4d)        # one launch (round 6; the same bits as the four below)
                else:
                    ops.copy_channels(g16, g4d, 0)
                    ops.area_downsample(g8, 2, out=g4d, out_ld=ld, out_off=c16)
                    ops.area_downsample(g4, 4, out=g4d, out_ld=ld, out_off=c16 + c8)
                    ops.area_downsample(logits, 4, out=g4d, out_ld=ld, out_off=c16 + c8 + c4)
                mid = gf.cout
                cat = torch.empty((K, h, w, mid + hd), dtype=ops.act_dtype(), device=g4.device)
                ops.conv2d(g4d, gf, out=cat, out_ld=cat.shape[3])
            else:
                ops.area_downsample(g4, 4, out=g4d, out_ld=g4d.shape[3])
                ops.area_downsample(logits, 4, out=g4d, out_ld=g4d.shape[3], out_off=c4)
                g8d = ops.area_downsample(g8, 2)
                t = ops.conv2d(g16, W['decoder.hidden_update.g16_conv'])
                t = ops.conv2d(g8d, W['decoder.hidden_update.g8_conv'], res=t)
                mid = t.shape[3]
                cat = torch.empty((K, h, w, mid + hd), dtype=ops.act_dtype(), device=g4.device)
                ops.conv2d(g4d, W['decoder.hidden_update.g4_conv'], res=t, out=cat, out_ld=cat.shape[3])
            ops.copy_channels(hidden, cat, mid)
            values = ops.conv2d(cat, W['decoder.hidden_update.transform'])
            new_hidden = ops.gru_gate(values, hidden, out=hidden)          # in place: the state tensor itself advances
        return new_hidden, logits

    # ---- reference-shaped wrappers (NCHW in / out) ----------------------------------------------
    @staticmethod
    def _as_nhwc(t):
        p = t.permute(0, 2, 3, 1)
        return p if p.is_contiguous() else ops.nchw_to_nhwc(t)

    def encode_key(self, frame, need_sk=True, need_ek=True):
        """model/network.py:40-70 for 4-D input [B,3,H,W] (H, W multiples of 16)."""
        if frame.dim() != 4:
            raise NotImplementedError
        B, _, H, Wd = frame.shape
        image4 = torch.cat([ops.pack_image(frame[b], H, Wd, 0, 0) for b in range(B)], 0) if B > 1 \
            else ops.pack_image(frame[0], H, Wd, 0, 0)
        key, shr, sel, f16, f8, f4 = self.encode_key_nhwc(image4, need_sk, need_ek)
        h, w = f16.shape[1], f16.shape[2]
        nchw = lambda t: t.permute(0, 3, 1, 2)
        key = nchw(key.view(B, h, w, self.key_dim))
        shr = shr.view(B, 1, h, w) if shr is not None else None
        sel = nchw(sel.view(B, h, w, self.key_dim)) if sel is not None else None
        self._last_image4 = (frame.data_ptr(), image4)
        return key, shr, sel, nchw(f16), nchw(f8), nchw(f4)

    def encode_value(self, frame, image_feat_f16, h16, masks, is_deep_update=True):
        """model/network.py:72-85: frame [1,3,H,W], f16 [1,1024,h,w], h16 [1,K,Ch,h,w], masks [1,K,H,W]."""
        cached = getattr(self, '_last_image4', None)
        if cached is not None and cached[0] == frame.data_ptr():
            image4 = cached[1]
        else:
            image4 = ops.pack_image(frame[0], frame.shape[2], frame.shape[3], 0, 0)
        f16 = self._as_nhwc(image_feat_f16)
        hidden = self._as_nhwc(h16[0]) if h16 is not None else None
        m = masks[0] if masks[0].is_contiguous() else masks[0].contiguous()
        value, hidden = self.encode_value_nhwc(image4, f16, hidden, m, is_deep_update)
        value = value.permute(0, 3, 1, 2).unsqueeze(0)
        hidden = hidden.permute(0, 3, 1, 2).unsqueeze(0) if hidden is not None else None
        return value, hidden

    def segment(self, multi_scale_features, memory_readout, hidden_state, selector=None, h_out=True, strip_bg=True):
        """model/network.py:107-120.  Returns (hidden, None, prob): the aggregated logits are a training-only
        output of the reference and are not produced here."""
        if selector is not None:
            raise NotImplementedError('selector is a training-time argument (model/trainer.py) - out of scope')
        f16, f8, f4 = [self._as_nhwc(t) for t in multi_scale_features]
        ro = self._as_nhwc(memory_readout[0])                      # [K,h,w,Cv]
        K, h, w, _ = ro.shape
        hidden = self._as_nhwc(hidden_state[0]) if hidden_state is not None else None
        cat16 = self.new_decoder_input(K, h, w, ro.device)
        ops.copy_channels(ro, cat16, 1024)
        if hidden is not None and hidden_state is not None and hidden.data_ptr() == hidden_state.data_ptr():
            hidden = hidden.clone()                     # the reference-shaped call must not edit the caller's tensor
        new_hidden, _, prob = self.segment_nhwc(f16, f8, f4, cat16, hidden, (16 * h, 16 * w), (0, 0), h_out)
        if new_hidden is not None:
            new_hidden = new_hidden.clone()             # the stage's state buffer is reused by the next call
        prob = prob.unsqueeze(0)
        if strip_bg:
            prob = prob[:, 1:]
        new_hidden = new_hidden.permute(0, 3, 1, 2).unsqueeze(0) if new_hidden is not None else None
        return new_hidden, None, prob

    def forward(self, mode, *args, **kwargs):
        if mode == 'encode_key':
            return self.encode_key(*args, **kwargs)
        if mode == 'encode_value':
            return self.encode_value(*args, **kwargs)
        if mode == 'segment':
            return self.segment(*args, **kwargs)
        raise NotImplementedError(mode)

    __call__ = forward
