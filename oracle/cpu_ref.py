"""ORACLE - test infrastructure only.  Never imported by the product path.

A plain PyTorch fp32 *CPU restatement* of XMem++'s per-frame space-time memory
path.  The arithmetic is floating point, so the
oracle is a torch-fp32 op sequence (not C / numpy): it issues the same ATen ops on
tensors of the same shapes and strides as the reference, which makes it bit-equal
to the imported reference at a fixed thread count.  That equality is what
``tests/golden/make_goldens.py`` pins (fixtures generated here by importing
/root/reference; the reference has no tests or golden vectors of its own - see
DESIGN.md "Oracle").  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file.

The network is expressed functionally over a state_dict (no nn.Modules); the
memory is a small set of classes that keep tensors in the reference's layouts
(keys ``1 x C x N`` etc.) because MKL/oneDNN results depend on operand strides.

Each function cites the reference lines it restates.
"""
import math
import warnings

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------
# model/memory_util.py
# ------------------------------------------------------------------------------------------


def get_similarity(mk, ms, qk, qe):
    """Anisotropic L2 similarity, model/memory_util.py:7-39.

    mk B x CK x N, ms B x 1 x N (or None), qk/qe B x CK x HW (qe may be None) -> B x N x HW.
    """
    ck = mk.shape[1]
    mk = mk.flatten(start_dim=2)
    ms = ms.flatten(start_dim=1).unsqueeze(2) if ms is not None else None
    qk = qk.flatten(start_dim=2)
    qe = qe.flatten(start_dim=2) if qe is not None else None
    if qe is not None:
        mkt = mk.transpose(1, 2)
        a_sq = mkt.pow(2) @ qe
        two_ab = 2 * (mkt @ (qk * qe))
        b_sq = (qe * qk.pow(2)).sum(1, keepdim=True)
        sim = -a_sq + two_ab - b_sq
    else:
        a_sq = mk.pow(2).sum(1).unsqueeze(2)
        two_ab = 2 * (mk.transpose(1, 2) @ qk)
        sim = -a_sq + two_ab
    if ms is not None:
        sim = sim * ms / math.sqrt(ck)
    else:
        sim = sim / math.sqrt(ck)
    return sim


def do_softmax(similarity, top_k=None, inplace=False, return_usage=False):
    """Top-k softmax without max-shift / full stable softmax, model/memory_util.py:41-65."""
    if top_k is not None:
        values, indices = torch.topk(similarity, k=top_k, dim=1)
        x_exp = values.exp_()
        x_exp /= torch.sum(x_exp, dim=1, keepdim=True)
        if inplace:
            similarity.zero_().scatter_(1, indices, x_exp)
            affinity = similarity
        else:
            affinity = torch.zeros_like(similarity).scatter_(1, indices, x_exp)
    else:
        maxes = torch.max(similarity, dim=1, keepdim=True)[0]
        x_exp = torch.exp(similarity - maxes)
        affinity = x_exp / torch.sum(x_exp, dim=1, keepdim=True)
    if return_usage:
        return affinity, affinity.sum(dim=2)
    return affinity


def topk_softmax_sparse(similarity, top_k):
    """The same top-k softmax as `do_softmax`, returned sparsely: weights and indices B x k x HW.

    This is the form the HIP kernel emits (it never materialises N x HW); tests compare it
    as sets (tie order of torch.topk is unspecified)."""
    values, indices = torch.topk(similarity, k=top_k, dim=1)
    x_exp = values.exp()
    return x_exp / torch.sum(x_exp, dim=1, keepdim=True), indices


# ------------------------------------------------------------------------------------------
# util/tensor_util.py, model/aggregate.py
# ------------------------------------------------------------------------------------------


def pad_divide_by(img, d):
    """Symmetric zero pad to a multiple of d, util/tensor_util.py:47-61."""
    h, w = img.shape[-2:]
    new_h = h + d - h % d if h % d > 0 else h
    new_w = w + d - w % d if w % d > 0 else w
    lh, uh = int((new_h - h) / 2), int(new_h - h) - int((new_h - h) / 2)
    lw, uw = int((new_w - w) / 2), int(new_w - w) - int((new_w - w) / 2)
    pad = (int(lw), int(uw), int(lh), int(uh))
    return F.pad(img, pad), pad


def unpad(img, pad):
    """Inverse crop, util/tensor_util.py:63-77 (3-D and 4-D only)."""
    if img.dim() not in (3, 4):
        raise NotImplementedError
    if pad[2] + pad[3] > 0:
        img = img[..., pad[2]:-pad[3], :]
    if pad[0] + pad[1] > 0:
        img = img[..., pad[0]:-pad[1]]
    return img


def aggregate(prob, dim, return_logits=False):
    """STM soft aggregation, model/aggregate.py:6-17."""
    new_prob = torch.cat([torch.prod(1 - prob, dim=dim, keepdim=True), prob], dim).clamp(1e-7, 1 - 1e-7)
    logits = torch.log((new_prob / (1 - new_prob)))
    prob = F.softmax(logits, dim=dim)
    return (logits, prob) if return_logits else prob


def compute_array_iou(seg, gt):
    """Mean per-object IoU of two index masks, util/tensor_util.py:6-44 (the parity metric)."""
    seg = np.squeeze(np.asarray(seg))
    gt = np.squeeze(np.asarray(gt))

    def iou(a, b):
        inter = float(np.logical_and(a, b).sum())
        union = float(np.logical_or(a, b).sum())
        return (inter + 1e-6) / (union + 1e-6)

    ious = [iou(seg == c, gt == c) for c in np.unique(seg) if c != 0]
    if not ious:
        ious = [iou(seg == 0, gt == 0)]
    return sum(ious) / len(ious)


# ------------------------------------------------------------------------------------------
# model/network.py + modules.py + resnet.py + group_modules.py + cbam.py, functional form
# ------------------------------------------------------------------------------------------


class RefNet:
    """Functional XMem forward over a state_dict (model/network.py:17-120)."""

    def __init__(self, state_dict, single_object=False):
        self.sd = {k: v for k, v in state_dict.items()}
        self.key_dim = self.sd['key_proj.key_proj.weight'].shape[0]
        self.value_dim = self.sd['value_encoder.fuser.block2.conv2.weight'].shape[0]
        self.hidden_dim = (self.sd['decoder.hidden_update.transform.weight'].shape[0] // 3
                           if 'decoder.hidden_update.transform.weight' in self.sd else 0)
        self.single_object = single_object

    # -- primitives --
    def _conv(self, x, name, stride=1, padding=0):
        return F.conv2d(x, self.sd[name + '.weight'], self.sd.get(name + '.bias'), stride, padding)

    def _bn(self, x, name):
        return F.batch_norm(x, self.sd[name + '.running_mean'], self.sd[name + '.running_var'],
                            self.sd[name + '.weight'], self.sd[name + '.bias'], False, 0.1, 1e-5)

    def _gconv(self, g, name, padding):
        """GConv2D: conv over B*K flattened groups, model/group_modules.py:25-29."""
        b, k = g.shape[:2]
        o = self._conv(g.flatten(0, 1), name, 1, padding)
        return o.view(b, k, *o.shape[1:])

    def _bottleneck(self, x, p, stride):
        """model/resnet.py:95-114."""
        out = F.relu(self._bn(self._conv(x, p + '.conv1'), p + '.bn1'))
        out = F.relu(self._bn(self._conv(out, p + '.conv2', stride, 1), p + '.bn2'))
        out = self._bn(self._conv(out, p + '.conv3'), p + '.bn3')
        if (p + '.downsample.0.weight') in self.sd:
            x = self._bn(self._conv(x, p + '.downsample.0', stride), p + '.downsample.1')
        out += x
        return F.relu(out)

    def _basic(self, x, p, stride):
        """model/resnet.py:59-75."""
        out = F.relu(self._bn(self._conv(x, p + '.conv1', stride, 1), p + '.bn1'))
        out = self._bn(self._conv(out, p + '.conv2', 1, 1), p + '.bn2')
        if (p + '.downsample.0.weight') in self.sd:
            x = self._bn(self._conv(x, p + '.downsample.0', stride), p + '.downsample.1')
        out += x
        return F.relu(out)

    def _stage(self, x, prefix, blocks, stride, fn):
        for b in range(blocks):
            x = fn(x, f'{prefix}.{b}', stride if b == 0 else 1)
        return x

    def _group_res(self, g, p):
        """GroupResBlock, model/group_modules.py:44-52."""
        out = self._gconv(F.relu(g), p + '.conv1', 1)
        out = self._gconv(F.relu(out), p + '.conv2', 1)
        if (p + '.downsample.weight') in self.sd:
            g = self._gconv(g, p + '.downsample', 1)
        return out + g

    def _cbam(self, x, p):
        """CBAM channel + spatial gate, model/cbam.py:21-77."""
        def mlp(v):
            v = v.view(v.size(0), -1)
            v = F.relu(F.linear(v, self.sd[p + '.ChannelGate.mlp.1.weight'], self.sd[p + '.ChannelGate.mlp.1.bias']))
            return F.linear(v, self.sd[p + '.ChannelGate.mlp.3.weight'], self.sd[p + '.ChannelGate.mlp.3.bias'])
        hw = (x.size(2), x.size(3))
        att = mlp(F.avg_pool2d(x, hw, stride=hw)) + mlp(F.max_pool2d(x, hw, stride=hw))
        x = x * torch.sigmoid(att).unsqueeze(2).unsqueeze(3).expand_as(x)
        comp = torch.cat((torch.max(x, 1)[0].unsqueeze(1), torch.mean(x, 1).unsqueeze(1)), dim=1)
        gate = self._conv(comp, p + '.SpatialGate.spatial.conv', 1, 3)
        return x * torch.sigmoid(gate)

    def _fusion(self, x, g, p):
        """FeatureFusionBlock, model/modules.py:31-41 (distributor = cat(x, g))."""
        b, k = g.shape[:2]
        g = torch.cat([x.unsqueeze(1).expand(-1, k, -1, -1, -1), g], 2)
        g = self._group_res(g, p + '.block1')
        r = self._cbam(g.flatten(0, 1), p + '.attention')
        r = r.view(b, k, *r.shape[1:])
        return self._group_res(g + r, p + '.block2')

    @staticmethod
    def _interp_groups(g, ratio, mode, align):
        b, k = g.shape[:2]
        o = F.interpolate(g.flatten(0, 1), scale_factor=ratio, mode=mode, align_corners=align)
        return o.view(b, k, *o.shape[1:])

    def _gru(self, g, h, name):
        """Shared gate arithmetic of HiddenUpdater / HiddenReinforcer, model/modules.py:56-99."""
        hd = self.hidden_dim
        values = self._gconv(torch.cat([g, h], 2), name, 1)
        forget = torch.sigmoid(values[:, :, :hd])
        update = torch.sigmoid(values[:, :, hd:hd * 2])
        new_value = torch.tanh(values[:, :, hd * 2:])
        return forget * h * (1 - update) + update * new_value

    # -- the three entry points --
    def encode_key(self, frame, need_sk=True, need_ek=True):
        """model/network.py:40-70 (4-D input only) -> key, shrinkage, selection, f16, f8, f4."""
        if frame.dim() != 4:
            raise NotImplementedError
        x = F.relu(self._bn(self._conv(frame, 'key_encoder.conv1', 2, 3), 'key_encoder.bn1'))
        x = F.max_pool2d(x, 3, 2, 1)
        f4 = self._stage(x, 'key_encoder.res2', 3, 1, self._bottleneck)
        f8 = self._stage(f4, 'key_encoder.layer2', 4, 2, self._bottleneck)
        f16 = self._stage(f8, 'key_encoder.layer3', 6, 2, self._bottleneck)
        shrinkage = self._conv(f16, 'key_proj.d_proj', 1, 1) ** 2 + 1 if need_sk else None
        selection = torch.sigmoid(self._conv(f16, 'key_proj.e_proj', 1, 1)) if need_ek else None
        key = self._conv(f16, 'key_proj.key_proj', 1, 1)
        return key, shrinkage, selection, f16, f8, f4

    def encode_value(self, frame, f16, h16, masks, is_deep_update=True):
        """model/network.py:72-85 + ValueEncoder.forward model/modules.py:124-150."""
        k = masks.shape[1]
        if k != 1:
            others = torch.cat([torch.sum(masks[:, [j for j in range(k) if i != j]], dim=1, keepdim=True)
                                for i in range(k)], 1)
        else:
            others = torch.zeros_like(masks)
        g = torch.stack([masks, others], 2) if not self.single_object else masks.unsqueeze(2)
        g = torch.cat([frame.unsqueeze(1).expand(-1, k, -1, -1, -1), g], 2)
        b = g.shape[0]
        g = g.flatten(0, 1)
        g = self._bn(self._conv(g, 'value_encoder.conv1', 2, 3), 'value_encoder.bn1')
        g = F.relu(F.max_pool2d(g, 3, 2, 1))
        g = self._stage(g, 'value_encoder.layer1', 2, 1, self._basic)
        g = self._stage(g, 'value_encoder.layer2', 2, 2, self._basic)
        g = self._stage(g, 'value_encoder.layer3', 2, 2, self._basic)
        g = g.view(b, k, *g.shape[1:])
        g = self._fusion(f16, g, 'value_encoder.fuser')
        if is_deep_update and self.hidden_dim > 0:
            h16 = self._gru(g, h16, 'value_encoder.hidden_reinforce.transform')
        return g, h16

    def segment(self, multi_scale_features, memory_readout, hidden_state, selector=None, h_out=True, strip_bg=True):
        """model/network.py:107-120 + Decoder.forward model/modules.py:229-250."""
        f16, f8, f4 = multi_scale_features
        b, k = memory_readout.shape[:2]
        if self.hidden_dim > 0:
            g16 = self._fusion(f16, torch.cat([memory_readout, hidden_state], 2), 'decoder.fuser')
        else:
            g16 = self._fusion(f16, memory_readout, 'decoder.fuser')

        def up_block(skip, up_g, p):   # UpsampleBlock, model/modules.py:186-191
            skip = self._conv(skip, p + '.skip_conv', 1, 1)
            g = self._interp_groups(up_g, 2, 'bilinear', False)
            g = skip.unsqueeze(1).expand(-1, k, -1, -1, -1) + g
            return self._group_res(g, p + '.out_conv')

        g8 = up_block(f8, g16, 'decoder.up_16_8')
        g4 = up_block(f4, g8, 'decoder.up_8_4')
        logits = self._conv(F.relu(g4.flatten(0, 1)), 'decoder.pred', 1, 1)
        if h_out and self.hidden_dim > 0:
            g4c = torch.cat([g4, logits.view(b, k, 1, *logits.shape[-2:])], 2)
            g = self._gconv(g16, 'decoder.hidden_update.g16_conv', 0) + \
                self._gconv(self._interp_groups(g8, 1 / 2, 'area', None), 'decoder.hidden_update.g8_conv', 0) + \
                self._gconv(self._interp_groups(g4c, 1 / 4, 'area', None), 'decoder.hidden_update.g4_conv', 0)
            hidden_state = self._gru(g, hidden_state, 'decoder.hidden_update.transform')
        else:
            hidden_state = None
        logits = F.interpolate(logits, scale_factor=4, mode='bilinear', align_corners=False)
        logits = logits.view(b, k, *logits.shape[-2:])
        prob = torch.sigmoid(logits)
        if selector is not None:
            prob = prob * selector
        logits, prob = aggregate(prob, dim=1, return_logits=True)
        if strip_bg:
            prob = prob[:, 1:]
        return hidden_state, logits, prob


# ------------------------------------------------------------------------------------------
# The reference's GPU mode: torch.cuda.amp.autocast around the frame loop (inference/run_on_video.py:76), fp32 preload (:59-66)
# ------------------------------------------------------------------------------------------
# PARITY UNPINNED: CUDA autocast cannot run in the build container (no GPU, and the reference's CUDA kernels are not the ROCm
# build's), so nothing below can be compared with a run of the reference.  It is a RESTATEMENT of autocast's published
# per-operator policy (torch/csrc/autocast_mode.cpp, device type 'cuda') for exactly the operators this network issues:
#   * lower_precision_fp (fp16 in, fp16 out, fp32 accumulation inside the kernel): conv2d, linear, matmul / @ / bmm;
#     the convolution's bias is a separate fp16 `add_` after the kernel (ATen _convolution, cudnn / miopen backends);
#   * fp32 (inputs cast up, fp32 out): pow, exp, log, prod, sum, softmax;
#   * promote (widest input type): cat, stack;
#   * every other operator runs in the type of its inputs - batch_norm, relu, max_pool2d, avg_pool2d, adaptive_avg_pool2d
#     ('area'), upsample_bilinear2d, sigmoid, tanh, add, mul, mean, max on fp16 tensors compute in fp32 registers and round
#     ONCE to fp16; binary operators on one fp16 and one fp32 tensor promote to fp32.
# fp16 values are carried as real torch.float16 CPU tensors so that type promotion is what torch does; the arithmetic of every
# operator is done on .float() copies (products of two fp16 numbers are exact in fp32, the accumulation is fp32 as on the GPU,
# up to summation order).  Used by tests/golden/make_autocast_goldens.py to measure how far the reference's OWN GPU mode is
# from its fp32 CPU path on the golden clips - the floor the fp16 loop of this repository (xmem2_amd precision='fp16') is
# gated against in tests/test_gpu_fp16_loop.py.


def _u(fn, x):
    """unary operator in the type of its input (fp32 opmath, one rounding for fp16)"""
    return fn(x.float()).to(x.dtype)


def _b(fn, a, b):
    """binary operator with torch's type promotion (fp16 op fp16 -> fp16 rounded once; fp16 op fp32 -> fp32)"""
    dt = torch.result_type(a, b)
    return fn(a.float() if torch.is_tensor(a) else a, b.float() if torch.is_tensor(b) else b).to(dt)


def _amm(a, b):
    """autocast matmul: operands cast to fp16, fp32 accumulation, fp16 result"""
    return (a.half().float() @ b.half().float()).half()


class RefNetAutocast(RefNet):
    """RefNet under CUDA autocast's operator policy (see the block comment above).  PARITY UNPINNED (a restatement that cannot
    be run against CUDA here).  Same entry points and argument meaning as RefNet; fp16-typed outputs are torch.float16."""

    def _conv(self, x, name, stride=1, padding=0):
        y = F.conv2d(x.half().float(), self.sd[name + '.weight'].half().float(), None, stride, padding).half()
        b = self.sd.get(name + '.bias')
        if b is not None:
            y = _b(torch.add, y, b.half().view(1, -1, 1, 1))
        return y

    def _bn(self, x, name):
        return _u(lambda t: RefNet._bn(self, t, name), x)

    def _bottleneck(self, x, p, stride):
        relu = lambda t: _u(F.relu, t)
        out = relu(self._bn(self._conv(x, p + '.conv1'), p + '.bn1'))
        out = relu(self._bn(self._conv(out, p + '.conv2', stride, 1), p + '.bn2'))
        out = self._bn(self._conv(out, p + '.conv3'), p + '.bn3')
        if (p + '.downsample.0.weight') in self.sd:
            x = self._bn(self._conv(x, p + '.downsample.0', stride), p + '.downsample.1')
        return relu(_b(torch.add, out, x))

    def _basic(self, x, p, stride):
        relu = lambda t: _u(F.relu, t)
        out = relu(self._bn(self._conv(x, p + '.conv1', stride, 1), p + '.bn1'))
        out = self._bn(self._conv(out, p + '.conv2', 1, 1), p + '.bn2')
        if (p + '.downsample.0.weight') in self.sd:
            x = self._bn(self._conv(x, p + '.downsample.0', stride), p + '.downsample.1')
        return relu(_b(torch.add, out, x))

    def _group_res(self, g, p):
        out = self._gconv(_u(F.relu, g), p + '.conv1', 1)
        out = self._gconv(_u(F.relu, out), p + '.conv2', 1)
        if (p + '.downsample.weight') in self.sd:
            g = self._gconv(g, p + '.downsample', 1)
        return _b(torch.add, out, g)

    def _cbam(self, x, p):
        def mlp(v):
            v = v.view(v.size(0), -1)
            w1, b1 = self.sd[p + '.ChannelGate.mlp.1.weight'], self.sd[p + '.ChannelGate.mlp.1.bias']
            w2, b2 = self.sd[p + '.ChannelGate.mlp.3.weight'], self.sd[p + '.ChannelGate.mlp.3.bias']
            v = _u(F.relu, F.linear(v.half().float(), w1.half().float(), b1.half().float()).half())      # addmm: bias inside the GEMM epilogue
            return F.linear(v.float(), w2.half().float(), b2.half().float()).half()
        hw = (x.size(2), x.size(3))
        att = _b(torch.add, mlp(_u(lambda t: F.avg_pool2d(t, hw, stride=hw), x)), mlp(_u(lambda t: F.max_pool2d(t, hw, stride=hw), x)))
        x = _b(torch.mul, x, _u(torch.sigmoid, att).unsqueeze(2).unsqueeze(3).expand_as(x))
        comp = torch.cat((_u(lambda t: torch.max(t, 1)[0], x).unsqueeze(1), _u(lambda t: torch.mean(t, 1), x).unsqueeze(1)), dim=1)
        gate = self._conv(comp, p + '.SpatialGate.spatial.conv', 1, 3)
        return _b(torch.mul, x, _u(torch.sigmoid, gate))

    def _fusion(self, x, g, p):
        b, k = g.shape[:2]
        g = torch.cat([x.unsqueeze(1).expand(-1, k, -1, -1, -1), g], 2)              # cat promotes: fp32 when g carries the fp32 hidden state
        g = self._group_res(g, p + '.block1')
        r = self._cbam(g.flatten(0, 1), p + '.attention')
        r = r.view(b, k, *r.shape[1:])
        return self._group_res(_b(torch.add, g, r), p + '.block2')

    @staticmethod
    def _interp_groups(g, ratio, mode, align):
        return _u(lambda t: RefNet._interp_groups(t, ratio, mode, align), g)

    def _gru(self, g, h, name):
        hd = self.hidden_dim
        values = self._gconv(torch.cat([g, h], 2), name, 1)                               # fp16
        forget = _u(torch.sigmoid, values[:, :, :hd])
        update = _u(torch.sigmoid, values[:, :, hd:hd * 2])
        new_value = _u(torch.tanh, values[:, :, hd * 2:])
        keep = _b(torch.mul, _b(torch.mul, forget, h), _b(lambda a, c: a - c, 1, update))  # fp16 * fp32 -> fp32; (1 - update) is an fp16 result
        return _b(torch.add, keep, _b(torch.mul, update, new_value))                       # update * new_value is an fp16 result

    def encode_key(self, frame, need_sk=True, need_ek=True):
        if frame.dim() != 4:
            raise NotImplementedError
        x = _u(F.relu, self._bn(self._conv(frame, 'key_encoder.conv1', 2, 3), 'key_encoder.bn1'))
        x = _u(lambda t: F.max_pool2d(t, 3, 2, 1), x)
        f4 = self._stage(x, 'key_encoder.res2', 3, 1, self._bottleneck)
        f8 = self._stage(f4, 'key_encoder.layer2', 4, 2, self._bottleneck)
        f16 = self._stage(f8, 'key_encoder.layer3', 6, 2, self._bottleneck)
        shrinkage = self._conv(f16, 'key_proj.d_proj', 1, 1).float() ** 2 + 1 if need_sk else None      # pow: fp32 policy
        selection = _u(torch.sigmoid, self._conv(f16, 'key_proj.e_proj', 1, 1)) if need_ek else None
        key = self._conv(f16, 'key_proj.key_proj', 1, 1)
        return key, shrinkage, selection, f16, f8, f4

    def encode_value(self, frame, f16, h16, masks, is_deep_update=True):
        k = masks.shape[1]
        if k != 1:
            others = torch.cat([torch.sum(masks[:, [j for j in range(k) if i != j]], dim=1, keepdim=True) for i in range(k)], 1)
        else:
            others = torch.zeros_like(masks)
        g = torch.stack([masks, others], 2) if not self.single_object else masks.unsqueeze(2)
        g = torch.cat([frame.unsqueeze(1).expand(-1, k, -1, -1, -1), g], 2)
        b = g.shape[0]
        g = g.flatten(0, 1)
        g = self._bn(self._conv(g, 'value_encoder.conv1', 2, 3), 'value_encoder.bn1')
        g = _u(F.relu, _u(lambda t: F.max_pool2d(t, 3, 2, 1), g))
        g = self._stage(g, 'value_encoder.layer1', 2, 1, self._basic)
        g = self._stage(g, 'value_encoder.layer2', 2, 2, self._basic)
        g = self._stage(g, 'value_encoder.layer3', 2, 2, self._basic)
        g = g.view(b, k, *g.shape[1:])
        g = self._fusion(f16, g, 'value_encoder.fuser')
        if is_deep_update and self.hidden_dim > 0:
            h16 = self._gru(g, h16, 'value_encoder.hidden_reinforce.transform')
        return g, h16

    def segment(self, multi_scale_features, memory_readout, hidden_state, selector=None, h_out=True, strip_bg=True):
        f16, f8, f4 = multi_scale_features
        b, k = memory_readout.shape[:2]
        if self.hidden_dim > 0:
            g16 = self._fusion(f16, torch.cat([memory_readout, hidden_state], 2), 'decoder.fuser')
        else:
            g16 = self._fusion(f16, memory_readout, 'decoder.fuser')

        def up_block(skip, up_g, p):
            skip = self._conv(skip, p + '.skip_conv', 1, 1)
            g = self._interp_groups(up_g, 2, 'bilinear', False)
            g = _b(torch.add, skip.unsqueeze(1).expand(-1, k, -1, -1, -1), g)
            return self._group_res(g, p + '.out_conv')

        g8 = up_block(f8, g16, 'decoder.up_16_8')
        g4 = up_block(f4, g8, 'decoder.up_8_4')
        logits = self._conv(_u(F.relu, g4.flatten(0, 1)), 'decoder.pred', 1, 1)
        if h_out and self.hidden_dim > 0:
            g4c = torch.cat([g4, logits.view(b, k, 1, *logits.shape[-2:])], 2)
            g = _b(torch.add, _b(torch.add, self._gconv(g16, 'decoder.hidden_update.g16_conv', 0),
                                 self._gconv(self._interp_groups(g8, 1 / 2, 'area', None), 'decoder.hidden_update.g8_conv', 0)),
                   self._gconv(self._interp_groups(g4c, 1 / 4, 'area', None), 'decoder.hidden_update.g4_conv', 0))
            hidden_state = self._gru(g, hidden_state, 'decoder.hidden_update.transform')
        else:
            hidden_state = None
        logits = _u(lambda t: F.interpolate(t, scale_factor=4, mode='bilinear', align_corners=False), logits)
        logits = logits.view(b, k, *logits.shape[-2:])
        prob = _u(torch.sigmoid, logits)                                                    # fp16
        if selector is not None:
            prob = _b(torch.mul, prob, selector)
        # aggregate (model/aggregate.py:6-17): prod is an fp32-policy operator, cat promotes, log / softmax run in fp32
        one_minus = _b(lambda a, c: a - c, 1, prob)
        new_prob = torch.cat([torch.prod(one_minus.float(), dim=1, keepdim=True), prob], 1).clamp(1e-7, 1 - 1e-7)
        logits = torch.log((new_prob / (1 - new_prob)))
        prob = F.softmax(logits, dim=1)
        if strip_bg:
            prob = prob[:, 1:]
        return hidden_state, logits, prob


def get_similarity_autocast(mk, ms, qk, qe):
    """get_similarity (model/memory_util.py:7-39) as CUDA autocast runs it: the two N x HW GEMMs take fp16 operands and return
    fp16 (`@` is a lower_precision_fp operator), `pow` and `sum` run in fp32, the fp16 sum -a_sq + two_ab is rounded to fp16
    before b_sq (fp32) and the shrinkage (fp32) promote it.  PARITY UNPINNED, see RefNetAutocast."""
    ck = mk.shape[1]
    mk = mk.flatten(start_dim=2)
    ms = ms.flatten(start_dim=1).unsqueeze(2) if ms is not None else None
    qk = qk.flatten(start_dim=2)
    qe = qe.flatten(start_dim=2) if qe is not None else None
    if qe is not None:
        mkt = mk.transpose(1, 2)
        a_sq = _amm(mkt.float().pow(2), qe)
        two_ab = _b(torch.mul, 2, _amm(mkt, _b(torch.mul, qk, qe)))
        b_sq = (qe.float() * qk.float().pow(2)).sum(1, keepdim=True)
        sim = _b(torch.add, _u(torch.neg, a_sq), two_ab).float() - b_sq
    else:
        a_sq = mk.float().pow(2).sum(1).unsqueeze(2)
        two_ab = _b(torch.mul, 2, _amm(mk.transpose(1, 2), qk))
        sim = -a_sq + two_ab.float()
    if ms is not None:
        sim = sim * ms.float() / math.sqrt(ck)
    else:
        sim = sim / math.sqrt(ck)
    return sim


# ------------------------------------------------------------------------------------------
# inference/kv_memory_store.py
# ------------------------------------------------------------------------------------------


class RefStore:
    """Growable key/value store with object groups, inference/kv_memory_store.py:4-240."""

    def __init__(self, count_usage):
        self.count_usage = count_usage
        self.k = None
        self.v = []
        self.obj_groups = []
        self.all_objects = []
        self.s = self.e = None
        if count_usage:
            self.use_count = self.life_count = None

    def add(self, key, value, shrinkage, selection, objects):
        """kv_memory_store.py:36-94."""
        new_count = torch.zeros((key.shape[0], 1, key.shape[2]), dtype=torch.float32)
        new_life = torch.zeros((key.shape[0], 1, key.shape[2]), dtype=torch.float32) + 1e-7
        if self.k is None:
            self.k, self.s, self.e = key, shrinkage, selection
            if self.count_usage:
                self.use_count, self.life_count = new_count, new_life
        else:
            self.k = torch.cat([self.k, key], -1)
            if shrinkage is not None:
                self.s = torch.cat([self.s, shrinkage], -1)
            if selection is not None:
                self.e = torch.cat([self.e, selection], -1)
            if self.count_usage:
                self.use_count = torch.cat([self.use_count, new_count], -1)
                self.life_count = torch.cat([self.life_count, new_life], -1)
        if objects is not None:
            assert isinstance(value, torch.Tensor)
            remaining = [o - 1 for o in objects]
            for gi, group in enumerate(self.obj_groups):
                for o in group:
                    remaining.remove(o)
                self.v[gi] = torch.cat([self.v[gi], value[group]], -1)
            if remaining:
                group = list(remaining)
                self.v.append(value[group])
                self.obj_groups.append(group)
                self.all_objects.extend(group)
                assert sorted(self.all_objects) == self.all_objects, 'Objects MUST be inserted in sorted order '
        else:
            assert isinstance(value, list)
            for gi, gv in enumerate(value):
                if gv is None:
                    continue
                if gi < self.num_groups:
                    self.v[gi] = torch.cat([self.v[gi], gv], -1)
                else:
                    self.v.append(gv)
        return int((self.k.shape[-1] + 1e-9) // (key.shape[-1] + 1e-9)) - 1

    def update_usage(self, usage):
        """kv_memory_store.py:96-103."""
        if not self.count_usage:
            return
        self.use_count += usage.view_as(self.use_count)
        self.life_count += 1

    def replace_at(self, start_pos, key, value, shrinkage=None, selection=None):
        """kv_memory_store.py:105-118."""
        n = key.shape[-1]
        a, b = start_pos * n, (start_pos + 1) * n
        self.k[:, :, a:b] = key
        for gi in range(self.num_groups):
            self.v[gi][:, :, a:b] = value[gi]
        if self.s is not None and shrinkage is not None:
            self.s[:, :, a:b] = shrinkage
        if self.e is not None and selection is not None:
            self.e[:, :, a:b] = selection

    def remove_at(self, start, elem_size):
        """kv_memory_store.py:120-123."""
        self.sieve_by_range(start, start + elem_size, min_size=0)

    def sieve_by_range(self, start, end, min_size):
        """Keep elements outside [start, end), kv_memory_store.py:125-158."""
        if end == 0:
            cut = lambda t: t[:, :, :start]
        else:
            cut = lambda t: torch.cat([t[:, :, :start], t[:, :, end:]], -1)
        self.k = cut(self.k)
        if self.count_usage:
            self.use_count = cut(self.use_count)
            self.life_count = cut(self.life_count)
        if self.s is not None:
            self.s = cut(self.s)
        if self.e is not None:
            self.e = cut(self.e)
        for gi in range(self.num_groups):
            if self.v[gi].shape[-1] >= min_size:
                self.v[gi] = cut(self.v[gi])

    def remove_obsolete_features(self, max_size):
        """Least-used eviction, kv_memory_store.py:160-181."""
        usage = self.get_usage().flatten()
        values, _ = torch.topk(usage, k=(self.size - max_size), largest=False, sorted=True)
        survived = usage > values[-1]
        self.k = self.k[:, :, survived]
        self.s = self.s[:, :, survived] if self.s is not None else None
        self.e = self.e[:, :, survived] if self.e is not None else None
        if self.num_groups > 1:
            raise NotImplementedError('feature removal with multiple object groups')
        for gi in range(self.num_groups):
            self.v[gi] = self.v[gi][:, :, survived]
        self.use_count = self.use_count[:, :, survived]
        self.life_count = self.life_count[:, :, survived]

    def get_usage(self):
        if not self.count_usage:
            raise RuntimeError('I did not count usage!')
        return self.use_count / self.life_count

    def get_all_sliced(self, start, end):
        """kv_memory_store.py:191-206."""
        sl = slice(start, None) if end == 0 else slice(start, end)
        pick = lambda t: t[:, :, sl] if t is not None else None
        return pick(self.k), pick(self.s), pick(self.e), self.get_usage()[:, :, sl]

    def get_v_size(self, ni):
        return self.v[ni].shape[2]

    def engaged(self):
        return self.k is not None

    @property
    def size(self):
        return 0 if self.k is None else self.k.shape[-1]

    @property
    def num_groups(self):
        return len(self.v)

    key = property(lambda self: self.k)
    value = property(lambda self: self.v)
    shrinkage = property(lambda self: self.s)
    selection = property(lambda self: self.e)


# ------------------------------------------------------------------------------------------
# inference/memory_manager.py
# ------------------------------------------------------------------------------------------


class RefMemory:
    """Temporary / permanent / long-term stores and the readout, inference/memory_manager.py:8-425."""

    def __init__(self, config):
        self.config = config
        self.hidden_dim = config['hidden_dim']
        self.top_k = config['top_k']
        self.enable_long_term = config['enable_long_term']
        self.enable_long_term_usage = config['enable_long_term_count_usage']
        if self.enable_long_term:
            self._read_lt_config(config)
        self.CK = self.CV = None
        self.H = self.W = None
        self.hidden = None
        self.temporary_work_mem = RefStore(count_usage=self.enable_long_term)
        self.permanent_work_mem = RefStore(count_usage=False)
        self.frame_id_to_permanent_mem_idx = dict()
        if self.enable_long_term:
            self.long_mem = RefStore(count_usage=self.enable_long_term_usage)
        self.reset_config = True
        self.autocast = False          # set by RefCore(autocast_network=...) around step(): the matmuls below as CUDA autocast runs them

    def _sim(self, mk, ms, qk, qe):
        if self.autocast:
            return get_similarity_autocast(mk, ms, qk, qe)
        f = lambda t: t.float() if t is not None else None          # (no-ops on the fp32 path; fp16 keys of an autocast network promote)
        return get_similarity(f(mk), f(ms), f(qk), f(qe))

    def _mm(self, a, b):
        return _amm(a, b) if self.autocast else a.float() @ b.float()

    def _read_lt_config(self, config):
        self.max_mt_frames = config['max_mid_term_frames']
        self.min_mt_frames = config['min_mid_term_frames']
        self.num_prototypes = config['num_prototypes']
        self.max_long_elements = config['max_long_term_elements']

    def update_config(self, config):
        """memory_manager.py:42-55."""
        self.reset_config = True
        self.hidden_dim = config['hidden_dim']
        self.top_k = config['top_k']
        assert self.enable_long_term == config['enable_long_term'], 'cannot update this'
        assert self.enable_long_term_usage == config['enable_long_term_count_usage'], 'cannot update this'
        self.enable_long_term_usage = config['enable_long_term_count_usage']
        if self.enable_long_term:
            self._read_lt_config(config)

    def match_memory(self, query_key, selection, disable_usage_updates=False):
        """memory_manager.py:61-190: similarity over [long|temp|perm], per-group top-k softmax, readout."""
        tmp, perm = self.temporary_work_mem, self.permanent_work_mem
        num_groups = max(tmp.num_groups, perm.num_groups)
        h, w = query_key.shape[-2:]
        query_key = query_key.flatten(start_dim=2)
        selection = selection.flatten(start_dim=2) if selection is not None else None
        n_tmp = tmp.size
        use_long = self.enable_long_term and self.long_mem.engaged()
        if use_long:
            lt = self.long_mem
            n_long = lt.size
            memory_key = torch.cat([lt.key, tmp.key, perm.key], -1)
            shrinkage = torch.cat([lt.shrinkage, tmp.shrinkage, perm.shrinkage], -1)
            similarity = self._sim(memory_key, shrinkage, query_key, selection)
            sim_long = similarity[:, :n_long]
            sim_tmp = similarity[:, n_long:n_long + n_tmp]
            sim_perm = similarity[:, n_long + n_tmp:]
            affinity, usage = do_softmax(
                torch.cat([sim_long[:, -lt.get_v_size(0):], sim_tmp, sim_perm], 1),
                top_k=self.top_k, inplace=True, return_usage=True)
            affinity = [affinity]
            for gi in range(1, num_groups):
                tv, pv = tmp.get_v_size(gi), perm.get_v_size(gi)
                parts = [sim_tmp[:, sim_tmp.shape[1] - tv:], sim_perm[:, sim_perm.shape[1] - pv:]]
                if gi < lt.num_groups:
                    parts = [sim_long[:, -lt.get_v_size(gi):]] + parts
                    aff = do_softmax(torch.cat(parts, dim=1), top_k=self.top_k, inplace=True)
                else:
                    aff = do_softmax(torch.cat(parts, 1), top_k=self.top_k, inplace=(gi == num_groups - 1))
                affinity.append(aff)
            all_value = []
            for gi in range(num_groups):
                if gi < lt.num_groups:
                    all_value.append(torch.cat([lt.value[gi], tmp.value[gi], perm.value[gi]], -1))
                else:
                    all_value.append(torch.cat([tmp.value[gi], perm.value[gi]], -1))
            if not disable_usage_updates:
                tmp.update_usage(usage[:, n_long:n_long + n_tmp].flatten())
                if self.enable_long_term_usage:
                    lt.update_usage(usage[:, :n_long].flatten())
        else:
            memory_key = torch.cat([tmp.key, perm.key], -1)
            shrinkage = torch.cat([tmp.shrinkage, perm.shrinkage], -1)
            similarity = self._sim(memory_key, shrinkage, query_key, selection)
            sim_tmp = similarity[:, :n_tmp]
            sim_perm = similarity[:, n_tmp:]
            if self.enable_long_term:
                affinity, usage = do_softmax(similarity, inplace=(num_groups == 1), top_k=self.top_k, return_usage=True)
                if not disable_usage_updates:
                    tmp.update_usage(usage[:, :n_tmp].flatten())
            else:
                affinity = do_softmax(similarity, inplace=(num_groups == 1), top_k=self.top_k, return_usage=False)
            affinity = [affinity]
            for gi in range(1, num_groups):
                tv, pv = tmp.get_v_size(gi), perm.get_v_size(gi)
                aff = do_softmax(torch.cat([sim_tmp[:, sim_tmp.shape[1] - tv:], sim_perm[:, sim_perm.shape[1] - pv:]], dim=1),
                                 top_k=self.top_k, inplace=(gi == num_groups - 1))
                affinity.append(aff)
            all_value = [torch.cat([tmp.value[gi], perm.value[gi]], -1) for gi in range(num_groups)]
        out = torch.cat([self._mm(gv, affinity[gi]) for gi, gv in enumerate(all_value)], 0)   # _readout, :57-59
        return out.view(out.shape[0], self.CV, h, w)

    def update_permanent_memory(self, frame_idx, key, shrinkage, value, selection=None):
        """memory_manager.py:192-202."""
        pos = self.frame_id_to_permanent_mem_idx[frame_idx]
        key = key.flatten(start_dim=2)
        shrinkage = shrinkage.flatten(start_dim=2)
        value = value[0].flatten(start_dim=2)
        if selection is not None:
            selection = selection.flatten(start_dim=2)
        self.permanent_work_mem.replace_at(pos, key, value, shrinkage, selection)

    def remove_from_permanent_memory(self, frame_idx):
        """memory_manager.py:204-210 (frame index used as element offset - reference quirk)."""
        pos = self.frame_id_to_permanent_mem_idx[frame_idx]
        self.permanent_work_mem.remove_at(pos, self.HW)
        del self.frame_id_to_permanent_mem_idx[frame_idx]

    def add_memory(self, key, shrinkage, value, objects, selection=None, permanent=False, ignore=False, ti=None):
        """memory_manager.py:212-281."""
        if self.H is None or self.reset_config:
            self.reset_config = False
            self.H, self.W = key.shape[-2:]
            self.HW = self.H * self.W
            if self.enable_long_term:
                self.min_work_elements = self.min_mt_frames * self.HW
                self.max_work_elements = self.max_mt_frames * self.HW
        key = key.flatten(start_dim=2)
        shrinkage = shrinkage.flatten(start_dim=2)
        value = value[0].flatten(start_dim=2)
        self.CK = key.shape[1]
        self.CV = value.shape[1]
        if selection is not None:
            if not self.enable_long_term:
                warnings.warn('the selection factor is only needed in long-term mode', UserWarning)
            selection = selection.flatten(start_dim=2)
        if ignore:
            pass
        elif permanent:
            pos = self.permanent_work_mem.add(key, value, shrinkage, selection, objects)
            if ti is not None:
                self.frame_id_to_permanent_mem_idx[ti] = pos
        else:
            self.temporary_work_mem.add(key, value, shrinkage, selection, objects)
        nt, npm = self.temporary_work_mem.num_groups, self.permanent_work_mem.num_groups
        if not self.temporary_work_mem.engaged() or (nt != npm):
            empty = (key[..., 0:0], value[..., 0:0], shrinkage[..., 0:0], selection[..., 0:0], objects)
            (self.temporary_work_mem if npm > nt else self.permanent_work_mem).add(*empty)
        if self.enable_long_term:
            if self.temporary_work_mem.size >= self.max_work_elements:
                if self.long_mem.size >= (self.max_long_elements - self.num_prototypes):
                    self.long_mem.remove_obsolete_features(self.max_long_elements - self.num_prototypes)
                self.compress_features()

    def create_hidden_state(self, n, sample_key):
        """memory_manager.py:283-294."""
        h, w = sample_key.shape[-2:]
        if self.hidden is None:
            self.hidden = torch.zeros((1, n, self.hidden_dim, h, w))
        elif self.hidden.shape[1] != n:
            self.hidden = torch.cat([self.hidden, torch.zeros((1, n - self.hidden.shape[1], self.hidden_dim, h, w))], 1)
        assert self.hidden.shape[1] == n

    def set_hidden(self, hidden):
        self.hidden = hidden

    def get_hidden(self):
        return self.hidden

    def frame_already_saved(self, ti):
        return ti in self.frame_id_to_permanent_mem_idx

    def compress_features(self):
        """memory_manager.py:316-347."""
        hw = self.HW
        total = self.temporary_work_mem.size
        candidate_value = []
        for gv in self.temporary_work_mem.value:
            n = gv.shape[-1]
            if n == total:
                candidate_value.append(gv[:, :, :-self.min_work_elements])
            else:
                assert hw <= n < total
                candidate_value.append(gv[:, :, :-self.min_work_elements] if n > self.min_work_elements else None)
        pk, pv, ps = self.consolidation(*self.temporary_work_mem.get_all_sliced(0, -self.min_work_elements), candidate_value)
        self.temporary_work_mem.sieve_by_range(0, -self.min_work_elements, min_size=self.min_work_elements + hw)
        self.long_mem.add(pk, pv, ps, selection=None, objects=None)

    def consolidation(self, cand_key, cand_shrinkage, cand_selection, usage, cand_value):
        """Prototype selection + potentiation, memory_manager.py:349-390."""
        n = cand_key.shape[-1]
        _, idx = torch.topk(usage, k=self.num_prototypes, dim=-1, sorted=True)
        proto_idx = idx.flatten()
        validity = [proto_idx >= (n - gv.shape[2]) if gv is not None else None for gv in cand_value]
        proto_key = cand_key[:, :, proto_idx]
        proto_sel = cand_selection[:, :, proto_idx] if cand_selection is not None else None
        similarity = self._sim(cand_key, cand_shrinkage, proto_key, proto_sel)
        affinity = [do_softmax(similarity[:, -gv.shape[2]:, validity[gi]]) if gv is not None else None
                    for gi, gv in enumerate(cand_value)]
        affinity = [a if a is None or a.shape[-1] > 0 else None for a in affinity]
        proto_value = [self._mm(gv, affinity[gi]) if affinity[gi] is not None else None for gi, gv in enumerate(cand_value)]
        proto_shrinkage = self._mm(cand_shrinkage, affinity[0]) if cand_shrinkage is not None else None
        return proto_key, proto_value, proto_shrinkage

    def copy_perm_mem_only(self):
        """memory_manager.py:392-425."""
        new = RefMemory(config=self.config)
        perm = self.permanent_work_mem
        if perm.key is None or perm.key.size(-1) == 0:
            return new
        new.permanent_work_mem = perm
        new.frame_id_to_permanent_mem_idx = self.frame_id_to_permanent_mem_idx
        new.temporary_work_mem.add(perm.key[..., 0:0], perm.value[0][..., 0:0],
                                   perm.shrinkage[..., 0:0] if perm.shrinkage is not None else None,
                                   perm.selection[..., 0:0] if perm.selection is not None else None,
                                   perm.all_objects)
        shape = perm.key.shape
        sample_key = perm.key[..., 0:self.HW].view(*shape[:-1], self.H, self.W)
        new.create_hidden_state(len(perm.all_objects), sample_key)
        new.temporary_work_mem.obj_groups = self.temporary_work_mem.obj_groups
        new.temporary_work_mem.all_objects = self.temporary_work_mem.all_objects
        new.CK, new.CV, new.H, new.W, new.HW = self.CK, self.CV, self.H, self.W, self.HW
        return new


# ------------------------------------------------------------------------------------------
# inference/inference_core.py
# ------------------------------------------------------------------------------------------


class RefCore:
    """Per-frame state machine, inference/inference_core.py:11-185 (without the cuda:0 warm-up :26)."""

    def __init__(self, network, config, autocast_network=None, autocast_memory=True):
        """autocast_network (a RefNetAutocast over the same state_dict): the reference's GPU mode - `step()` runs under CUDA
        autocast's operator policy (network AND the memory's matmuls, inference/run_on_video.py:76) while
        `put_to_permanent_memory` stays fp32 (:59-66).  autocast_memory=False keeps the memory's matmuls (similarity, readout,
        consolidation) in fp32 while the network follows the policy - what xmem2_amd's fp16 loop does.  PARITY UNPINNED for
        that mode, see RefNetAutocast."""
        self.config = config
        self.network = network
        self._fp32_network, self._autocast_network, self._autocast_memory = network, autocast_network, bool(autocast_memory)
        self._read_config(config)
        self.clear_memory()
        self.all_labels = None

    def _read_config(self, config):
        self.mem_every = config['mem_every']
        self.deep_update_every = config['deep_update_every']
        self.enable_long_term = config['enable_long_term']
        self.deep_update_sync = (self.deep_update_every < 0)

    def clear_memory(self, keep_permanent=False):
        """inference_core.py:28-38."""
        self.curr_ti = -1
        self.last_mem_ti = 0
        if not self.deep_update_sync:
            self.last_deep_update_ti = -self.deep_update_every
        self.memory = self.memory.copy_perm_mem_only() if keep_permanent else RefMemory(config=self.config)

    def update_config(self, config):
        self._read_config(config)
        self.memory.update_config(config)

    def set_all_labels(self, all_labels):
        self.all_labels = all_labels

    def encode_frame_key(self, image):
        """inference_core.py:53-61."""
        image, self.pad = pad_divide_by(image, 16)
        key, shrinkage, selection, _, _, _ = self.network.encode_key(image.unsqueeze(0), need_ek=True, need_sk=True)
        return key, shrinkage, selection

    def step(self, image, mask=None, valid_labels=None, end=False, manually_curated_masks=False,
             disable_memory_updates=False, do_not_add_mask_to_memory=False, return_key_and_stuff=False):
        """inference_core.py:62-152."""
        if self._autocast_network is not None:
            self.network, self.memory.autocast = self._autocast_network, self._autocast_memory
            try:
                return self._step(image, mask, valid_labels, end, manually_curated_masks, disable_memory_updates,
                                  do_not_add_mask_to_memory, return_key_and_stuff)
            finally:
                self.network, self.memory.autocast = self._fp32_network, False
        return self._step(image, mask, valid_labels, end, manually_curated_masks, disable_memory_updates,
                          do_not_add_mask_to_memory, return_key_and_stuff)

    def _step(self, image, mask, valid_labels, end, manually_curated_masks, disable_memory_updates, do_not_add_mask_to_memory,
              return_key_and_stuff):
        self.curr_ti += 1
        image, self.pad = pad_divide_by(image, 16)
        image = image.unsqueeze(0)
        if manually_curated_masks:
            is_mem_frame = (mask is not None) and (not end)
        else:
            is_mem_frame = ((self.curr_ti - self.last_mem_ti >= self.mem_every) or (mask is not None)) and (not end)
        need_segment = (valid_labels is None) or (len(self.all_labels) != len(valid_labels))
        is_deep_update = ((self.deep_update_sync and is_mem_frame) or
                          (not self.deep_update_sync and self.curr_ti - self.last_deep_update_ti >= self.deep_update_every)
                          ) and (not end)
        is_normal_update = (not self.deep_update_sync or not is_deep_update) and (not end)
        key, shrinkage, selection, f16, f8, f4 = self.network.encode_key(
            image, need_ek=(self.enable_long_term or need_segment), need_sk=True)
        if disable_memory_updates:
            is_normal_update = is_deep_update = is_mem_frame = False
            self.curr_ti -= 1
        if need_segment:
            readout = self.memory.match_memory(key, selection, disable_usage_updates=disable_memory_updates).unsqueeze(0)
            hidden, _, prob_bg = self.network.segment((f16, f8, f4), readout, self.memory.get_hidden(),
                                                      h_out=is_normal_update, strip_bg=False)
            prob_bg = prob_bg[0]
            prob_no_bg = prob_bg[1:]
            if is_normal_update:
                self.memory.set_hidden(hidden)
        else:
            prob_no_bg = prob_bg = None
        if mask is not None:
            mask, _ = pad_divide_by(mask, 16)
            if prob_no_bg is not None:
                regions = (mask.sum(0) > 0.5)
                prob_no_bg[:, regions] = 0
                mask = mask.type_as(prob_no_bg)
                if valid_labels is not None:
                    keep = [i for i in range(prob_no_bg.shape[0]) if (i + 1) not in valid_labels]
                    mask[keep] = prob_no_bg[keep]
            prob_bg = aggregate(mask, dim=0)
            if not disable_memory_updates:
                self.memory.create_hidden_state(len(self.all_labels), key)
        if is_mem_frame:
            value, hidden = self.network.encode_value(image, f16, self.memory.get_hidden(),
                                                      prob_bg[1:].unsqueeze(0), is_deep_update=is_deep_update)
            self.memory.add_memory(key, shrinkage, value, self.all_labels,
                                   selection=selection if self.enable_long_term else None,
                                   ignore=do_not_add_mask_to_memory)
            self.last_mem_ti = self.curr_ti
            if is_deep_update:
                self.memory.set_hidden(hidden)
                self.last_deep_update_ti = self.curr_ti
        res = unpad(prob_bg, self.pad)
        return (res, key, shrinkage, selection) if return_key_and_stuff else res

    def put_to_permanent_memory(self, image, mask, ti=None):
        """inference_core.py:154-179."""
        image, self.pad = pad_divide_by(image, 16)
        image = image.unsqueeze(0)
        key, shrinkage, selection, f16, _, _ = self.network.encode_key(image, need_ek=True, need_sk=True)
        mask, _ = pad_divide_by(mask, 16)
        prob_bg = aggregate(mask, dim=0)
        self.memory.create_hidden_state(len(self.all_labels), key)
        value, _ = self.network.encode_value(image, f16, self.memory.get_hidden(), prob_bg[1:].unsqueeze(0),
                                             is_deep_update=False)
        is_update = self.memory.frame_already_saved(ti)
        sel = selection if self.enable_long_term else None
        if is_update:
            self.memory.update_permanent_memory(ti, key, shrinkage, value, selection=sel)
        else:
            self.memory.add_memory(key, shrinkage, value, self.all_labels, selection=sel, permanent=True, ti=ti)
        return is_update

    def remove_from_permanent_memory(self, frame_idx):
        self.memory.remove_from_permanent_memory(frame_idx)

    @property
    def permanent_memory_frames(self):
        return list(self.memory.frame_id_to_permanent_mem_idx.keys())


def post_process(prob, shape=None):
    """Resize (if needed) + argmax -> uint8 index mask, inference/run_on_video.py:165-173."""
    if shape is not None and tuple(prob.shape[-2:]) != tuple(shape):
        prob = F.interpolate(prob.unsqueeze(1), tuple(shape), mode='bilinear', align_corners=False)[:, 0]
    return torch.argmax(prob, dim=0).cpu().numpy().astype(np.uint8)


class RefMaskMapper:
    """Index mask -> one-hot with label remapping, inference/data/mask_mapper.py:7-63."""

    def __init__(self):
        self.labels = []
        self.remappings = {}
        self.coherent = True

    def convert_mask(self, mask, exhaustive=False):
        labels = np.unique(mask).astype(np.uint8)
        labels = labels[labels != 0].tolist()
        new_labels = list(set(labels) - set(self.labels))
        if not exhaustive:
            assert len(new_labels) == len(labels), 'Old labels found in non-exhaustive mode'
        for i, l in enumerate(new_labels):
            self.remappings[l] = i + len(self.labels) + 1
            if self.coherent and i + len(self.labels) + 1 != l:
                self.coherent = False
        if exhaustive:
            mapped = range(1, len(self.labels) + len(new_labels) + 1)
        elif self.coherent:
            mapped = new_labels
        else:
            mapped = range(len(self.labels) + 1, len(self.labels) + len(new_labels) + 1)
        self.labels.extend(new_labels)
        onehot = np.stack([(mask == l).astype(np.uint8) for l in self.labels]) if self.labels else \
            np.zeros((0,) + mask.shape, np.uint8)
        return torch.from_numpy(onehot).float(), mapped

    def remap_index_mask(self, mask):
        if self.coherent:
            return mask
        out = np.zeros_like(mask)
        for l, i in self.remappings.items():
            out[mask == i] = l
        return out


# ---------------------------------------------------------------------------------------------
# Annotation-candidate selector
# ---------------------------------------------------------------------------------------------
# NOTE parity: inference/frame_selection/* cannot be imported in the build container (torchvision / cv2 are absent), so
# this restatement is NOT pinned against a run of the reference.  Its arithmetic core is `get_similarity` above (pinned
# bit-equal); the loop below follows frame_selection.py:99-244 line by line, with torchvision's tensor
# Resize(NEAREST) restated as F.interpolate(mode='nearest'), which is what torchvision dispatches to for tensors.

def cycle_dissimilarity(key_a, shr_a, sel_a, key_b, shr_b, sel_b):
    """frame_selection.py:218-226.  key_* C_k x h x w (composite), shr_* 1 x h x w, sel_* C_k x h x w -> 0-d tensor."""
    fwd = get_similarity(key_a.unsqueeze(0), shr_a.unsqueeze(0), key_b.unsqueeze(0), sel_b.unsqueeze(0))
    rev = get_similarity(key_b.unsqueeze(0), shr_b.unsqueeze(0), key_a.unsqueeze(0), sel_a.unsqueeze(0))
    diff = (fwd - rev).to(torch.float32)
    return F.relu(diff).sum() / diff.numel()


def composite_keys_and_validity(keys, masks, previously_chosen, alpha, min_mask_presence_percent, epsilon):
    """frame_selection.py:150-186: mask presence test + mask-weighted composite keys (None for ignored frames)."""
    n = len(keys)
    h, w = keys[0].shape[1:3]
    valid = np.full(n, True)
    composite = []
    for i, mask in enumerate(masks):
        m3 = mask if mask.ndim == 3 else mask.unsqueeze(0)
        merged = m3.max(dim=0).values
        percent = (merged > epsilon).sum() / merged.numel() * 100
        if percent < min_mask_presence_percent and i not in previously_chosen:
            valid[i] = False
            composite.append(None)
            continue
        small = F.interpolate(m3.unsqueeze(0).float(), size=(h, w), mode='nearest')[0]
        ck = keys[i] * small.max(dim=0, keepdim=True).values
        ck = ck * alpha + keys[i] * (1 - alpha)
        composite.append(ck.to(dtype=keys[i].dtype))
    return composite, valid


def select_next_candidates(keys, shrinkages, selections, masks, num_next_candidates, previously_chosen_candidates=(0,),
                           alpha=0.5, min_mask_presence_percent=0.25, only_new_candidates=True, epsilon=0.5):
    """frame_selection.py:99-244 (greedy farthest-point selection under the cycle dissimilarity), O(k^2 N) like the
    reference.  keys F x C_k x h x w, shrinkages F x 1 x h x w, selections F x C_k x h x w, masks list of C x H x W."""
    assert len(keys) == len(masks) and len(keys) > 0
    assert num_next_candidates > 0 and len(previously_chosen_candidates) > 0
    assert 0.0 <= alpha <= 1.0 and min_mask_presence_percent >= 0
    assert len(previously_chosen_candidates) < len(keys)
    n = len(keys)
    composite, valid = composite_keys_and_validity(keys, masks, previously_chosen_candidates, alpha,
                                                   min_mask_presence_percent, epsilon)
    chosen = list(previously_chosen_candidates)
    trace = []
    for _ in range(num_next_candidates):
        scores = []
        for j in range(n):
            if not valid[j]:
                scores.append(0)
                continue
            per_chosen = [cycle_dissimilarity(composite[m], shrinkages[m], selections[m],
                                              composite[j], shrinkages[j], selections[j]) for m in chosen]
            scores.append(min(per_chosen))
        scores_t = torch.tensor(scores)
        trace.append(scores_t.to(torch.float32).numpy().copy())
        chosen.append(int(torch.argmax(scores_t)))
    select_next_candidates.last_scores = trace
    return chosen[len(previously_chosen_candidates):] if only_new_candidates else chosen
