"""Architecture table of the XMem network on the hot path.

The reference builds its network from nn.Modules (model/network.py:18-38,
model/modules.py:102-250, model/resnet.py:117-164).  Here the same network is
described as a flat table: every parameter tensor with the name and shape an
upstream ``XMem.pth`` checkpoint carries (412 tensors), generated in the same
order ``nn.Module.state_dict()`` would emit them, so an upstream checkpoint
loads unchanged and a synthetic one can be produced without torch modules.
"""
from collections import OrderedDict

BN_EPS = 1e-5  # nn.BatchNorm2d default used by model/resnet.py


def _conv(spec, name, cout, cin, k, bias):
    spec[name + '.weight'] = (cout, cin, k, k)
    if bias:
        spec[name + '.bias'] = (cout,)


def _bn(spec, name, c):
    spec[name + '.weight'] = (c,)
    spec[name + '.bias'] = (c,)
    spec[name + '.running_mean'] = (c,)
    spec[name + '.running_var'] = (c,)
    spec[name + '.num_batches_tracked'] = ()


def _bottleneck_layer(spec, prefix, inplanes, planes, blocks, stride):
    """ResNet-50 stage (model/resnet.py:78-114, 141-152)."""
    for b in range(blocks):
        p = f'{prefix}.{b}'
        cin = inplanes if b == 0 else planes * 4
        _conv(spec, p + '.conv1', planes, cin, 1, False); _bn(spec, p + '.bn1', planes)
        _conv(spec, p + '.conv2', planes, planes, 3, False); _bn(spec, p + '.bn2', planes)
        _conv(spec, p + '.conv3', planes * 4, planes, 1, False); _bn(spec, p + '.bn3', planes * 4)
        if b == 0 and (stride != 1 or inplanes != planes * 4):
            _conv(spec, p + '.downsample.0', planes * 4, inplanes, 1, False)
            _bn(spec, p + '.downsample.1', planes * 4)
    return planes * 4


def _basic_layer(spec, prefix, inplanes, planes, blocks, stride):
    """ResNet-18 stage (model/resnet.py:46-75, 141-152)."""
    for b in range(blocks):
        p = f'{prefix}.{b}'
        cin = inplanes if b == 0 else planes
        _conv(spec, p + '.conv1', planes, cin, 3, False); _bn(spec, p + '.bn1', planes)
        _conv(spec, p + '.conv2', planes, planes, 3, False); _bn(spec, p + '.bn2', planes)
        if b == 0 and (stride != 1 or inplanes != planes):
            _conv(spec, p + '.downsample.0', planes, inplanes, 1, False)
            _bn(spec, p + '.downsample.1', planes)
    return planes


def _group_res_block(spec, prefix, cin, cout):
    """GroupResBlock (model/group_modules.py:32-52): downsample registered first."""
    if cin != cout:
        _conv(spec, prefix + '.downsample', cout, cin, 3, True)
    _conv(spec, prefix + '.conv1', cout, cin, 3, True)
    _conv(spec, prefix + '.conv2', cout, cout, 3, True)


def _fusion_block(spec, prefix, x_in, g_in, g_mid, g_out):
    """FeatureFusionBlock (model/modules.py:22-41) with CBAM (model/cbam.py:21-77)."""
    _group_res_block(spec, prefix + '.block1', x_in + g_in, g_mid)
    spec[prefix + '.attention.ChannelGate.mlp.1.weight'] = (g_mid // 16, g_mid)
    spec[prefix + '.attention.ChannelGate.mlp.1.bias'] = (g_mid // 16,)
    spec[prefix + '.attention.ChannelGate.mlp.3.weight'] = (g_mid, g_mid // 16)
    spec[prefix + '.attention.ChannelGate.mlp.3.bias'] = (g_mid,)
    _conv(spec, prefix + '.attention.SpatialGate.spatial.conv', 1, 2, 7, True)
    _group_res_block(spec, prefix + '.block2', g_mid, g_out)


def state_dict_spec(key_dim=64, value_dim=512, hidden_dim=64, single_object=False):
    """name -> shape for every tensor of the reference's state_dict, in order."""
    s = OrderedDict()
    # KeyEncoder: ResNet-50 stem..layer3 (model/modules.py:153-175)
    _conv(s, 'key_encoder.conv1', 64, 3, 7, False); _bn(s, 'key_encoder.bn1', 64)
    c = _bottleneck_layer(s, 'key_encoder.res2', 64, 64, 3, 1)
    c = _bottleneck_layer(s, 'key_encoder.layer2', c, 128, 4, 2)
    c = _bottleneck_layer(s, 'key_encoder.layer3', c, 256, 6, 2)
    # ValueEncoder: ResNet-18 stem..layer3 + fuser + GRU (model/modules.py:102-150)
    extra = 1 if single_object else 2
    _conv(s, 'value_encoder.conv1', 64, 3 + extra, 7, False); _bn(s, 'value_encoder.bn1', 64)
    c = _basic_layer(s, 'value_encoder.layer1', 64, 64, 2, 1)
    c = _basic_layer(s, 'value_encoder.layer2', c, 128, 2, 2)
    c = _basic_layer(s, 'value_encoder.layer3', c, 256, 2, 2)
    _fusion_block(s, 'value_encoder.fuser', 1024, 256, value_dim, value_dim)
    if hidden_dim > 0:
        _conv(s, 'value_encoder.hidden_reinforce.transform', hidden_dim * 3, value_dim + hidden_dim, 3, True)
    # KeyProjection (model/modules.py:194-211)
    _conv(s, 'key_proj.key_proj', key_dim, 1024, 3, True)
    _conv(s, 'key_proj.d_proj', 1, 1024, 3, True)
    _conv(s, 'key_proj.e_proj', key_dim, 1024, 3, True)
    # Decoder (model/modules.py:214-250)
    _fusion_block(s, 'decoder.fuser', 1024, value_dim + hidden_dim, 512, 512)
    if hidden_dim > 0:
        _conv(s, 'decoder.hidden_update.g16_conv', 256, 512, 1, True)
        _conv(s, 'decoder.hidden_update.g8_conv', 256, 256, 1, True)
        _conv(s, 'decoder.hidden_update.g4_conv', 256, 257, 1, True)
        _conv(s, 'decoder.hidden_update.transform', hidden_dim * 3, 256 + hidden_dim, 3, True)
    _conv(s, 'decoder.up_16_8.skip_conv', 512, 512, 3, True)
    _group_res_block(s, 'decoder.up_16_8.out_conv', 512, 256)
    _conv(s, 'decoder.up_8_4.skip_conv', 256, 256, 3, True)
    _group_res_block(s, 'decoder.up_8_4.out_conv', 256, 256)
    _conv(s, 'decoder.pred', 1, 256, 3, True)
    return s


def infer_dims(state_dict):
    """C_k / C_v / C_h from checkpoint tensor shapes (model/network.py:142-154)."""
    key_dim = state_dict['key_proj.key_proj.weight'].shape[0]
    value_dim = state_dict['value_encoder.fuser.block2.conv2.weight'].shape[0]
    if 'decoder.hidden_update.transform.weight' in state_dict:
        hidden_dim = state_dict['decoder.hidden_update.transform.weight'].shape[0] // 3
    else:
        hidden_dim = 0
    return key_dim, value_dim, hidden_dim
