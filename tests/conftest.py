import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')

torch.set_grad_enabled(False)
if os.environ.get('XMEM_POISON_EMPTY'):
    # debugging aid: every torch.empty (activations, arenas, scratch) comes back filled with NaN / 0xFF instead of whatever the allocator
    # held - a read of memory nobody wrote shows up as NaN in the outputs instead of depending on the allocation history
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def synth_sd():
    """Conditioned synthetic checkpoint (seed 0), cached on disk because the hash generator takes ~10 s."""
    from xmem2_amd.synth import synthetic_state_dict
    cache = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'xmem2_amd_synth_sd_seed0.pt')
    if os.path.exists(cache):
        try:
            return torch.load(cache)
        except Exception:
            pass
    sd = synthetic_state_dict(0)
    try:
        torch.save(sd, cache)
    except Exception:
        pass
    return sd


@pytest.fixture(scope='session')
def synth_sd_mo():
    """The multi-object conditioning of the synthetic checkpoint (xmem2_amd.synth, conditioning='multi_object'): the object-specific path
    dominates the shared image features, so that several objects do not saturate on the same pixels (config-3 clips)."""
    from xmem2_amd.synth import synthetic_state_dict
    return synthetic_state_dict(0, conditioning='multi_object')


@pytest.fixture(scope='session')
def hip_net_mo(synth_sd_mo):
    from xmem2_amd.network import XMem
    net = XMem({'key_dim': 64, 'value_dim': 512, 'hidden_dim': 64, 'precision': 'fp32'}, None).to('cuda').eval()
    net.load_weights(synth_sd_mo)
    return net


@pytest.fixture(scope='session')
def ref_net_mo(synth_sd_mo):
    from oracle import cpu_ref
    return cpu_ref.RefNet(synth_sd_mo)


@pytest.fixture(scope='session')
def device():
    return torch.device('cuda', 0)


@pytest.fixture(scope='session')
def hip_net(request, synth_sd):
    """The HIP network in the default fp32 mode; `@pytest.mark.parametrize('hip_net', ['fp32', 'fp32x'], indirect=True)` runs a test
    in the opt-in split-operand mode as well (same gates: it is fp32-class)."""
    from xmem2_amd.network import XMem
    cfg = {'key_dim': 64, 'value_dim': 512, 'hidden_dim': 64, 'precision': getattr(request, 'param', 'fp32')}
    net = XMem(cfg, None).to('cuda').eval()
    net.load_weights(synth_sd)
    return net


BOTH_FP32_CLASS_MODES = pytest.mark.parametrize('hip_net', ['fp32', 'fp32x'], indirect=True)


@pytest.fixture(scope='session')
def ref_net(synth_sd):
    from oracle import cpu_ref
    return cpu_ref.RefNet(synth_sd)


def base_config(**over):
    cfg = dict(mem_every=10, deep_update_every=-1, enable_long_term=True, enable_long_term_count_usage=True,
               hidden_dim=64, key_dim=64, value_dim=512, top_k=30, max_mid_term_frames=10, min_mid_term_frames=5,
               num_prototypes=128, max_long_term_elements=10000)
    cfg.update(over)
    return cfg
