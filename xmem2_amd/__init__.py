"""xmem2_amd - MI355X-native per-frame space-time memory path of XMem++ (mbzuai-metaverse/XMem2).

Drop-in surface: ``InferenceCore`` / ``XMem`` / ``MemoryManager`` / ``run_on_video`` /
``VIDEO_INFERENCE_CONFIG``; all arithmetic runs in hand-written gfx950 HIP kernels reached through the
C ABI of include/xmem_hip.h (xmem2_amd/csrc/libxmem_hip.so).  Importing the package is side-effect free;
the first kernel call loads the library and fails loudly if it is missing (no CPU fallback).
"""
from .configuration import VIDEO_INFERENCE_CONFIG

__all__ = ['VIDEO_INFERENCE_CONFIG', 'XMem', 'InferenceCore', 'MemoryManager', 'KeyValueMemoryStore']
# the harness keeps the reference's module path: `from xmem2_amd.run_on_video import run_on_video`


def __getattr__(name):
    if name == 'XMem':
        from .network import XMem
        return XMem
    if name == 'InferenceCore':
        from .inference_core import InferenceCore
        return InferenceCore
    if name == 'MemoryManager':
        from .memory_manager import MemoryManager
        return MemoryManager
    if name == 'KeyValueMemoryStore':
        from .kv_memory_store import KeyValueMemoryStore
        return KeyValueMemoryStore
    raise AttributeError(name)
