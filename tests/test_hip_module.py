"""renderer_amd/hip.py without a GPU: it binds to the HIP runtime that libvvm_hip.so has brought into the process (never a second
copy), and a call that needs a device fails with the runtime's own message."""
import ctypes

import pytest

from renderer_amd import abi, hip


def test_binds_to_the_runtime_already_in_the_process():
    abi.load_hip_backend()
    rt = hip.runtime()
    assert isinstance(rt, ctypes.CDLL) and "libamdhip64" in rt._name
    assert hip.runtime() is rt


def test_a_stream_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    abi.load_hip_backend()
    with pytest.raises(RuntimeError) as e:
        hip.Stream()
    assert "hipStreamCreateWithFlags" in str(e.value)
