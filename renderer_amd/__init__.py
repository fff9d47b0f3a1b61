"""MI355X-native executor for the VectorVM particle-simulation bytecode.

The package is a thin host-side mirror of the reference's runtime interface
(Runtime/VectorVM/Public/VectorVMRuntime.h:10-33) over the C ABI in ``include/vvm_abi.h``; the
compute lives in ``csrc/`` (HIP kernels for gfx950 plus the C++ launcher) and is loaded from
``csrc/libvvm_hip.so``.
"""
from . import abi, assembler, opcodes, scripts  # noqa: F401

__all__ = ["abi", "assembler", "opcodes", "scripts"]
