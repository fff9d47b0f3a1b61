"""Loads the CPU oracle (oracle/libvvm_oracle.so).  Test infrastructure only: nothing under renderer_amd/ imports this."""
import ctypes as C
import os
import subprocess

from renderer_amd import abi

_ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_LIB = os.path.join(_ORACLE_DIR, "libvvm_oracle.so")
_backend = None


def build_oracle(force: bool = False) -> str:
    src = os.path.join(_ORACLE_DIR, "vvm_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _LIB


class OracleBackend(abi.Backend):
    def __init__(self, lib):
        super().__init__(lib, "vvm_oracle_alloc_state", "vvm_oracle_free_state", "vvm_oracle_exec")
        lib.vvm_oracle_set_strict.argtypes = [C.c_int]
        lib.vvm_oracle_set_page_kb.argtypes = [C.c_int]
        lib.vvm_oracle_set_options.argtypes = [C.c_uint64, C.c_int32, C.c_int32]
        lib.vvm_oracle_set_xorwow.argtypes = [C.POINTER(C.c_uint32)]
        lib.vvm_oracle_xorwow_seed.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
        lib.vvm_oracle_perlin3.restype = C.c_float
        lib.vvm_oracle_perlin3.argtypes = [C.c_float] * 3
        lib.vvm_oracle_curl_noise3.argtypes = [C.c_float] * 3 + [C.POINTER(C.c_float)]
        lib.vvm_oracle_rand_u32.restype = C.c_uint32
        lib.vvm_oracle_rand_u32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        lib.vvm_oracle_store_half.restype = C.c_uint16
        lib.vvm_oracle_store_half.argtypes = [C.c_float]
        lib.vvm_oracle_load_half.restype = C.c_float
        lib.vvm_oracle_load_half.argtypes = [C.c_uint16]

    def set_strict(self, on: bool):
        self.lib.vvm_oracle_set_strict(1 if on else 0)

    def set_page_kb(self, kb: int):
        self.lib.vvm_oracle_set_page_kb(kb)

    def set_xorwow(self, state24=None):
        """state24: the reference generator's start state (24 words), or None for the counter-based generator."""
        self.lib.vvm_oracle_set_xorwow((C.c_uint32 * 24)(*state24) if state24 is not None else None)

    def xorwow_seed(self, cycles_first: int, cycles_second: int, batch_state_address: int):
        out = (C.c_uint32 * 24)()
        self.lib.vvm_oracle_xorwow_seed(cycles_first, cycles_second, batch_state_address, out)
        return list(out)

    def set_options(self, seed: int = 0, exec_index_base: int = 0, rand_instance_base: int = 0):
        self.lib.vvm_oracle_set_options(seed, exec_index_base, rand_instance_base)


def load_oracle() -> OracleBackend:
    global _backend
    if _backend is None:
        _backend = OracleBackend(C.CDLL(build_oracle()))
    return _backend
