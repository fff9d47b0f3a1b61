"""The few HIP runtime calls a Python caller of the executor needs beside the library's own entry points: streams, and capturing
the launches of queued Execs into a graph that is then replayed.  Binds to the copy of the HIP runtime that the process has
already loaded (libvvm_hip.so links it; PyTorch brings one of its own), never to a second one."""
import ctypes as C
import os
from typing import Optional

hipStreamNonBlocking = 1
hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed = 0, 1, 2

_rt: Optional[C.CDLL] = None


def runtime() -> C.CDLL:
    global _rt
    if _rt is None:
        path = None
        with open("/proc/self/maps") as f:
            for line in f:
                name = line.rsplit(" ", 1)[-1].strip()
                if os.path.basename(name).startswith("libamdhip64.so"):
                    path = name
                    break
        if path is None:
            raise RuntimeError("no HIP runtime is loaded in this process: load libvvm_hip.so first (renderer_amd.abi.load_hip_backend)")
        _rt = C.CDLL(path)
        _rt.hipGetErrorString.restype = C.c_char_p
    return _rt


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError("%s: HIP error %d (%s)" % (what, rc, runtime().hipGetErrorString(rc).decode()))


class Stream:
    def __init__(self, flags: int = hipStreamNonBlocking):
        self.handle = C.c_void_p()
        _check(runtime().hipStreamCreateWithFlags(C.byref(self.handle), C.c_uint(flags)), "hipStreamCreateWithFlags")

    @property
    def value(self) -> int:
        return self.handle.value

    def synchronize(self) -> None:
        _check(runtime().hipStreamSynchronize(self.handle), "hipStreamSynchronize")

    def begin_capture(self, mode: int = hipStreamCaptureModeRelaxed) -> None:
        _check(runtime().hipStreamBeginCapture(self.handle, C.c_int(mode)), "hipStreamBeginCapture")

    def end_capture(self) -> "Graph":
        g = C.c_void_p()
        _check(runtime().hipStreamEndCapture(self.handle, C.byref(g)), "hipStreamEndCapture")
        return Graph(g)

    def destroy(self) -> None:
        if self.handle:
            runtime().hipStreamDestroy(self.handle)
            self.handle = C.c_void_p()


class Graph:
    """A captured graph and its executable form."""

    def __init__(self, graph: C.c_void_p):
        self.graph = graph
        self.exe = C.c_void_p()
        _check(runtime().hipGraphInstantiate(C.byref(self.exe), self.graph, None, None, C.c_size_t(0)), "hipGraphInstantiate")

    def num_nodes(self) -> int:
        n = C.c_size_t(0)
        _check(runtime().hipGraphGetNodes(self.graph, None, C.byref(n)), "hipGraphGetNodes")
        return n.value

    def launch(self, stream: Stream) -> None:
        _check(runtime().hipGraphLaunch(self.exe, stream.handle), "hipGraphLaunch")

    def destroy(self) -> None:
        if self.exe:
            runtime().hipGraphExecDestroy(self.exe)
            self.exe = C.c_void_p()
        if self.graph:
            runtime().hipGraphDestroy(self.graph)
            self.graph = C.c_void_p()
