"""A script that compacts and writes two datasets (the particle dataset and an event-like secondary one), shared by the
oracle test (CPU) and the parity test (GPU).  Each dataset has its own acquireindex, keep mask and survivor count
(VectorVMRuntime.cpp:824-878, outputs :555-569)."""
import struct

import numpy as np

from renderer_amd.assembler import FLOAT, INT32, Script


def two_dataset_script(secondary_first: bool = False, many_outputs: bool = False):
    s = Script()
    x, y = s.input(0, FLOAT, 0), s.input(0, FLOAT, 1)

    def primary():
        keep = s.temp(); s.op("cmplt", x, s.const(0), keep)
        s.acquireindex(keep, 0)
        a = s.temp(); s.op("mul", x, y, a)
        pairs = [(a, s.output(0, FLOAT, 0)), (y, s.output(0, FLOAT, 1))]
        if many_outputs:
            for i in range(9):
                t = s.temp(); s.op("mad", x, s.const(2), y, t)
                pairs.append((t, s.output(0, FLOAT, 2 + i)))
        s.output_data(FLOAT, 0, pairs)

    def secondary():
        keep = s.temp(); s.op("cmpgt", y, s.const(1), keep)
        s.acquireindex(keep, 1)
        idx = s.temp(); s.op("exec_index", idx)
        b = s.temp(); s.op("add", x, y, b)
        s.output_data(FLOAT, 1, [(b, s.output(1, FLOAT, 0))])
        s.output_data(INT32, 1, [(idx, s.output(1, INT32, 0))])

    for part in ((secondary, primary) if secondary_first else (primary, secondary)):
        part()
    return s.build()


def constants(limit0: float, limit1: float) -> bytes:
    return struct.pack("<fff", limit0, limit1, 3.0) + bytes(4)


def expected(x: np.ndarray, y: np.ndarray, limit0, limit1, many_outputs: bool = False):
    f32 = np.float32
    k0, k1 = x < f32(limit0), y > f32(limit1)
    d0 = [(x * y)[k0], y[k0]]
    if many_outputs:
        d0 += [(x * f32(3.0) + y)[k0]] * 9
    d1 = [(x + y)[k1], np.nonzero(k1)[0].astype(np.int32)]
    return d0, d1


def many_dataset_script(k, order=None, trig_in=None):
    """x, y, z in; dataset d keeps the instances with (x, y or z) below limit d and writes two floats and, on odd datasets, the
    instance's index.  trig_in: that dataset's value goes through a sine (its pass then needs the kernel with the transcendental
    functions, which has one chain)."""
    s = Script()
    xs = [s.input(0, FLOAT, i) for i in range(3)]
    for d in (order or range(k)):
        keep = s.temp()
        s.op("cmplt", xs[d % 3], s.const(d), keep)
        s.acquireindex(keep, d)
        a, b = s.temps(2)
        s.op("mad", xs[d % 3], s.const(k), xs[(d + 1) % 3], a)
        if trig_in == d:
            s.op("sin", a, a)
        s.op("sub", xs[(d + 2) % 3], a, b)
        s.output_data(FLOAT, d, [(a, s.output(d, FLOAT, 0)), (b, s.output(d, FLOAT, 1))])
        if d % 2:
            idx = s.temp()
            s.op("exec_index", idx)
            s.output_data(INT32, d, [(idx, s.output(d, INT32, 0))])
    return s.build()


def many_dataset_consts(k):
    v = np.array([0.25 + 0.6 * ((d * 7) % k) / k for d in range(k)] + [1.5], np.float32)
    pad = (-len(v)) % 4
    return v.tobytes() + bytes(4 * pad)
