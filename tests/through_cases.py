"""A particle update on a wide record: the fountain update (VectorVMRuntime.cpp semantics as in scripts.fountain) on particles that
carry more attributes, float and int32, which the script hands through unchanged.  Shared by the oracle-side and the GPU tests."""
import struct

import numpy as np

from renderer_amd.assembler import FLOAT, HALF, INT32, Script


def wide_record_script(extra_float: int, extra_int: int, touch_one: bool = False, more_math: int = 0, trig: bool = False, extra_half: int = 0):
    """8 float streams of the fountain + extra_float + extra_int handed-through streams.  touch_one: one of the extra float
    streams is also read by the arithmetic (so the program loads it and the copy kernel reads it again).  more_math: that many of
    the extra float streams are not handed through but scaled and offset (what the script computes then needs more registers than
    the light kernel has).  trig: the new x velocity goes through a sine and the age through an exponential (the kernels with the
    transcendental functions).  extra_half: that many 16-bit streams, handed through as they are."""
    s = Script()
    dt, drag = s.const(0), s.const(4)
    pos = [s.input(0, FLOAT, k) for k in range(3)]
    vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
    age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
    more_f = [s.input(0, FLOAT, 8 + k) for k in range(extra_float)]
    more_i = [s.input(0, INT32, k) for k in range(extra_int)]
    more_h = [s.input(0, HALF, k) for k in range(extra_half)]
    age1 = s.temp(); s.op("add", age, dt, age1)
    alive = s.temp(); s.op("cmplt", age1, life, alive)
    s.acquireindex(alive, 0)
    v1, p1 = s.temps(3), s.temps(3)
    for c in range(3):
        s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
        # (touch_one: the first extra stream stands in for pos.x, which is then not read at all, so that the computed part still
        # reads eight streams)
        s.op("mad", v1[c], dt, more_f[0] if (touch_one and more_f and c == 0) else pos[c], p1[c])
    if trig:
        t = s.temp(); s.op("sin", v1[0], t); v1[0] = t
        t = s.temp(); s.op("exp", age1, t)
        u = s.temp(); s.op("mul", p1[1], t, u); p1[1] = u
    changed = []
    for k in range(min(more_math, extra_float)):
        t = s.temp(); s.op("mad", more_f[k], drag, age1, t)
        changed.append(t)
    s.output_data(FLOAT, 0, [(p1[c], s.output(0, FLOAT, c)) for c in range(3)] + [(v1[c], s.output(0, FLOAT, 3 + c)) for c in range(3)] +
                  [(age1, s.output(0, FLOAT, 6)), (life, s.output(0, FLOAT, 7))] +
                  [(changed[k] if k < len(changed) else more_f[k], s.output(0, FLOAT, 8 + k)) for k in range(extra_float)])
    if extra_int:
        s.output_data(INT32, 0, [(more_i[k], s.output(0, INT32, k)) for k in range(extra_int)])
    if extra_half:
        s.output_data(HALF, 0, [(more_h[k], s.output(0, HALF, k)) for k in range(extra_half)], from_half=True)
    return s.build(alias_outputs=True)


CONSTANTS = struct.pack("<6f", 1.0 / 60.0, 0.0, 0.0, -980.0, 0.98, 1.0) + bytes(8)


def streams(n: int, extra_float: int, extra_int: int, death: float, seed: int = 3, extra_half: int = 0):
    rng = np.random.default_rng(seed)
    f32 = np.float32
    fl = [rng.uniform(-100, 100, n).astype(f32) for _ in range(6)]
    life = rng.uniform(1, 5, n).astype(f32)
    age = np.where(rng.random(n) < death, life, life * f32(0.25)).astype(f32)
    fl += [age, life]
    fl += [rng.uniform(-1, 1, n).astype(f32) for _ in range(extra_float)]
    il = [rng.integers(-2 ** 31, 2 ** 31 - 1, n, dtype=np.int64).astype(np.int32) for _ in range(extra_int)]
    hl = [rng.integers(0, 2 ** 16, n, dtype=np.int64).astype(np.uint16).view(np.float16) for _ in range(extra_half)]      # (any bits: they are only moved)
    return fl + il + hl


def out_dtypes(extra_float: int, extra_int: int, extra_half: int = 0):
    return [np.float32] * (8 + extra_float) + [np.int32] * extra_int + [np.float16] * extra_half
