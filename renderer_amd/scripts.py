"""Canonical emitter scripts and their synthetic particle streams.

The reference ships no scripts (its only client, the Niagara plugin, is outside the tree), so the
benchmark workloads are defined here: a fountain update (gravity + drag + age, kill by
``acquireindex``), a transcendental-heavy update, and per-emitter variants of the fountain for the
batched case.  Streams are generated from a counter hash so that any index range can be produced
independently on the host (numpy) or on the device (torch) with identical bits.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

from .assembler import FLOAT, INT32, HALF, Script, BuiltScript

SEED = 0x5EEDF00D
_M64 = (1 << 64) - 1


@dataclass
class Workload:
    name: str
    script: BuiltScript
    constants: bytes                 # one constant table
    num_float_inputs: int
    num_float_outputs: int
    ext_fns: Sequence[int] = ()      # built-in ids, one per external function of the script


# ---------------------------------------------------------------------------------------------------
# scripts
# ---------------------------------------------------------------------------------------------------

def fountain(merged: bool = True, alias_outputs: bool = True, variant: int = 0) -> Workload:
    """F1: 8 float inputs (pos.xyz, vel.xyz, age, lifetime) -> 8 float outputs, 17 flops per particle.

        v' = (g*dt + v) * drag ;  p' = v'*dt + p ;  age' = age + dt ;  keep = age' < lifetime

    ``variant`` (0..63) permutes register/constant assignment and toggles the merged-op form and an
    extra age-scale multiply so that batched emitters run distinct bytecode with one I/O footprint.
    """
    s = Script()
    perm = variant % 6
    axes = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)][perm]
    use_merged = merged if variant == 0 else bool((variant >> 3) & 1)
    age_scale = bool((variant >> 4) & 1)
    shuffle_consts = bool((variant >> 5) & 1)
    # constant table layout (dwords): dt gx gy gz drag agescale | shuffled: agescale drag gz gy gx dt
    ci = {"dt": 0, "g": (1, 2, 3), "drag": 4, "agescale": 5}
    if shuffle_consts:
        ci = {"dt": 5, "g": (4, 3, 2), "drag": 1, "agescale": 0}
    dt, drag = s.const(ci["dt"]), s.const(ci["drag"])
    pos = [s.input(0, FLOAT, k) for k in range(3)]
    vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
    age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)

    age1 = s.temp()
    if age_scale:
        tmp = s.temp()
        s.op("mul", dt, s.const(ci["agescale"]), tmp)
        s.op("add", age, tmp, age1)
    else:
        s.op("add", age, dt, age1)
    alive = s.temp()
    s.op("cmplt", age1, life, alive)
    s.acquireindex(alive, 0)
    v1 = s.temps(3)
    p1 = s.temps(3)
    for c in axes:
        g = s.const(ci["g"][c])
        if use_merged:
            s.op("mad_mul", g, dt, vel[c], drag, v1[c])
        else:
            t = s.temp()
            s.op("mad", g, dt, vel[c], t)
            s.op("mul", t, drag, v1[c])
        s.op("mad", v1[c], dt, pos[c], p1[c])
    pairs = [(p1[k], s.output(0, FLOAT, k)) for k in range(3)]
    pairs += [(v1[k], s.output(0, FLOAT, 3 + k)) for k in range(3)]
    pairs += [(age1, s.output(0, FLOAT, 6)), (life, s.output(0, FLOAT, 7))]
    s.output_data(FLOAT, 0, pairs)
    built = s.build(alias_outputs=alias_outputs)
    return Workload("fountain" if variant == 0 else "fountain_v%d" % variant, built,
                    fountain_constants(variant=variant), 8, 8)


def fountain_constants(dt=1.0 / 60.0, gravity=(0.0, 0.0, -980.0), drag=0.98, age_scale=1.0, variant: int = 0) -> bytes:
    vals = [dt, gravity[0], gravity[1], gravity[2], drag, age_scale]
    if (variant >> 5) & 1:
        vals = vals[::-1]
    return struct.pack("<6f", *[np.float32(v) for v in vals]) + bytes(8)


def transcendental(use_noise: bool = True) -> Workload:
    """T1: inputs pos.xyz, phase -> outputs pos'.xyz, phase'.

        s,c = sin_cos(phase*freq) ; e = exp(-(phase*k)) ; n = curl_noise3(pos*scale)
        pos' = pos + (n*amp + (s,c,0)) * e * dt ; phase' = phase + dt ; keep all

    constants (dwords): freq k scale amp dt all_ones
    """
    s = Script()
    freq, k, scale, amp, dt, ones = (s.const(i) for i in range(6))
    pos = [s.input(0, FLOAT, i) for i in range(3)]
    phase = s.input(0, FLOAT, 3)
    s.acquireindex(ones, 0)
    ph = s.temp(); s.op("mul", phase, freq, ph)
    sn, cs = s.temp(), s.temp(); s.op("sin_cos", ph, sn, cs)
    pk = s.temp(); s.op("mul", phase, k, pk)
    npk = s.temp(); s.op("neg", pk, npk)
    e = s.temp(); s.op("exp", npk, e)
    edt = s.temp(); s.op("mul", e, dt, edt)
    n = s.temps(3)
    if use_noise:
        sp = s.temps(3)
        for i in range(3):
            s.op("mul", pos[i], scale, sp[i])
        f = s.ext_fn(3, 3)
        s.call(f, sp, n)
    else:
        for i in range(3):
            s.op("mul", pos[i], scale, n[i])
    disp = s.temps(3)
    s.op("mad", n[0], amp, sn, disp[0])
    s.op("mad", n[1], amp, cs, disp[1])
    s.op("mul", n[2], amp, disp[2])
    p1 = s.temps(3)
    for i in range(3):
        s.op("mad", disp[i], edt, pos[i], p1[i])
    ph1 = s.temp(); s.op("add", phase, dt, ph1)
    s.output_data(FLOAT, 0, [(p1[i], s.output(0, FLOAT, i)) for i in range(3)] + [(ph1, s.output(0, FLOAT, 3))])
    built = s.build(alias_outputs=True)
    from . import abi
    return Workload("transcendental", built, transcendental_constants(), 4, 4,
                    ext_fns=(abi.VVM_EXTFN_CURL_NOISE3,) if use_noise else ())


def transcendental_constants(freq=3.0, k=0.35, scale=0.01, amp=25.0, dt=1.0 / 60.0) -> bytes:
    return struct.pack("<5fI", *[np.float32(v) for v in (freq, k, scale, amp, dt)], 0xFFFFFFFF) + bytes(8)


# ---------------------------------------------------------------------------------------------------
# synthetic streams:  u(i,k) = splitmix64(seed ^ (k << 40) ^ i) ;  U01 = (u >> 40) * 2^-24
# ---------------------------------------------------------------------------------------------------

def _splitmix64_np(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def hash_u64_np(start: int, n: int, k: int, seed: int = SEED) -> np.ndarray:
    i = np.arange(start, start + n, dtype=np.uint64)
    return _splitmix64_np(np.uint64(seed) ^ np.uint64((k << 40) & _M64) ^ i)


def u01_np(start: int, n: int, k: int, seed: int = SEED) -> np.ndarray:
    u = hash_u64_np(start, n, k, seed)
    return (u >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def fountain_streams_np(start: int, n: int, age_fraction: float = 0.5, kill_half: bool = False,
                        dt: float = 1.0 / 60.0, seed: int = SEED) -> List[np.ndarray]:
    """Eight float32 streams for instances [start, start+n).  ``kill_half`` makes exactly the instances with
    bit 0 of u(i,9) set fail ``age + dt < lifetime`` (about half, spatially random)."""
    f32 = np.float32
    out = []
    for k in range(3):
        out.append((u01_np(start, n, k, seed) - f32(0.5)) * f32(2000.0))
    for k in range(3):
        v = (u01_np(start, n, 3 + k, seed) - f32(0.5)) * f32(200.0)
        if k == 2:
            v = v + f32(150.0)
        out.append(v)
    life = f32(1.0) + f32(4.0) * u01_np(start, n, 7, seed)
    age = (u01_np(start, n, 6, seed) * life) * f32(age_fraction)
    if kill_half:
        dead = (hash_u64_np(start, n, 9, seed) & np.uint64(1)).astype(bool)
        age = np.where(dead, life, age).astype(np.float32)
    out.append(age.astype(np.float32))
    out.append(life.astype(np.float32))
    return out


def transcendental_streams_np(start: int, n: int, seed: int = SEED) -> List[np.ndarray]:
    f32 = np.float32
    out = [(u01_np(start, n, k, seed) - f32(0.5)) * f32(2000.0) for k in range(3)]
    out.append(u01_np(start, n, 3, seed) * f32(6.0))
    return out


def _splitmix64_torch(x):
    import torch

    def c(v):  # two's complement int64 constant
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(t, s):  # logical shift right on int64
        return (t >> s) & ((1 << (64 - s)) - 1)

    z = x + c(0x9E3779B97F4A7C15)
    z = (z ^ lsr(z, 30)) * c(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * c(0x94D049BB133111EB)
    return z ^ lsr(z, 31)


def hash_i64_torch(start: int, n: int, k: int, device, seed: int = SEED):
    import torch
    i = torch.arange(start, start + n, dtype=torch.int64, device=device)
    key = seed ^ ((k << 40) & _M64)
    return _splitmix64_torch(i ^ key)


def u01_torch(start: int, n: int, k: int, device, seed: int = SEED):
    import torch
    u = hash_i64_torch(start, n, k, device, seed)
    return ((u >> 40) & 0xFFFFFF).to(torch.float32) * (2.0 ** -24)


def fountain_streams_torch(start: int, n: int, device, age_fraction: float = 0.5, kill_half: bool = False,
                           seed: int = SEED, pad: int = 0):
    """Device twin of :func:`fountain_streams_np` (same bits).  Each stream gets ``pad`` extra elements."""
    import torch

    def padded(t):
        if pad == 0:
            return t.contiguous()
        buf = torch.zeros(n + pad, dtype=t.dtype, device=device)
        buf[:n] = t
        return buf

    out = []
    for k in range(3):
        out.append(padded((u01_torch(start, n, k, device, seed) - 0.5) * 2000.0))
    for k in range(3):
        v = (u01_torch(start, n, 3 + k, device, seed) - 0.5) * 200.0
        if k == 2:
            v = v + 150.0
        out.append(padded(v))
    life = 1.0 + 4.0 * u01_torch(start, n, 7, device, seed)
    age = (u01_torch(start, n, 6, device, seed) * life) * age_fraction
    if kill_half:
        dead = (hash_i64_torch(start, n, 9, device, seed) & 1).bool()
        age = torch.where(dead, life, age)
    out.append(padded(age))
    out.append(padded(life))
    return out


def transcendental_streams_torch(start: int, n: int, device, seed: int = SEED, pad: int = 0):
    import torch
    out = [(u01_torch(start, n, k, device, seed) - 0.5) * 2000.0 for k in range(3)]
    out.append(u01_torch(start, n, 3, device, seed) * 6.0)
    if pad:
        res = []
        for t in out:
            buf = torch.zeros(n + pad, dtype=t.dtype, device=device)
            buf[:n] = t
            res.append(buf)
        return res
    return [t.contiguous() for t in out]
