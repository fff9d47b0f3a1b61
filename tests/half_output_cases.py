"""outputdata_float_from_half (op 177): a half input stream copied to a float output.  The reference's handler for it is an
empty stub (VVM_Output32_from_16, VectorVMRuntime.cpp:573-577: nothing is written); the oracle reproduces that in strict mode
only, and otherwise, like the GPU executor, does what the instruction's name says: converts every survivor's half to float
(FPlatformMath::LoadHalf, GenericPlatformMath.h:34-61, exact for every half including subnormals) and compacts like
outputdata_float.  A documented deviation, pinned here against numpy's conversion.  Shared by the oracle test and the GPU test."""
import numpy as np

from renderer_amd.assembler import FLOAT, HALF, Script

f32 = np.float32
OUT_DTYPES = [f32, f32, np.float16]


def script():
    """keep = sign bit of a float stream; outputs: the half stream as float (177), a computed float, the half stream as it is (178)."""
    s = Script()
    h = s.input(0, HALF, 0)
    keep = s.input(0, FLOAT, 0)
    s.acquireindex(keep, 0)
    f = s.temp(); s.op("half_to_float", h, f)
    g = s.temp(); s.op("add", f, keep, g)
    s.output_data(FLOAT, 0, [(h, s.output(0, FLOAT, 0))], from_half=True)
    s.output_data(FLOAT, 0, [(g, s.output(0, FLOAT, 1))])
    s.output_data(HALF, 0, [(h, s.output(0, HALF, 0))], from_half=True)
    return s.build()


def inputs(n, death, seed):
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 0x10000, n, dtype=np.uint32).astype(np.uint16)      # every half there is: normals, subnormals, Inf, NaN
    special = np.array([0x0000, 0x8000, 0x0001, 0x8001, 0x03FF, 0x0400, 0x7BFF, 0xFBFF, 0x7C00, 0xFC00, 0x7E00, 0x7C01, 0xFE00, 0x3C00], np.uint16)
    bits[:min(n, len(special))] = special[:min(n, len(special))]
    keep = np.where(rng.random(n) < death, f32(1.5), f32(-1.5)).astype(f32)      # sign bit set = the instance survives
    if death == 0.0:
        keep[:] = f32(-1.5)
    return keep, bits.view(np.float16)


def check(run, n, death, seed, **kw):
    """run = run_oracle or run_hip (extra keywords, e.g. flags, pass through).  Returns the executor's info object."""
    keep, h = inputs(n, death, seed)
    outs, surv, info = run(script(), [bytes(16)], [keep, h], OUT_DTYPES, n, **kw)
    alive = keep < 0
    assert surv == int(alive.sum())
    want = h[alive].astype(f32)                # numpy's half -> float conversion is exact
    got = outs[0][:surv]
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    # NaN halves keep sign and payload (shifted into the float's mantissa), as the reference's LoadHalf does
    hb = h[alive].view(np.uint16)[nan].astype(np.uint32)
    assert np.array_equal(got.view(np.uint32)[nan], ((hb & 0x8000) << 16) | 0x7F800000 | ((hb & 0x3FF) << 13))
    # the computed float beside it and the half copy use the same positions
    with np.errstate(invalid="ignore"):
        g = (h.astype(f32) + keep)[alive]
    gn = np.isnan(g)
    assert np.array_equal(outs[1][:surv].view(np.uint32)[~gn], g.view(np.uint32)[~gn])
    assert np.array_equal(outs[2][:surv].view(np.uint16), h[alive].view(np.uint16))
    # nothing is written past the survivors (the harness fills outputs with 0xCD)
    assert np.all(outs[0][surv:n].view(np.uint32) == 0xCDCDCDCD)
    assert np.all(outs[2][surv:n].view(np.uint16) == 0xCDCD)
    return info
