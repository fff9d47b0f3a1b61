"""Randomly generated scripts in the compiler's format: the optimized blob against the instruction-by-instruction translation
(tests/original_cases.py), both on the CPU oracle, bit for bit.  The generator reuses temp registers (never in place), mixes float,
compare and integer ops, constants, a kill predicate and float / int32 outputs.  Scripts in which the optimizer picks one of the two
merged opcodes whose reference definition differs from the pair it replaces are skipped: mad_sqrt (a bare square root where sqrt
returns 0 below 1e-8, VectorVMRuntime.cpp:1914, 2033) and neg_cmplt (loses the operand order, VectorVMOptimizer.cpp:2435-2449)."""
import random

import numpy as np
import pytest

import original_cases as oc
from renderer_amd.assembler import FLOAT, INT32
from renderer_amd.disasm import thaw
from renderer_amd.original import K, OriginalScript, op_inputs
from test_optimizer import ALL_ONES, Blob, optimize
from vm_harness import run_oracle

f32 = np.float32
FOPS = ["add", "sub", "mul", "mad", "min", "max", "neg", "abs", "lerp", "clamp", "floor", "frac", "select", "div", "sqrt", "rsq", "step", "sign"]
CMPS = ["cmplt", "cmple", "cmpgt", "cmpge", "cmpeq", "cmpneq"]
IOPS = ["addi", "subi", "muli", "mini", "maxi", "bit_and", "bit_or", "bit_xor", "negi", "absi"]

def gen(seed):
    r = random.Random(seed)
    o = OriginalScript()
    nf, ni = r.randint(1, 4), r.randint(0, 2)
    fvals = [o.input(0, FLOAT, i, o.temp()) for i in range(nf)]
    ivals = [o.input(0, INT32, i, o.temp()) for i in range(ni)]
    masks = []
    pool_regs = []
    def newreg(live):
        # sometimes reuse a register whose value is no longer wanted
        if pool_regs and r.random() < 0.4:
            return pool_regs.pop(r.randrange(len(pool_regs)))
        return o.temp()
    nops = r.randint(3, 25)
    for _ in range(nops):
        kind = r.random()
        if kind < 0.6 or not ivals:
            name = r.choice(FOPS)
            n = op_inputs(name)
            srcs = []
            for k in range(n):
                if name == "select" and k == 0:
                    if not masks:
                        break
                    srcs.append(r.choice(masks))
                elif r.random() < 0.25:
                    srcs.append(K(r.randint(1, 4)))
                else:
                    srcs.append(r.choice(fvals))
            if len(srcs) != n: continue
            d = newreg(None)
            if d in srcs: d = o.temp()
            o.op(name, *srcs, d)
            for lst in (fvals, ivals, masks):
                while d in lst: lst.remove(d)
            fvals.append(d)
        elif kind < 0.8:
            name = r.choice(CMPS)
            srcs = [r.choice(fvals), r.choice(fvals) if r.random() < 0.7 else K(r.randint(1, 4))]
            d = newreg(None)
            if d in srcs: d = o.temp()
            o.op(name, *srcs, d)
            for lst in (fvals, ivals, masks):
                while d in lst: lst.remove(d)
            masks.append(d)
        else:
            name = r.choice(IOPS)
            n = op_inputs(name)
            srcs = [r.choice(ivals) if r.random() < 0.8 else K(5) for _ in range(n)]
            d = newreg(None)
            if d in srcs: d = o.temp()
            o.op(name, *srcs, d)
            for lst in (fvals, ivals, masks):
                while d in lst: lst.remove(d)
            ivals.append(d)
        # retire a value sometimes (its register may be reused)
        if len(fvals) > nf + 2 and r.random() < 0.3:
            v = fvals.pop(r.randrange(nf, len(fvals)))
            pool_regs.append(v)
    pred = r.choice(masks) if masks and r.random() < 0.7 else K(0)
    idx = o.temp(); o.acquireindex(pred, idx, 0)
    nout = r.randint(1, 6)
    dts = []
    fi = ii = 0
    for k in range(nout):
        if ivals and r.random() < 0.3:
            o.output(INT32, 0, idx, r.choice(ivals), ii); ii += 1
        else:
            src = r.choice(fvals) if r.random() < 0.9 else K(2)
            o.output(FLOAT, 0, idx, src, fi); fi += 1
    return o, nf, ni, fi, ii



@pytest.mark.parametrize("block", range(8))
def test_random_scripts(block):
    checked = 0
    for seed in range(block * 50, block * 50 + 50):
        o, nf, ni, fo, io = gen(seed)
        n = 133
        rng = np.random.default_rng(seed)
        ins = [rng.uniform(-3, 3, n).astype(f32) for _ in range(nf)] + [rng.integers(-20, 20, n).astype(np.int32) for _ in range(ni)]
        consts = ALL_ONES + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()
        dts = [f32] * fo + [np.int32] * io
        blob = optimize(o)
        if {c[0] for c in thaw(blob).code} & {"mad_sqrt", "neg_cmplt"}:
            continue
        a = run_oracle(Blob(blob), [consts], ins, dts, n)
        b = run_oracle(oc.translate(o).build(), [consts], ins, dts, n)
        assert a[1] == b[1], seed
        for k in range(len(dts)):
            x, y = a[0][k][:a[1]].view(np.uint32), b[0][k][:b[1]].view(np.uint32)
            both_nan = np.isnan(x.view(f32)) & np.isnan(y.view(f32)) if k < fo else np.zeros(a[1], bool)
            assert np.array_equal(x[~both_nan], y[~both_nan]), (seed, k)
        checked += 1
    assert checked >= 40
