"""Whole random scripts on the C oracle against a numpy evaluation of the same script: a second, independent statement at program
level (operand decoding, register reuse after the assembler's allocation, output aliasing, keep mask, stable compaction, constant
and no-advance operands, half conversion, chunking at two page sizes), built on the per-op restatement of tests/np_semantics.py.
Only ops with exactly defined results take part."""
import random

import numpy as np
import pytest

import np_semantics as nps
from renderer_amd.assembler import FLOAT, HALF, INT32, Script
from renderer_amd.opcodes import OPS
from vm_harness import run_oracle

f32, u32 = np.float32, np.uint32
FLOAT_OPS = ("add sub mul div mad lerp min max neg abs floor ceil frac trunc round sign step clamp sqrt rsq mad_add mad_sub0 mad_sub1 "
             "mad_mul mad_mad0 mad_mad1 mul_mad0 mul_mad1 mul_add mul_sub0 mul_sub1 mul_mul mul_max add_mad1 add_add sub_neg sub_mul "
             "div_mad0 div_mul").split()
CMP = "cmplt cmple cmpgt cmpge cmpeq cmpneq sub_cmplt1 neg_cmplt".split()
INT_OPS = "addi subi muli divi mini maxi bit_and bit_or bit_xor negi absi muli_addi addi_muli bit_or_muli clampi".split()
CONSTS = np.array([0xFFFFFFFF], u32).tobytes() + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()


def gen(seed):
    """(script, program, inputs description): `program` is the list of steps the numpy evaluation replays."""
    r = random.Random(seed)
    s = Script()
    prog = []
    nf, ni, nh, nna = r.randint(1, 6), r.randint(0, 2), r.randint(0, 2), r.randint(0, 1)
    vals = {}                                   # assembler operand -> key of the numpy value
    fvals = [s.input(0, FLOAT, i) for i in range(nf)]
    for i, o in enumerate(fvals):
        vals[o] = ("in", i)
    ivals = [s.input(0, INT32, i) for i in range(ni)]
    for i, o in enumerate(ivals):
        vals[o] = ("in", nf + nna + i)
    if nna:
        o = s.input(0, FLOAT, nf, noadvance=True)
        vals[o] = ("first", nf)
        fvals.append(o)
    for i in range(nh):
        t = s.temp()
        s.op("half_to_float", s.input(0, HALF, i), t)
        vals[t] = ("half", nf + nna + ni + i)
        fvals.append(t)
    masks, counter = [], [0]

    def fsrc():
        if r.random() < 0.2:
            k = r.randint(1, 4)
            c = s.const(k)
            vals[c] = ("const", k)
            return c
        return r.choice(fvals)

    def emit(name, srcs, pool):
        t = s.temp()
        s.op(name, *srcs, t)
        counter[0] += 1
        vals[t] = ("t", counter[0])
        prog.append((name, [vals[x] for x in srcs], vals[t]))
        pool.append(t)

    pred = None
    pred_at = r.randint(0, 6)
    n_ops = r.randint(3, 40)
    ones = s.const(0); vals[ones] = ("const", 0)
    for k in range(n_ops):
        if k == pred_at:
            pred = r.choice(masks) if masks and r.random() < 0.7 else ones
            s.acquireindex(pred, 0)
        c = r.random()
        if c < 0.6:
            name = r.choice(FLOAT_OPS)
            emit(name, [fsrc() for _ in range(OPS[name][1])], fvals)
        elif c < 0.7 and masks:
            emit("select", [r.choice(masks), fsrc(), fsrc()], fvals)
        elif c < 0.82:
            name = r.choice(CMP)
            emit(name, [fsrc() for _ in range(OPS[name][1])], masks)
        elif c < 0.86:
            t = s.temp()
            name = r.choice(["exec_index", "exec_indexf"])
            s.op(name, t)
            counter[0] += 1
            vals[t] = ("t", counter[0])
            prog.append((name, [], vals[t]))
            (ivals if name == "exec_index" else fvals).append(t)
        elif ivals:
            name = r.choice(INT_OPS)
            k5 = s.const(5); vals[k5] = ("const", 5)
            emit(name, [r.choice(ivals) if r.random() < 0.8 else k5 for _ in range(OPS[name][1])], ivals)
        if len(fvals) > nf + 3 and r.random() < 0.4:
            fvals.pop(r.randrange(nf, len(fvals)))
    if pred is None:
        pred = r.choice(masks) if masks and r.random() < 0.7 else ones
        s.acquireindex(pred, 0)
    outs = []
    for _ in range(r.randint(1, 12)):
        c = r.random()
        if c < 0.2 and ivals:
            outs.append((INT32, r.choice(ivals)))
        elif c < 0.35:
            outs.append((HALF, r.choice(fvals)))
        else:
            outs.append((FLOAT, r.choice(fvals)))
    dts, wanted = [], []
    for typ in (FLOAT, INT32, HALF):
        group = [(v, s.output(0, typ, i)) for i, v in enumerate(v for t_, v in outs if t_ == typ)]
        for v, _ in group:
            dts.append({FLOAT: f32, INT32: np.int32, HALF: np.float16}[typ])
            wanted.append((typ, vals[v]))
        for i in range(0, len(group), 4):
            s.output_data(typ, 0, group[i:i + 4])
    return s.build(alias_outputs=r.random() < 0.5), prog, vals[pred], wanted, dts, (nf, nna, ni, nh)


def evaluate(prog, pred, wanted, ins, n):
    consts = np.frombuffer(CONSTS, u32)
    env = {}

    def get(key):
        kind, k = key
        if kind == "in":
            return ins[k].view(u32)[:n]
        if kind == "first":
            return np.full(n, ins[k].view(u32)[0], u32)
        if kind == "half":
            return ins[k][:n].astype(f32).view(u32)
        if kind == "const":
            return np.full(n, consts[k], u32)
        return env[k]

    for name, srcs, dst in prog:
        if name == "exec_index":
            env[dst[1]] = np.arange(n, dtype=np.int32).view(u32)
        elif name == "exec_indexf":
            env[dst[1]] = np.arange(n, dtype=np.int32).astype(f32).view(u32)
        else:
            env[dst[1]] = np.asarray(nps.OPS[name](*[get(k) for k in srcs]), u32)
    keep = (get(pred) >> 31).astype(bool)
    out = []
    for typ, key in wanted:
        v = get(key)[keep]
        if typ == HALF:
            with np.errstate(all="ignore"):
                v = v.view(f32).astype(np.float16)
        out.append(v)
    return out, int(keep.sum())


@pytest.mark.parametrize("block", range(6))
def test_random_programs_against_numpy(block):
    for seed in range(block * 40, block * 40 + 40):
        built, prog, pred, wanted, dts, (nf, nna, ni, nh) = gen(seed)
        n = 1 + (seed * 53) % 700
        rng = np.random.default_rng(seed)
        ins = [rng.uniform(-3, 3, n).astype(f32) for _ in range(nf + nna)] + [rng.integers(-20, 20, n).astype(np.int32) for _ in range(ni)] + \
              [rng.uniform(-2, 2, n).astype(np.float16) for _ in range(nh)]
        want, want_n = evaluate(prog, pred, wanted, ins, n)
        for page_kb in (64, 2):
            got, got_n, _ = run_oracle(built, [CONSTS], ins, dts, n, page_kb=page_kb)
            assert got_n == want_n, (seed, page_kb)
            for k, (typ, _) in enumerate(wanted):
                if typ == HALF:
                    a, b = got[k][:got_n], want[k]
                    nan = np.isnan(a) & np.isnan(b)
                    assert np.array_equal(a.view(np.uint16)[~nan], b.view(np.uint16)[~nan]), (seed, page_kb, k)
                else:
                    a, b = got[k][:got_n].view(u32), want[k].view(u32)
                    if typ == FLOAT:
                        a, b = nps.canon_nan(a), nps.canon_nan(b)
                    assert np.array_equal(a, b), (seed, page_kb, k)
