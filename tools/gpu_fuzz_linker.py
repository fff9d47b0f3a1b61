#!/usr/bin/env python3
"""usage: gpu_fuzz_linker.py [--large] <first seed> <last seed>   (on a GPU box)  random scripts written straight in the optimized format
(renderer_amd.assembler), wider than the optimizer fuzz: up to 60 instructions over float / int32 / half / no-advance inputs, merged
opcodes, two-output ops, the bit-exact transcendentals, exec_index, random, up to 24 outputs of three kinds (several staging groups),
many values alive at once (the 32-register kernels, LDS register slots, the one-instance-per-lane kernel), short scripts that fit the
light kernel, wide records whose extra streams are handed through (the lean form and its copy kernel), the reference's random generator, Execs inside a vvm_exec_batch call.  Each script runs on libvvm_hip.so and on the CPU
oracle; results must agree bit for bit (NaN payloads aside)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np                                                      # noqa: E402
from renderer_amd import abi                                            # noqa: E402
from renderer_amd.assembler import FLOAT, HALF, INT32, Script          # noqa: E402
from renderer_amd.opcodes import OPS                                    # noqa: E402
from vm_harness import IdState, run_hip, run_oracle                     # noqa: E402

f32 = np.float32
FLOAT_OPS = ("add sub mul div mad lerp min max neg abs floor ceil frac trunc round sign step clamp sqrt rsq sin cos exp log pow atan2 "
             "asin acos fmod mad_add mad_sub0 mad_sub1 mad_mul mad_sqrt mad_mad0 mad_mad1 mul_mad0 mul_mad1 mul_add mul_sub0 mul_sub1 "
             "mul_mul mul_max add_mad1 add_add sub_neg sub_mul div_mad0 div_mul fmod_add").split()
MASK_FIRST = "select select_mul select_add".split()                     # (mask, float...) -> float
CMP = "cmplt cmple cmpgt cmpge cmpeq cmpneq sub_cmplt1 neg_cmplt".split()            # floats -> mask
CMP_SELECT = "cmplt_select cmple_select cmpeq_select".split()           # (a, b, x, y) -> float
CMP_LOGIC = ["cmp%s_logic_%s" % (c, l) for c in "lt le gt ge eq ne".split() for l in ("and", "or")]      # (a, b, mask) -> mask
INT_OPS = "addi subi muli divi mini maxi bit_and bit_or bit_xor negi absi muli_addi addi_muli bit_or_muli".split()
CMPI = "cmplti cmplei cmpgti cmpgei cmpeqi cmpneqi".split()


def gen(seed):
    r = random.Random(seed)
    s = Script()
    nf, ni, nh, nna = r.randint(1, 10), r.randint(0, 3), r.randint(0, 2), r.randint(0, 1)
    if seed % 2 == 0 and seed % 5 != 0 and seed % 3 != 0:
        nf, ni, nh = r.randint(1, 5), r.randint(0, 1), 0
    if seed % 5 == 0:
        nf = r.randint(17, 48)                  # every stream is in a register from the top of the tile: the big-register kernels
    fvals = [s.input(0, FLOAT, i) for i in range(nf)]
    ivals = [s.input(0, INT32, i) for i in range(ni)]
    # one script in sixteen takes the IDs of its instances from the dataset's free list AND returns those of the dead ones (acquire_id
    # here, update_id below): such an Exec runs chunk after chunk of the reference's chunking
    gen.both_ids = seed % 16 == 3
    if gen.both_ids:
        new_id, new_tag = s.temps(2)
        s.acquire_id(new_id, new_tag, 0)
        ivals += [new_id, new_tag]
    if nna:
        fvals.append(s.input(0, FLOAT, nf, noadvance=True))
    hins = [s.input(0, HALF, i) for i in range(nh)]
    for h in hins:
        t = s.temp()
        s.op("half_to_float", h, t)
        fvals.append(t)
    masks = []
    ext = []                                        # built-in ids in table order
    if seed % 3 == 0:
        perlin = s.ext_fn(3, 1); ext.append(abi.VVM_EXTFN_PERLIN3)
        if seed % 6 == 0:
            curl = s.ext_fn(3, 3); ext.append(abi.VVM_EXTFN_CURL_NOISE3)
    gen.ext = ext
    pred_at = r.randint(0, 8)
    acquired = False
    keep_alive = r.random() < 0.4               # retire few values: many registers
    def fsrc():
        return s.const(r.randint(1, 4)) if r.random() < 0.2 else r.choice(fvals)
    n_ops = r.randint(3, 60)
    small = seed % 2 == 0 and seed % 5 != 0 and seed % 3 != 0      # short scripts over few streams: most of these fit the light eight-wide kernel
    if small:
        n_ops = r.randint(2, 9)
    gen.small = small
    for k in range(n_ops):
        if k == pred_at:
            s.acquireindex(r.choice(masks) if masks and r.random() < 0.7 else s.const(0), 0)
            acquired = True
        kind = r.random()
        if kind < 0.55:
            name = r.choice(FLOAT_OPS)
            _, a, b = OPS[name]
            t = s.temp()
            s.op(name, *[fsrc() for _ in range(a)], t)
            fvals.append(t)
        elif kind < 0.62 and masks:
            name = r.choice(MASK_FIRST)
            _, a, b = OPS[name]
            t = s.temp()
            s.op(name, r.choice(masks), *[fsrc() for _ in range(a - 1)], t)
            fvals.append(t)
        elif kind < 0.72:
            name = r.choice(CMP)
            _, a, b = OPS[name]
            t = s.temp()
            s.op(name, *[fsrc() for _ in range(a)], t)
            masks.append(t)
        elif kind < 0.76:
            t = s.temp()
            s.op(r.choice(CMP_SELECT), fsrc(), fsrc(), fsrc(), fsrc(), t)
            fvals.append(t)
        elif kind < 0.80 and masks:
            t = s.temp()
            s.op(r.choice(CMP_LOGIC), fsrc(), fsrc(), r.choice(masks), t)
            masks.append(t)
        elif kind < 0.84:
            a_, b_ = s.temp(), s.temp()
            if r.random() < 0.5:
                s.op("sin_cos", fsrc(), a_, b_)
            else:
                s.op("mul_2x", fsrc(), fsrc(), a_, b_)
            fvals += [a_, b_]
        elif kind < 0.855 and ext:
            from renderer_amd.assembler import INVALID
            if len(ext) == 2 and r.random() < 0.5:
                outs3 = [s.temp() if r.random() < 0.7 else INVALID for _ in range(3)]
                if all(o is INVALID for o in outs3):
                    outs3[1] = s.temp()
                s.call(curl, [r.choice(fvals) for _ in range(3)], outs3)
                fvals += [o for o in outs3 if o is not INVALID]
            else:
                t = s.temp()
                s.call(perlin, [r.choice(fvals) for _ in range(3)], [t])
                fvals.append(t)
        elif kind < 0.87:
            t = s.temp()
            if r.random() < 0.5:
                s.op("exec_indexf", t); fvals.append(t)
            else:
                s.op("exec_index", t); ivals.append(t)
        elif kind < 0.90:
            t = s.temp()
            s.op("random", fsrc(), t)
            fvals.append(t)
        elif kind < 0.93:
            t = s.temp()
            if ivals and r.random() < 0.5:
                s.op("i2f", r.choice(ivals), t); fvals.append(t)
            else:
                s.op("f2i", r.choice(fvals), t); ivals.append(t)
        elif ivals:
            if r.random() < 0.25:
                t = s.temp()
                s.op(r.choice(CMPI), r.choice(ivals), r.choice(ivals), t)
                masks.append(t)
            else:
                name = r.choice(INT_OPS)
                _, a, b = OPS[name]
                t = s.temp()
                s.op(name, *[r.choice(ivals) if r.random() < 0.8 else s.const(5) for _ in range(a)], t)
                ivals.append(t)
        if not keep_alive and len(fvals) > nf + 3 and r.random() < 0.5:
            fvals.pop(r.randrange(nf, len(fvals)))
    if seed % 5 == 0:
        acc = fvals[0]                              # a sum over every stream at the very end keeps all of them alive
        for i in range(1, nf):
            t = s.temp()
            s.op("add", acc, s.input(0, FLOAT, i), t)
            acc = t
        fvals = [acc] + fvals[-6:]
    if not acquired:
        s.acquireindex(r.choice(masks) if masks and r.random() < 0.7 else s.const(0), 0)
    gen.with_ids = seed % 8 in (3, 5)
    if gen.both_ids:
        s.update_id(new_id, new_tag, 0)
    elif gen.with_ids:                              # update_id over an ID stream of its own (the last int32 input)
        s.update_id(s.input(0, INT32, ni), r.choice(ivals) if ivals else s.input(0, INT32, ni), 0)
        ni += 1
    n_out = r.randint(1, 24) if keep_alive else r.randint(1, 9)
    dts, fo, io, ho = [], 0, 0, 0
    outs = []
    # a wide record: one short script in eight also hands nine to eighteen more float streams through unchanged; where what it
    # computes fits the light kernel the library keeps a lean form of it, and these seeds run with eight instances per lane forced
    wide = []
    if small and seed % 16 == 2:
        wide = [s.input(0, FLOAT, nf + nna + i) for i in range(r.randint(9, 18))]
        n_out = r.randint(1, 7)
    for _ in range(n_out):
        c = r.random()
        if c < 0.2 and ivals:
            outs.append((INT32, r.choice(ivals)))
        elif c < 0.35:
            outs.append((HALF, r.choice(fvals)))
        else:
            outs.append((FLOAT, r.choice(fvals) if r.random() < 0.95 else s.const(2)))
    for v in wide:
        outs.insert(r.randint(0, len(outs)), (FLOAT, v))
    # half inputs copied straight to outputs: as float (outputdata_float_from_half) and as they are (outputdata_half_from_half)
    straight = []
    for h in hins:
        if r.random() < 0.6:
            straight.append((FLOAT if r.random() < 0.6 else HALF, h))
    # streams are ordered float, int32, half
    for typ in (FLOAT, INT32, HALF):
        idx = 0
        group = []
        for t_, v in outs:
            if t_ != typ:
                continue
            group.append((v, s.output(0, typ, idx)))
            idx += 1
            dts.append({FLOAT: f32, INT32: np.int32, HALF: np.float16}[typ])
        for i in range(0, len(group), 5):
            s.output_data(typ, 0, group[i:i + 5])
        group = []
        for t_, v in straight:
            if t_ != typ:
                continue
            group.append((v, s.output(0, typ, idx)))
            idx += 1
            dts.append({FLOAT: f32, HALF: np.float16}[typ])
        if group:
            s.output_data(typ, 0, group, from_half=True)
    more_dts = []
    if seed % 4 == 1 or (gen.both_ids and seed % 32 == 3):      # a second acquired dataset with outputs of its own; of the scripts without ID instructions a third have a third, another third a fourth as well
        for d in range(1, 2 + ((seed // 8) % 3 if seed % 8 == 1 else 1 if seed % 32 == 5 else 0)):      # (seed % 8 == 5: with ID instructions)
            s.acquireindex(r.choice(masks) if masks and r.random() < 0.8 else s.const(0), d)
            pairs_f = [(r.choice(fvals), s.output(d, FLOAT, i)) for i in range(r.randint(1, 3))]
            s.output_data(FLOAT, d, pairs_f)
            dts_d = [f32] * len(pairs_f)
            if ivals:
                s.output_data(INT32, d, [(r.choice(ivals), s.output(d, INT32, 0))])
                dts_d.append(np.int32)
            more_dts.append(dts_d)
    gen.more_datasets = more_dts
    return s.build(alias_outputs=r.random() < 0.5), nf + nna + len(wide), ni, nh, dts


class Neighbours:
    """Four fountain emitters (one large enough to run on the light kernel inside a batch) that share a vvm_exec_batch call with every
    thirteenth script; their own results are checked after each such call."""

    def __init__(self):
        from renderer_amd import scripts
        from renderer_amd.runtime import VectorVMState
        from test_gpu_more import _DeviceStreams
        self.be = abi.load_hip_backend()
        self.frames, self.states, self.streams, self.refs = [], [], [], []
        for e, n in enumerate((3000, 9001, 4022, 1)):
            w = scripts.fountain(variant=e)
            ins = scripts.fountain_streams_np(e * 7777, n, kill_half=True)
            self.refs.append(run_oracle(w.script, [w.constants], ins, [f32] * 8, n))
            ds = _DeviceStreams(self.be, ins, 8, n)
            st = VectorVMState(w.script.blob, self.be)
            self.streams.append(ds); self.states.append(st)
            self.frames.append(st.frame([abi.DataSet(ds.ins, ds.outs, 0)], n, [w.constants]))

    def close(self):
        for st in self.states:
            st.close()
        for ds in self.streams:
            ds.free()

    def check(self):
        for e, (ref, ref_n, _) in enumerate(self.refs):
            if self.frames[e].survivors(0) != ref_n:
                return False
            for k in range(8):
                if not np.array_equal(self.streams[e].download(k)[:ref_n].view(np.uint32), ref[k][:ref_n].view(np.uint32)):
                    return False
        return True


# --large: every seventeenth script runs over 200 K to 600 K instances.  (This is what showed that the look-back words of a script's
# later pass were sized for its first pass: a pass on a kernel with smaller tiles wrote past the end of the array.  Scripts this
# odd at this size reach code that little else does; run it on a box you can afford to lose.)
LARGE = "--large" in sys.argv
if LARGE:
    sys.argv.remove("--large")


def run(first_seed, last_seed, report_every=100):
    """Seeds [first_seed, last_seed): returns (scripts that differed from the oracle, scripts skipped as too large, counts per kind)."""
    bad = skipped = 0
    shapes = {}
    neighbours = Neighbours()
    for seed in range(first_seed, last_seed):
        try:
            built, nf, ni, nh, dts = gen(seed)
        except AssertionError:
            skipped += 1
            continue
        n = (seed // 7) % 10 if seed % 7 == 0 else 2048 + 1 + (seed * 37) % 3000          # (some tiny Execs, empty ones included)
        if LARGE and seed % 17 == 0 and seed % 7 != 0:
            n = 200_000 + (seed * 7919) % 400_000         # hundreds to thousands of tiles: the eight-wide kernels by the library's own choice, several tiles per workgroup
        if LARGE and seed % 97 == 0 and seed % 7 != 0:
            n = 1_200_000 + (seed * 7919) % 2_000_000     # more tiles than resident workgroups on every kernel
        offset = 1 + seed % 9 if seed % 3 == 0 else 0                                        # InstanceOffset
        rng = np.random.default_rng(seed)
        ins = [rng.uniform(-3, 3, max(n, 1)).astype(f32) for _ in range(nf)] + [rng.integers(-20, 20, max(n, 1)).astype(np.int32) for _ in range(ni)] + \
              [rng.uniform(-2, 2, max(n, 1)).astype(np.float16) for _ in range(nh)]
        consts = bytes([255] * 4) + np.array([0.5, -1.25, 2.0, 0.0], f32).tobytes() + np.array([3], np.int32).tobytes()
        try:
            more = gen.more_datasets
            want_ids = got_ids = None
            if gen.both_ids:
                m = max(n, 1)
                want_ids = IdState(np.full(m + 7, -1, np.int32), np.full(3 * m + 40, -7, np.int32), num_free=m + 5, max_used=3, num_spawned=1, tag=11)
                want_ids.free_table[:m + 5] = rng.permutation(m + 7)[:m + 5]
                got_ids = want_ids.copy()
            elif gen.with_ids:
                m = max(n, 1)
                ins[nf + ni - 1] = rng.permutation(m + 7)[:m].astype(np.int32)          # distinct IDs
                want_ids = IdState(np.full(m + 7, -1, np.int32), np.full(2 * m + 16, -7, np.int32), num_free=5, max_used=3)
                want_ids.free_table[:5] = [100001, 100002, 100003, 100004, 100005]
                got_ids = want_ids.copy()
            ref_more, got_more = [], []
            # every eleventh script draws from the reference's own generator (vvm_exec_options.RandXorwowState) instead of the counter-based
            # one (scripts whose datasets go two to a pass then run one program per dataset: the kernels that read the words have one chain)
            xorwow = [(seed * 2654435761 + 97 * k) & 0xFFFFFFFF for k in range(24)] if seed % 11 == 0 and not gen.both_ids else None
            in_batch = seed % 13 == 0 and xorwow is None
            # every nineteenth script goes through vvm_exec_noalloc where that entry point takes it (no ID instructions, a parameter
            # block of at most 1 KB, no injected generator state); a refusal for one of those reasons sends it through vvm_exec_ex
            no_alloc = seed % 19 == 0 and not in_batch and xorwow is None and not gen.with_ids
            ref = run_oracle(built, [consts], ins, dts, n, seed=seed, instance_offset=offset, more_datasets=more, more_results=ref_more, ext_fns=list(gen.ext), ids=want_ids, xorwow=xorwow)
            hip_flags = (0, abi.VVM_EXEC_FORCE_EIGHT_WIDE, abi.VVM_EXEC_SEPARATE_PASSES, abi.VVM_EXEC_FORCE_EIGHT_WIDE | abi.VVM_EXEC_NO_LIGHT_KERNEL)[(seed // 2) % 4 if gen.small else seed % 4]
            if seed % 23 == 0:
                hip_flags |= abi.VVM_EXEC_PARAMS_IN_MEMORY
            if no_alloc:
                try:
                    got = run_hip(built, [consts], ins, dts, n, seed=seed, instance_offset=offset, flags=hip_flags, more_datasets=more, more_results=got_more,
                                  ext_fns=list(gen.ext), no_alloc=True)
                except abi.VVMError as e:
                    if e.code not in (abi.VVM_E_TOO_LARGE, abi.VVM_E_UNSUPPORTED_OP):
                        raise
                    no_alloc = False
                    got_more.clear()
            if not no_alloc:
                got = run_hip(built, [consts], ins, dts, n, seed=seed, instance_offset=offset, flags=hip_flags, more_datasets=more, more_results=got_more,
                              ext_fns=list(gen.ext), ids=got_ids, xorwow=xorwow, batch_with=neighbours.frames if in_batch else None,
                              batch_handle=in_batch and seed % 26 == 0 and not gen.with_ids)      # (an ID script's tables take one Exec: the handle runs two)
        except abi.VVMError as e:
            if e.code == abi.VVM_E_TOO_LARGE:
                skipped += 1
                continue
            raise
        info = got[2]
        key = (info.InstancesPerLane, info.NumVirtualRegisters > 16, info.LdsSpillRegisters > 0)
        shapes[key] = shapes.get(key, 0) + 1
        if got[2].Flags & 0x20000 and ((seed // 2) % 4 if gen.small else seed % 4) == 1:
            shapes["ran on the light kernel"] = shapes.get("ran on the light kernel", 0) + 1
        for bit, name in ((0x20000, "fits the light kernel"), (0x40000, "two chains in one pass"), (0x80000, "has a lean form (runs when eight-wide is forced)")):
            if info.Flags & bit:
                shapes[name] = shapes.get(name, 0) + 1
        if gen.both_ids:
            shapes["acquire_id and update_id on one dataset"] = shapes.get("acquire_id and update_id on one dataset", 0) + 1
        if xorwow is not None:
            shapes["reference generator"] = shapes.get("reference generator", 0) + 1
        if in_batch:
            shapes["inside a batch"] = shapes.get("inside a batch", 0) + 1
            if seed % 26 == 0 and not gen.with_ids:
                shapes["... kept in a vvm_batch handle"] = shapes.get("... kept in a vvm_batch handle", 0) + 1
        if no_alloc:
            shapes["through vvm_exec_noalloc"] = shapes.get("through vvm_exec_noalloc", 0) + 1
        if n >= 200_000:
            shapes["200 K instances or more"] = shapes.get("200 K instances or more", 0) + 1
        ok = got[1] == ref[1]
        for k in range(len(dts)):
            if not ok:
                break
            if dts[k] == np.float16:
                x, y = got[0][k][offset:offset + got[1]].view(np.uint16), ref[0][k][offset:offset + ref[1]].view(np.uint16)
                nan = np.isnan(got[0][k][offset:offset + got[1]]) & np.isnan(ref[0][k][offset:offset + ref[1]])
            else:
                x, y = got[0][k][offset:offset + got[1]].view(np.uint32), ref[0][k][offset:offset + ref[1]].view(np.uint32)
                nan = np.isnan(x.view(f32)) & np.isnan(y.view(f32)) if dts[k] == f32 else np.zeros(got[1], bool)
            if not np.array_equal(x[~nan], y[~nan]):
                ok = False
                where = np.nonzero((x != y) & ~nan)[0]
                print("seed", seed, "stream", k, np.dtype(dts[k]).name, "differs at", where.size, "places, first", where[:6], "last", where[-1],
                      "got", [hex(int(v)) for v in x[where[:3]]], "want", [hex(int(v)) for v in y[where[:3]]], "n", n, "xorwow", xorwow is not None, "batch", in_batch, flush=True)
        if ok and want_ids is not None:
            ok = np.array_equal(got_ids.id_table, want_ids.id_table) and np.array_equal(got_ids.free_table, want_ids.free_table) and \
                (got_ids.num_free, got_ids.max_used) == (want_ids.num_free, want_ids.max_used)
        for d, dts_d in enumerate(more):
            if not ok:
                break
            ok = got_more[d][1] == ref_more[d][1]
            for k, dt in enumerate(dts_d):
                if not ok:
                    break
                m = got_more[d][1]
                x, y = got_more[d][0][k][offset:offset + m].view(np.uint32), ref_more[d][0][k][offset:offset + m].view(np.uint32)
                nan = np.isnan(x.view(f32)) & np.isnan(y.view(f32)) if dt == f32 else np.zeros(m, bool)
                ok = np.array_equal(x[~nan], y[~nan])
        if len(more) >= 2:
            key = "three or more acquired datasets, %d passes" % ((info.Flags >> 20) & 15)
            shapes[key] = shapes.get(key, 0) + 1
        if ok and in_batch and not neighbours.check():
            ok = False
            print("seed", seed, "the neighbours of the batch differ", flush=True)
        if not ok:
            bad += 1
            print("seed", seed, "MISMATCH (survivors %d / %d)" % (got[1], ref[1]), "vregs", info.NumVirtualRegisters, "width", info.InstancesPerLane, flush=True)
        if report_every and seed % report_every == 0:
            print("seed", seed, "failures so far", bad, "skipped", skipped, "kernel shapes", shapes, flush=True)
    neighbours.close()
    return bad, skipped, shapes


def main():
    bad, skipped, shapes = run(int(sys.argv[1]), int(sys.argv[2]))
    print("done: failures", bad, "skipped", skipped, "kernel shapes (instances per lane, more than 16 registers, LDS registers):", shapes)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
