#!/usr/bin/env python3
"""The benchmark configurations other than the headline one (bench.py holds that):

  transcendental   16 Mi particles, sin_cos + exp + curl-noise script            (BASELINE.json configs[2])
  kill_half        64 Mi particles, fountain with 50 % death                      (configs[3], one GPU's share scaled down)
  death_sweep      64 Mi particles, fountain with 1 % to 75 % death
  kill_half_250m   250 M particles, fountain with 50 % death                      (configs[3], one GPU's share of 2 G particles)
  one_million      1 Mi particles, fountain                                       (configs[0], the reference's CPU-runnable case)
  reference_random 16 Mi particles, three random draws: the counter-based generator against the reference's xorwow
  batch            4096 emitters x 16 Ki particles, 64 distinct bytecode bodies, one launch per kernel shape  (configs[4])

Prints one JSON object per configuration.  Times are wall-clock around synchronous calls (vvm_exec / vvm_exec_batch), so
they include the launch and the parameter upload; `kernel_ms` is the device time from HIP events where available.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="transcendental,kill_half,batch")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-light", action="store_true", help="keep eight-wide Execs on the general kernel")
    ap.add_argument("--no-through", action="store_true", help="wide_record: run the script whole instead of in its lean form (copy kernel for the handed-through streams)")
    ap.add_argument("--force-eight", action="store_true", help="single Execs: eight instances per lane whatever the size (and with it the lean form of a wide-record script)")
    ap.add_argument("--extras", default="16,8", help="wide_record: numbers of handed-through attributes to run")
    ap.add_argument("--wide-particles", type=int, default=32 * 1024 * 1024, help="wide_record: instances")
    ap.add_argument("--trig", action="store_true", help="wide_record: a sine and an exponential in what the script computes (the kernels with the transcendental functions)")
    ap.add_argument("--batch-per", type=int, default=16384, help="batch_wide: particles per emitter")
    ap.add_argument("--wide-kill-half", action="store_true", help="wide_record: half of the particles die")
    ap.add_argument("--more-math", type=int, default=0, help="wide_record: that many of the extra attributes are scaled and offset instead of handed through")
    ap.add_argument("--separate-passes", action="store_true", help="two_datasets: one launch per dataset instead of one launch for both")
    ap.add_argument("--batch-variants", type=int, default=64, help="bytecode variants of the batch (the first 16 all fit the light kernel)")
    ap.add_argument("--batch-flags", type=int, default=0, help="vvm_exec_options.Flags of every member of the batch (4: force eight instances per lane)")
    ap.add_argument("--arith-ops", default="16,64", help="arith: instruction counts to run")
    args = ap.parse_args()
    import torch
    from renderer_amd import abi, scripts
    from renderer_amd.runtime import VectorVMState
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    be = abi.load_hip_backend()
    PAD = 64

    def run_single(name, w, ins, n_out, n, ext=()):
        outs = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(n_out)]
        torch.cuda.synchronize()
        st = VectorVMState(w.script.blob, be)
        frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [w.constants], list(ext))
        o = abi.ExecOptions(0, 0, 0, None, (abi.VVM_EXEC_NO_LIGHT_KERNEL if args.no_light else 0) | (abi.VVM_EXEC_NO_THROUGH_KERNEL if args.no_through else 0) |
                             (abi.VVM_EXEC_FORCE_EIGHT_WIDE if args.force_eight else 0))
        for _ in range(3):
            surv = st.exec(frame, o)
        st.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.exec_async(frame, o)
        surv = st.finish(frame)
        el = time.perf_counter() - t0
        ms, launches = st.collect_kernel_ms()
        info = st.info()
        width = st.last_exec_width()
        st.close()
        kernel_ms = ms / max(launches, 1)
        alg = info.BytesInPerInstance * n + info.BytesOutPerInstance * surv
        print(json.dumps({"config": name, "particles": n, "survivors": surv, "steps": args.steps, "ms_per_step": el / args.steps * 1e3,
                          "kernel_ms": kernel_ms, "updates_per_s": n * args.steps / el, "algorithmic_GBps": alg / kernel_ms / 1e6,
                          "frac_of_8TBps": alg / kernel_ms / 1e6 / 8000.0, "instances_per_lane": width,
                          "streams_left_to_the_copy_kernel": (info.Flags >> 24) if (info.Flags & 0x80000 and not args.no_through and not args.no_light) else 0}), flush=True)

    def arith_script(n_ops):
        """4 streams in, 4 out, n_ops dependent mad / mul / add instructions spread over the streams: interpreter throughput."""
        from renderer_amd.assembler import FLOAT, Script
        s = Script()
        ins_ = [s.input(0, FLOAT, i) for i in range(4)]
        s.acquireindex(s.const(0), 0)
        cur = list(ins_)
        for k in range(n_ops):
            t = s.temp()
            a, b = cur[k % 4], cur[(k + 1) % 4]
            if k % 3 == 0:
                s.op("mad", a, s.const(1), b, t)
            elif k % 3 == 1:
                s.op("mul", a, b, t)
            else:
                s.op("add", a, s.const(2), t)
            cur[k % 4] = t
        s.output_data(FLOAT, 0, [(cur[i], s.output(0, FLOAT, i)) for i in range(4)])
        return s.build(alias_outputs=True)

    which = args.which.split(",")
    if "arith" in which:
        import struct
        from renderer_amd import scripts as _s
        n = 64 * 1024 * 1024
        ins = _s.fountain_streams_torch(0, n, dev, pad=PAD)
        for n_ops in [int(x) for x in args.arith_ops.split(",")]:
            class W:      # (run_single wants .script and .constants)
                pass
            w = W()
            w.script = arith_script(n_ops)
            w.constants = struct.pack("<I2f", 0xFFFFFFFF, 1.0009765625, 0.125) + bytes(4)
            t_before = time.perf_counter()
            run_single("arithmetic %d ops, 4 streams in / 4 out, 64Mi" % n_ops, w, ins[:4], 4, n)
            print(json.dumps({"config": "arithmetic %d ops: interpreted instructions x instances per second = value of 'updates_per_s' x %d" % (n_ops, n_ops)}), flush=True)

    if "reference_random" in which:
        # 16 Mi particles, one stream in, three draws (random, random_2x) and three streams out: the counter-based generator
        # against the reference's xorwow laid out ahead of the Exec (vvm_exec_options.RandXorwowState)
        import ctypes as C
        import struct
        from renderer_amd.assembler import FLOAT, Script
        s = Script()
        x = s.input(0, FLOAT, 0)
        s.acquireindex(s.const(0), 0)
        r0, r1, r2 = s.temps(3)
        s.op("random", x, r0); s.op("random_2x", x, r1, r2)
        s.output_data(FLOAT, 0, [(r, s.output(0, FLOAT, i)) for i, r in enumerate((r0, r1, r2))])
        built = s.build()
        n = 16 * 1024 * 1024
        xs = torch.rand(n + PAD, dtype=torch.float32, device=dev)
        outs = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(3)]
        torch.cuda.synchronize()
        words = (C.c_uint32 * 24)(*be.xorwow_seed(0x1234, 0x5678, 0x7F0000001000))
        for name, use in (("counter-based generator", False), ("reference xorwow from an injected state", True)):
            st = VectorVMState(built.blob, be)
            frame = st.frame([abi.DataSet([xs.data_ptr()], [t.data_ptr() for t in outs], 0)], n, [struct.pack("<I", 0xFFFFFFFF) + bytes(12)])
            o = abi.ExecOptions(7, 0, 0, None, 0)
            if use:
                o.RandXorwowState = words
            for _ in range(3):
                st.exec(frame, o)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                st.exec_async(frame, o)
            st.finish(frame)
            el = time.perf_counter() - t0
            width = st.last_exec_width()
            st.close()
            print(json.dumps({"config": "3 random draws, 1 stream in / 3 out, 16Mi: " + name, "ms_per_step": el / args.steps * 1e3,
                              "updates_per_s": n * args.steps / el, "instances_per_lane": width}), flush=True)

    if "six_out" in which:
        # the fountain script with two of its eight outputs left out (eight streams in, six out): what the light eight-wide kernel,
        # whose staging holds six outputs, does on fountain-like traffic, against the general kernel (--no-light)
        from renderer_amd.assembler import FLOAT, Script
        import struct
        s = Script()
        dt, drag = s.const(0), s.const(4)
        pos = [s.input(0, FLOAT, k) for k in range(3)]
        vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
        age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
        age1 = s.temp(); s.op("add", age, dt, age1)
        alive = s.temp(); s.op("cmplt", age1, life, alive)
        s.acquireindex(alive, 0)
        v1, p1 = s.temps(3), s.temps(3)
        for c in range(3):
            s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
            s.op("mad", v1[c], dt, pos[c], p1[c])
        s.output_data(FLOAT, 0, [(p1[0], s.output(0, FLOAT, 0)), (p1[1], s.output(0, FLOAT, 1)), (p1[2], s.output(0, FLOAT, 2)),
                                 (v1[0], s.output(0, FLOAT, 3)), (v1[1], s.output(0, FLOAT, 4)), (age1, s.output(0, FLOAT, 5))])
        class W6:
            pass
        w = W6()
        w.script = s.build(alias_outputs=True)
        w.constants = struct.pack("<6f", 1.0 / 60.0, 0.0, 0.0, -980.0, 0.98, 1.0) + bytes(8)
        n = 64 * 1024 * 1024
        run_single("fountain with six outputs, 64Mi", w, scripts.fountain_streams_torch(0, n, dev, pad=PAD), 6, n)
        run_single("fountain with six outputs, 64Mi, 50% death", w, scripts.fountain_streams_torch(0, n, dev, kill_half=True, pad=PAD), 6, n)
    if "wide_record" in which:
        # the fountain update on particles that carry sixteen more attributes (colour, size, rotation, ...) which the script hands
        # through unchanged, as a Niagara update script does with every attribute it does not touch (the buffers are double-buffered,
        # so they are still read and written): 24 streams in, 24 out.  Every input of the whole script is in a per-lane register from
        # the top of the tile (the 32-register four-wide kernel); in its lean form the light kernel runs what the script computes and
        # a copy kernel moves the rest.
        from renderer_amd.assembler import FLOAT, Script
        import struct
        for extra in [int(x) for x in args.extras.split(",")]:
            s = Script()
            dt, drag = s.const(0), s.const(4)
            pos = [s.input(0, FLOAT, k) for k in range(3)]
            vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
            age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
            rest = [s.input(0, FLOAT, 8 + k) for k in range(extra)]
            age1 = s.temp(); s.op("add", age, dt, age1)
            alive = s.temp(); s.op("cmplt", age1, life, alive)
            s.acquireindex(alive, 0)
            v1, p1 = s.temps(3), s.temps(3)
            for c in range(3):
                s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
                s.op("mad", v1[c], dt, pos[c], p1[c])
            if args.trig:
                t = s.temp(); s.op("sin", v1[0], t); v1[0] = t
                t = s.temp(); s.op("exp", age1, t)
                u = s.temp(); s.op("mul", p1[1], t, u); p1[1] = u
            changed = []
            for k in range(min(args.more_math, extra)):
                t = s.temp(); s.op("mad", rest[k], drag, age1, t)
                changed.append(t)
            s.output_data(FLOAT, 0, [(p1[c], s.output(0, FLOAT, c)) for c in range(3)] + [(v1[c], s.output(0, FLOAT, 3 + c)) for c in range(3)] +
                          [(age1, s.output(0, FLOAT, 6)), (life, s.output(0, FLOAT, 7))] +
                          [(changed[k] if k < len(changed) else rest[k], s.output(0, FLOAT, 8 + k)) for k in range(extra)])
            class WR:
                pass
            w = WR()
            w.script = s.build(alias_outputs=True)
            w.constants = struct.pack("<6f", 1.0 / 60.0, 0.0, 0.0, -980.0, 0.98, 1.0) + bytes(8)
            n = args.wide_particles
            base = scripts.fountain_streams_torch(0, n, dev, kill_half=args.wide_kill_half, pad=PAD)
            more = [torch.rand(n + PAD, dtype=torch.float32, device=dev) for _ in range(extra)]
            run_single(("fountain with a sine and an exponential, " if args.trig else "fountain with ") + "%d more attributes, %d of them computed with, the others handed through (%d streams in / %d out), %d instances" % (extra, min(args.more_math, extra), 8 + extra, 8 + extra, n), w, base + more, 8 + extra, n)
            del more
    if "many_datasets" in which:
        # the fountain update with a dataset for the particles that die this tick and one or two more event-like datasets (particles
        # below a height, particles beyond a distance): 3 or 4 acquired datasets, every one reading the fountain's streams.  Run with
        # VVM_NO_CHAINS=1 (one program and one launch per dataset) for the comparison.
        from renderer_amd.assembler import FLOAT, Script
        import struct
        for K in (3, 4):
            s = Script()
            dt, drag = s.const(0), s.const(4)
            pos = [s.input(0, FLOAT, k) for k in range(3)]
            vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
            age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
            age1 = s.temp(); s.op("add", age, dt, age1)
            alive, dead = s.temps(2)
            s.op("cmplt", age1, life, alive)
            s.op("cmpge", age1, life, dead)
            s.acquireindex(alive, 0)
            s.acquireindex(dead, 1)
            v1, p1 = s.temps(3), s.temps(3)
            for c in range(3):
                s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
                s.op("mad", v1[c], dt, pos[c], p1[c])
            s.output_data(FLOAT, 0, [(p1[c], s.output(0, FLOAT, c)) for c in range(3)] + [(v1[c], s.output(0, FLOAT, 3 + c)) for c in range(3)] +
                          [(age1, s.output(0, FLOAT, 6)), (life, s.output(0, FLOAT, 7))])
            s.output_data(FLOAT, 1, [(p1[c], s.output(1, FLOAT, c)) for c in range(3)] + [(age1, s.output(1, FLOAT, 3))])
            for d in range(2, K):
                hit = s.temp()
                s.op("cmplt", p1[d - 2], s.const(5), hit)
                s.acquireindex(hit, d)
                s.output_data(FLOAT, d, [(p1[c], s.output(d, FLOAT, c)) for c in range(3)] + [(age1, s.output(d, FLOAT, 3))])
            built = s.build()
            consts = struct.pack("<6f", 1.0 / 60.0, 0.0, 0.0, -980.0, 0.98, 0.25) + bytes(8)
            n = 64 * 1024 * 1024
            ins = scripts.fountain_streams_torch(0, n, dev, kill_half=True, pad=PAD)
            outs = [[torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(8 if d == 0 else 4)] for d in range(K)]
            torch.cuda.synchronize()
            st = VectorVMState(built.blob, be)
            frame = st.frame([abi.DataSet([t.data_ptr() for t in ins] if d == 0 else [], [t.data_ptr() for t in outs[d]], 0) for d in range(K)], n, [consts])
            o = abi.ExecOptions(0, 0, 0, None, 0)
            for _ in range(3):
                st.exec(frame, o)
            st.enable_timing(True)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                st.exec_async(frame, o)
            st.finish(frame)
            el = time.perf_counter() - t0
            ms, launches = st.collect_kernel_ms()
            surv = [frame.survivors(d) for d in range(K)]
            passes = (st.info().Flags >> 20) & 15
            st.close()
            alg = 32.0 * n + 32.0 * surv[0] + 16.0 * sum(surv[1:])
            print(json.dumps({"config": "fountain with %d more datasets (8 streams in, 8 + %s out), 64Mi" % (K - 1, " + ".join(["4"] * (K - 1))), "particles": n, "survivors": surv,
                              "passes": passes, "ms_per_step": el / args.steps * 1e3, "kernel_ms_per_exec": ms / max(launches, 1), "updates_per_s": n * args.steps / el,
                              "algorithmic_GBps": alg / (ms / max(launches, 1)) / 1e6}), flush=True)
            del outs
    if "two_datasets" in which:
        # the fountain update with a second dataset that receives the particles that die this tick (position and age): both
        # datasets' programs read all eight streams
        from renderer_amd.assembler import FLOAT, Script
        import struct
        s = Script()
        dt, drag = s.const(0), s.const(4)
        pos = [s.input(0, FLOAT, k) for k in range(3)]
        vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
        age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
        age1 = s.temp(); s.op("add", age, dt, age1)
        alive, dead = s.temps(2)
        s.op("cmplt", age1, life, alive)
        s.op("cmpge", age1, life, dead)
        s.acquireindex(alive, 0)
        s.acquireindex(dead, 1)
        v1, p1 = s.temps(3), s.temps(3)
        for c in range(3):
            s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
            s.op("mad", v1[c], dt, pos[c], p1[c])
        s.output_data(FLOAT, 0, [(p1[c], s.output(0, FLOAT, c)) for c in range(3)] + [(v1[c], s.output(0, FLOAT, 3 + c)) for c in range(3)] +
                      [(age1, s.output(0, FLOAT, 6)), (life, s.output(0, FLOAT, 7))])
        s.output_data(FLOAT, 1, [(p1[c], s.output(1, FLOAT, c)) for c in range(3)] + [(age1, s.output(1, FLOAT, 3))])
        built = s.build()
        consts = struct.pack("<6f", 1.0 / 60.0, 0.0, 0.0, -980.0, 0.98, 1.0) + bytes(8)
        n = 64 * 1024 * 1024
        ins = scripts.fountain_streams_torch(0, n, dev, kill_half=True, pad=PAD)
        outs0 = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
        outs1 = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(4)]
        torch.cuda.synchronize()
        st = VectorVMState(built.blob, be)
        frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs0], 0), abi.DataSet([], [t.data_ptr() for t in outs1], 0)], n, [consts])
        o = abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_SEPARATE_PASSES if args.separate_passes else 0)
        for _ in range(3):
            st.exec(frame, o)
        st.enable_timing(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.exec_async(frame, o)
        s0 = st.finish(frame)
        el = time.perf_counter() - t0
        ms, launches = st.collect_kernel_ms()
        s1 = frame.survivors(1)
        st.close()
        alg = 32.0 * n + 32.0 * s0 + 16.0 * s1
        print(json.dumps({"config": "fountain with a second dataset for the particles that die (8 streams in, 8 + 4 out), 64Mi, " +
                          ("one launch per dataset" if args.separate_passes else "both datasets in one launch"),
                          "particles": n, "survivors": [s0, s1], "steps": args.steps, "ms_per_step": el / args.steps * 1e3, "kernel_ms_per_exec": ms / max(launches, 1),
                          "algorithmic_GBps": alg / (ms / max(launches, 1)) / 1e6, "algorithmic_bytes": alg}), flush=True)
    if "transcendental" in which:
        n = 16 * 1024 * 1024
        w = scripts.transcendental(True)
        run_single("transcendental 16Mi (sin_cos, exp, curl_noise3)", w, scripts.transcendental_streams_torch(0, n, dev, pad=PAD), 4, n, w.ext_fns)
    if "kill_half" in which:
        n = 64 * 1024 * 1024
        w = scripts.fountain()
        run_single("fountain 64Mi, 50% death", w, scripts.fountain_streams_torch(0, n, dev, kill_half=True, pad=PAD), 8, n)
    if "death_sweep" in which:
        # the fountain with a given share of the particles dying in the Exec (spatially random)
        n = 64 * 1024 * 1024
        w = scripts.fountain()
        for rate in (0.0, 0.01, 0.05, 0.1, 0.2, 0.25, 0.3, 0.4, 0.5, 0.75):
            ins = scripts.fountain_streams_torch(0, n, dev, pad=PAD)
            g = torch.Generator(device=dev); g.manual_seed(11)
            dead = torch.rand(n + PAD, device=dev, generator=g) < rate
            ins[6] = torch.where(dead, ins[7], ins[6])
            run_single("fountain 64Mi, %g%% death" % (rate * 100), w, ins, 8, n)
            del ins, dead
    if "kill_half_250m" in which:
        n = 250_000_000
        w = scripts.fountain()
        run_single("fountain 250M, 50% death (one GPU's share of 2G particles)", w, scripts.fountain_streams_torch(0, n, dev, kill_half=True, pad=PAD), 8, n)
    if "small_sizes" in which:
        # one Exec of the fountain script at small sizes, four-wide against the (forced) light eight-wide kernel
        w = scripts.fountain()
        for n in (16 * 1024, 64 * 1024, 128 * 1024, 256 * 1024, 1024 * 1024):
            ins = scripts.fountain_streams_torch(0, n, dev, pad=PAD)
            for label, flag in (("four-wide", abi.VVM_EXEC_NO_EIGHT_WIDE), ("eight-wide forced", abi.VVM_EXEC_FORCE_EIGHT_WIDE)):
                args.no_light, keep = False, args.no_light
                outs = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
                st = VectorVMState(w.script.blob, be)
                frame = st.frame([abi.DataSet([t.data_ptr() for t in ins], [t.data_ptr() for t in outs], 0)], n, [w.constants])
                o = abi.ExecOptions(0, 0, 0, None, flag)
                for _ in range(3):
                    st.exec(frame, o)
                st.enable_timing(True)
                for _ in range(50):
                    st.exec_async(frame, o)
                st.finish(frame)
                ms, launches = st.collect_kernel_ms()
                print(json.dumps({"config": "fountain %d instances, %s" % (n, label), "kernel_us": ms / launches * 1e3, "width": st.last_exec_width()}), flush=True)
                st.close()
                args.no_light = keep
    if "one_million" in which:
        n = 1024 * 1024
        w = scripts.fountain()
        ins1m = scripts.fountain_streams_torch(0, n, dev, pad=PAD)
        run_single("fountain 1Mi", w, ins1m, 8, n)
        # (run_single brackets every Exec with two timing events; the same queue of Execs without them)
        outs_plain = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
        sp = VectorVMState(w.script.blob, be)
        fp = sp.frame([abi.DataSet([t.data_ptr() for t in ins1m], [t.data_ptr() for t in outs_plain], 0)], n, [w.constants])
        op = abi.ExecOptions(0, 0, 0, None, 0)
        for _ in range(3):
            sp.exec(fp, op)
        reps = max(args.steps, 10) * 8
        t0 = time.perf_counter()
        for _ in range(reps):
            sp.exec_async(fp, op)
        sp.finish(fp)
        el = time.perf_counter() - t0
        print(json.dumps({"config": "fountain 1Mi, %d Execs queued through vvm_exec_async, no timing events" % reps, "particles": n, "us_per_exec": el / reps * 1e6}), flush=True)
        sp.close()
        # the same Exec as nodes of a HIP graph: two states that hand the particles back and forth (A's outputs are B's inputs), sixteen
        # ticks of the pair captured once and replayed; no host work per Exec, so what is left is the kernel and the gap between kernels
        from renderer_amd import hip
        mid = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
        back = [torch.zeros(n + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
        torch.cuda.synchronize()
        stream = hip.Stream()
        sa, sb = VectorVMState(w.script.blob, be), VectorVMState(w.script.blob, be)
        fa = sa.frame([abi.DataSet([t.data_ptr() for t in ins1m], [t.data_ptr() for t in mid], 0)], n, [w.constants])
        fb = sb.frame([abi.DataSet([t.data_ptr() for t in mid], [t.data_ptr() for t in back], 0)], n, [w.constants])
        o = abi.ExecOptions(0, 0, 0, stream.value, 0)
        for _ in range(3):
            sa.exec(fa, o); sb.exec(fb, o)
        per_graph = 32
        stream.begin_capture(hip.hipStreamCaptureModeGlobal)
        for _ in range(per_graph // 2):
            sa.exec_async(fa, o); sb.exec_async(fb, o)
        graph = stream.end_capture()
        graph.launch(stream); stream.synchronize()
        replays = max(args.steps, 10)
        t0 = time.perf_counter()
        for _ in range(replays):
            graph.launch(stream)
        stream.synchronize()
        el = time.perf_counter() - t0
        surv = (sa.finish(fa), sb.finish(fb))
        assert surv == (n, n), surv
        print(json.dumps({"config": "fountain 1Mi, %d Execs captured into a HIP graph (%d nodes), %d replays" % (per_graph, graph.num_nodes(), replays),
                          "particles": n, "us_per_exec": el / (replays * per_graph) * 1e6, "updates_per_s": n * replays * per_graph / el}), flush=True)
        graph.destroy(); sa.close(); sb.close(); stream.destroy()
    if "batch_wide" in which:
        # 4096 emitters x 16 Ki particles whose records carry sixteen attributes that the script hands through (24 streams each way)
        import struct
        from renderer_amd.assembler import FLOAT, Script
        emitters, per, extra = 4096, args.batch_per, 16
        total = emitters * per
        ins = scripts.fountain_streams_torch(0, total, dev, kill_half=True, pad=PAD) + [torch.rand(total + PAD, dtype=torch.float32, device=dev) for _ in range(extra)]
        outs = [torch.zeros(total + PAD, dtype=torch.float32, device=dev) for _ in range(8 + extra)]
        torch.cuda.synchronize()
        s = Script()
        dt, drag = s.const(0), s.const(4)
        pos = [s.input(0, FLOAT, k) for k in range(3)]
        vel = [s.input(0, FLOAT, 3 + k) for k in range(3)]
        age, life = s.input(0, FLOAT, 6), s.input(0, FLOAT, 7)
        rest = [s.input(0, FLOAT, 8 + k) for k in range(extra)]
        age1 = s.temp(); s.op("add", age, dt, age1)
        alive = s.temp(); s.op("cmplt", age1, life, alive)
        s.acquireindex(alive, 0)
        v1, p1 = s.temps(3), s.temps(3)
        for c in range(3):
            s.op("mad_mul", s.const(1 + c), dt, vel[c], drag, v1[c])
            s.op("mad", v1[c], dt, pos[c], p1[c])
        s.output_data(FLOAT, 0, [(p1[c], s.output(0, FLOAT, c)) for c in range(3)] + [(v1[c], s.output(0, FLOAT, 3 + c)) for c in range(3)] +
                      [(age1, s.output(0, FLOAT, 6)), (life, s.output(0, FLOAT, 7))] + [(rest[k], s.output(0, FLOAT, 8 + k)) for k in range(extra)])
        blob = s.build(alias_outputs=True).blob
        consts = struct.pack("<6f", 1.0 / 60.0, 0.0, 0.0, -980.0, 0.98, 1.0) + bytes(8)
        states, frames, opts = [], [], []
        for e in range(emitters):
            st = VectorVMState(blob, be)
            states.append(st)
            off = e * per * 4
            frames.append(st.frame([abi.DataSet([t.data_ptr() + off for t in ins], [t.data_ptr() + off for t in outs], 0)], per, [consts]))
            opts.append(abi.ExecOptions(0, 0, 0, None, args.batch_flags))
        arr, oarr = be.pack_batch(frames, opts)
        handle = be.batch_create(arr, oarr)
        for _ in range(5):
            be.batch_exec(handle, 0)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            be.batch_exec(handle, 0)
        el = time.perf_counter() - t0
        surv = sum(f.survivors(0) for f in frames)
        alg = (32.0 + 4.0 * extra) * (total + surv)
        print(json.dumps({"config": "batch 4096 emitters x %d, %d streams each way, through vvm_batch_exec, nothing changes" % (per, 8 + extra), "particles": total, "survivors": surv,
                          "steps": args.steps, "ms_per_step": el / args.steps * 1e3, "updates_per_s": total * args.steps / el,
                          "algorithmic_GBps_wallclock": alg * args.steps / el / 1e9, "width": states[0].last_exec_width()}), flush=True)
        be.batch_destroy(handle)
        for st in states:
            st.close()
    if "batch" in which:
        emitters, per = 4096, 16384
        total = emitters * per
        ins = scripts.fountain_streams_torch(0, total, dev, kill_half=True, pad=PAD)
        outs = [torch.zeros(total + PAD, dtype=torch.float32, device=dev) for _ in range(8)]
        torch.cuda.synchronize()
        workloads = [scripts.fountain(variant=v % args.batch_variants) for v in range(64)]
        states, frames, opts = [], [], []
        for e in range(emitters):
            w = workloads[e % 64]
            st = VectorVMState(w.script.blob, be)
            states.append(st)
            off = e * per * 4
            frames.append(st.frame([abi.DataSet([t.data_ptr() + off for t in ins], [t.data_ptr() + off for t in outs], 0)], per, [w.constants]))
            opts.append(abi.ExecOptions(0, 0, 0, None, args.batch_flags))
        arr, oarr = be.pack_batch(frames, opts)
        for _ in range(25):      # (host-bound: the first calls also measure the host's clocks ramping up)
            be.exec_batch_packed(arr, oarr)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            be.exec_batch_packed(arr, oarr)
        el = time.perf_counter() - t0
        surv = sum(f.survivors(0) for f in frames)
        alg = 32.0 * total + 32.0 * surv
        print(json.dumps({"config": "batch 4096 emitters x 16Ki, 64 bytecode variants", "particles": total, "survivors": surv, "steps": args.steps,
                          "ms_per_step": el / args.steps * 1e3, "updates_per_s": total * args.steps / el,
                          "algorithmic_GBps_wallclock": alg * args.steps / el / 1e9}), flush=True)
        # the same batch kept in a handle: nothing changes between ticks / the constants change every tick
        handle = be.batch_create(arr, oarr)
        for changed, label in ((0, "nothing changes"), (abi.VVM_BATCH_CHANGED_CONSTANTS, "constants change every tick"),
                               (abi.VVM_BATCH_CHANGED_INSTANCES | abi.VVM_BATCH_CHANGED_CONSTANTS, "sizes are re-read and constants change every tick")):
            for _ in range(2):
                be.batch_exec(handle, changed)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                be.batch_exec(handle, changed)
            el = time.perf_counter() - t0
            surv2 = sum(f.survivors(0) for f in frames)
            print(json.dumps({"config": "batch 4096 emitters x 16Ki through vvm_batch_exec, " + label, "particles": total, "survivors": surv2, "steps": args.steps,
                              "ms_per_step": el / args.steps * 1e3, "updates_per_s": total * args.steps / el,
                              "algorithmic_GBps_wallclock": (32.0 * total + 32.0 * surv2) * args.steps / el / 1e9}), flush=True)
        be.batch_destroy(handle)
        for st in states:
            st.close()


if __name__ == "__main__":
    main()
