#!/usr/bin/env python3
"""ULP report of the GPU executor against the CPU oracle (run on the GPU box; the oracle runs on the host cores).

  * the transcendental script (sin_cos + exp + curl-noise built-in) at the full benchmark size, 16 Mi particles: histogram of
    the ULP distance per output stream, maximum, and the bound the tests hold it to;
  * every single op that is not bit-exact by construction (libm-class functions and the reciprocal estimate): maximum ULP
    distance over the edge-value and random operand streams of tests/op_cases.py.

Prints one JSON object.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=16 * 1024 * 1024)
    args = ap.parse_args()
    import numpy as np
    import op_cases
    from renderer_amd import scripts
    from vm_harness import run_hip, run_oracle, ulp_diff
    f32 = np.float32
    report = {"oracle": "CPU restatement of the reference (oracle/vvm_oracle.c), not the reference's own binary"}

    w = scripts.transcendental(True)
    n = args.particles
    ins = scripts.transcendental_streams_np(0, n)
    ref, ref_n, _ = run_oracle(w.script, [w.constants], ins, [f32] * 4, n, ext_fns=list(w.ext_fns))
    got, got_n, _ = run_hip(w.script, [w.constants], ins, [f32] * 4, n, ext_fns=list(w.ext_fns))
    streams = {}
    for k, name in enumerate(("pos.x", "pos.y", "pos.z", "phase")):
        d = ulp_diff(got[k][:n], ref[k][:n])
        hist = np.bincount(np.minimum(d, 4).astype(np.int64), minlength=5)
        streams[name] = {"max_ulp": int(d.max()), "histogram_0_1_2_3_ge4": [int(v) for v in hist]}
    report["transcendental_script"] = {"particles": n, "survivors_gpu": int(got_n), "survivors_oracle": int(ref_n), "bound_ulp": 2, "streams": streams}

    ops = {}
    for name in sorted(set(op_cases.ULP_TOLERANCE) | set(op_cases.REL_TOLERANCE) | {"sin", "cos", "exp", "log", "pow", "sin_cos", "sqrt", "rsq", "div", "atan2", "asin", "acos"}):
        script, n_in, n_out = op_cases.single_op_script(name)
        cols = op_cases.operand_streams(name, n_in, 65536)
        r, rn, _ = run_oracle(script, [op_cases.ALL_ONES], cols, [f32] * n_out, 65536)
        g, gn, _ = run_hip(script, [op_cases.ALL_ONES], cols, [f32] * n_out, 65536)
        worst = 0
        for a, b in zip(g, r):
            fin = np.isfinite(b[:65536])
            d = ulp_diff(a[:65536], b[:65536])
            worst = max(worst, int(d[fin].max()) if fin.any() else 0)
        ops[name] = {"max_ulp_on_finite_results": worst, "test_bound_ulp": op_cases.ULP_TOLERANCE.get(name, 0),
                     "note": "relative bound 2^-11, both sides are estimates" if name in op_cases.REL_TOLERANCE else ""}
    report["single_ops"] = ops

    # the libm-class functions over their whole useful domains (glibc on the host, the device math library on the GPU): nothing is
    # clipped away, inputs whose results differ by more than the bound are listed
    rng = np.random.default_rng(2024)
    half_pi = float(np.float32(np.pi / 2))
    domains = {
        "tan": [np.linspace(-half_pi + 1e-4, half_pi - 1e-4, 400_000), rng.uniform(-20.0, 20.0, 400_000),
                half_pi - np.geomspace(1e-7, 1e-2, 50_000), -half_pi + np.geomspace(1e-7, 1e-2, 50_000), rng.uniform(-1e4, 1e4, 100_000)],
        "exp2": [np.linspace(-150.0, 128.0, 600_000), rng.uniform(-1.0, 1.0, 200_000), np.arange(-150, 129, dtype=np.float64), rng.uniform(-1000, 1000, 100_000)],
        "log2": [np.geomspace(1e-45, 3e38, 600_000), rng.uniform(0.5, 2.0, 300_000)],
        "atan": [np.linspace(-50.0, 50.0, 400_000), np.geomspace(1e-30, 1e30, 200_000), -np.geomspace(1e-30, 1e30, 200_000)],
    }
    wide = {}
    for name, parts in domains.items():
        x = np.concatenate(parts).astype(f32)
        script, n_in, n_out = op_cases.single_op_script(name)
        r, _, _ = run_oracle(script, [op_cases.ALL_ONES], [x], [f32], len(x))
        g, _, _ = run_hip(script, [op_cases.ALL_ONES], [x], [f32], len(x))
        d = ulp_diff(g[0][:len(x)], r[0][:len(x)])
        fin = np.isfinite(r[0][:len(x)])
        d = np.where(fin, d, 0)
        bound = op_cases.ULP_TOLERANCE[name]
        over = np.nonzero(d > bound)[0]
        worst = np.argsort(d)[-5:][::-1]
        wide[name] = {"inputs": int(len(x)), "max_ulp": int(d.max()), "bound_ulp": bound, "above_bound": int(over.size),
                      "histogram_0_1_2_3_ge4": [int(v) for v in np.bincount(np.minimum(d, 4).astype(np.int64), minlength=5)],
                      "worst_inputs": [{"x": float(x[i]), "gpu": float(g[0][i]), "oracle": float(r[0][i]), "ulp": int(d[i])} for i in worst if d[i] > 0]}
    report["libm_class_wide_domains"] = wide
    print(json.dumps(report))


if __name__ == "__main__":
    main()
