"""The reference-held vectors of tests/golden/mathtest_vectors.json (UnrealMathTest.cpp's test of the SIMD math layer the VM ops
call), run through one-instruction scripts on any executor and compared the way the reference's test compares."""
import json
import os

import numpy as np

from op_cases import ALL_ONES, single_op_script

f32 = np.float32
_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mathtest_vectors.json")
RECORDS = json.load(open(_PATH))["records"]
IDS = ["%s@%s" % (r["op"], r["source"].split(":")[1]) for r in RECORDS]

# the scalar functions the reference's test evaluates beside the vector ones (FMath::Sin = sinf, ...; glibc here)
_LIBM = {
    "sinf": np.sin, "cosf": np.cos, "tanf": np.tan, "atanf": np.arctan, "atan2f": np.arctan2, "expf": np.exp, "exp2f": np.exp2,
    "logf": np.log, "log2f": np.log2, "fmodf": np.fmod,
}


def check_record(r, run):
    """run(script, const_tables, inputs, out_dtypes, n) -> (outputs, survivors, info): run_oracle or run_hip."""
    script, n_in, n_out = single_op_script(r["op"])
    ins = [np.array(v, dtype=np.uint32).view(f32) for v in r["inputs"]]
    assert len(ins) == n_in
    n = len(ins[0])
    outs, surv, _ = run(script, [ALL_ONES], ins, [f32] * n_out, n)
    assert surv == n
    got = [o[:n] for o in outs]
    if r["libm"]:
        want = [_LIBM[name](*[a.astype(f32) for a in ins]).astype(f32) for name in r["libm"]]      # float32 in, float32 out: the libm float function
    elif r["expected"] is not None:
        want = [np.array(v, dtype=np.uint32).view(f32) for v in r["expected"]]
    else:
        want = None
    tol = r["tolerance"]
    check = r["check"]
    if check == "bitwise":
        for g, w in zip(got, want):
            assert np.array_equal(g.view(np.uint32), w.view(np.uint32)), (r["op"], g, w)
    elif check == "sum_abs":
        for g, w in zip(got, want):
            err = np.abs(g.astype(f32) - w.astype(f32)).reshape(-1, 4).astype(f32).sum(axis=1)
            assert (err <= f32(tol)).all(), (r["op"], err.max(), tol)
    elif check == "component":
        for g, w in zip(got, want):
            err = np.abs(g.astype(f32) - w.astype(f32))
            assert (err <= f32(tol)).all(), (r["op"], float(err.max()), tol, ins[0][int(err.argmax())])
    elif check == "product_one":
        err = np.abs(f32(1.0) - got[0] * ins[0]).reshape(-1, 4).sum(axis=1)
        assert (err <= f32(tol)).all(), (r["op"], err)
    elif check == "square_product_one":
        err = np.abs(f32(1.0) - (got[0] * got[0]) * ins[0]).reshape(-1, 4).sum(axis=1)
        assert (err <= f32(tol)).all(), (r["op"], err)
    else:
        raise AssertionError("unknown check " + check)
