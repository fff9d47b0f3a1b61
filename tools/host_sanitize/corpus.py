#!/usr/bin/env python3
"""usage: corpus.py <directory>   writes the scripts the sanitizer drivers run on: frozen blobs (*.bin: the canonical scripts, the
single-op scripts, the ID and two-dataset cases, 400 optimizer-fuzz and 3000 linker-fuzz scripts) and compiler-format scripts
(*.bc with the external functions' input / output counts in *.bc.ext: 1500 optimizer-fuzz scripts)."""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    out = sys.argv[1]
    os.makedirs(out, exist_ok=True)
    import dataset_cases
    import gpu_fuzz_linker
    import id_cases
    import op_cases
    import original_cases
    from renderer_amd import scripts
    from test_optimizer import optimize
    from test_optimizer_fuzz import gen
    blobs, originals = [], []
    blobs += [scripts.fountain(variant=v).script.blob for v in range(64)] + [scripts.transcendental(True).script.blob]
    for name in list(op_cases.ULP_TOLERANCE) + "add mad select f2i cmplt_select mul_2x b2i_2x div sqrt mad_sqrt bit_rshift divi clamp round".split():
        blobs.append(op_cases.single_op_script(name)[0].blob)
    for make in (id_cases.update_script, id_cases.spawn_script, id_cases.update_and_spawn_script, id_cases.two_spawns_script, id_cases.two_updates_script):
        blobs.append(make().blob)
    blobs += [dataset_cases.two_dataset_script(a, b).blob for a in (False, True) for b in (False, True)]
    for seed in range(1500):
        o = gen(seed)[0]
        originals.append(o)
        if seed < 400:
            blobs.append(optimize(o))
    for seed in range(3000):
        try:
            blobs.append(gpu_fuzz_linker.gen(seed)[0].blob)
        except AssertionError:
            pass
    import through_cases
    blobs += [through_cases.wide_record_script(f, i, t, m).blob for f, i, t, m in ((16, 0, False, 0), (14, 2, False, 0), (12, 0, True, 0), (18, 1, False, 6), (40, 7, False, 0), (60, 0, False, 0), (2, 0, False, 0))]
    originals.append(original_cases.fountain()[0])
    for i, b in enumerate(blobs):
        with open(os.path.join(out, "b%05d.bin" % i), "wb") as f:
            f.write(bytes(b))
    for i, o in enumerate(originals):
        path = os.path.join(out, "o%05d.bc" % i)
        with open(path, "wb") as f:
            f.write(bytes(o.done()))
        with open(path + ".ext", "wb") as f:
            f.write(b"".join(struct.pack("<ii", a, b) for a, b in o.ext))
    print(len(blobs), "blobs,", len(originals), "compiler-format scripts")


if __name__ == "__main__":
    main()
