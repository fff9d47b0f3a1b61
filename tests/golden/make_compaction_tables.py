#!/usr/bin/env python3
"""Writes tests/golden/compaction_tables.json: the four tables of numbers by which the reference's output instructions compact
the survivors of a group of four instances (Runtime/VectorVM/Private/VectorVMRuntime.cpp):

  shuffle32   VVM_PSHUFB_OUTPUT_TABLE    :132-150   per 4-bit keep mask, the 16 byte selectors of _mm_shuffle_epi8 (0xFF = zero)
  shuffle16   VVM_PSHUFB_OUTPUT_TABLE16  :152-170   the same for four 16-bit elements (the low 8 bytes matter)
  advance32   VVM_OUTPUT_ADVANCE_TABLE   :172-178   bytes the destination moves on after the group
  advance16   VVM_OUTPUT_ADVANCE_TABLE16 :180-186

These numbers ARE the reference's definition of the order in which survivors are written (VVM_Output32 :744-822, VVM_Output16
:579-742 index them by the mask byte that VVM_acquireindex :824-878 wrote), so they pin what no arithmetic vector can: bit i of
the mask is instance i of the group, survivors move to the front in instance order, the destination advances by the
survivors' bytes.  Nothing of the reference is compiled or executed: the script reads the array literals as text, where the
reference's sources are present (this container), and stores the numbers; the tests read only the JSON.
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "compaction_tables.json")
SOURCE = "/root/reference/Runtime/VectorVM/Private/VectorVMRuntime.cpp"
TABLES = [("shuffle32", "VVM_PSHUFB_OUTPUT_TABLE", 256), ("shuffle16", "VVM_PSHUFB_OUTPUT_TABLE16", 256),
          ("advance32", "VVM_OUTPUT_ADVANCE_TABLE", 16), ("advance16", "VVM_OUTPUT_ADVANCE_TABLE16", 16)]


def main():
    with open(SOURCE, encoding="utf-8", errors="replace") as f:
        text = f.read()
    lines = text.split("\n")
    out = {"about": "keep-mask -> byte shuffle / destination advance, as VectorVMRuntime.cpp holds them; written by make_compaction_tables.py",
           "tables": {}}
    for key, name, count in TABLES:
        m = re.search(r"\b%s\s*\[\s*\]\s*(?:GCC_ALIGN\(\d+\)\s*)?=\s*\{(.*?)\};" % re.escape(name), text, re.S)
        if not m:
            sys.exit("table %s not found" % name)
        body = re.sub(r"//[^\n]*", "", m.group(1))
        numbers = [int(tok, 0) for tok in re.findall(r"0[xX][0-9a-fA-F]+|\d+", body)]
        if len(numbers) != count:
            sys.exit("table %s: %d numbers, expected %d" % (name, len(numbers), count))
        first = text[:m.start()].count("\n") + 1
        last = text[:m.end()].count("\n") + 1
        assert name in lines[first - 1]
        out["tables"][key] = {"name": name, "source": "VectorVMRuntime.cpp:%d-%d" % (first, last), "values": numbers}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
