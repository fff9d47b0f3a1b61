#!/usr/bin/env python3
"""Writes tests/golden/opcode_table.json: the reference's list of VectorVM opcodes as data (Runtime/VectorVM/Public/VectorVM.h,
the VVM_OP_XM_LIST table): for every opcode its position in the list (= its numeric value, EVectorVMOp is generated from the list
in order), name, category and operand counts as the table states them (the table's other columns are not kept).  Nothing of the reference is compiled or executed: the script reads the table rows as
text, where the reference's sources are present (this container), and stores the fields; the tests read only the JSON."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "opcode_table.json")
SOURCE = "/root/reference/Runtime/VectorVM/Public/VectorVM.h"


def main():
    rows = []
    first = last = None
    with open(SOURCE, encoding="utf-8", errors="replace") as f:
        for lineno, line in enumerate(f, 1):
            m = re.match(r"\s*VVM_OP_XM\(\s*(\w+)\s*,\s*(\w+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\w+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\w+)\s*\)", line)
            if not m:
                continue
            first = first or lineno
            last = lineno
            name, cat, nin, nout = m.groups()[:4]
            rows.append({"value": len(rows), "name": name, "category": cat, "num_inputs": int(nin), "num_outputs": int(nout)})
    out = {"about": "VVM_OP_XM_LIST as data; written by make_opcode_table.py", "source": "VectorVM.h:%d-%d" % (first, last), "ops": rows}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=0)
        f.write("\n")
    print("wrote", OUT, len(rows), "ops", out["source"])


if __name__ == "__main__":
    main()
