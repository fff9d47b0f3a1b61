#!/usr/bin/env python3
"""Script files on the command line (no GPU needed):

  vvm_tool.py optimize <compiler-format bytecode file> <blob out> [--ext IN,OUT ...]   bytecode -> frozen blob (vvm_optimize_script)
  vvm_tool.py list <blob>                                                              side tables and instructions as text
  vvm_tool.py inspect <blob>                                                           what the GPU linker makes of it
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    o = sub.add_parser("optimize")
    o.add_argument("bytecode"); o.add_argument("blob")
    o.add_argument("--ext", action="append", default=[], help="NumInputs,NumOutputs of an external function, in table order")
    l = sub.add_parser("list"); l.add_argument("blob")
    i = sub.add_parser("inspect"); i.add_argument("blob")
    args = ap.parse_args()
    from renderer_amd import abi, disasm
    if args.cmd == "optimize":
        ext = [tuple(int(v) for v in e.split(",")) for e in args.ext]
        blob = abi.load_hip_backend().optimize_script(open(args.bytecode, "rb").read(), ext)
        open(args.blob, "wb").write(blob)
        print("%d bytes" % len(blob))
    elif args.cmd == "list":
        print(disasm.listing(open(args.blob, "rb").read()))
    else:
        info = abi.load_hip_backend().inspect_blob(open(args.blob, "rb").read())
        for name, _ in info._fields_:
            print("%-22s %d" % (name, getattr(info, name)))


if __name__ == "__main__":
    main()
