"""Ways of handing one logical constant array to an Exec as several constant tables (VectorVMRuntime.cpp:490-508: the tables
are concatenated in dword index space, each contributing NumBytes >> 2 dwords)."""
import numpy as np

from renderer_amd.assembler import FLOAT, Script

f32 = np.float32


def splits(blob: bytes):
    """(name, tables) pairs that all describe the dwords of `blob`."""
    n = len(blob) // 4
    dw = [blob[4 * i:4 * i + 4] for i in range(n)]

    def cut(points, tails=(), empties=()):
        tables, prev = [], 0
        for k, p in enumerate(list(points) + [n]):
            t = b"".join(dw[prev:p])
            if k in tails:
                t += b"\xAB\xCD\xEF"[: 1 + k % 3]          # 1..3 bytes past the last whole dword: not a constant
            tables.append(t)
            if k in empties:
                tables.append(b"")                            # a table with no dword at all
            prev = p
        return tables

    out = [("one", [blob])]
    if n >= 2:
        out.append(("two", cut([1])))
        out.append(("last-alone", cut([n - 1])))             # the last constant is the only dword of the last table
    if n >= 4:
        out.append(("three-unequal", cut([1, n - 2])))
        out.append(("four-with-tails", cut([2, 3, n - 1], tails=(0, 1, 2))))
        out.append(("empty-in-the-middle", cut([2, n - 1], empties=(0, 1))))
    return out


def many_constant_script(count=9):
    """out = ((x * c1 + c2) * c3 + c4) ... over `count` constants after the keep mask: neighbouring constants of one instruction
    land in different tables under most splits."""
    s = Script()
    s.acquireindex(s.const(0), 0)
    x = s.input(0, FLOAT, 0)
    acc = x
    for i in range(1, count, 2):
        t = s.temp()
        s.op("mad", acc, s.const(i), s.const(i + 1), t)
        acc = t
    s.output_data(FLOAT, 0, [(acc, s.output(0, FLOAT, 0))])
    consts = bytes([255] * 4) + (np.arange(1, count + 1, dtype=f32) * f32(0.37) - f32(1.0)).astype(f32).tobytes()
    return s.build(), consts
