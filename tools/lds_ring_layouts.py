#!/usr/bin/env python3
"""The LDS ring layouts of csrc/le_rows.hip (Lay64<A0,S0,S1>, Lay32<A0,S0>) checked against tools/lds_bank_model.py: prints the
bank-conflict cycles of every access pattern of every ring (all must be 0), and with --search re-derives the layouts by brute
force over the parity masks.  The access patterns are listed in le_rows.hip's layout comment (R1, S2, W1, W2)."""
import sys

from lds_bank_model import cycles

L = [(lane & 31, lane >> 5) for lane in range(64)]


def par(x):
    return bin(x).count("1") & 1


def lay64(A0, S0, S1):
    return lambda c, k: ((c ^ par(c & A0)) << 6) | ((k ^ (par(c & S0) | (par(c & S1) << 1))) << 4)


def lay32(A0, S0):
    return lambda c, h: ((c ^ par(c & A0)) << 5) | ((h ^ par(c & S0)) << 4)


def R1(a, bases=(0, 32)):      # MFMA fragment reads, stride 1
    return sum(cycles("ds_read_b128", [a(c0 + l + kx, (ks << 1) | lh) for l, lh in L])[1] for c0 in bases for kx in range(3) for ks in range(2))


def S2(a):                     # ... stride 2
    return sum(cycles("ds_read_b128", [a(2 * l + kx, (ks << 1) | lh) for l, lh in L])[1] for kx in range(3) for ks in range(2))


def W1(a, instr="ds_write_b128", offs=(0,)):      # chunk writes (reads: instr = ds_read_b128) of slot c0 + l31 + off
    return sum(cycles(instr, [a(c0 + l + o, lh + 2 * q) for l, lh in L])[1] for c0 in (0, 32) for o in offs for q in range(2))


def W2(a, instr="ds_write_b128"):                 # ... of slot 2 l31 + gh
    return sum(cycles(instr, [a(2 * l + gh, lh + 2 * q) for l, lh in L])[1] for gh in (0, 1) for q in range(2))


def C1(a):
    return sum(cycles("ds_read_b128", [a(c0 + l + o, lh) for l, lh in L])[1] for c0 in (0, 32) for o in (0, 1))


def C2(a):
    return sum(cycles("ds_read_b128", [a(2 * l + gh, lh) for l, lh in L])[1] for gh in (0, 1))


RINGS = {
    "LStd   = Lay64<0, 4, 10>": (lay64(0, 4, 10), [("R1", R1), ("W1 write", W1), ("W1 read, slots +0..+2", lambda a: W1(a, "ds_read_b128", (0, 1, 2)))]),
    "LTailY = Lay64<4, 3, 8> ": (lay64(4, 3, 8), [("R1", R1), ("W2 write", W2)]),
    "LTailF = Lay64<4, 8, 16>": (lay64(4, 8, 16), [("W2 read", lambda a: W2(a, "ds_read_b128"))]),
    "LHeadF = Lay64<4, 9, 18>": (lay64(4, 9, 18), [("S2", S2), ("W1 write", W1)]),
    "LCond  = Lay32<0, 8>    ": (lay32(0, 8), [("stride-1 reads", C1)]),
    "LCondT = Lay32<8, 16>   ": (lay32(8, 16), [("stride-2 reads", C2)]),
    "round 4: s = bits 2..3  ": (lay64(0, 0, 0) if False else (lambda c, k: (c << 6) | ((k ^ ((c >> 2) & 3)) << 4)), [("R1", R1), ("S2", S2), ("W1 write", W1), ("W2 write", W2)]),
}

if __name__ == "__main__":
    bad = 0
    for name, (a, pats) in RINGS.items():
        for pn, f in pats:
            c = f(a)
            print(f"{name}  {pn:24s} conflict cycles {c}")
            bad += c if not name.startswith("round 4") else 0
    if "--search" in sys.argv:
        sol = [(s0, s1) for s0 in range(64) for s1 in range(64)
               if R1(lay64(0, s0, s1)) == 0 and W1(lay64(0, s0, s1)) == 0 and W1(lay64(0, s0, s1), "ds_read_b128", (0, 1, 2)) == 0]
        print("LStd candidates (A0 = 0):", sol[:8], "...", len(sol))
        for label, cond in (("LTailY", lambda a: R1(a) == 0 and W2(a) == 0), ("LTailF", lambda a: W2(a, "ds_read_b128") == 0),
                            ("LHeadF", lambda a: S2(a) == 0 and W1(a) == 0)):
            sol = [(4, s0, s1) for s0 in range(64) for s1 in range(64) if cond(lay64(4, s0, s1))]
            print(f"{label} candidates (A0 = 4):", sol[:8], "...", len(sol))
    sys.exit(1 if bad else 0)
