#!/usr/bin/env python3
"""Instruction histogram per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): tuning aid for the
ISA-level instruction budgets quoted in DESIGN.md / profiles/.  usage: isa_hist.py file.s [name-substring ...]"""
import collections
import re
import sys


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and line.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        if name:
            s = line.strip()
            if not s or s[0] in ".;/" or s.endswith(":"):
                continue
            body.append(s.split(";")[0].strip())


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    for name, body in kernels(path):
        if pats and not any(p in name for p in pats):
            continue
        c = collections.Counter(l.split()[0] for l in body)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop", "s_barrier")))
        print(f"{name}: {len(body)} instructions, VALU {valu}, SALU {salu}, s_nop {c['s_nop']}, s_waitcnt {c['s_waitcnt']}")
        print("   " + ", ".join(f"{k} {v}" for k, v in c.most_common(30)))


if __name__ == "__main__":
    main()
