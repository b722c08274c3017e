#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only): total instructions and the
count of given mnemonics.  Used to spot no-op repacking (v_perm_b32 / v_lshrrev pairs the compiler emits when a
__bf16 vector lives across divergent control flow).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S x.hip -o /tmp/x.s
  python scripts/asm_mix.py /tmp/x.s [v_perm_b32 ...]"""
import collections
import re
import sys


def main():
    path, pats = sys.argv[1], (sys.argv[2:] or ["v_perm_b32"])
    cur = None
    tot, cnt = collections.Counter(), collections.Counter()
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        t = line.strip()
        if cur is None or not t or t[0] in ".;" or t.endswith(":"):
            continue
        tot[cur] += 1
        if any(t.startswith(p) for p in pats):
            cnt[cur] += 1
    for k, v in cnt.most_common(40):
        print("%6d of %6d  %s" % (v, tot[k], k[:120]))


if __name__ == "__main__":
    main()
