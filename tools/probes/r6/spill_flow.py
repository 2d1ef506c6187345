#!/usr/bin/env python3
"""spill_flow.py <kernel.s> — which region of the kernel STORES a scratch slot and which regions LOAD it (slots are dwords, offsets
from the scratch_* instruction's immediate). A slot stored in `chain kinematics` and loaded in `CRB + bias` is a value kept across the
pair pass; a slot stored and loaded inside `pair pass` at depth 5 is a hot spill of the collider. Companion of spill_map.py."""
import re
import sys
from collections import Counter, defaultdict
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import spill_map as sm


def main():
    lines = open(sys.argv[1]).read().splitlines()
    files = {}
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
    core = [k for k, v in files.items() if v.endswith("lm_core.h")][0]
    cur = (-1, 0)
    slots = defaultdict(lambda: {"st": Counter(), "ld": Counter()})
    for ln in lines:
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
        if m:
            f_, l_ = int(m.group(1)), int(m.group(2))
            if f_ == core and l_ >= sm.REGIONS[1][0]:
                cur = (f_, l_)
            elif f_ != core and l_ >= 247 and files.get(f_, "").endswith("lm_step.h"):
                cur = (f_, l_)
            continue
        s = ln.strip()
        m = re.match(r'scratch_(load|store)_dword(x(\d))?\s+(.*)', s)
        if not m:
            continue
        n = int(m.group(3) or 1)
        mo = re.search(r'offset:(\d+)', s)
        off = int(mo.group(1)) if mo else 0
        kind = "ld" if m.group(1) == "load" else "st"
        r = sm.region_of(core, cur[0], cur[1])
        for k in range(n):
            slots[off + 4 * k][kind][r] += 1
    flow = Counter()
    for off, d in slots.items():
        st = "+".join(sorted(d["st"])) or "-"
        for r in d["ld"]:
            flow[(st, r)] += 1
        if not d["ld"]:
            flow[(st, "(never loaded)")] += 1
    print("%d slots (dwords) in use" % len(slots))
    for (st, ld), n in sorted(flow.items(), key=lambda kv: -kv[1])[:60]:
        print("%4d  stored in [%s]  -> loaded in [%s]" % (n, st, ld))


if __name__ == "__main__":
    main()
