#!/usr/bin/env python3
"""spill_map.py <kernel.s> — where the scratch (spill) instructions of a step kernel sit.

The .s comes from tools/probes/r6/ru.sh "<args>" out.s -gline-tables-only. Every scratch_load / scratch_store (and v_accvgpr_*, the
AGPR copies) is attributed to the source line of the last .loc in front of it, binned into the regions of lm_core.h::forward, with
the loop depth of its basic block (back edges of the label graph). Static counts x an assumed trip count say which spills are hot.
"""
import re
import sys
from collections import Counter, defaultdict

MARKERS = [  # (text that opens the region in lm_core.h, name): line numbers are looked up in the source, so the map follows edits
    ("LM_DEV void forward(", "fwd: root kinematics"), ("// chain: kinematics + velocity recursion", "fwd: chain kinematics"),
    ("// floor contacts of this chain's geoms", "fwd: floor contacts"), ("// ======== self-collisions (PAIRS)", "fwd: pair pass"),
    ("// DEFER (the kernels with a pair pass)", "fwd: CRB + bias"), ("// ======== muscles:", "fwd: muscles"),
    ("// ======== smooth forces, unconstrained acceleration", "fwd: smooth/park M/factor"), ("// cross-chain contacts: who is coupled", "fwd: unit rows"),
    ("// ================= constraint solve: Newton", "newton: setup/cost_at/warm"), ("for (int it = 0; it <= P.iterations; it++)", "newton: gradient"),
    ("// ---- Hessian H = M + J^T W J", "newton: hessian"), ("float Lr[21];", "newton: factor+solve"),
    ("// ---- Newton decrement", "newton: jv + linesearch"), ("if (want_grf) {", "fwd: grf/dbg/euler"), ("LM_DEV void substep(", "substep (RK4)"),
]


def _regions():
    import os
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "loco_mujoco_amd", "csrc", "lm_core.h")).read().splitlines()
    out = [(0, "helpers")]
    for text, name in MARKERS:
        hit = [i + 1 for i, ln in enumerate(src) if text in ln]
        assert hit, text
        out.append((hit[0], name))
    return out


REGIONS = _regions()


def region_of(fileno_core, fileno, line):
    if fileno != fileno_core:
        return "lm_step.h / other"
    name = REGIONS[0][1]
    for first, n in REGIONS:
        if line >= first:
            name = n
    return name


def main():
    path = sys.argv[1]
    lines = open(path).read().splitlines()
    files = {}
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
    core = [k for k, v in files.items() if v.endswith("lm_core.h")]
    core = core[0] if core else -1
    # basic blocks and loop depth: a branch to a label defined EARLIER is a back edge; every block between target and branch is in the loop
    label_at = {}
    for i, ln in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', ln)
        if m:
            label_at[m.group(1)] = i
    depth = [0] * (len(lines) + 1)
    for i, ln in enumerate(lines):
        m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\d+_\d+)', ln) or re.match(r'\s*s_add_u32 s\d+, s\d+, \((\.LBB\d+_\d+)-\.Lpost_getpc', ln)   # (long branches: s_getpc + s_add + s_setpc)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            for j in range(label_at[m.group(1)], i + 1):
                depth[j] += 1
    cur = (-1, 0)
    by = defaultdict(Counter)
    tot = Counter()
    for i, ln in enumerate(lines):
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
        if m:
            f_, l_ = int(m.group(1)), int(m.group(2))
            # inlined helpers (V3 operators, arrow_factor, the colliders: lines above forward()) keep the region of the last
            # line of forward() / substep() seen: the .s carries no inlined-at chain
            if f_ == core and l_ >= REGIONS[1][0]:
                cur = (f_, l_)
            elif f_ != core and l_ >= 247 and files.get(f_, "").endswith("lm_step.h"):      # (above: the quad policy's helpers, inlined everywhere)
                cur = (f_, l_)
            continue
        s = ln.strip()
        kind = None
        if s.startswith("scratch_load"):
            kind = "ld"
        elif s.startswith("scratch_store"):
            kind = "st"
        elif s.startswith("v_accvgpr"):
            kind = "acc"
        elif s.startswith("v_readlane") or s.startswith("v_writelane"):
            kind = "lane"
        elif s and not s.startswith((".", ";", "//")) and not s.endswith(":"):
            kind = "inst"
        if kind:
            r = region_of(core, cur[0], cur[1])
            d = min(depth[i], 5)
            by[r][(kind, d)] += 1
            tot[kind] += 1
    print("totals:", dict(tot))
    print("%-46s %7s | %s" % ("region", "insts", "scratch ld / st by loop depth 0..5+   | accvgpr | lane"))
    for _, name in REGIONS + [(0, "lm_step.h / other")]:
        c = by.get(name)
        if not c:
            continue
        insts = sum(v for (k, d), v in c.items())
        ld = [c.get(("ld", d), 0) for d in range(6)]
        st = [c.get(("st", d), 0) for d in range(6)]
        acc = sum(c.get(("acc", d), 0) for d in range(6))
        lane = sum(c.get(("lane", d), 0) for d in range(6))
        print("%-46s %7d | ld %s st %s | %5d | %5d" % (name, insts, ld, st, acc, lane))


if __name__ == "__main__":
    main()
