#!/usr/bin/env python3
"""tools/blocks.py <kernel.s> : per basic block — label, loop depth, counts of v_/s_/ds_/global_/scratch_ and the branches out"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
blocks = []; cur = None
depth = 0
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
    if m:
        cur = {"label": m.group(1), "line": i + 1, "v": 0, "s": 0, "ds": 0, "g": 0, "sc": 0, "br": [], "depth": None, "rl": 0}
        blocks.append(cur)
        d = re.search(r"Depth=(\d+)", l)
        if d: cur["depth"] = int(d.group(1))
        continue
    if cur is None: continue
    d = re.search(r"Depth=(\d+)", l)
    if d and cur["depth"] is None and l.strip().startswith(";"): cur["depth"] = int(d.group(1))
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    if op.startswith("v_"): cur["v"] += 1; cur["rl"] += op.startswith("v_readlane")
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): cur["br"].append(t.split()[0][2:] + ">" + t.split()[1])
    elif op.startswith("s_"): cur["s"] += 1
    elif op.startswith("ds_"): cur["ds"] += 1
    elif op.startswith("global_"): cur["g"] += 1
    elif op.startswith("scratch_"): cur["sc"] += 1
for b in blocks:
    print("%-12s L%-5d d=%s v %3d (rl %2d) s %3d ds %2d g %2d sc %d  %s" % (b["label"], b["line"], b["depth"], b["v"], b["rl"], b["s"], b["ds"], b["g"], b["sc"], " ".join(b["br"])))
