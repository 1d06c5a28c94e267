#!/usr/bin/env python3
"""tools/isa.py <file.s> <kernel substring>...  -> static instruction counts, registers, spills; writes <kernel>.s next to the input"""
import re, sys, os
src = sys.argv[1]
text = open(src).read().split("\n")
for name in sys.argv[2:]:
    start = next(i for i, l in enumerate(text) if re.match(r"^_ZN4femk\d+%sE\w*:" % re.escape(name), l))
    end = next(i for i in range(start, len(text)) if ".end_amdhsa_kernel" in text[i])
    body = text[start:end + 40]
    out = os.path.join(os.path.dirname(src), name + ".s")
    open(out, "w").write("\n".join(body))
    ins = [l.strip() for l in body if re.match(r"^\s+[a-z_0-9]+\s", l) and not l.strip().startswith(".")]
    cnt = lambda p: sum(1 for l in ins if l.startswith(p))
    info = {k: next((l.split()[-1] for l in body if k in l), "?") for k in (".sgpr_spill_count", ".vgpr_spill_count", "next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size")}
    meta = "\n".join(text[end:end + 400])
    print("%-32s v_ %4d  s_ %4d  ds_ %3d  global_ %3d  readlane %3d  scratch_ %2d | vgpr %s sgpr %s scratch %s" % (
        name, cnt("v_"), cnt("s_"), cnt("ds_"), cnt("global_"), sum(1 for l in ins if l.startswith("v_readlane")), cnt("scratch_"),
        info["next_free_vgpr"], info["next_free_sgpr"], info["private_segment_fixed_size"]))
