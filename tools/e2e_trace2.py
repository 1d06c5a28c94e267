"""tools/e2e_trace2.py <tag> [min_us]: kernels + copies of an e2e_prof run on one time axis (ms since the first seed_select)"""
import csv, sys, re
tag = sys.argv[1]; min_us = int(sys.argv[2]) if len(sys.argv) > 2 else 100
def short(n):
    n = n.replace("femt::(anonymous namespace)::", "").replace("femk::", "").replace("void ", ""); n = re.sub(r"\(.*", "", n)
    if "rocprim" in n: n = "rocprim:" + ("init" if "init_lookback" in n else "scan")
    return n[:30]
ev = []
for r in csv.DictReader(open("gpurun_out/e2e_%s_kernel_trace.csv" % tag)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + short(r["Kernel_Name"]), "s" + r["Stream_Id"]))
for r in csv.DictReader(open("gpurun_out/e2e_%s_copy_trace.csv" % tag)):
    d = "H2D" if "HOST_TO_DEVICE" in r["Direction"] else "D2H" if "DEVICE_TO_HOST" in r["Direction"] else "D2D"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + d, "s" + r["Stream_Id"]))
t0 = min(s for s, e, n, st in ev if "seed_select" in n)
ev = sorted(x for x in ev if x[0] >= t0 - 5_000_000)
def union(xs):
    busy, cs, ce = 0, None, None
    for s, e in sorted(xs):
        if ce is None or s > ce:
            if ce is not None: busy += ce - cs
            cs, ce = s, e
        else: ce = max(ce, e)
    return busy + (ce - cs if ce else 0)
span = max(e for s, e, *_ in ev) - t0
print("span %.1f ms; kernels busy %.1f; H2D busy %.1f; D2H busy %.1f" % (span / 1e6, union([(s, e) for s, e, n, _ in ev if n[0] == "K"]) / 1e6,
      union([(s, e) for s, e, n, _ in ev if n == "C H2D"]) / 1e6, union([(s, e) for s, e, n, _ in ev if n == "C D2H"]) / 1e6))
for s, e, n, st in ev:
    if e - s >= min_us * 1000:
        print("%8.2f %8.2f %7.2f  %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, st, n))
