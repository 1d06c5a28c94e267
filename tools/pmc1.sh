#!/bin/bash
# usage: tools/pmc1.sh <workload> name... : one counter pass (instruction counts) per variant -> gpurun_out/pmc1_<wl>.txt
cd /tmp && export TMPDIR=/tmp
wl=$1; shift
root=$GRAFT_REPO_ROOT
cd $root
for n in "$@"; do
  out=$root/gpurun_out/pmc1_$n
  rm -rf $out; mkdir -p $out
  export FEM_HIP_LIBRARY=$root/build/abl/libfemhip_$n.so FEM_TESTING=1 FEM_NO_PARTS=1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/a -- python3 bench.py --workload $wl --profile-replay 6 > $out/a.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((x for x in ("seed_select_kernel", "seed_join", "verify_kernel") if x in r["Kernel_Name"]), None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
line = "%-8s" % "$n"
for k in ("seed_join", "seed_select_kernel"):
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"):
        v = acc[k][c]; t = v[len(v)//2:] or [0]
        line += " %s.%s %.1f" % (k[5:9], c[9:], sum(t)/len(t)/2.5e6)
print(line)
open("$root/gpurun_out/pmc1_$wl.txt", "a").write(line + "\n")
PY
done
