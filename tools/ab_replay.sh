#!/bin/bash
# tools/ab_replay.sh <workload> <steps> name...   -> one line per variant: Mreads/s, join ms, select ms
wl=$1; steps=$2; shift 2
for n in "$@"; do
  FEM_HIP_LIBRARY=$PWD/build/abl/libfemhip_$n.so python bench.py --workload $wl --profile-replay $steps 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
e = d['event_times']
print('$wl %-8s %7.2f Mreads/s  join %.3f  select %.3f  verify %.3f  filter %.3f' % ('$n', d['mreads_per_s'], e['seed_join_kernel']['mean_ms'], e['seed_select_kernel']['mean_ms'], e['verify_kernel']['mean_ms'], e['seed_filter_kernel']['mean_ms']))"
done
