#!/bin/bash
# tools/e2e_prof.sh <c2|c3> <n_reads> <tag>: kernel trace of one FEM map run to /dev/null
set -e
key=$1; n=$2; tag=$3
d=/dev/shm/fem_e2e_prof
trap "rm -rf $d" EXIT
python tools/e2e_files.py $key $n $d
e=3
export TMPDIR=/tmp
FEM_STAGE_TIMES=1 fem_amd/csrc/FEM map -e $e -t 16 --ref $d/ref.fa --index $d/ref.idx --read1 $d/reads.fq -o /dev/null 2> gpurun_out/e2e_${tag}_plain.log
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/e2e_${tag}_prof -o e2e -- fem_amd/csrc/FEM map -e $e -t 16 --ref $d/ref.fa --index $d/ref.idx --read1 $d/reads.fq -o /dev/null > gpurun_out/e2e_${tag}_rocprof.log 2>&1
find gpurun_out/e2e_${tag}_prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/e2e_${tag}_kernel_stats.csv \;
find gpurun_out/e2e_${tag}_prof -name "*kernel_trace.csv" -exec cp {} gpurun_out/e2e_${tag}_kernel_trace.csv \;
find gpurun_out/e2e_${tag}_prof -name "*.db" -delete
find gpurun_out/e2e_${tag}_prof -name "*memory_copy_trace.csv" -exec cp {} gpurun_out/e2e_${tag}_copy_trace.csv \;
head -40 gpurun_out/e2e_${tag}_kernel_stats.csv
grep -h "Time\|FEM\]" gpurun_out/e2e_${tag}_plain.log | tail -20
