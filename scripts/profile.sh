#!/bin/bash
# Collects the rocprofv3 evidence behind profiles/ on the GPU box (run through gpurun from the repository root):
#   scripts/profile.sh <run-name>        ->  gpurun_out/<run-name>_{fse,huf}/{trace,pmc_fetch,pmc_write}
# then, back in the container:  python scripts/pmc_summary.py gpurun_out/<run>_fse r01_fse 20000   (and _huf)
# Counters are collected in passes of their own (--pmc never together with tracing domains other than the kernel trace).
R=$(pwd)
RUN=${1:-prof}
cd /tmp && export TMPDIR=/tmp
for codec in fse huf; do
    O=$R/gpurun_out/${RUN}_$codec
    mkdir -p $O
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --codec $codec --steps 5 --warmup 2 > $O/trace.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python $R/bench.py --codec $codec --steps 2 --warmup 1 --blocks 20000 --no-cpu-baseline > $O/fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python $R/bench.py --codec $codec --steps 2 --warmup 1 --blocks 20000 --no-cpu-baseline > $O/write.log 2>&1
    tail -1 $O/trace.log | cut -c1-400
done
find $R/gpurun_out/${RUN}_fse $R/gpurun_out/${RUN}_huf -name "*.csv" | head -20
