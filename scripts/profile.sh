#!/bin/bash
# Collects the rocprofv3 evidence behind profiles/ on the GPU box (run through gpurun from the repository root):
#   scripts/profile.sh <run-name>        ->  gpurun_out/<run-name>_{fse,huf}/{trace,pmc_fetch,pmc_write,pmc_rd,pmc_wr}
# then, back in the container:  python scripts/pmc_summary.py gpurun_out/<run>_fse r02_fse 20000   (and _huf)
# Counters are collected in passes of their own (--pmc never together with tracing domains other than the kernel trace).
#   pmc_fetch / pmc_write : FETCH_SIZE / WRITE_SIZE, the derived counters MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950)
#   pmc_rd / pmc_wr       : the L2's memory-side request counters by request size (32 / 64 / 128 B), an exact byte count that
#                           needs no access-pattern calibration -- used to cross-check the corrected derived counters
R=$(pwd)
RUN=${1:-prof}
cd /tmp && export TMPDIR=/tmp
for codec in fse huf; do
    O=$R/gpurun_out/${RUN}_$codec
    mkdir -p $O
    B="python $R/bench.py --codec $codec --no-configs --no-cpu-baseline"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B --steps 2 --warmup 1 --blocks 20000 > $O/fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B --steps 2 --warmup 1 --blocks 20000 > $O/write.log 2>&1
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_rd -o bench -- $B --steps 2 --warmup 1 --blocks 20000 > $O/rd.log 2>&1
    timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc_wr -o bench -- $B --steps 2 --warmup 1 --blocks 20000 > $O/wr.log 2>&1
    tail -1 $O/trace.log | cut -c1-300
done
find $R/gpurun_out/${RUN}_fse $R/gpurun_out/${RUN}_huf -name "*.csv" | head -30
