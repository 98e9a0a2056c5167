#!/bin/bash
# Collects the rocprofv3 evidence behind profiles/ on the GPU box (run through gpurun from the repository root):
#   scripts/profile.sh <run-name>        ->  gpurun_out/<run-name>_{fse,huf}/{trace,pmc_fetch,pmc_write,pmc_rd,pmc_wr,pmc_sq1,pmc_sq2}
# then, back in the container:  python scripts/pmc_summary.py gpurun_out/<run>_fse r03_fse 20000   (and _huf)
# bench.py runs with --plain: warm-up + timed steps only, so every kernel is launched exactly warmup + steps times.
# Counters are collected in passes of their own (--pmc never together with tracing domains other than the kernel trace).
#   pmc_fetch / pmc_write : FETCH_SIZE / WRITE_SIZE, the derived counters MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950)
#   pmc_rd / pmc_wr       : the L2's memory-side request counters by request size (32 / 64 / 128 B), an exact byte count that
#                           needs no access-pattern calibration -- used to cross-check the corrected derived counters
#   pmc_sq1 / pmc_sq2     : what the kernels are really bound by (SURVEY 8(d) "report both"): wave cycles, busy cycles, VALU / LDS
#                           instruction activity, LDS-array cycles and bank conflicts, cycles parked in s_waitcnt, issue stalls
R=$(pwd)
RUN=${1:-prof}
cd /tmp && export TMPDIR=/tmp
P="--steps 2 --warmup 1 --blocks 20000"       # the counter passes: 3 launches of every kernel over 20000 blocks
for codec in $([ -n "$PROFILE_U16_ONLY" ] || echo ${PROFILE_CODECS:-fse huf}); do
    O=$R/gpurun_out/${RUN}_$codec
    mkdir -p $O
    B="python $R/bench.py --codec $codec --no-configs --plain"
    P="--steps 2 --warmup 1 --blocks 20000"
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B $P > $O/fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B $P > $O/write.log 2>&1
    timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_rd -o bench -- $B $P > $O/rd.log 2>&1
    timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc_wr -o bench -- $B $P > $O/wr.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq1 -o bench -- $B $P > $O/sq1.log 2>&1
    timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq2 -o bench -- $B $P > $O/sq2.log 2>&1
    tail -1 $O/trace.log | cut -c1-300
done
[ -n "$PROFILE_CODECS" ] && exit 0      # PROFILE_CODECS=fse scripts/profile.sh <run>: only the codec passes above (a kernel of theirs changed)
if [ -z "$PROFILE_U16_ONLY" ]; then     # PROFILE_U16_ONLY=1 PROFILE_CODECS= scripts/profile.sh <run>: only the 16-bit coder's passes
# BASELINE config 3 (Proba80, FSE): kernel trace + HBM traffic passes;  16-bit symbols and the using-table calls: kernel traces (their
# kernels run beside the headline's in one bench run; the rows are told apart by kernel name / call count)
O=$R/gpurun_out/${RUN}_p80
mkdir -p $O
B="python $R/bench.py --codec fse --proba 80 --no-configs --plain"
P="--steps 2 --warmup 1 --blocks 20000"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B $P > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $O/pmc_rd -o bench -- $B $P > $O/rd.log 2>&1
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $O/pmc_wr -o bench -- $B $P > $O/wr.log 2>&1
# BASELINE config 5's mix as the corpus (20k mixed P02 / P14 / P80 blocks through both codecs): HBM traffic of the mixed workload
O=$R/gpurun_out/${RUN}_mixed
mkdir -p $O
B="python $R/bench.py --codec both --workload mixed --no-configs --plain"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B $P > $O/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
fi
# the 16-bit-symbol coder: its kernels (k_u16_*) beside the headline's in one run
O=$R/gpurun_out/${RUN}_u16pmc
mkdir -p $O
B="python $R/bench.py --codec fse --configs fse_u16 --u16-blocks 20000 --plain"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B $P > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B $P > $O/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- $B --steps 5 --warmup 2 > $O/trace.log 2>&1
for extra in fse_u16 ${PROFILE_U16_ONLY:+} $([ -z "$PROFILE_U16_ONLY" ] && echo using_tables); do
    O=$R/gpurun_out/${RUN}_$extra
    mkdir -p $O
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $R/bench.py --codec fse --configs $extra --plain --steps 5 --warmup 2 > $O/trace.log 2>&1
done
# the using-table calls, one record per run (the headline of the same run uses the record's distribution, so that kernels both of them launch see one kind
# of data); python scripts/pmc_summary.py --ut gpurun_out/<run> <tag> 20000 turns them into profiles/traffic_<kernel>_ut_<key>.json
[ -n "$PROFILE_U16_ONLY" ] && exit 0
for rec in fse_p14:fse:14 fse_p80:fse:80 huf_p14:fse:14; do
    key=${rec%%:*}; rest=${rec#*:}; codec=${rest%%:*}; proba=${rest#*:}
    O=$R/gpurun_out/${RUN}_ut_$key
    mkdir -p $O
    B="python $R/bench.py --codec $codec --proba $proba --configs using_tables --ut-keys $key --plain"
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- $B $P > $O/fetch.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- $B $P > $O/write.log 2>&1
done
find $R/gpurun_out/${RUN}_fse $R/gpurun_out/${RUN}_huf $R/gpurun_out/${RUN}_p80 -name "*.csv" | head -60
