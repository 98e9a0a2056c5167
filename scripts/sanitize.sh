#!/bin/bash
# The reference ships `make sanitize` / `memtest` for its host code (Makefile:75-79, programs/Makefile:165-170).  This is that leg for
# libfsehip.so: the HOST side of the library (C ABI, per-thread arenas, the frame calls' thread pool and rings) built with
# -fsanitize=address,undefined (make -C finitestateentropy_amd/csrc SAN=1 -> variants/san/libfsehip.so; device code unchanged) and run
# under tests/host/san_driver.c (host threads over the host-pointer calls, the frame thread pool, the _wksp names; every result against
# the reference) and the reference's three fuzzers bound to the device.  (C programs, not pytest: the sanitizer runtime's HSA interceptor
# fails inside the ROCm runtime bundled with the torch wheel -- "allocator is trying to allocate" at the first hipMalloc -- while programs
# on /opt/rocm's own runtime run fine.)
#   scripts/sanitize.sh [out-dir]       (on the GPU box, from the repository root; the variant is built in the container and travels)
# Leak checking is off (the Python interpreter and the HIP runtime keep process-lifetime allocations); everything else is fatal.
R=$(pwd)
O=${1:-$R/gpurun_out/san}
mkdir -p $O
LIB=$R/finitestateentropy_amd/csrc/variants/san
[ -f $LIB/libfsehip.so ] || make -C $R/finitestateentropy_amd/csrc SAN=1 -j8 > $O/build.log 2>&1 || { echo "sanitized build failed"; exit 1; }
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:exitcode=97
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=98
rc=0
# tests/host/san_driver.c: host threads over the calls on host pointers, the frame calls' thread pool, the _wksp names -- against the reference
LD_PRELOAD=$RT LD_LIBRARY_PATH=$LIB timeout 900 $R/oracle/_ref/san_driver 4 12 > $O/san_driver.log 2>&1 || { rc=1; echo "san_driver under the sanitizers: FAILED"; }
tail -2 $O/san_driver.log
# the reference's own fuzzers, bound to the device (oracle/Makefile), on the sanitized library
for f in fuzzer fuzzerHuff0 fuzzerU16; do
    LD_PRELOAD=$RT LD_LIBRARY_PATH=$LIB timeout 900 $R/oracle/_ref/$f-mi355x -s1 -i300 > $O/$f.log 2>&1 || { rc=1; echo "$f under the sanitizers: FAILED"; }
    tail -c 200 $O/$f.log; echo
done
# the single calls of the table glue and of the header-reading Huff0 family (round 6), through the reference's fullbench bound to the device
for c in 4 5 6 21 22 40 41 45 80 81; do
    LD_PRELOAD=$RT LD_LIBRARY_PATH=$LIB timeout 300 $R/oracle/_ref/fullbench-mi355x -i1 -b$c > $O/fullbench_$c.log 2>&1 || { rc=1; echo "fullbench -b$c under the sanitizers: FAILED"; }
done
tail -c 120 $O/fullbench_45.log; echo
grep -l "ERROR: AddressSanitizer\|runtime error:" $O/*.log && rc=1
# the sanitized library must really be the one that ran: its mapping shows in the driver's process
LD_PRELOAD=$RT LD_LIBRARY_PATH=$LIB LD_DEBUG=libs $R/oracle/_ref/san_driver 1 1 2>&1 | grep -m1 "variants/san/libfsehip.so" >> $O/san_driver.log || { rc=1; echo "the sanitized library was not the one loaded"; }
echo "sanitize.sh: rc=$rc" | tee $O/verdict.txt
exit $rc
