#!/bin/bash
# development aid: Huff0 round trip at P14 / P80 / P02 with per-kernel times -> gpurun_out/$1/
out=gpurun_out/${1:-hd}; mkdir -p $out
for p in 14 80 2; do
  timeout 200 python bench.py --codec huf --proba $p --steps 5 --warmup 2 --no-configs --no-cpu-baseline > $out/huf_p$p.json 2> $out/huf_p$p.err
  python scripts/showbench.py $out/huf_p$p.json
done
