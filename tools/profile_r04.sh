#!/bin/bash
# Round-4 profiles of the bench command itself (run on the GPU box; results under gpurun_out/r4/, copy what is to be judged
# into profiles/):  kernel trace + stats of the headline and of the full default run, HBM traffic and SQ counters in passes
# of their own (rocprofv3 --pmc with --kernel-trace only).
set -u
out=gpurun_out/r4
mkdir -p $out
export TMPDIR=/tmp
head="--no-secondary --no-cpu-baseline --steps 50 --warmup 5 --repeats 6 --sustained 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_headline -o p -- python3 bench.py $head > $out/bench_headline.json 2> $out/bench_headline.err
cp $(find $out/prof_headline -name '*kernel_stats.csv' | head -1) $out/r04_kernel_stats_headline.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_full -o p -- python3 bench.py --no-cpu-baseline --repeats 6 --sustained 0 > $out/bench_full_profiled.json 2> $out/bench_full_profiled.err
cp $(find $out/prof_full -name '*kernel_stats.csv' | head -1) $out/r04_kernel_stats_full.csv
pmc_args="--no-secondary --no-cpu-baseline --no-seam --steps 30 --warmup 3 --repeats 1 --sustained 0"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python3 bench.py $pmc_args > /dev/null 2> $out/pmc_$c.err
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/pmc_sq -o p -- python3 bench.py $pmc_args > /dev/null 2> $out/pmc_sq.err
python3 tools/pmc_summary.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_sq > $out/r04_pmc.txt 2>&1
tail -40 $out/r04_pmc.txt
