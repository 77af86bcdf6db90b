set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/prof
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_kt.json 2> $O/bench_kt.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq1 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_sq1.json 2> $O/bench_sq1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_sq2.json 2> $O/bench_sq2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fetch.json 2> $O/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_write.json 2> $O/bench_write.err
find $O -name "*.csv" | head -40
