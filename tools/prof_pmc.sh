# SQ counter passes of the default bench workload (no secondary measurements): bash tools/prof_pmc.sh <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/prof_$1
mkdir -p $O
cd $R
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq1 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_sq1.json 2> $O/bench_sq1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_sq2.json 2> $O/bench_sq2.err
python3 tools/pmc_summary.py $O/sq1 $O/sq2 > $O/summary.txt
cat $O/summary.txt
