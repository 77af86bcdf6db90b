# Round-2 profile set: bash tools/prof_r02.sh <tag>   (on the GPU box, through gpurun)
#  kt      rocprofv3 --kernel-trace --stats of the default bench (all kernels the JSON line quotes)
#  kt_main the same without the secondary measurements: the headline kernel alone (its average must match roofline.kernel_ms)
#  sq1/sq2 SQ counters of the headline kernel;  mfma: matrix-pipe counters of the dense kernels;  fetch/write: HBM traffic
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/prof_$1
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_kt.json 2> $O/bench_kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_main -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary --no-seam > $O/bench_kt_main.json 2> $O/bench_kt_main.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/sq1 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_sq1.json 2> $O/bench_sq1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_sq2.json 2> $O/bench_sq2.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_mfma.json 2> $O/bench_mfma.err || echo "mfma counter pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_fetch.json 2> $O/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_write.json 2> $O/bench_write.err
python3 tools/pmc_summary.py $O/sq1 $O/sq2 $O/mfma $O/fetch $O/write > $O/pmc_summary.txt
cat $O/kt/*/*_kernel_stats.csv | head -12
cat $O/kt_main/*/*_kernel_stats.csv | head -6
grep -c . $O/pmc_summary.txt
