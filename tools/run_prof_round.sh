# Profiles of a round, on the GPU box: bench line, rocprofv3 kernel-trace summary of the same
# command, and PMC passes (one per counter group) over the RANSAC call of the benchmark's shape.
# usage: bash tools/run_prof_round.sh r02   (raw output under gpurun_out/prof_<round>)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_line.json 2> $O/bench.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pnp -- $R/tools/pnp_time_bin 16 4096 2000 > $O/kt_pnp.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- $R/tools/pnp_time_bin 16 4096 2000 > $O/sq1.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $O/sq2 -- $R/tools/pnp_time_bin 16 4096 2000 > $O/sq2.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $R/tools/pnp_time_bin 16 4096 2000 > $O/fetch.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $R/tools/pnp_time_bin 16 4096 2000 > $O/write.out 2>&1
grep -h -E "knn_mfma_frames|ransac_" $O/kt/*/*_kernel_stats.csv | cut -d, -f1-4
echo done
