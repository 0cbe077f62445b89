set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r01d
timeout -k 10 300 python3 bench.py > $R/gpurun_out/bench_frames.json 2> $R/gpurun_out/bench_frames.err
tail -c 3000 $R/gpurun_out/bench_frames.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01d/kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_r01d/bench_under_rocprof.json 2> $R/gpurun_out/prof_r01d/kt.err
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01d/fetch -- $R/tools/knn_time 10000 10000 16 > $R/gpurun_out/prof_r01d/fetch.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01d/write -- $R/tools/knn_time 10000 10000 16 > $R/gpurun_out/prof_r01d/write.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/prof_r01d/sq -- $R/tools/knn_time 10000 10000 16 > $R/gpurun_out/prof_r01d/sq.out 2>&1
echo done
