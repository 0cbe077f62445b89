set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01e
mkdir -p $O
timeout -k 10 120 python3 -u tools/frames_time.py 10000 10000 16 | tee $O/frames_time.txt
timeout -k 10 300 python3 -u -m pytest tests/test_matcher_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread -k "frames" 2>&1 | tail -3
cd /tmp
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $R/tools/knn_time 10000 10000 16 > $O/fetch.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $R/tools/knn_time 10000 10000 16 > $O/write.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- $R/tools/knn_time 10000 10000 16 > $O/sq.out 2>&1
tail -1 $O/sq.out
echo done
