set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01g
mkdir -p $O
timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
cd /tmp
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $R/tools/knn_time 10000 10000 > $O/fetch.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $R/tools/knn_time 10000 10000 > $O/write.out 2>&1
echo done
