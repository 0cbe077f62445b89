# Profiles of the matrix-core matcher at the bench shape (F frames of 10k x 10k, F = $1 or 64):
# kernel trace of bench.py, then HBM-traffic and SQ counter passes on tools/knn_time.
# usage: bash tools/run_prof_mfma.sh [F] [tag]   (raw output under gpurun_out/prof_<tag>_F<F>; tag r01m by default)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F=${1:-64}
TAG=${2:-r01m}
O=$R/gpurun_out/prof_${TAG}_F$F
mkdir -p $O
timeout -k 10 300 python3 bench.py --frames-per-step $F > $O/bench_line.json 2> $O/bench.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --frames-per-step $F --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
echo "kernel trace done"
timeout -s KILL 60 rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -i -E "mfma|MOPS" $O/counters.txt | head -40 > $O/counters_mfma.txt || true
timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $R/tools/knn_time 10000 10000 $F > $O/fetch.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $R/tools/knn_time 10000 10000 $F > $O/write.out 2>&1
echo "traffic passes done"
timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq -- $R/tools/knn_time 10000 10000 $F > $O/sq.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- $R/tools/knn_time 10000 10000 $F > $O/sq2.out 2>&1 || echo "second SQ pass failed"
tail -2 $O/sq.out
python3 - <<PY
import json
j=json.load(open('$O/bench_line.json')); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['kernel_ms'], j['roofline']['prepass_ms'])
PY
echo done
