set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01z
mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench_line.json 2> $O/bench.err
timeout -k 10 500 python3 tests/tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.err
python3 -c "
import json
j=json.load(open('$O/bench_line.json')); print(j['value'], j['roofline']['frac'], j['roofline']['traffic'], j['ransac_epnp']['value'])
j=json.loads(open('$O/bench_under_rocprof.json').read().strip().splitlines()[-1]); print('under rocprof kernel_ms', j['roofline']['kernel_ms'])"
grep -h "knn_mfma_frames" $O/kt/*/*_kernel_stats.csv | cut -d, -f2-4
