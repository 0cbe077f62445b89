# PMC counters and kernel trace of a lone small RANSAC call: 100 hypotheses on 500 correspondences
# (ransac_solve3_kernel<4> + ransac_score_kernel).  Output: gpurun_out/prof_r01l
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01l
mkdir -p $O
timeout -k 10 60 $R/tools/pnp_time_bin 1 100 500 | tee $O/time.json
cd /tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $R/tools/pnp_time_bin 1 100 500 > $O/kt.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- $R/tools/pnp_time_bin 1 100 500 > $O/sq1.out 2>&1
timeout -s KILL 60 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM --output-format csv -d $O/sq2 -- $R/tools/pnp_time_bin 1 100 500 > $O/sq2.out 2>&1
echo done
