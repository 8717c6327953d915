#!/bin/bash
# round-5: where do the gather kernels wait?  PMC passes (counters only, no tracing flags) over the rotating-operand probe
# (k_tcg_hess_sym with HBM-only operands) and over a short bench run (the cycle's kernels in the loop)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_deep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o p -- python $R/tools/rotating_probe.py 30 > $OUT/p$i.log 2>&1
done
cd $R
python tools/pmc_deep.py gpurun_out/r05_v4_pmc_deep_hess.json k_tcg_hess_sym -- $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6 2>&1 | tee gpurun_out/r05_v4_pmc_deep_hess.txt
tail -3 $OUT/p*.log | cut -c1-200 > gpurun_out/r05_v4_pmc_deep_logs.txt
rm -rf $OUT
