#!/bin/bash
# kernel durations + PMC counters of the mid-size product at cfg 2 (8 100 blobs, free space): the round-3 kernel
# k_apply_M_sym<false,1,1,0> and the wave-unit kernel k_apply_M_symw<false,4> in the same run (tools/bench_midsize.py alternates them)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r04e}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export ONLY=cfg2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid_stats -- python3 $R/tools/bench_midsize.py > $O/mid_stats.log 2>&1 || echo "stats pass failed"
for c in "SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  d=$O/mid_pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/bench_midsize.py > $d.log 2>&1 || echo "pass '$c' failed"
done
cd $R
python3 tools/pmc_summary.py $O/mid_pmc_summary.txt $O/mid_pmc_* $O/mid_stats
grep -v "^#" $O/mid_pmc_summary.txt | grep "k_apply_M_sym\|k_reduce"
