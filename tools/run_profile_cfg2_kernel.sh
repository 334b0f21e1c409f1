#!/bin/bash
# PMC look at the mid-size product kernel (cfg 2: k_apply_M_sym<false,1,1,0>, 8 128 one-tile-pair units): issue share, waves in flight.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${CFG2_OUT:-r03m}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config cfg2 --steps 50 --warmup 5 --cpu-budget 0 --timestep-steps 0 --other-configs 0"
for c in "SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_LEVEL_WAVES SQ_WAIT_INST_LDS"; do
  d=$O/cfg2_pmc_$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- $B > $d.log 2>&1 || echo "pass '$c' failed"
done
cd $R
python3 tools/pmc_summary.py $O/cfg2_pmc_summary.txt $O/cfg2_pmc_*
grep -v "^#" $O/cfg2_pmc_summary.txt | grep "k_apply_M_sym\|k_reduce"
