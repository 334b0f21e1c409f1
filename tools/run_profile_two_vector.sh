#!/bin/bash
# counter passes over tools/bench_two_vector.py: what the two-vector product does with its issue slots and its LDS next to the one-vector one
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_two_vector; mkdir -p $O
cd $R
python3 tools/bench_two_vector.py > $O/wall.txt 2>&1 || exit 1
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=$O/pmc_$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 tools/bench_two_vector.py > $d.log 2>&1 || echo "pass $c failed"
done
python3 tools/pmc_summary.py $O/pmc_summary.txt $O/pmc_*
grep "k_apply_M_sym" $O/pmc_summary.txt
cat $O/wall.txt | tail -1
