#!/bin/bash
# Round-3 rocprofv3 passes (run on the GPU box from the repo root: bash tools/run_profiles_r03.sh [part ...]).
# Every --pmc pass is its own run with --kernel-trace only (counter groups that do not fit one pass abort the profiler);
# summaries are made by tools/pmc_summary.py / tools/pmc_to_json.py and copied to profiles/ by hand.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
parts=${@:-"cfg3 dense mrhs steps"}
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --cpu-budget 0 --timestep-steps 0 --other-configs 0"
for p in $parts; do
case $p in
cfg3)   # headline kernel: per-kernel stats of the default line's workload, then the PMC passes
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3_stats -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-budget 0 --timestep-steps 2 --other-configs 0 > $O/cfg3_stats.log 2>&1 || exit 1
  for c in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$O/cfg3_pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- $BENCH > $d.log 2>&1 || exit 1
  done ;;
dense)  # dense path: Cholesky at n = 57 780 (MFMA counters) and the pairwise build at cfg 5 (HBM write)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/chol_stats -- python3 $R/tools/bench_dense.py 30 642 free > $O/chol_stats.log 2>&1 || exit 1
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
    d=$O/chol_pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/bench_dense.py 30 642 free > $d.log 2>&1 || exit 1
  done
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/build_pmc_WRITE_SIZE -- python3 $R/tools/bench_dense.py 20 2562 free --build-only > $O/build_pmc_WRITE_SIZE.log 2>&1 || exit 1 ;;
mrhs)   # 16 right-hand sides on the fp64 matrix cores
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/mrhs_stats -- python3 $R/tools/bench_mrhs.py 200 642 wall 3 > $O/mrhs_stats.log 2>&1 || exit 1
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    d=$O/mrhs_pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/bench_mrhs.py 200 642 wall 3 > $d.log 2>&1 || exit 1
  done ;;
steps)  # whole time steps: converged Brownian step at cfg 3 and at cfg 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/brownian_cfg3_stats -- python3 $R/bench.py --mode timestep --kBT 1 --pc block --rtol 1e-8 --steps 3 --warmup 1 > $O/brownian_cfg3_stats.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/brownian_cfg2_stats -- python3 $R/bench.py --mode timestep --config cfg2 --kBT 1 --pc block --rtol 1e-8 --steps 10 --warmup 2 > $O/brownian_cfg2_stats.log 2>&1 || exit 1 ;;
esac
echo "part $p done" >> $O/progress.txt
done
cd $R
python3 tools/pmc_summary.py $O/cfg3_pmc_summary.txt $O/cfg3_pmc_* 2>/dev/null
python3 tools/pmc_summary.py $O/chol_pmc_summary.txt $O/chol_pmc_* $O/build_pmc_* 2>/dev/null
python3 tools/pmc_summary.py $O/mrhs_pmc_summary.txt $O/mrhs_pmc_* 2>/dev/null
ls $O | head -50
