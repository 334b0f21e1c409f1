#!/bin/bash
# Round-4 rocprofv3 passes (run on the GPU box from the repo root: bash tools/run_profiles_r04.sh [part ...]).
# Every --pmc pass is its own run with --kernel-trace only; summaries by tools/pmc_summary.py / tools/pmc_to_json.py, copied to
# profiles/ by hand.  Parts: cfg3 (headline kernel: stats + PMC + traffic JSON), mid (cfg 2 product kernels, both forms),
# steps (kernel traces of converged Brownian steps at cfg 2 and cfg 3), lines (the default bench line and the N-rank code path on one rank).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
parts=${@:-"lines cfg3 mid steps"}
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --cpu-budget 0 --timestep-steps 0 --other-configs 0"
for p in $parts; do
case $p in
lines)
  python3 $R/bench.py > $O/bench_line.json 2> $O/bench_line.err || echo "default bench failed"
  python3 $R/bench.py --force-comm --cpu-budget 0 --timestep-steps 3 > $O/bench_force_comm_line.json 2> $O/bench_force_comm_line.err || echo "force-comm bench failed" ;;
cfg3)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3_stats -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-budget 0 --timestep-steps 2 --other-configs 0 > $O/cfg3_stats.log 2>&1 || exit 1
  for c in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" "SQ_BUSY_CU_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$O/cfg3_pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- $BENCH > $d.log 2>&1 || exit 1
  done ;;
mid)
  export ONLY=cfg2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/mid_stats -- python3 $R/tools/bench_midsize.py > $O/mid_stats.log 2>&1 || echo "mid stats failed"
  for c in "SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAVES" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    d=$O/mid_pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/bench_midsize.py > $d.log 2>&1 || echo "pass '$c' failed"
  done
  unset ONLY ;;
steps)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/brownian_cfg3_stats -- python3 $R/bench.py --mode timestep --kBT 1 --pc block --rtol 1e-8 --steps 3 --warmup 1 > $O/brownian_cfg3_stats.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/brownian_cfg2_stats -- python3 $R/bench.py --mode timestep --config cfg2 --kBT 1 --pc block --rtol 1e-8 --steps 10 --warmup 2 > $O/brownian_cfg2_stats.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --output-format csv -d $O/brownian_cfg2_unfused -- python3 $R/bench.py --mode timestep --config cfg2 --kBT 1 --pc block --rtol 1e-8 --steps 10 --warmup 2 --opt fused_krylov=0 --opt sym_wave_units=0 > $O/brownian_cfg2_unfused.log 2>&1 || echo "unfused trace failed" ;;
esac
echo "part $p done" >> $O/progress.txt
done
cd $R
python3 tools/pmc_summary.py $O/cfg3_pmc_summary.txt $O/cfg3_pmc_* 2>/dev/null
python3 tools/pmc_to_json.py "k_apply_M_sym<true, 2, 4, 0" cfg3 $O/cfg3_pmc.json $O/cfg3_pmc_* 2>/dev/null
python3 tools/pmc_summary.py $O/mid_pmc_summary.txt $O/mid_pmc_* $O/mid_stats 2>/dev/null
for f in $(find $O/brownian_cfg2_stats -name "*kernel_trace.csv" | head -1); do python3 tools/step_launches.py $f "cfg 2 converged Brownian step, round 4 (rocprofv3 --kernel-trace)" > $O/cfg2_step_launches.md; done
for f in $(find $O/brownian_cfg2_unfused -name "*kernel_trace.csv" | head -1); do python3 tools/step_launches.py $f "cfg 2 converged Brownian step, round-3 kernels (fused_krylov=0, sym_wave_units=0)" > $O/cfg2_step_launches_unfused.md; done
ls $O | head -60
