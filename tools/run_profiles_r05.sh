#!/bin/bash
# Round-5 rocprofv3 passes (GPU box, repo root: bash tools/run_profiles_r05.sh [part ...]).  Every --pmc pass is its own run with
# --kernel-trace only.  Parts: line (default bench line + sidecar), cfg3 (headline kernel: stats + PMC for the traffic JSON),
# tile (per-body factor build: stats, MFMA counters, in-kernel phase stamps), multi (the lock-step multi-RHS solve: stats + MFMA busy).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
parts=${@:-"line cfg3 tile multi"}
BENCH="python3 $R/bench.py --steps 5 --warmup 1 --cpu-budget 0 --timestep-steps 0 --other-configs 0"
for p in $parts; do
case $p in
line)
  python3 $R/bench.py --steps 20 --warmup 5 --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench_line.err || echo "default bench failed" ;;
cfg3)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg3_stats -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-budget 0 --timestep-steps 2 --other-configs 0 > $O/cfg3_stats.log 2>&1 || exit 1
  for c in "SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    d=$O/cfg3_pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- $BENCH > $d.log 2>&1 || exit 1
  done ;;
tile)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/tile_stats -- python3 $R/tools/bench_block_factor.py 200 642 wall > $O/tile_stats.log 2>&1 || exit 1
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU"; do
    d=$O/tile_pmc_$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/bench_block_factor.py 200 642 wall > $d.log 2>&1 || exit 1
  done
  (cd $R && python3 tools/tile_phase_profile.py 200 642 1 > $O/tile_phases.txt 2>&1; python3 tools/tile_phase_profile.py 200 642 0 >> $O/tile_phases.txt 2>&1; python3 tools/tile_phase_profile.py 25 642 1 >> $O/tile_phases.txt 2>&1) ;;
multi)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/multi_stats -- python3 $R/tools/bench_multi_rhs.py 200 642 wall 16 > $O/multi_stats.log 2>&1 || exit 1
  for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_INSTS_VALU"; do
    d=$O/multi_pmc_$(echo $c | tr ' ' '_')
    ONLY_MULTI=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/bench_multi_rhs.py 200 642 wall 16 > $d.log 2>&1 || exit 1
  done ;;
esac
echo "part $p done" >> $O/progress.txt
done
cd $R
python3 tools/pmc_summary.py $O/cfg3_pmc_summary.txt $O/cfg3_pmc_* 2>/dev/null
python3 tools/pmc_to_json.py "k_apply_M_sym<true, 2, 4, 0" cfg3 $O/cfg3_pmc.json $O/cfg3_pmc_* 2>/dev/null
python3 tools/pmc_summary.py $O/tile_pmc_summary.txt $O/tile_pmc_* 2>/dev/null
python3 tools/pmc_summary.py $O/multi_pmc_summary.txt $O/multi_pmc_* 2>/dev/null
ls $O | head -40
