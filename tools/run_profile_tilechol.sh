#!/bin/bash
# rocprofv3 passes of the per-body factor build (tools/bench_block_factor.py): kernel stats, then MFMA / VALU busy counters of k_tile_chol
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_tilechol; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/bench_block_factor.py ${1:-200} ${2:-642} wall > $O/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 tools/bench_block_factor.py ${1:-200} ${2:-642} wall > $O/pmc1.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY --output-format csv -d $O/pmc2 -- python3 tools/bench_block_factor.py ${1:-200} ${2:-642} wall > $O/pmc2.log 2>&1 || echo pmc2 failed
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-200'
