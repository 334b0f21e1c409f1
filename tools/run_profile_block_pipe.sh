#!/bin/bash
# rocprofv3 passes of the per-body substitution kernels (tools/bench_block_pipe.py): kernel stats of the A/B run, then the L2's
# memory-side read bytes of the forward (ONLY_MODE=1) and the backward (ONLY_MODE=2) sweep -- the same kernel name, one mode a run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_block_pipe; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/bench_block_pipe.py ${1:-200} ${2:-642} wall > $O/stats.log 2>&1 || exit 1
export ONLY_MODE=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fwd -- python3 tools/bench_block_pipe.py ${1:-200} ${2:-642} wall > $O/pmc_fwd.log 2>&1 || exit 1
export ONLY_MODE=2
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_bwd -- python3 tools/bench_block_pipe.py ${1:-200} ${2:-642} wall > $O/pmc_bwd.log 2>&1 || exit 1
python3 tools/pmc_summary.py $O/pmc_summary.txt $O/pmc_fwd $O/pmc_bwd
grep "k_block_solve" $O/pmc_summary.txt
