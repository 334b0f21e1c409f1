#!/bin/bash
# A/B of librbl builds (tools/build_variant.sh <name> <flags>) on the headline product, interleaved, three rounds, one box:
#   tools/build_variant.sh base; tools/build_variant.sh bias100 -mllvm -amdgpu-schedule-metric-bias=100; bash tools/ab_variants.sh
# (end of round 4: the scheduler's occupancy / latency bias 0 and 100 against the default: 20.80-20.92 / 20.83-20.89 / 20.78-20.82 ms -- nothing)
for rnd in 1 2 3; do for v in base bias0 bias100; do
  ms=$(RBL_LIBRARY=$GRAFT_REPO_ROOT/rigid_body_light_amd/build/variants/librbl_$v.so python bench.py --steps 10 --warmup 2 --cpu-budget 0 --timestep-steps 0 --other-configs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f %s' % (d['ms_per_step'], d['repeats']['ms_per_step']))")
  echo "round $rnd $v: $ms"
done; done
