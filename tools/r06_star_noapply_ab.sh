#!/bin/bash
# A/B of round 6's PartitionedNorm backward without k_star_pnb_apply (StarPnBwdArgs::fused == 2) on Amazon-13 bs 8,192,
# whole epochs, full rows, two alternating repeats: MAMDR_STAR_PNB_APPLY=1 keeps the launch (the round-5 chain).
# (The same script measured the two rejected forms of k_star_prep riding in the tower's launch --
#  tools/patches/r06_star_prep_in_tower.patch, profiles/r06_star_prep_rides_ab_*.txt.)
OUT=gpurun_out/${1:-r06_noapply}
mkdir -p $OUT
for rep in 1 2; do
  for mode in 1 0; do
    MAMDR_STAR_PNB_APPLY=$mode timeout 600 python bench.py --workload amazon13 --steps 1 --warmup 1 --cpu-budget 0 --no-targets --no-profile --lanes 0 > $OUT/bench_apply${mode}_rep$rep.json 2> $OUT/bench_apply${mode}_rep$rep.err
    python - <<PY
import json
b=json.loads(open("$OUT/bench_apply${mode}_rep$rep.json").read().strip().splitlines()[-1])
print("k_star_pnb_apply launch=$mode rep $rep: %.1f domain-steps/s, %.2f us/step" % (b["value"], b["us_per_domain_step"]))
PY
  done
done | tee $OUT/ab.txt
