#!/bin/bash
# pick-level outcome of a few full-pipeline configurations on the current arithmetic: bash scratch/r4/fp_variants.sh
R=$GRAFT_REPO_ROOT
cd $R
for it in 112000 128000 160000; do
  python3 full_pipeline.py --micrographs 16 --iterations $it --batch 16 --dtypes f32,mixed16 --agreement f16 --print-interval 16000 --out gpurun_out/fp_var_$it.json > gpurun_out/fp_var_$it.log 2>&1
  python3 - <<PY
import json
d = json.load(open("gpurun_out/fp_var_$it.json"))
for k, r in d["runs"].items():
    e = r["eval"]["picks_vs_planted_centres"]
    print("$it", k, "AP %.3f picks %d" % (e["average_precision"], e["n_picks"]), {t: (round(v["precision"], 3), round(v["recall"], 3)) for t, v in e["at"].items()},
          "train %.0f s" % r["train"]["wall_s"], flush=True)
print("$it agreement", {t: round(v["jaccard"], 3) for t, v in d["pick_agreement_with_fp32"].items() if isinstance(v, dict)}, flush=True)
PY
done
