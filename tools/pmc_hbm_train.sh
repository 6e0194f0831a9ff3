#!/usr/bin/env bash
# HBM-side bytes per kernel of the training step (two PMC passes, kernel-trace only):  bash tools/pmc_hbm_train.sh r02
set -u
TAG="${1:-rXX}"; R="$(pwd)"; OUT="$R/gpurun_out/profiles_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/train_profile.py --steps 4"
rm -rf "$OUT/raw_tf" "$OUT/raw_tw"
YM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/raw_tf" -o f -- $CMD > /dev/null 2>&1
YM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/raw_tw" -o w -- $CMD > /dev/null 2>&1
f=$(find "$OUT/raw_tf" -name '*.db' | head -1); w=$(find "$OUT/raw_tw" -name '*.db' | head -1)
python $R/tools/pmc_hbm_kernels.py "$f" "$w" "$OUT/${TAG}_pmc_hbm_train_bs8_res101.json" "YM_WGRAD_STREAM=0 tools/train_profile.py --steps 4 (res101_coco 544 bs=8 training)"
rm -rf "$OUT/raw_tf" "$OUT/raw_tw"
