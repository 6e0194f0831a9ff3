#!/usr/bin/env bash
# kernel trace of the training step -> kernel stats + overlap-aware gap analysis (run on the GPU box from the repo root)
set -u
TAG="${1:-rXX}"; R="$(pwd)"; OUT="$R/gpurun_out/profiles_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/raw_train"
rocprofv3 --kernel-trace -d "$OUT/raw_train" -o train -- python $R/tools/train_profile.py --steps 10 > "$OUT/train.stdout" 2> "$OUT/train.stderr"
db=$(find "$OUT/raw_train" -name '*.db' | head -1)
python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_train_res101_bs8_kernel_stats.md" > /dev/null
python $R/tools/gap_summary.py "$db" 30 k_sgd > "$OUT/${TAG}_train_res101_bs8_gaps.txt"
rm -rf "$OUT/raw_train"
cat "$OUT/train.stdout"; head -4 "$OUT/${TAG}_train_res101_bs8_gaps.txt"
