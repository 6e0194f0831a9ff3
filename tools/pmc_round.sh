#!/usr/bin/env bash
# MFMA-utilisation PMC passes (own runs, kernel-trace only):  bash tools/pmc_round.sh r02
set -u
TAG="${1:-rXX}"; R="$(pwd)"; OUT="$R/gpurun_out/profiles_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
BENCH="python $R/bench.py --no-extra --no-cpu-baseline --no-train --lean --inflight 1"     # (one kernel at a time owns the counters; YM_GRAPH=0: eager launches -- rocprofv3 --pmc crashes on hipGraph replays on this pool)
one() {   # name, env, args...
    local name="$1"; shift; local envs="$1"; shift
    rm -rf "$OUT/raw_$name"
    env $envs rocprofv3 --kernel-trace --pmc $C -d "$OUT/raw_$name" -o p --output-format csv -- "$@" > /dev/null 2>&1
    local f=$(find "$OUT/raw_$name" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python $R/tools/pmc_mfma_summary.py "$f" "$OUT/${TAG}_pmc_mfma_$name.json" "rocprofv3 --pmc $C -- $envs $*"
    rm -rf "$OUT/raw_$name"
}
one infer_bs1_res101 "YM_GRAPH=0" $BENCH --steps 10 --warmup 3
one infer_bs8_res101 "YM_GRAPH=0" $BENCH --batch 8 --steps 6 --warmup 2
one infer_bs8_res101_bf16x3 "YM_GRAPH=0 YM_CONV_MMA=3" $BENCH --batch 8 --steps 6 --warmup 2
# (one stream: with the weight gradients on their side stream two kernels share the counters)
one train_bs8_res101 "YM_WGRAD_STREAM=0" python $R/tools/train_profile.py --steps 4
one train_bs8_swin "YM_WGRAD_STREAM=0" python $R/tools/train_profile.py --cfg swin_tiny_coco --steps 4
