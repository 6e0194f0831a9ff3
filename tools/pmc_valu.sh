#!/usr/bin/env bash
# Instruction mix per kernel (own PMC pass, kernel-trace only): how many vector instructions other than MFMAs a kernel issues per
# MFMA -- they share the SIMD's issue port.   bash tools/pmc_valu.sh r03
set -u
TAG="${1:-rXX}"; R="$(pwd)"; OUT="$R/gpurun_out/profiles_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
one() {   # name, env, args...
    local name="$1"; shift; local envs="$1"; shift
    rm -rf "/tmp/raw_$name"
    env $envs rocprofv3 --kernel-trace --pmc $C -d "/tmp/raw_$name" -o p --output-format csv -- "$@" > /dev/null 2>&1 < /dev/null
    local f=$(find "/tmp/raw_$name" -name '*counter_collection.csv' | head -n 1)
    if [ -n "$f" ]; then python $R/tools/pmc_valu_summary.py "$f" "$OUT/${TAG}_pmc_valu_$name.txt"; else echo "$name: no counters"; fi
    rm -rf "/tmp/raw_$name"
}
one infer_bs8_res101 "YM_X=0" python $R/bench.py --no-extra --no-cpu-baseline --no-train --lean --inflight 1 --batch 8 --steps 3 --warmup 1
one infer_bs1_res101 "YM_X=0" python $R/bench.py --no-extra --no-cpu-baseline --no-train --lean --inflight 1 --steps 5 --warmup 2
one train_bs8_res101 "YM_WGRAD_STREAM=0" python $R/tools/train_profile.py --steps 2
