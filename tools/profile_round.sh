#!/usr/bin/env bash
# Per-round rocprofv3 evidence (run on the GPU box from the repo root):  bash tools/profile_round.sh r02
# Writes kernel-stat tables, steady-state gap analyses and the HBM-side PMC summary under gpurun_out/profiles_<tag>/ ; copy what
# is to be judged into profiles/ (tracked).
set -u
TAG="${1:-rXX}"
R="$(pwd)"
OUT="$R/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {   # name, command...
    local name="$1"; shift
    rm -rf "$OUT/raw_$name"
    rocprofv3 --kernel-trace -d "$OUT/raw_$name" -o "$name" -- "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
    find "$OUT/raw_$name" -name '*.db' | head -1
}
# --inflight 1: one request at a time, so that a kernel's begin..end is its own duration (the default, 4 requests in flight, is
# traced separately below: its kernels overlap)
BENCH="python $R/bench.py --no-extra --no-cpu-baseline --no-train --lean --inflight 1"
BENCH4="python $R/bench.py --no-extra --no-cpu-baseline --no-train --lean"
db=$(run infer_bs1 $BENCH --steps 30 --warmup 5)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_infer_bs1_res101_kernel_stats.md" > /dev/null && python $R/tools/gap_summary.py "$db" 30 k_stem_pool > "$OUT/${TAG}_infer_bs1_res101_gaps.txt"
db=$(run infer_bs1_inflight4 $BENCH4 --steps 60 --warmup 8)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_infer_bs1_inflight4_res101_kernel_stats.md" > /dev/null && python $R/tools/gap_summary.py "$db" 30 k_stem_pool > "$OUT/${TAG}_infer_bs1_inflight4_res101_gaps.txt"
db=$(YM_CONV_MMA=3 run infer_bs8_bf16x3 $BENCH --batch 8 --steps 20 --warmup 5)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_infer_bs8_res101_bf16x3_kernel_stats.md" > /dev/null
db=$(run infer_bs8 $BENCH --batch 8 --steps 20 --warmup 5)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_infer_bs8_res101_kernel_stats.md" > /dev/null
# BASELINE configs 2 and 5: res50_coco / swin_tiny_coco bs=8 inference, one batch at a time
db=$(run infer_bs8_res50 $BENCH --cfg res50_coco --batch 8 --steps 20 --warmup 5)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_infer_bs8_res50_kernel_stats.md" > /dev/null
db=$(run infer_bs8_swin $BENCH --cfg swin_tiny_coco --batch 8 --steps 20 --warmup 5)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_infer_bs8_swin_kernel_stats.md" > /dev/null
db=$(run train python $R/tools/train_profile.py --steps 10)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_train_res101_bs8_kernel_stats.md" > /dev/null && python $R/tools/gap_summary.py "$db" 30 k_sgd > "$OUT/${TAG}_train_res101_bs8_gaps.txt"
# the reference's own training loop (torch DDP + torch.optim through dropin/, timer fences): tools/ref_loop_profile.py
db=$(run ref_loop python $R/tools/ref_loop_profile.py --steps 8)
[ -n "$db" ] && python $R/tools/prof_summary.py "$db" "$OUT/${TAG}_reference_loop_res101_bs8_kernel_stats.md" > /dev/null && python $R/tools/gap_summary.py "$db" 30 k_nchw_to_nhwc4 > "$OUT/${TAG}_reference_loop_res101_bs8_gaps.txt"
# HBM-side bytes per conv launch (separate PMC passes, kernel-trace only)
rm -rf "$OUT/raw_pmc_f" "$OUT/raw_pmc_w"
YM_GRAPH=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/raw_pmc_f" -o f -- $BENCH --steps 20 --warmup 5 > /dev/null 2>&1      # (eager launches: rocprofv3 --pmc crashes on hipGraph replays on this pool)
YM_GRAPH=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/raw_pmc_w" -o w -- $BENCH --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find "$OUT/raw_pmc_f" -name '*.db' | head -1); w=$(find "$OUT/raw_pmc_w" -name '*.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && python $R/tools/pmc_summary.py "$f" "$w" "$OUT/${TAG}_pmc_hbm_infer_bs1_res101.json" "bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-train --lean --inflight 1 (res101_coco 544 bs=1, one request at a time)" > /dev/null
rm -rf "$OUT"/raw_*          # the raw databases are large; only the summaries travel back
ls -la "$OUT"
