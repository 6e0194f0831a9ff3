set -u
R="$(pwd)"; OUT="$R/gpurun_out/swin_prof"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/raw" -o swin -- python $R/tools/train_profile.py --cfg swin_tiny_coco --steps 8 > "$OUT/stdout.txt" 2> "$OUT/stderr.txt"
db=$(find "$OUT/raw" -name '*.db' | head -1)
python $R/tools/prof_summary.py "$db" "$OUT/r02_train_swin_bs8_kernel_stats.md" > /dev/null
python $R/tools/gap_summary.py "$db" 30 k_adamw > "$OUT/r02_train_swin_bs8_gaps.txt" 2>&1 || true
rm -rf "$OUT/raw"
cat "$OUT/stdout.txt" | tail -2
head -45 "$OUT/r02_train_swin_bs8_kernel_stats.md" | cut -c1-160
