cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
for S in 2 3 4 5 6 8; do
  timeout 300 python bench.py --no-extra --no-train --no-cpu-baseline --steps 300 --inflight $S > gpurun_out/s10/b_$S.txt 2>&1
done
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --no-extra --no-train --no-cpu-baseline --steps 300 --inflight 8 > gpurun_out/s10/b_q16_8.txt 2>&1
