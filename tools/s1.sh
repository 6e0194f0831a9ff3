cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s16
for ws in 1 2 3; do
  YM_WGRAD_STREAMS=$ws timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 20 --train-steps 12 > gpurun_out/s16/t_$ws.txt 2>&1
done
GPU_MAX_HW_QUEUES=4 YM_WGRAD_STREAMS=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 20 --train-steps 12 --inflight 1 > gpurun_out/s16/t_q4_1.txt 2>&1
YM_WGRAD_STREAMS=2 timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 20 --train-steps 8 --train-batch 16 > gpurun_out/s16/t16_2.txt 2>&1
YM_WGRAD_STREAMS=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 20 --train-steps 8 --train-batch 16 > gpurun_out/s16/t16_1.txt 2>&1
