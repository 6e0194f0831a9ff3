mkdir -p gpurun_out/r9i
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r9i/probe.jsonl
python tools/eval_loop_probe.py --tag default_a 2>/dev/null | tail -1 >> $O
python tools/eval_loop_probe.py --tag default_b 2>/dev/null | tail -1 >> $O
YM_READ_SYNC=1 python tools/eval_loop_probe.py --tag read_sync 2>/dev/null | tail -1 >> $O
HSA_ENABLE_INTERRUPT=0 python tools/eval_loop_probe.py --tag hsa_no_interrupt 2>/dev/null | tail -1 >> $O
ROC_ACTIVE_WAIT_TIMEOUT=2000 python tools/eval_loop_probe.py --tag active_wait_2ms 2>/dev/null | tail -1 >> $O
GPU_MAX_HW_QUEUES=4 python tools/eval_loop_probe.py --tag four_queues 2>/dev/null | tail -1 >> $O
cat $O
