export GPU_MAX_HW_QUEUES=4
for i in 1 2; do
python bench.py --leg train_reference_loop --train-steps 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('policy', d['ms_per_step'], d['stage_ms'], d['without_timer_fences'], d['inputs_from_host']['ms_per_step'])"
YM_DROPIN_GC=0 python bench.py --leg train_reference_loop --train-steps 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defaults', d['ms_per_step'], d['stage_ms'], d['without_timer_fences'], d['inputs_from_host']['ms_per_step'])"
done
