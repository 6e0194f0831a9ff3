export GPU_MAX_HW_QUEUES=8
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', {k:(v['img_s'], v.get('stage_ms',{}).get('metric')) for k,v in d.items() if isinstance(v,dict) and 'img_s' in v})"; }
python bench.py --leg eval_loop 2>/dev/null | show policy
YM_DROPIN_GC=0 python bench.py --leg eval_loop 2>/dev/null | show defaults
YM_GC_OFF=1 python bench.py --leg eval_loop 2>/dev/null | show gc_off
