import json, os, sys
sys.path.insert(0, '/root/repo')
os.environ['YM_TUNE_TRAIN'] = '1'
from yolact_minimal_amd.engine import tuned_table
t = tuned_table()
for k in [k for k in t if k.startswith('W_')]:
    del t[k]
import runpy
sys.argv = ['autotune_train.py', '--out', 'gpurun_out/tuned_W.json']
runpy.run_path('/root/repo/tools/autotune_train.py', run_name='__main__')
