"""dropin/ makes the reference's harness scripts import the MI355X path without edits: execute their import lines verbatim
(`/root/reference/eval.py:10-15`, `train.py:17-18`, `detect.py:13-17`) in a fresh interpreter with dropin/ in front."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REFERENCE_IMPORT_LINES = textwrap.dedent('''
    from utils.coco import COCODetection, val_collate
    from modules.yolact import Yolact
    from utils import timer
    from utils.output_utils import after_nms, nms
    from utils.common_utils import ProgressBar, MakeJson, APDataObject, prep_metrics, calc_map
    from config import get_config
    from utils.coco import COCODetection, train_collate
    from utils.common_utils import save_best, save_latest
    from utils.coco import COCODetection, detect_collate
    from utils.output_utils import nms, after_nms, draw_img
    from utils.common_utils import ProgressBar
    from utils.augmentations import val_aug
    from config import COLORS
    from utils.box_utils import box_iou, mask_iou, make_anchors
''')

CHECK = textwrap.dedent('''
    import yolact_minimal_amd.modules.yolact as Y, yolact_minimal_amd.utils.output_utils as O, yolact_minimal_amd.config as C
    assert Yolact is Y.Yolact and nms is O.nms and after_nms is O.after_nms and get_config is C.get_config
    import utils, modules, config
    assert config is C and utils.timer.counter and modules.yolact is Y
    bar = ProgressBar(10, 4)
    assert len(bar.get_bar(2)) == 10
    timer.reset(); timer.start()
    with timer.counter('forward'):
        pass
    timer.add_batch_time(1.0)
    assert len(timer.get_times(['batch', 'data', 'forward'])) == 3
    print('DROPIN_OK')
''')


def _run(code, cwd):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, 'dropin'), REPO]))
    return subprocess.run([sys.executable, '-c', code], cwd=cwd, env=env, capture_output=True, text=True, timeout=300)


def test_reference_import_lines_run_verbatim(tmp_path):
    r = _run(REFERENCE_IMPORT_LINES + CHECK, str(tmp_path))
    assert r.returncode == 0 and 'DROPIN_OK' in r.stdout, r.stderr[-2000:]


def test_launcher_runs_a_script_inside_a_checkout_unmodified(tmp_path):
    """The real situation: the script sits in a checkout next to its OWN `modules/`, `utils/`, `config.py` (which Python would
    pick first).  Through dropin/run.py the hot-path imports resolve to yolact_minimal_amd, other `utils.*` to the checkout."""
    for pkg in ('utils', 'modules'):
        (tmp_path / pkg).mkdir()
        (tmp_path / pkg / '__init__.py').write_text('')
    (tmp_path / 'utils' / 'only_in_checkout.py').write_text('VALUE = 41\n')
    (tmp_path / 'utils' / 'output_utils.py').write_text('raise RuntimeError("the checkout\'s own output_utils was imported")\n')
    (tmp_path / 'modules' / 'yolact.py').write_text('raise RuntimeError("the checkout\'s own yolact was imported")\n')
    (tmp_path / 'config.py').write_text('raise RuntimeError("the checkout\'s own config was imported")\n')
    (tmp_path / 'eval_like.py').write_text(REFERENCE_IMPORT_LINES + CHECK +
                                           'import sys\nfrom utils.only_in_checkout import VALUE\nprint("FALL", VALUE, sys.argv[1:])\n')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'dropin', 'run.py'), 'eval_like.py', '--weight', 'w.pth'],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'DROPIN_OK' in r.stdout and "FALL 41 ['--weight', 'w.pth']" in r.stdout, r.stderr[-2000:]


def test_launcher_picks_hardware_queues_by_script():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_dropin_run', os.path.join(REPO, 'dropin', 'run.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.hw_queues_for('/x/eval.py') == '8' and mod.hw_queues_for('detect.py') == '8' and mod.hw_queues_for('/x/train.py') is None


def test_launcher_sets_the_host_gc_policy_before_the_script(tmp_path):
    """dropin/run.py::host_gc_policy: the import-time heap is frozen and the young-generation threshold raised before the script
    runs (the `--coco_api` eval loop with device RLE: 234 -> 340 img/s); `YM_DROPIN_GC=0` keeps the interpreter's defaults."""
    (tmp_path / 'eval_gc.py').write_text('import gc\nprint("GC", gc.get_threshold()[0], gc.get_freeze_count() > 10000, gc.isenabled())\n')
    run = [sys.executable, os.path.join(REPO, 'dropin', 'run.py'), 'eval_gc.py']
    r = subprocess.run(run, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'GC 20000 True True' in r.stdout, (r.stdout, r.stderr[-1500:])
    r = subprocess.run(run, cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=dict(os.environ, YM_DROPIN_GC='0'))
    assert r.returncode == 0 and 'GC 700 False True' in r.stdout, (r.stdout, r.stderr[-1500:])
