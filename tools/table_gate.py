"""Gate between a tuning tool and yolact_minimal_amd/tuned_gfx950.json.

A row of the table changes which kernel computes a layer (tile, K split, wave / persistent / LDS-tiled kernel, tail split), i.e. the
fp32 summation order of that layer.  Rows a tuner proposes are therefore merged only after the plan that READS them has reproduced
the reference's outputs: the 544 px digest tests of tests/test_gpu_forward.py (goldens from the real reference, both plan modes:
`latency` rows and the `_tp` rows behind bench.py's `value`) are run against the CANDIDATE table (YM_TUNED_PATH) in a subprocess,
and a failure leaves the committed table untouched.
"""
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GATE_TESTS = ['tests/test_gpu_forward.py::test_forward_544_digest',
              'tests/test_gpu_forward.py::test_forward_544_bs8_digest_under_the_tuned_plan']


def run_digest_tests(candidate_path, extra_tests=()):
    """pytest on the reference-digest tests with the candidate table in place of the committed one; returns the exit code."""
    env = dict(os.environ, YM_TUNED_PATH=candidate_path)
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', *GATE_TESTS, *extra_tests]
    return subprocess.call(cmd, cwd=REPO, env=env)


class GateRefused(RuntimeError):
    pass


def merge_rows(rows, table_path, runner=run_digest_tests, extra_tests=()):
    """Merge `rows` ({key: row}) into the table at `table_path` iff the digest tests pass on the merged candidate.
    Raises GateRefused (table untouched) otherwise.  Returns the merged table."""
    with open(table_path) as f:
        table = json.load(f)
    if all(table.get(k) == v for k, v in rows.items()):
        return table                                     # nothing would change
    cand = dict(table)
    cand.update(rows)
    fd, tmp = tempfile.mkstemp(suffix='.json', prefix='tuned_candidate_')
    try:
        with os.fdopen(fd, 'w') as f:
            json.dump(cand, f, indent=0, sort_keys=True)
        rc = runner(tmp, extra_tests) if extra_tests else runner(tmp)
        if rc != 0:
            raise GateRefused(f'the 544 px reference digests failed (exit {rc}) under the candidate table: {sorted(rows)} NOT written '
                              f'to {table_path}')
        os.replace(tmp, table_path)
        tmp = None
    finally:
        if tmp is not None and os.path.exists(tmp):
            os.unlink(tmp)
    return cand
