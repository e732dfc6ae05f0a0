"""The driver keeps an ~8 KB tail of bench.py's stdout: the ONE JSON line must stay far below that (BENCH_r04.json: parsed = null for a 24.9 KB line).
bench.compact_line() is fed the largest committed full result (profiles/r04_w_bench_default.json: by_shape x 12 + attention_gemms x 10 for three
configurations) and stub results with every optional block present."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FULL = os.path.join(ROOT, 'profiles', 'r04_w_bench_default.json')
REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
            'roofline', 'cpu_baseline')


def _full():
    return json.load(open(FULL))


def test_line_from_the_largest_recorded_result_is_short_and_round_trips():
    res = _full()
    assert len(json.dumps(res)) > 20000                      # the input really is the line that broke the r04 record
    s = bench.compact_line(res)
    assert '\n' not in s and len(s) < bench.LINE_LIMIT <= 4096
    line = json.loads(s)
    for k in REQUIRED:
        assert k in line, k
    assert line['value'] == res['value'] and line['ms_per_step'] == res['ms_per_step'] and line['n_gpus'] == 1
    assert line['config']['workload'] == res['config']['workload']
    roof = line['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert roof[k] == res['roofline'][k], k
    assert roof['algorithmic_bytes'] == res['roofline']['algorithmic_bytes_per_launch']
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    rows = roof['attention_gemms']['rows']
    assert 1 <= len(rows) <= 6 and all(len(r) == len(roof['attention_gemms']['cols']) for r in rows)
    assert {(r[0], r[1], r[2]) for r in rows} >= {(4096, 256, 1792), (4096, 1792, 256)}     # QK^T and P.V of cfg2 (A = 256, C = 1792)
    cb = line['cpu_baseline']
    assert cb['value'] == res['cpu_baseline']['value'] and cb['kind'] == 'port' and cb['cores'] == res['cpu_baseline']['cores']
    assert set(cb['s_per_step_median']) == set(res['cpu_baseline']['detail'])
    for cfg in ('cfg4', 'cfg5'):
        b = line['brats'][cfg]
        assert b['value'] == res['brats'][cfg]['value'] and b['roofline']['frac'] == res['brats'][cfg]['roofline']['frac']


def test_line_with_every_optional_block_at_eight_ranks_stays_short():
    res = _full()
    res['n_gpus'] = 8
    res['config'].update(collective_backend='nccl (RCCL)', ranks=8, overlap={'buckets': 12, 'launched_in_backward': 11, 'bucket_mb': 64},
                         eager_cfg1={'ms_per_step': 23.1}, hipgraph_replay={'error': 'rc 1: ' + 'x' * 200})
    blk = copy.deepcopy(res['brats']['cfg5'])
    blk['config']['overlap'] = {'buckets': 3, 'launched_in_backward': 2, 'bucket_mb': 64}
    res['polyp'] = {'cfg3_bs6_per_gpu': copy.deepcopy(blk), 'cfg3_global_bs6': copy.deepcopy(blk)}
    res['cpu_baseline']['sample'] = 'y' * 3000
    s = bench.compact_line(res)
    assert len(s) < bench.LINE_LIMIT
    line = json.loads(s)
    assert line['config']['hipgraph_replay_ms'] == 'error' and line['config']['cfg1_eager_ms'] == 23.1
    assert line['polyp']['cfg3_global_bs6']['overlap']['launched_in_backward'] == 2
    assert len(line['cpu_baseline']['sample']) <= 200


def test_line_drops_optional_parts_rather_than_overflowing():
    res = _full()
    res['brats'] = {('cfg%d' % i): copy.deepcopy(res['brats']['cfg5']) for i in range(40)}
    s = bench.compact_line(res)
    assert len(s) < bench.LINE_LIMIT
    line = json.loads(s)
    assert line['value'] == res['value'] and 'roofline' in line and 'brats' not in line


def test_a_failed_cpu_baseline_and_missing_roofline_still_give_a_line():
    res = {'metric': 'm', 'value': 1.0, 'unit': 'images/s', 'n_gpus': 1, 'steps': 1, 'warmup': 0, 'ms_per_step': 1.0, 'config': {'workload': 'w'}, 'roofline': None,
           'cpu_baseline': {'value': None, 'unit': 'images/s', 'cores': 4, 'kind': 'port', 'sample': 'failed: timeout'}}
    line = json.loads(bench.compact_line(res))
    assert line['roofline'] is None and line['cpu_baseline']['value'] is None


@pytest.mark.parametrize('name', ['r04_w_bench_default.json', 'r04_p_bench_default.json', 'r04_t_bench_default.json'])
def test_every_committed_full_result_of_round_4_fits(name):
    p = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(p):
        pytest.skip(name)
    assert len(bench.compact_line(json.load(open(p)))) < bench.LINE_LIMIT
