"""GPU (-m gpu): the multi-rank code path of bench.py (overlapped bucketed gradient all-reduce from autograd hooks, synchronised BatchNorm
with device tensors, max-over-ranks timing) exercised with TWO ranks sharing the one GPU of the box.  RCCL refuses duplicate devices, so
the ranks talk over gloo (SEGX_DIST_BACKEND); everything else -- hooks, collectives on device tensors, the kernels -- is what runs over
RCCL on a multi-GPU node."""
import json
import os
import socket
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_one_gpu():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK='0', WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   SEGX_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'cfg1', '--steps', '3', '--warmup', '2'],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1 and not [l for l in outs[1][0].decode().splitlines() if l.startswith('{')], 'rank 0 alone prints the JSON line'
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['global_batch'] == 4 and res['config']['parallelism'] == 'dp2'
    assert res['value'] > 0 and res['config']['final_loss'] == res['config']['final_loss']
    assert 'cpu_baseline' not in res                       # N = 1 only
