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
                   SEGX_DIST_BACKEND='gloo', SEGX_BENCH_SHARE_GPU='1')
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


def test_bench_gpus_2_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus N` with WORLD_SIZE unset must start N ranks itself (here: two gloo ranks sharing the box's
    one GPU) and report n_gpus = the ranks that ran; without the sharing override it must REFUSE rather than print a one-GPU number."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'cfg1', '--steps', '3', '--warmup', '2']
    p = subprocess.run(cmd, env=dict(env, SEGX_DIST_BACKEND='gloo', SEGX_BENCH_SHARE_GPU='1'), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['config']['ranks'] == 2 and res['config']['parallelism'] == 'dp2' and res['value'] > 0
    assert len(lines[0]) < 4096                                # the driver keeps an ~8 KB tail of stdout
    ov = res['config']['overlap']                              # overlap evidence: buckets whose all-reduce was launched from inside backward (last step)
    assert ov['buckets'] >= 1 and 0 <= ov['launched_in_backward'] <= ov['buckets']
    import torch
    if torch.cuda.device_count() < 2:
        q = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT, timeout=300)
        assert q.returncode != 0 and not [l for l in q.stdout.decode().splitlines() if l.startswith('{')]
        assert 'GPU(s) visible' in q.stderr.decode()


def test_rccl_backend_world_of_one_runs_every_collective():
    """VERDICT r01 next #4a: the 'nccl' (= RCCL) branch had never executed.  tools/rccl_world1.py initialises RCCL with ONE rank on the box's
    GPU and drives the bucketed AVG all-reduce (launched from autograd hooks while backward runs), the synchronised-BatchNorm all-gather /
    all-reduce on device tensors and the scalar reduction through it, for a 2-D and a 3-D model; with one rank every collective is the
    identity, so the loss trajectory must equal the collective-free run's (fp32 rounding: the SyncBN path merges statistics with Chan's formula)."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('SEGX_DIST_BACKEND', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'rccl_world1.py')], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    res = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert res['backend'] == 'nccl' and res['scalars'] == [1.0, 1.0, 1.0]
    for k in ('2d', '3d'):
        r = res[k]
        assert r['avg'] is True and r['buckets'] >= 2
        assert r['launched_in_backward'] >= 1, 'no bucket was all-reduced from an autograd hook (overlap path)'
        # step 1: the same arithmetic up to the order of the BatchNorm statistics sums (one process: channel-resident two-pass variance; synchronised: shifted
        # sums + Chan merge); later steps amplify that rounding through the optimizer (the 64 x 64 toy trajectory is not a stable one: loss 0.48 -> 1.04)
        assert abs(r['plain'][0] - r['rccl'][0]) < 5e-5 and max(abs(a - b) for a, b in zip(r['plain'], r['rccl'])) < 2e-3, r
        assert all(v == v for v in r['rccl'])
