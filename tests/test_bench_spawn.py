"""bench.py --gpus N really means N ranks: the spawn command carries N, a node
with fewer devices fails loudly instead of reporting a 1-GPU number, and a WORLD_SIZE that disagrees
with --gpus is an error."""
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_spawn_command_builds_n_workers():
    import bench
    args = types.SimpleNamespace(gpus=8)
    cmd = bench.spawn_command(args, ['--gpus', '8', '--steps', '20', '--warmup', '5'])
    assert cmd[1:3] == ['-m', 'torch.distributed.run']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']
    assert os.path.basename(cmd[cmd.index('--master-port') + 2]) == 'bench.py'


def _run(extra_env, *argv):
    env = dict(os.environ)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=env,
                          capture_output=True, text=True, timeout=300)


def test_more_gpus_than_devices_fails_loudly():
    import torch
    n = torch.cuda.device_count()
    p = _run({}, '--gpus', str(n + 2))
    assert p.returncode != 0
    assert 'needs %d devices' % (n + 2) in p.stderr


def test_world_size_must_agree_with_gpus():
    p = _run({'WORLD_SIZE': '4', 'RANK': '0', 'LOCAL_RANK': '0'}, '--gpus', '2')
    assert p.returncode != 0
    assert 'WORLD_SIZE=4' in p.stderr


def test_stdout_carries_only_the_result_line():
    """The driver parses ONE JSON line from stdout; RCCL prints its version banner to stdout when a
    communicator is created (under torch.distributed.run).  bench.own_stdout() points file descriptor 1 at
    stderr and keeps the original for emit_line()."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.own_stdout(); "
            "print('library noise'); os.write(1, b'raw noise on fd 1\\n'); bench.emit_line('{\"value\": 1}')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"value": 1}\n'
    assert 'library noise' in r.stderr and 'raw noise on fd 1' in r.stderr
