"""bench.py through the launch contract of the driver (`python -m torch.distributed.run ... bench.py --gpus N`) on the one GPU of the test box:
two gloo ranks sharing cuda:0 (the only N > 1 configuration such a box can run), one RCCL rank (backend nccl at world size 1: communicator
set-up, the bucketed all-reduces, the agreement reduce and the staged two-stream step WITH a process group), and the single-process line.
Guards against the collapse found in round 2 (second stream x process group x batch-prefetch stream: 1.4-2.9 s per step instead of ~75 ms,
DESIGN.md section 6) and against hangs.

These tests start process trees.  Every launch runs in its own session with bench.py's watchdog on (PCACC_HANG_DUMP: no step for that long ->
every thread's stack on stderr, exit 1), is given a limit well under the suite's, and on timeout the WHOLE tree is killed (the launcher's
ranks live in sessions of their own: they are found through /proc and killed by pid; bench.py additionally asks the kernel to kill a rank whose
launcher died).  conftest.py collects this file last, so `pytest -x` reaches every parity test before the first process tree is started."""
import json
import os
import signal
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _descendants(pid):
    """pids of every live descendant of `pid` (children sit in other sessions / process groups: walk /proc by parent pid)."""
    parent = {}
    for d in os.listdir('/proc'):
        if d.isdigit():
            try:
                with open('/proc/%s/stat' % d) as f:
                    parent[int(d)] = int(f.read().rsplit(')', 1)[1].split()[1])
            except (OSError, ValueError, IndexError):
                pass
    out, todo = [], [pid]
    while todo:
        p = todo.pop()
        kids = [c for c, pp in parent.items() if pp == p]
        out += kids
        todo += kids
    return out


def run_tree(cmd, env, limit):
    """Run `cmd` in its own session; -> (returncode or None on timeout, stdout, stderr).  Whatever happens, nothing of its tree survives."""
    proc = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    timed_out = False
    try:
        try:
            out, err = proc.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            timed_out = True
            tree = _descendants(proc.pid)
            proc.send_signal(signal.SIGTERM)                    # the launcher's handler ends its ranks
            try:
                out, err = proc.communicate(timeout=10)
            except subprocess.TimeoutExpired:
                for p in tree + [proc.pid]:
                    try:
                        os.kill(p, signal.SIGKILL)
                    except OSError:
                        pass
                out, err = proc.communicate()
    finally:
        for p in _descendants(proc.pid):                        # stragglers (none in a clean exit)
            try:
                os.kill(p, signal.SIGKILL)
            except OSError:
                pass
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
    return (None if timed_out else proc.returncode), out, err


def _launch(nproc, extra, env_extra, limit=170, hang_dump='45'):
    env = dict(os.environ, PCACC_HANG_DUMP=hang_dump, PCACC_BENCH_TRACE='1', **env_extra)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc)] + extra
    t0 = time.time()
    rc, out, err = run_tree(cmd, env, limit)
    assert rc == 0, 'rc %r after %.0f s\n--- stderr tail ---\n%s' % (rc, time.time() - t0, err[-6000:])
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]                          # rank 0 prints, the others do not
    return json.loads(lines[0])


def test_two_gloo_ranks_share_the_gpu():
    d = _launch(2, ['--steps', '3', '--warmup', '2', '--batch', '2', '--no-cpu-baseline', '--no-fp32-leg'], {'PCACC_DIST_BACKEND': 'gloo'})
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp2'
    assert d['value'] > 0 and 'roofline' in d
    # the instrumented extra step behind `roofline_step` is a training step with collectives: both ranks take it (rank 0 alone would sit in an all-reduce
    # no peer answers -- with RCCL for good), and it is a step that ran, not one the stepper skipped
    assert 'error' not in d['roofline_step'] and d['roofline_step']['native_calls_per_step'] > 100, d['roofline_step']
    assert d['distributed']['backend'] == 'gloo' and d['distributed']['ranks_per_device'] == 2 and d['distributed']['gpu_bring_up_in_turn'] is True
    # Ranks SHARING a device keep the plain one-stream step (two processes x (main, side, prefetch) streams oversubscribe the device's
    # hardware queues -- 388 ms per step at 4 queues per process, 2 555 at 8, 62.5 at 2); one rank per device, the
    # production layout, runs the staged two-stream step of N = 1 (distributed.DataParallelStep).
    assert d['config']['step_variant'] == 'one backward', d['config']['step_variant']
    # two sequences per rank, two ranks time-slicing one device + a gloo all-reduce of 44.5 MB through host memory: ~45-55 ms measured; the collapse was 388 - 2 555 ms
    assert d['ms_per_step'] < 150, d['ms_per_step']


def test_one_rccl_rank_runs_the_production_step():
    """backend nccl (= RCCL) at world size 1 under the driver's launcher: the communicator is created, every gradient bucket goes through
    ncclAllReduce (AVG) on the process group's stream, the agreement MIN-reduce runs, and -- one rank per device -- the step is the staged
    two-stream step of N = 1.  The collectives of one rank move no data over xGMI; what this pins is that the production code path executes."""
    d = _launch(1, ['--steps', '4', '--warmup', '7', '--batch', '2', '--no-cpu-baseline', '--no-fp32-leg', '--no-configs'],
                {'PCACC_FORCE_PROCESS_GROUP': '1'})
    assert d['distributed']['backend'] == 'nccl' and d['distributed']['world_size'] == 1 and d['n_gpus'] == 1
    assert d['distributed']['collectives_per_step'] >= 2, d['distributed']                      # gradient buckets + the agreement reduce
    assert d['config']['step_variant'].startswith('staged') and d['config']['step_variant'].endswith('second stream'), d['config']['step_variant']
    assert d['value'] > 0 and d['steps'] == 4
    assert 'error' not in d['roofline_step'] and d['roofline_step']['native_calls_per_step'] > 100, d['roofline_step']      # the instrumented step under RCCL too


def test_single_rank_line_keeps_the_contract():
    """`python bench.py --steps K --warmup W` (one rank, short): ONE JSON line with the contract's keys, the roofline object of the pillar-scatter
    kernel (HBM-bound, fraction = achieved / peak, positive measured duration), the configured step variant, the host-side settings, and EXACTLY
    K timed steps."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '3', '--batch', '2', '--no-cpu-baseline', '--no-fp32-leg', '--no-configs',
           '--no-step-model']
    rc, out, err = run_tree(cmd, dict(os.environ, PCACC_HANG_DUMP='60'), 240)
    assert rc == 0, 'rc %r\n%s' % (rc, err[-4000:])
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 4 and d['warmup'] == 3 and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'mixed' and d['data'] == 'synthetic' and 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 2 * 5 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']          # frames per second of 2 sequences x 5 frames per step
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and r['unit'] == 'GB/s' and 0.05 < r['frac'] < 1.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert d['config']['step_variant'].startswith('staged') and 'cpu_affinity' in d['config'] and 'early_backward_thread' in d['config']


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher around it: the parent -- which imports nothing that touches the GPU before
    it decides -- starts `python -m torch.distributed.run --nproc-per-node 2 bench.py ...` as a child process, relays rank 0's ONE JSON line and the exit
    status.  Two gloo ranks on the one GPU of the test box; the line carries what makes a first real scaling run diagnosable (every rank's own step time,
    the exposed part of the all-reduce, which ranks reported)."""
    env = dict(os.environ, PCACC_DIST_BACKEND='gloo', PCACC_HANG_DUMP='45', PCACC_BENCH_TRACE='1', PCACC_BENCH_LAUNCH_TIMEOUT='160')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'TORCHELASTIC_RUN_ID', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '2', '--no-cpu-baseline', '--no-fp32-leg',
           '--no-step-model']
    t0 = time.time()
    rc, out, err = run_tree(cmd, env, 200)
    assert rc == 0, 'rc %r after %.0f s\n--- stderr tail ---\n%s' % (rc, time.time() - t0, err[-6000:])
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['config']['parallelism'] == 'dp2' and d['value'] > 0
    dist = d['distributed']
    assert dist['backend'] == 'gloo' and dist['ranks_seen'] == [0, 1]
    assert len(dist['per_rank_ms_per_step']) == 2 and all(t > 0 for t in dist['per_rank_ms_per_step'])
    assert len(dist['exposed_allreduce_ms']) == 2 and all(t >= 0 for t in dist['exposed_allreduce_ms'])
    assert d["ms_per_step"] >= max(dist["per_rank_ms_per_step"]) - 1e-3          # the line's time is the maximum over the ranks
