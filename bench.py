#!/usr/bin/env python
"""Headline benchmark: LiDAR-frames/s of one training step of the hot path (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over `--batch` (default 4: the reference's train.batch_size) synthetic 5-frame x 160k-point sequences per
GPU (BASELINE config c3 shape): GPU voxelisation + collate layout -> MotionNet forward -> FuseLoss -> backward into a flat gradient buffer
(N > 1: bucketed all-reduce overlapped with backward) -> non-finite check -> grad clip -> Adam step (pdist.DataParallelStep, the cadence of
libs/trainer.py:165-237).  Default compute mode `mixed`: the fp32-accurate forward (outputs within 1e-3 of the reference) with a bf16 gradient
graph; the all-bf16 step is timed beside it (`bf16` object).  Inputs (raw points, labels, poses) are resident in HBM before the timed region.
Scenes are independent, so ranks run different scenes and the only collective is the gradient all-reduce: weak scaling.

Host side: the process is bound to the CPUs of one last-level-cache domain (`--cpu-affinity l3`, a different domain per local rank; restored
for the CPU baseline), and the stepper decides over its first five steps -- inside the warm-up at the default `--warmup 5` -- whether it
issues its early backward from a helper thread (`config.early_backward_thread` reports the measurement).  The line carries the spread of
the timed steps beside their mean (`ms_per_step_p10 / _p50 / _p90`: device time between the steps' first launches).

Extra objects on the JSON line: `roofline` for the pillar-scatter kernel (the kernel BASELINE.json's north_star names), timed live with HIP
events on the launch stream inside the timed region; `cpu_baseline`: the same step on the host cores with the oracle-backed CPU backend (rank 0,
N = 1 only); `bf16` / `matched_accuracy`: the same step in the other compute mode (N = 1 only); `configs`: the other BASELINE configurations.
"""
import argparse
import json
import os
import sys
import time

# Kernel arguments written straight to device memory by the runtime: 2-3 us less per launch on this stack, and a training step is ~1 300 launches
# with long chains of small dependent kernels (mixed, four sequences: median 30.9 -> 30.5 ms over six interleaved pairs, tools/gpu_r04_kernarg.sh).
# Read by the HIP runtime when it initialises: set before the first GPU call; an explicit setting in the environment wins.
os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
# This pool's host driver only supports dmabuf IPC: with the legacy mode RCCL (and any sharing of device memory between processes) fails with
# `hipIpcGetMemHandle: invalid argument`.  The image exports it; a launcher that builds its own environment may not.  An explicit setting wins.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')



def _descendants(pid):
    """pids of every live descendant of `pid` (torch.distributed.run gives each rank a session of its own: walk /proc by parent pid)."""
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


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE / TORCHELASTIC_RUN_ID in the environment): this process -- which has
    imported nothing that touches a GPU -- starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py <same arguments>`
    as a CHILD (never an exec), relays its output (rank 0's JSON line on stdout) and its exit status, and kills the whole tree (the ranks live in sessions of their
    own) when PCACC_BENCH_LAUNCH_TIMEOUT seconds (default 1800) pass or this process is told to stop.  Returns the exit status, or None when this process
    is to run the bench itself (N = 1, or already one rank of a launch)."""
    n = 1
    for i, a in enumerate(argv):
        if a == '--gpus' and i + 1 < len(argv):
            n = argv[i + 1]
        elif a.startswith('--gpus='):
            n = a.split('=', 1)[1]
    try:
        n = int(n)
    except ValueError:
        return None                                             # argparse reports it
    if n <= 1 or 'WORLD_SIZE' in os.environ or 'TORCHELASTIC_RUN_ID' in os.environ or 'RANK' in os.environ:
        return None
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    try:
        limit = float(os.environ.get('PCACC_BENCH_LAUNCH_TIMEOUT', '1800'))
    except ValueError:
        limit = 1800.0
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + list(argv)
    print('[bench] --gpus %d without a launcher: starting %s' % (n, ' '.join(cmd[1:9])), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, start_new_session=True)      # stdout / stderr inherited: rank 0's line is this process's line

    def kill_tree(*_):
        tree = _descendants(proc.pid)
        try:
            proc.send_signal(signal.SIGTERM)                    # the launcher's handler ends its ranks
            proc.wait(timeout=10)
        except (OSError, subprocess.TimeoutExpired):
            pass
        for p in tree + _descendants(proc.pid) + [proc.pid]:
            try:
                os.kill(p, signal.SIGKILL)
            except OSError:
                pass
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, lambda *a: (kill_tree(), os._exit(128 + a[0])))
    try:
        rc = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print('[bench] the %d-rank launch did not finish within %.0f s: killing its process tree' % (n, limit), file=sys.stderr, flush=True)
        kill_tree()
        return 124
    finally:
        for p in _descendants(proc.pid):                        # stragglers (none after a clean exit)
            try:
                os.kill(p, signal.SIGKILL)
            except OSError:
                pass
    return rc if rc >= 0 else 128 - rc


if __name__ == '__main__':
    _rc = self_launch(sys.argv[1:])                             # before torch is imported: the parent of a launch makes no GPU call at all
    if _rc is not None:
        sys.exit(_rc)

_PARENT_AT_IMPORT = os.getppid()

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pcaccumulation_amd import distributed as pdist  # noqa: E402
from pcaccumulation_amd import native  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.loss import FuseLoss  # noqa: E402
from pcaccumulation_amd.motionnet import MotionNet  # noqa: E402
from pcaccumulation_amd.pipeline import DeviceBatcher, sample_to_device  # noqa: E402
from pcaccumulation_amd.synthetic import make_sequence  # noqa: E402

T_FRAMES = 5
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)


def build(cfg, device, seed=0):
    torch.manual_seed(seed)
    model = MotionNet(cfg).to(device)
    if device.type == 'cuda':
        model.channels_last_()
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg['Adam']['learning_rate'], weight_decay=cfg['Adam']['weight_decay'],
                           fused=(device.type == 'cuda'))
    return model, opt, FuseLoss(cfg['loss'])


class BatchFeed(object):
    """One batch ahead: next() hands out the batch whose voxelisation was queued (on a side stream) during the previous step,
    prefetch() queues the following one -- every step still voxelises and collates exactly one batch.  train_step calls
    prefetch() while the forward waits for its host sync (after the pillar-scatter launch the roofline object times has been issued; the
    side stream's kernels then run into the host-bound stretch behind the sync), and again after the forward for a forward without that hook
    (a no-op when the batch is already queued)."""

    def __init__(self, batcher, batch_of, ahead, prepare=None):
        self.batcher, self.batch_of, self.ahead, self.i, self.prepare = batcher, batch_of, ahead, 0, prepare
        self.pending = batcher.start(batch_of(0), side_stream=True) if ahead else None

    def next(self):
        if not self.ahead:
            inp = self.batcher(self.batch_of(self.i))
            self.i += 1
            return inp
        inp = self.batcher.finish(self.pending)
        self.pending = None
        self.i += 1
        return inp

    def prefetch(self):
        if self.ahead and self.pending is None:
            self.pending = self.batcher.start(self.batch_of(self.i), side_stream=True)

    def prepare_next(self):
        """After the step's launches are issued: collate the prefetched batch and build what the forward derives from the batch alone
        (pillar index, CSR, per-pillar means, point features) on the prefetch stream, under the rest of the current step."""
        if self.ahead and self.pending is not None and self.prepare is not None:
            self.batcher.finish_early(self.pending, self.prepare)


class Watchdog(object):
    """No progress for S seconds -> every thread's Python stack on stderr and the process exits with status 1 (a fresh exit of THIS process; the
    launcher then ends the other ranks).  The timer is faulthandler's C thread: it fires although the main thread sits in a runtime call that holds the
    interpreter lock.  arm() restarts it -- called at every phase boundary and after every step -- so S bounds one step, not the run.
    PCACC_HANG_DUMP = S (seconds; 0 = off).  Default: 120 s with a process group (a rank that waits for a peer that died or never arrives must not sit
    there until somebody's 10-minute limit), off for a single process; the first interval (imports of the other ranks, rendezvous, library load, the
    first step's code-object load and library find) gets 3 S."""

    def __init__(self):
        v = os.environ.get('PCACC_HANG_DUMP')
        if v is None:
            v = '120' if int(os.environ.get('WORLD_SIZE', '1')) > 1 else '0'
        try:
            self.seconds = float(v)
        except ValueError:
            self.seconds = 0.0
        self.trace = bool(os.environ.get('PCACC_BENCH_TRACE'))
        self.t0 = time.time()

    def arm(self, what=None, first=False):
        if self.trace and what:
            print('[bench rank %s +%.1fs] %s' % (os.environ.get('RANK', '0'), time.time() - self.t0, what), file=sys.stderr, flush=True)
        if self.seconds > 0:
            import faulthandler
            faulthandler.dump_traceback_later(self.seconds * (3 if first else 1), exit=True)

    def off(self):
        if self.seconds > 0:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()


watchdog = Watchdog()


def die_with_launcher():
    """A rank started by torch.distributed.run lives in its own session (elastic's subprocess handler): killing the launcher -- a test's or a driver's
    timeout -- would leave the ranks behind, holding the GPU.  Ask the kernel to SIGKILL this process when its parent goes (prctl PR_SET_PDEATHSIG)."""
    if 'TORCHELASTIC_RUN_ID' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return
    try:
        import ctypes
        import signal
        ctypes.CDLL('libc.so.6', use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)
        # the launcher went between fork and prctl: the parent CHANGED since this module was imported.  (A parent that was pid 1 all along is a launcher that
        # is the container's entry point -- `python -m torch.distributed.run` as PID 1 under docker / k8s -- and very much alive: ADVICE round 5.)
        if os.getppid() != _PARENT_AT_IMPORT:
            os._exit(1)
    except (OSError, AttributeError):
        pass


def bring_up_in_turn(device, local_rank, n_dev):
    """Ranks that SHARE a device (more local ranks than GPUs: the gloo test configuration of a one-GPU box) create their GPU context, load
    libpcacc_hip.so's code object and run their first kernels ONE AFTER THE OTHER (an exclusive file lock per job and device).  A precaution, not a
    diagnosis: both round-4 hangs of the two-rank launch were the first GPU work of a fresh box with two processes initialising at once, and a later
    process could not initialise the device either while they sat there -- which points below this code; 40 fresh-box launches with stack dumps armed
    did not reproduce it (profiles/r05_2rank_hang.txt).  One rank per device (production) takes no lock."""
    try:
        local_world = int(os.environ.get('LOCAL_WORLD_SIZE', '1'))
    except ValueError:
        local_world = 1
    if local_world <= n_dev:
        return False
    import fcntl
    import tempfile
    path = os.path.join(tempfile.gettempdir(), 'pcacc_bringup_%s_%d.lock' % (os.environ.get('MASTER_PORT', '0'), device.index or 0))
    with open(path, 'w') as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            torch.cuda.init()
            x = torch.zeros(1 << 16, device=device)
            native.lib()
            float(native.absmax256(x).max())                      # a kernel of the library: its code object is loaded and has run
            torch.cuda.synchronize()
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)
    return True


def train_step(stepper, batcher, scenes):
    """One micro-step of the reference's loop (libs/trainer.py:165-237) through pdist.DataParallelStep: voxelise + collate, forward,
    FuseLoss, backward into the flat gradient buffer (bucketed all-reduce overlapped with backward when N > 1), and -- every
    `iter_size` micro-steps -- agreement across ranks, non-finite check, clip, Adam, zero."""
    feed = isinstance(scenes, BatchFeed)
    inp = scenes.next() if feed else batcher(scenes)
    early = feed and os.environ.get('PCACC_PREFETCH_AT', 'sync') == 'sync'       # 'forward': after the forward's launches, as rounds 2-4 did (A/B)
    stats = stepper(inp, before_sync=scenes.prefetch if early else None, after_forward=scenes.prefetch if feed else None,
                    after_backward=scenes.prepare_next if feed else None)
    if stats is not None and hasattr(stats, 'resolve'):
        stats.resolve()                 # the metrics the reference reads with .item(): on the host before the step counts as done
    return stats


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2])


def cpu_baseline(cfg, pts_per_frame, budget_s=170.0):
    """The same workload on the host cores: product host code + oracle CPU backend (kind 'port'), fp32, one sequence, all host
    threads.  SURVEY 8d: warm second call, 5 repeats, median -- bounded by `budget_s` of CPU work, so the (3x heavier) train step
    gets as many warm repeats as fit (at least one) and the eval forward, the figure the survey timed the reference itself at
    (1.93 frames/s on 8 cores), gets the full five."""
    from oracle import cpu_backend
    import oracle
    oracle.lib()
    cpu_backend.install()
    cfg = json.loads(json.dumps(cfg))
    cfg['misc']['compute_dtype'] = 'fp32'
    dev = torch.device('cpu')
    threads = torch.get_num_threads()
    model, opt, loss_fn = build(cfg, dev)
    batcher = DeviceBatcher(cfg)
    batcher.voxeliser.device = dev
    scene = sample_to_device(make_sequence(999, T_FRAMES, pts_per_frame, cfg), dev)
    t_start = time.time()

    def fwd():
        t0 = time.time()
        torch.manual_seed(0)
        with torch.no_grad():
            model(batcher([scene]))
        return time.time() - t0
    model.eval()
    fwd()                                                              # cold call (allocator, oneDNN primitive caches)
    # thread count: all hardware threads is rarely the fastest setting for this mix of mid-sized convolutions and row-wise ops
    # (128 threads on the GPU box's EPYC host ran the forward 2.5x slower than 8 cores of the build container); probe once
    cores = os.cpu_count() or threads
    best = (None, float('inf'))
    for n in sorted({min(c, cores) for c in (8, 16, 32)}):             # three probes (all threads of the GPU box's host ran 2.5x slower than 16)
        torch.set_num_threads(n)
        os.environ['OMP_NUM_THREADS'] = str(n)
        t = fwd()
        if t < best[1]:
            best = (n, t)
    threads = best[0]
    torch.set_num_threads(threads)
    try:
        import ctypes
        ctypes.CDLL('libgomp.so.1').omp_set_num_threads(threads)       # the twin's OpenMP runtime
    except OSError:
        pass
    f_times = [fwd() for _ in range(3)]
    model.train()
    stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=cfg['train']['grad_clip'], catch=False)

    def step():
        t0 = time.time()
        torch.manual_seed(0)
        train_step(stepper, batcher, [scene])
        return time.time() - t0
    cold = step()
    s_times = []
    while len(s_times) < 5 and (not s_times or time.time() - t_start + _median(s_times) < budget_s):
        s_times.append(step())
    dt, fdt = _median(s_times), _median(f_times)
    return {'value': T_FRAMES / dt, 'unit': 'LiDAR-frames/s', 'cores': threads, 'threads_used': threads, 'host_threads_available': cores, 'kind': 'port',
            'sample': 'train step (fwd + loss + bwd + Adam) on one %dx%d-point sequence, fp32: warm, median of %d (%.2f s; cold first step '
                      '%.2f s)' % (T_FRAMES, pts_per_frame, len(s_times), dt, cold),
            'forward_only': {'value': T_FRAMES / fdt, 'unit': 'LiDAR-frames/s',
                             'sample': 'eval forward on the same sequence: warm, median of 3 (%.2f s)' % fdt}}


def step_model(stepper, batcher, feed):
    """One extra, untimed training step with every call into libpcacc_hip.so instrumented: algorithmic bytes (every tensor operand and
    result once -- tools/native_call_table.py's model) and algorithmic FLOPs of the GEMM-shaped calls (convolutions, row-linear layers,
    their weight gradients: 2 x multiply-adds of the mathematical operation, whatever the kernel spends on it -- the fp32x3 kernels
    issue 3 MFMAs per product).  Torch / library kernels are not in the model (their share of GPU time: profiles/)."""
    import types
    tot = {'bytes': 0.0, 'flops': 0.0, 'calls': 0}

    def tensors(obj):
        if torch.is_tensor(obj):
            yield obj
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                yield from tensors(o)

    def flops_of(name, a, k):
        try:
            if name in ('conv3x3', 'conv3x3_split'):
                x, wp = a[0], a[1]
                wp = wp[0] if isinstance(wp, (tuple, list)) else wp
                taps, co, ci = wp.shape[-3], wp.shape[-2], wp.shape[-1]
                return 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * ci * co * taps
            if name in ('conv3x3_wgrad', 'conv3x3_wgrad_deep', 'conv3x3_wgrad_split'):
                dy, x = a[0], a[1]
                return 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * x.shape[3] * 9
            if name in ('rows_linear', 'rows_linear_split'):
                x, w = a[0], (a[2] if name == 'rows_linear_split' else a[1])
                return 2.0 * x.shape[0] * w.shape[0] * w.shape[1]
            if name in ('rows_linear_cat', 'rows_linear_cat_split'):
                w = a[5] if name.endswith('split') else a[3]
                return 2.0 * a[0].shape[0] * w.shape[0] * w.shape[1]
            if name in ('rows_linear_cat_backward', 'rows_linear_cat_backward_split'):
                w = a[2] if name.endswith('split') else a[1]
                return 2.0 * a[0].shape[0] * w.shape[0] * w.shape[1]
            if name in ('rows_wgrad', 'rows_wgrad_split'):
                dy, x = a[0], (a[2] if name == 'rows_wgrad_split' else a[1])
                return 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1]
            if name in ('rows_wgrad_cat', 'rows_wgrad_cat_split'):
                dy = a[0]
                kk = (a[2].shape[1] + a[4].shape[1]) if name.endswith('split') else (a[1].shape[1] + a[2].shape[1])
                return 2.0 * dy.shape[0] * dy.shape[1] * kk
            if name in ('pfn_block_forward', 'pfn_block_split_forward'):
                return 2.0 * a[0].shape[0] * (64 * 32 * 2 + 32 * 32)
            if name == 'pfn_block_backward':
                return 2.0 * a[0].shape[0] * (64 * 32 * 2 + 32 * 32) * 2
            if name == 'pfn_block_split_dgrad':                      # the data gradients; its three weight gradients are rows_wgrad*_split calls
                return 2.0 * a[0].shape[0] * (64 * 32 * 2 + 32 * 32)
            if name in ('chamfer_forward',):
                return 8.0 * a[0].shape[0] * a[1].shape[0] * 2
        except Exception:
            return 0.0
        return 0.0

    def wrap(name, fn):
        def w(*a, **k):
            r = fn(*a, **k)
            ins = list(tensors(a)) + list(tensors(list(k.values())))
            tot['bytes'] += sum(t.numel() * t.element_size() for t in ins + list(tensors(r)))
            tot['flops'] += flops_of(name, a, k)
            tot['calls'] += 1
            return r
        return w
    saved = {}
    for name, fn in list(vars(native).items()):
        if isinstance(fn, types.FunctionType) and not name.startswith('_') and name not in ('lib', 'upload_small') and not name.endswith('_supported'):
            saved[name] = fn
            setattr(native, name, wrap(name, fn))
    try:
        train_step(stepper, batcher, feed)
        torch.cuda.synchronize()
    finally:
        for name, fn in saved.items():
            setattr(native, name, fn)
    return tot


def config_throughput(kind, name, dataset, T, ppf, batch, mode, steps, warmup, device, points='uniform'):
    """Throughput of one BASELINE.json configuration: kind 'train' = this file's training step (voxelise + collate, forward, FuseLoss, backward,
    clip, fused Adam through DataParallelStep, next batch prefetched), kind 'eval' = voxelise + collate + MotionNet forward under no_grad."""
    cfg = default_config(dataset, 'train' if kind == 'train' else 'val', n_sweeps=T)
    cfg['misc']['compute_dtype'] = mode
    cfg['pose_estimation']['kpt_sampler'] = 'device'
    batcher = DeviceBatcher(cfg)
    if kind == 'train':
        model, opt, loss_fn = build(cfg, device)
        scenes = [sample_to_device(make_sequence(100 + i, T, ppf, cfg, mode='lidar_scan' if points == 'lidar' else 'uniform'), device) for i in range(2 * batch)]
        batch_of = lambda i: [scenes[(i * batch + j) % len(scenes)] for j in range(batch)]
        stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=1, grad_clip=cfg['train']['grad_clip'])
        feed = BatchFeed(batcher, batch_of, True)
        run = lambda: train_step(stepper, batcher, feed)
    else:
        torch.manual_seed(0)
        model = MotionNet(cfg).to(device).channels_last_().eval()
        scenes = [sample_to_device(make_sequence(100 + i, T, ppf, cfg, mode='lidar_scan' if points == 'lidar' else 'uniform'), device) for i in range(batch)]

        def run():
            with torch.no_grad():
                model(batcher(scenes))
    for _ in range(max(warmup, 5) if kind == 'train' else warmup):     # a train stepper settles how it issues its early backward over its first five steps
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dict(config=name, kind='train step' if kind == 'train' else 'eval forward (voxelise + MotionNet)', dataset=dataset, frames=T, pts_per_frame=ppf,
                sequences_per_step=batch, dtype=mode, points=points, ms_per_step=round(dt * 1e3, 3), lidar_frames_per_s=round(batch * T / dt, 1), steps=steps, warmup=warmup)


# the other BASELINE.json configurations beside the headline (c3, 4 sequences per step): (kind, name, dataset, frames, points per frame, sequences per step)
OTHER_CONFIGS = [('eval', 'c2', 'nuscene', 5, 80000, 4, 'uniform'), ('train', 'c3', 'waymo', 5, 160000, 1, 'uniform'), ('eval', 'c4', 'waymo', 10, 200000, 2, 'uniform'),
                 ('train', 'c5', 'nuscene', 5, 80000, 4, 'uniform'),
                 # the headline's step on LiDAR-distributed points (1/r range density, 64 beams, scan order; 10 % foreground in 20 boxes): real sweeps are not
                 # uniform, and a kernel that is fast on uniform points can hide a factor behind them (the bilinear backward did for two rounds)
                 ('train', 'c3_lidar', 'waymo', 5, 160000, 4, 'lidar')]


def scatter_flushed(dtype, batch, pillars):
    """The pillar-scatter launch of the roofline object with cold caches: 1 GiB read and written between launches (the in-step launch
    follows the pillar encoder's last layers and finds part of its feature table in the 256 MB Infinity Cache)."""
    dev = torch.device('cuda')
    n_cells, c = batch * T_FRAMES * 288 * 288, 32
    m = int(min(pillars, n_cells))
    rows_dtype = torch.bfloat16 if dtype == 'bf16' else torch.float32
    feats = torch.randn(m, c, device=dev).to(rows_dtype)
    c2p = torch.full((n_cells,), -1, dtype=torch.int32, device=dev)
    c2p[torch.randperm(n_cells, device=dev)[:m].sort().values] = torch.arange(m, dtype=torch.int32, device=dev)     # pillar ids in cell order, as the model numbers them
    flush = torch.zeros(256 * 1024 * 1024, device=dev)
    for _ in range(3):
        native.pillar_scatter(feats, c2p, rows_dtype)
    native.scatter_timer = []
    try:
        for _ in range(12):
            flush.add_(1.0)
            native.pillar_scatter(feats, c2p, rows_dtype)
        torch.cuda.synchronize()
        us = sorted(t[0].elapsed_us() for t in native.scatter_timer)
    finally:
        native.scatter_timer = None
    s = 2 if dtype == 'bf16' else 4
    alg = n_cells * c * s + m * c * s + 4 * m
    med = us[len(us) // 2]
    return {'frac': alg / med / 1e3 / HBM_PEAK_GBPS, 'achieved': alg / med / 1e3, 'median_launch_us': med, 'launches': len(us),
            'note': 'same launch size, 1 GiB streamed between launches (cold L2 / Infinity Cache)'}


def fused_alg_bytes(n_cells, c, m, n):
    """Algorithmic bytes of one pcacc_segment_max_canvas launch: read the point rows (4 c n), their CSR order (4 n), the segment offsets
    (4 (m + 1)) and the cell table (4 n_cells); write the fp32 canvas (4 c n_cells), its bf16 shadow (2 c n_cells) and the winners (4 c m) -- every
    operand once.  SURVEY 8d's pillar-scatter bytes (canvas written incl. zero fill, C s per pillar read, 4 per pillar of index) are the canvas part of it."""
    return 4 * c * n + 4 * n + 4 * (m + 1) + 4 * n_cells + 4 * c * n_cells + 2 * c * n_cells + 4 * c * m


def fused_flushed(batch, pillars, points):
    """The fused pooling + canvas launch of the roofline object with cold caches (see scatter_flushed): synthetic rows of the step's sizes -- `points`
    point rows in `pillars` pillars (every pillar at least one point, the rest uniform), pillars numbered in cell order as the model does."""
    dev = torch.device('cuda')
    n_cells, c = batch * T_FRAMES * 288 * 288, 32
    m = int(min(pillars, n_cells))
    n = int(max(points, m))
    src = torch.randn(n, c, device=dev)
    c2p = torch.full((n_cells,), -1, dtype=torch.int32, device=dev)
    c2p[torch.randperm(n_cells, device=dev)[:m].sort().values] = torch.arange(m, dtype=torch.int32, device=dev)
    p2v = torch.cat([torch.arange(m, device=dev), torch.randint(0, m, (n - m,), device=dev)])[torch.randperm(n, device=dev)].to(torch.int32)
    offs, order = native.csr_build(p2v, m)
    flush = torch.zeros(256 * 1024 * 1024, device=dev)
    for _ in range(3):
        native.segment_max_canvas(src, offs, order, m, c2p)
    native.scatter_timer = []
    try:
        for _ in range(12):
            flush.add_(1.0)
            native.segment_max_canvas(src, offs, order, m, c2p)
        torch.cuda.synchronize()
        us = sorted(t[0].elapsed_us() for t in native.scatter_timer)
    finally:
        native.scatter_timer = None
    alg = fused_alg_bytes(n_cells, c, m, n)
    med = us[len(us) // 2]
    return {'frac': alg / med / 1e3 / HBM_PEAK_GBPS, 'achieved': alg / med / 1e3, 'median_launch_us': med, 'launches': len(us),
            'note': 'same launch size on synthetic rows (uniform pillar sizes), 1 GiB streamed between launches (cold L2 / Infinity Cache)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--pts-per-frame', type=int, default=160000)
    ap.add_argument('--batch', type=int, default=4, help='sequences per GPU per step (reference default: train.batch_size = 4, configs/default.yaml:33)')
    ap.add_argument('--dtype', default='mixed', choices=['mixed', 'mixed2', 'bf16', 'fp32', 'fp32x3'],
                    help="compute mode of the timed step.  'mixed' (default): fp32 tensors and fp32-accurate matrix-core products (scaled fp16 hi / lo halves) in the forward -- the "
                         "mode that matches the reference within north_star's 1e-3 -- with a bf16 gradient graph in the backward; 'bf16': everything bf16 (reported beside the headline "
                         "as `bf16`); 'fp32x3': fp32-accurate products both ways; 'fp32': library fp32 convolutions")
    ap.add_argument('--iter-size', type=int, default=1, help='micro-steps per optimizer step (gradient accumulation; the all-reduce fires on the last one; reference yaml: 2)')
    ap.add_argument('--points', default='uniform', choices=['uniform', 'lidar'], help="synthetic point distribution: 'uniform' (BASELINE.json: synthetic; the headline) or 'lidar' = 1/r range density, 64 beams, scan-ordered within a frame (SURVEY 8d; synthetic.make_sequence(mode='lidar_scan'))")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the short runs of the other BASELINE.json configurations (c2 / c4 eval forward, c3 at one sequence per step, c5 train step) reported as `configs` at N = 1')
    ap.add_argument('--no-step-model', action='store_true', help='skip the instrumented extra step behind roofline_step and the cold-cache scatter launches')
    ap.add_argument('--no-fp32-leg', '--no-second-leg', dest='no_fp32_leg', action='store_true', help='skip the second timing of the same step in the other mode (bf16 beside the mixed headline; fp32x3 beside a bf16 run) that follows at N = 1')
    ap.add_argument('--miopen-find', action='store_true', help='library find mode (torch.backends.cudnn.benchmark) for whatever still goes to the convolution library; off by default: '
                    'no convolution of the default step does, and the headline must not depend on an untracked find-db directory')
    ap.add_argument('--no-miopen-find', action='store_true', help='(default since round 5; kept for the command lines of earlier rounds)')
    ap.add_argument('--pipeline', action='store_true', help='force the staged step (default: staged with a second stream at N = 1, one backward at N > 1)')
    ap.add_argument('--no-pipeline', action='store_true', help='one backward at the end of the forward instead of the early backward of the ego / fb / perm terms (DataParallelStep.pipelined)')
    ap.add_argument('--one-stream', action='store_true', help='motion heads and TubeNet on the main stream behind the early backward instead of beside it on a second stream')
    ap.add_argument('--prepare-ahead', action='store_true', help='build the pillar index / CSR / point features of the next batch on the prefetch stream during the current step (MotionNet.prepare_inputs) instead of inside its own forward; measured neutral: 29.16 vs 29.22 ms over 8 interleaved runs each, sd 0.5')
    ap.add_argument('--cpu-affinity', default='l3', choices=['l3', 'none'], help="'l3' (default): bind the process to the CPUs of one last-level-cache domain (a different one per local rank) for the GPU part of the run -- the step's two host threads then share a cache instead of landing on two sockets in some runs (distributed.bind_to_l3_domain); restored for the CPU baseline")
    ap.add_argument('--no-prefetch', action='store_true', help='voxelise each batch at the start of its own step instead of one step ahead on a side stream')
    args = ap.parse_args()
    die_with_launcher()
    watchdog.arm(first=True)

    # PCACC_DIST_BACKEND=gloo lets the N > 1 path be exercised on a box with fewer GPUs than ranks (ranks then share
    # cuda:0); the driver's multi-GPU runs use the default: nccl = RCCL over xGMI, one rank per GPU.
    backend = os.environ.get('PCACC_DIST_BACKEND') or None
    n_dev = max(torch.cuda.device_count(), 1)
    if backend is None and int(os.environ.get('WORLD_SIZE', '1')) > n_dev:
        raise SystemExit('WORLD_SIZE exceeds the %d visible GPU(s); set PCACC_DIST_BACKEND=gloo to share devices' % n_dev)
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % n_dev)
    affinity_before = None
    if args.cpu_affinity == 'l3':
        affinity_before = pdist.bind_to_l3_domain(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1'))))
    rank, world, local_rank = pdist.init_from_env(backend)
    watchdog.arm('process group up (world %d)' % world, first=True)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d inside a launch of %d rank(s): start `python bench.py --gpus %d` without a launcher (it starts its own ranks) or give '
                         'torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus, args.gpus))
    device = torch.device('cuda', local_rank % n_dev)
    torch.cuda.set_device(device)
    pdist.per_rank_library_cache(rank, world)
    took_turns = bring_up_in_turn(device, local_rank, n_dev)
    native.lib()
    watchdog.arm('libpcacc_hip.so loaded', first=True)
    if args.miopen_find and not args.no_miopen_find:
        torch.backends.cudnn.benchmark = True

    cfg = default_config('waymo', 'train', n_sweeps=T_FRAMES)
    cfg['misc']['compute_dtype'] = args.dtype
    cfg['pose_estimation']['kpt_sampler'] = 'device'          # same uniform key-point subset, drawn on the GPU generator
    model, opt, loss_fn = build(cfg, device)
    batcher = DeviceBatcher(cfg)
    n_scenes = 2 * args.batch
    mode = 'lidar_scan' if args.points == 'lidar' else 'uniform'
    scenes = [sample_to_device(make_sequence(1000 * rank + i, T_FRAMES, args.pts_per_frame, cfg, mode=mode), device) for i in range(n_scenes)]

    def batch_of(i):
        return [scenes[(i * args.batch + j) % n_scenes] for j in range(args.batch)]
    stepper = pdist.DataParallelStep(model, opt, loss_fn, iter_size=args.iter_size, grad_clip=cfg['train']['grad_clip'],
                                     pipelined=False if args.no_pipeline else (True if args.pipeline else None), two_streams=not args.one_stream)

    step_variant = ('staged (early backward of the ego / fb / perm terms)' + (' + second stream' if stepper.side is not None else '')
                    ) if stepper.pipelined else 'one backward'
    torch.manual_seed(1234 + rank)
    feed = BatchFeed(batcher, batch_of, not args.no_prefetch, prepare=model.prepare_inputs if args.prepare_ahead else None)
    watchdog.arm('model, scenes and first prefetch queued', first=True)
    for i in range(args.warmup):
        train_step(stepper, batcher, feed)
        if watchdog.trace:
            torch.cuda.synchronize()
        watchdog.arm('warm-up step %d done' % i)

    native.scatter_timer = []
    pdist.barrier()
    torch.cuda.synchronize()
    watchdog.arm('warm-up done, barrier passed')
    collectives_before = stepper.reducer.collectives
    stepper.reducer.time_exposed = world > 1 or stepper.reducer.active
    gc_log = []
    if os.environ.get('PCACC_BENCH_DEBUG'):                             # which collections of the interpreter's garbage collector fall into the timed steps, and how long they take
        import gc
        # (measured, profiles/r06_gc_pauses.txt: ~1 young-generation collection per step, 0.07 ms per step in total, no full collection in 40 steps -- the collector
        # is not where the slow steps come from; the steps of a run get slower by ~0.03 ms each over 40 steps whatever the collector does)
        def gc_cb(phase, info, _t=[0.0]):
            if phase == 'start':
                _t[0] = time.perf_counter()
            else:
                gc_log.append((info['generation'], (time.perf_counter() - _t[0]) * 1e3, info['collected'], time.perf_counter()))
        gc.callbacks.append(gc_cb)
    t0 = time.perf_counter()
    marks = []                                                          # one event per step boundary on the main stream: the spread of the K steps
    for i in range(args.steps):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks.append(ev)
        train_step(stepper, batcher, feed)
        watchdog.arm()
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    marks.append(ev)
    torch.cuda.synchronize()
    pdist.barrier()
    watchdog.arm('timed steps done')
    collectives_per_step = (stepper.reducer.collectives - collectives_before) / max(args.steps, 1)
    stepper_buckets = list(stepper.reducer.buckets)
    dt_own = time.perf_counter() - t0
    dt = pdist.max_over_ranks(dt_own, device)
    stepper.reducer.time_exposed = False
    exposed = [a.elapsed_time(b) for a, b in stepper.reducer.exposed_events]
    # what makes the first real N > 1 run diagnosable: every rank's own wall clock per step, the part of the gradient all-reduce that the backward did not
    # cover (device time the main stream spent waiting in reducer.finish), and who reported -- gathered from all ranks, printed by rank 0
    mine = {'rank': rank, 'ms_per_step': dt_own / max(args.steps, 1) * 1e3, 'exposed_allreduce_ms': (sum(exposed) / len(exposed)) if exposed else 0.0,
            'device': '%s:%d' % (os.uname().nodename, device.index or 0)}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)
    timer, native.scatter_timer = native.scatter_timer, None
    per_step_in_order = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
    per_step = sorted(per_step_in_order)
    if os.environ.get('PCACC_BENCH_DEBUG'):
        print('per-step ms, in order:', ['%.2f' % t for t in per_step_in_order], file=sys.stderr)
        by_gen = {}
        for g_, ms_, n_, _ in gc_log:
            by_gen.setdefault(g_, []).append(ms_)
        print('garbage collections inside the timed steps: ' + ', '.join('gen %d: %d, %.2f ms in total, longest %.2f ms' % (g_, len(v), sum(v), max(v)) for g_, v in sorted(by_gen.items())),
              '| thresholds', __import__('gc').get_threshold(), file=sys.stderr)

    def pct(q):
        return per_step[min(len(per_step) - 1, int(round(q * (len(per_step) - 1))))] if per_step else None
    if stepper.skipped:
        raise SystemExit('bench: %d optimizer step(s) were skipped (rank %d: %r)' % (stepper.skipped, rank, stepper.last_error))
    thread_choice = getattr(stepper, 'early_thread_choice', None)     # how the stepper settled on issuing its early backward (distributed.DataParallelStep)
    model_tot = flushed = None
    if world == 1:
        watchdog.off()                                                  # the single-process legs below (other modes, other configs, CPU baseline) run minutes by design
    if not args.no_step_model:
        # The instrumented extra step is a whole training step: with a process group it issues the step's collectives (bucketed all-reduce, the
        # agreement flag), so EVERY rank takes it -- rank 0 alone would wait for peers that have already left (an unmatched RCCL all-reduce never
        # returns: the N > 1 launch would end in the watchdog, not in a bench line).  Only rank 0 counts calls; the barrier keeps the ranks together
        # until rank 0's cold-cache launches are done too.
        try:
            if rank == 0:
                model_tot = step_model(stepper, batcher, feed)
            else:
                train_step(stepper, batcher, feed)
                torch.cuda.synchronize()
            watchdog.arm('instrumented step done')
            if rank == 0:
                fused_t = [t for t in timer if t[4] == 'fused']
                if fused_t:
                    flushed = fused_flushed(args.batch, sum(t[3] for t in fused_t) / len(fused_t), sum(t[5] for t in fused_t) / len(fused_t))
                else:
                    flushed = scatter_flushed('bf16' if args.dtype in ('bf16', 'mixed', 'mixed2') else args.dtype, args.batch, sum(t[3] for t in timer) / max(len(timer), 1)) if timer else None
        except Exception as e:                                         # diagnostics must never take the bench line down
            model_tot, flushed = {'error': repr(e)}, None
        if stepper.skipped and rank == 0:
            model_tot = {'error': 'the instrumented step was skipped: %r' % (stepper.last_error,)}
        if world > 1:
            pdist.barrier()
            watchdog.arm('diagnostics done, barrier passed')

    # The same step in the other mode, N = 1 only: beside the default 'mixed' headline (forward at the accuracy that matches the reference within
    # north_star's 1e-3, tests/test_config_parity.py::test_gpu_config_fp32[mixed-*]) the all-bf16 step; beside a bf16 run the fp32x3 step.
    second_leg = None
    second_mode = {'mixed': 'bf16', 'mixed2': 'bf16', 'bf16': 'fp32x3'}.get(args.dtype)
    if world == 1 and second_mode and not args.no_fp32_leg:
        del stepper, model, opt
        torch.cuda.empty_cache()
        cfg2 = json.loads(json.dumps(cfg))
        cfg2['misc']['compute_dtype'] = second_mode
        m2, o2, l2 = build(cfg2, device)
        st2 = pdist.DataParallelStep(m2, o2, l2, iter_size=args.iter_size, grad_clip=cfg['train']['grad_clip'],
                                     pipelined=False if args.no_pipeline else (True if args.pipeline else None), two_streams=not args.one_stream)
        feed2 = BatchFeed(batcher, batch_of, not args.no_prefetch, prepare=m2.prepare_inputs if args.prepare_ahead else None)
        k2 = args.steps
        for i in range(args.warmup):
            train_step(st2, batcher, feed2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(k2):
            train_step(st2, batcher, feed2)
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        notes = {'bf16': 'same step with bf16 tensors and bf16 matrix-core products both ways (BASELINE config c3 names bf16): NOT within 1e-3 of the reference '
                         '(validation-set level on trained weights: EPE 6e-3 m, mos_iou 3e-3; DESIGN.md section 4) -- bounded, not matched',
                 'fp32x3': 'same step with fp32 tensors and fp32-accurate matrix-core products (scaled fp16 hi / lo halves) in the forward AND the backward'}
        second_leg = {'dtype': second_mode, 'value': args.batch * T_FRAMES * k2 / dt2, 'unit': 'LiDAR-frames/s', 'ms_per_step': dt2 / k2 * 1e3,
                      'steps': k2, 'warmup': args.warmup, 'early_backward_thread': getattr(st2, 'early_thread_choice', None), 'note': notes[second_mode]}
        del st2, m2, o2

    if rank == 0:
        frames = world * args.batch * T_FRAMES * args.steps
        # events attached to the pillar-scatter dispatches themselves (pcacc_pillar_scatter_timed -> hipExtLaunchKernel): the
        # kernel's own begin-to-end time, the quantity rocprofv3's kernel trace reports for the same launch
        # 'mixed' fills two canvases per step (fp32 twin from the fp32 rows, bf16 shadow from the bf16 rows): the roofline object is the bf16 fill --
        # the kernel the bf16 mode runs and earlier rounds reported --, the fp32 fill rides along as `roofline.f32_fill`
        fused = [t for t in timer if t[4] == 'fused']            # [r6] 'mixed': the encoder's last pooling writes both canvases itself (pcacc_segment_max_canvas)
        fills16 = [t for t in timer if t[4] == torch.bfloat16]
        fills32 = [t for t in timer if t[4] not in (torch.bfloat16, 'fused')]
        timer_main = fused or fills16 or fills32
        esize = lambda t: 2 if t[4] == torch.bfloat16 else 4
        durs = [t[0].elapsed_us() * 1e-6 for t in timer_main]
        if os.environ.get('PCACC_BENCH_DEBUG'):
            print('scatter launches (us):', ['%.1f' % (d * 1e6) for d in durs], file=sys.stderr)
        # SURVEY 8d 'pillar scatter': write C*s*cells (canvas incl. zero fill) + read C*s*M (feature rows) + read 4*M (index), s = bytes
        # per element of the activation dtype
        alg = [fused_alg_bytes(t[1], t[2], t[3], t[5]) for t in fused] if fused else [t[1] * t[2] * esize(t) + t[3] * t[2] * esize(t) + 4 * t[3] for t in timer_main]
        main_bf16 = bool(fills16) and not fused
        achieved = (sum(alg) / len(alg)) / (sum(durs) / len(durs)) / 1e9 if durs else 0.0
        # HBM traffic of the same kernel from the committed PMC passes (profiles/r02_pmc_scatter_summary.json, taken at this
        # launch's size: 4 sequences): measured bytes / algorithmic bytes, applied to this run's per-launch algorithmic bytes
        traffic = None
        pmc_file = next((f for f in ('r06_pmc_scatter_summary.json', 'r05_pmc_scatter_summary.json', 'r04_pmc_scatter_summary.json', 'r02_pmc_scatter_summary.json') if os.path.exists(os.path.join(ROOT, 'profiles', f))), None)
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', pmc_file)))
            key = 'seg_max_canvas' if fused else ('pillar_scatter_rows16' if main_bf16 else 'pillar_scatter_vec4<0>')
            traffic = pmc[key]['traffic_over_algorithmic'] * (sum(alg) / len(alg)) if alg else None
        except Exception:
            traffic = None
        line = {
            'metric': 'LiDAR-frames/sec (5-frame seq, 160k pts) fwd+bwd', 'value': frames / dt, 'unit': 'LiDAR-frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'ms_per_step_p10': pct(0.1), 'ms_per_step_p50': pct(0.5), 'ms_per_step_p90': pct(0.9), 'ms_per_step_max': per_step[-1] if per_step else None,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'c3 shape: Waymo geometry 288x288x%d, %d pts/frame %s synthetic, %d sequences per GPU per step, '
                                   'train step = GPU voxelise + MotionNet fwd + FuseLoss + bwd + bucketed grad all-reduce (overlapped) + clip + Adam'
                                   % (T_FRAMES, args.pts_per_frame, 'uniform' if args.points == 'uniform' else 'LiDAR-distributed (1/r, 64 beams, scan order)',
                                      args.batch),
                       'frames_per_sequence': T_FRAMES, 'pts_per_frame': args.pts_per_frame, 'sequences_per_gpu': args.batch, 'iter_size': args.iter_size,
                       'points': args.points, 'parallelism': 'dp%d' % world,
                       'kpt_sampler': "device: the ego head's 1024 key points per frame are drawn by pcacc_sample_subsets (keyed Feistel permutation, one "
                                      "launch) instead of the reference's host torch.randperm stream (models/egomotion.py:157) -- same uniform "
                                      "distribution over subsets, different draw; the parity tests use the host stream",
                       'step_variant': step_variant, 'early_backward_thread': thread_choice,
                       'cpu_affinity': ('one L3 domain: %d CPUs' % len(os.sched_getaffinity(0))) if affinity_before is not None else 'unbound',
                       'HIP_FORCE_DEV_KERNARG': os.environ.get('HIP_FORCE_DEV_KERNARG')},
            'distributed': {'world_size': world, 'backend': (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
                            'device': str(device), 'ranks_per_device': max(1, world // n_dev) if world > n_dev else 1, 'gpu_bring_up_in_turn': took_turns,
                            'collectives_per_step': collectives_per_step, 'gradient_buckets': len(stepper_buckets), 'bucket_mb': [round(4e-6 * (e - b), 2) for b, e in stepper_buckets],
                            'ranks_seen': sorted(r['rank'] for r in per_rank if r), 'per_rank_ms_per_step': [round(r['ms_per_step'], 3) for r in per_rank if r],
                            'exposed_allreduce_ms': [round(r['exposed_allreduce_ms'], 3) for r in per_rank if r], 'rank_devices': [r['device'] for r in per_rank if r]},
            'roofline': {'kernel': ('seg_max_canvas_kernel<8> (pillar scatter fused into the encoder\'s last max-pooling: point rows -> fp32 BEV canvas + bf16 shadow + winners, one pass)' if fused else
                                    'pillar_scatter_rows16_k<1, true, true> (BEV canvas fill, bf16 rows -> bf16 canvas, streaming loads / stores)' if main_bf16 else 'pillar_scatter_vec4<0> (BEV canvas fill)'),
                         'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic,
                         'traffic_source': 'PMC FETCH_SIZE x2 + WRITE_SIZE (calibrated on a 128 MiB copy), profiles/%s' % pmc_file,
                         'launches_timed': len(durs), 'avg_launch_us': (sum(durs) / len(durs) * 1e6) if durs else None,
                         'timing': 'HIP events attached to each dispatch (hipExtLaunchKernel start/stop)',
                         'algorithmic_bytes_per_launch': (sum(alg) / len(alg)) if alg else None,
                         'cold_cache': flushed},
        }
        if model_tot is not None and 'error' not in model_tot:
            step_s = dt / args.steps
            line['roofline_step'] = {
                'flops_per_step': model_tot['flops'], 'bytes_per_step': model_tot['bytes'], 'native_calls_per_step': model_tot['calls'],
                'mfma': {'achieved': model_tot['flops'] / step_s / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': model_tot['flops'] / step_s / 1e12 / 2500.0},
                'hbm': {'achieved': model_tot['bytes'] / step_s / 1e9, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': model_tot['bytes'] / step_s / 1e9 / HBM_PEAK_GBPS},
                'model': 'whole step: algorithmic FLOPs of the GEMM-shaped native calls (3x3 / 3x3x3 convolutions, per-point linear layers, their '
                         'weight gradients; 2 x MACs of the operation) and algorithmic bytes of ALL native calls (every tensor operand and result '
                         'once), both divided by the measured step time; torch / library kernels are outside the model'}
        elif model_tot is not None:
            line['roofline_step'] = model_tot
        if fills16 and fills32:
            d32 = [t[0].elapsed_us() * 1e-6 for t in fills32]
            a32 = [t[1] * t[2] * 4 + t[3] * t[2] * 4 + 4 * t[3] for t in fills32]
            ach = (sum(a32) / len(a32)) / (sum(d32) / len(d32)) / 1e9
            line['roofline']['f32_fill'] = {'kernel': 'pillar_scatter_vec4<0> (the fp32 twin canvas of the mixed mode; cached stores: the first convolution reads it next)', 'achieved': ach, 'frac': ach / HBM_PEAK_GBPS,
                                            'avg_launch_us': sum(d32) / len(d32) * 1e6, 'algorithmic_bytes_per_launch': sum(a32) / len(a32), 'launches_timed': len(d32)}
        if second_leg is not None:
            line[second_leg['dtype'] if args.dtype in ('mixed', 'mixed2') else 'matched_accuracy'] = second_leg
        if args.dtype in ('mixed', 'mixed2'):
            line['dtype_note'] = ('mixed = fp32 tensors with fp32-accurate products on the 16-bit matrix cores (scaled fp16 hi / lo halves, 3 MFMAs per product) in the '
                                  'FORWARD: mos_iou / ego errors / EPE within 1e-3 of the reference on c2-c5 + nus11 + c3_lidar '
                                  '(tests/test_config_parity.py::test_gpu_config_fp32[mixed-*]); bf16 tensors and products (fp32 accumulation, fp32 weight '
                                  'gradients) in the BACKWARD of the pillar encoder and the convolution stacks')
        switches = sorted(k for k in os.environ if k.startswith('PCACC_') and k not in ('PCACC_DIST_BACKEND', 'PCACC_BENCH_DEBUG'))
        if switches:
            line['config']['kernel_switches'] = {k: os.environ[k] for k in switches}      # A/B switches in effect (none in a default run)
        if world == 1 and not args.no_configs:
            # the other BASELINE configurations on the same tree, a few steps each: the certified mode ('mixed' for train steps, its forward alone =
            # 'fp32x3' for the eval-only configurations); profiles/r04_bench_configs.jsonl holds the longer runs incl. bf16
            rows = []
            for kind, name, ds, T, ppf, b, pts in OTHER_CONFIGS:
                try:
                    rows.append(config_throughput(kind, name, ds, T, ppf, b, 'mixed' if kind == 'train' else 'fp32x3', 4, 2, device, points=pts))
                except Exception as e:                                 # diagnostics must never take the bench line down
                    rows.append({'config': name, 'error': repr(e)})
                torch.cuda.empty_cache()
            line['configs'] = rows
        if world == 1 and not args.no_cpu_baseline:
            try:
                if affinity_before is not None:
                    pdist.set_affinity_all_threads(affinity_before)  # the CPU leg probes 8 / 16 / 32 threads: give it -- every existing thread, pools included -- the whole host back
                line['cpu_baseline'] = cpu_baseline(cfg, args.pts_per_frame)
            except Exception as e:                                     # the baseline must never take the bench line down
                line['cpu_baseline'] = {'value': None, 'unit': 'LiDAR-frames/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                                        'sample': 'failed: %r' % (e,)}
        print(json.dumps(line))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
