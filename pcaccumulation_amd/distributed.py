"""Data-parallel glue: one process per GPU, scenes sharded by rank, the gradient all-reduce over RCCL/xGMI (backend "nccl" on
ROCm; "gloo" for the CPU tests) bucketed and overlapped with the tail of backward, fired every `iter_size` micro-steps.

The reference is single-GPU (its only collectives sit in dead code, models/utils/norm.py:8-20), so nothing is translated here;
what is mirrored is the optimizer-step cadence of libs/trainer.py:165-237 (loss / iter_size, gradient check, clip, step, zero).

Design for MotionNet on an 8 x MI355X node:
  * ONE flat fp32 gradient buffer (11.1 M elements, 44.5 MB).  Every parameter's `.grad` is a view into it for the life of
    the reducer, so autograd accumulates straight into the buffer (no copy in, no copy out) and zeroing the gradients is one
    memset.  MotionNet has data-dependent branches (models/motionnet.py:222,243) that leave whole sub-modules without gradients
    on some ranks: their slices simply stay zero, every rank reduces the same bytes.
  * The buffer is laid out in REVERSE parameter order (backward produces gradients roughly last-layer first) and cut into
    buckets of ~8 MB; a bucket's all-reduce is launched (async, on the process group's own stream) from a post-accumulate hook
    as soon as every gradient it expects has been written, the rest at the end of backward.  xGMI is point to point
    (7 links x ~153 GB/s per GPU): a 44.5 MB ring all-reduce is ~0.5 ms per-link bound, a few MB per bucket keeps each launch
    bandwidth- rather than latency-bound while leaving the unet's 31 MB (the last gradients to arrive) in flight under the
    pillar encoder's backward.
  * Collectives must be issued in the same order on every rank although readiness is data dependent: buckets are launched
    strictly in index order, and parameters that cannot receive a gradient in this backward (not reachable from the loss) are
    marked ready up front, so a skipped branch does not hold the earlier buckets back.
  * The reference's loop swallows exceptions per iteration (libs/trainer.py:234-235).  A rank that failed still launches every
    bucket (`flush`) so the collectives match, and a small MIN all-reduce (`agree`) carries the "ok" flag together with the
    per-parameter "somebody produced a gradient" mask: parameters nobody touched keep `.grad = None` for the optimizer step,
    which is what single-process Adam sees (it skips them instead of applying stale momentum).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def _reachable_parameters(loss):
    """ids of the leaf tensors whose AccumulateGrad node is reachable from `loss` (what this backward can write)."""
    seen, found, stack = set(), set(), [loss.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, 'variable', None)
        if var is not None:
            found.add(id(var))
        for nxt, _ in fn.next_functions:
            if nxt is not None and nxt not in seen:
                stack.append(nxt)
    return found


class BucketedGradReducer(object):
    """Mean of the gradients of `params` across ranks: flat buffer, reverse-order buckets, all-reduces overlapped with backward.

        reducer.zero()                          # start of an accumulation window (replaces optimizer.zero_grad)
        for each micro-step of the window:
            reducer.prepare(loss, sync=last)    # sync=True on the window's last micro-step only
            loss.backward()                     # hooks launch ready buckets (sync steps)
        reducer.finish()                        # wait for the collectives (current stream waits; the host does not)
        ok = reducer.agree(ok)                  # all ranks ok?  + per-parameter "any rank has a gradient"
        with reducer.sparse_grads(): clip; optimizer.step()

    With one rank everything degenerates to the flat buffer (still one memset instead of 195 `grad = None`)."""

    def __init__(self, params, bucket_bytes=8 << 20):
        self.params = [p for p in params if p.requires_grad]
        dev, dtype = self.params[0].device, self.params[0].dtype
        order = list(reversed(range(len(self.params))))
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dtype, device=dev)
        self.views = [None] * len(self.params)
        self.bucket_of = [0] * len(self.params)
        self.buckets = []                                    # (start, end) element ranges of the flat buffer
        off = start = 0
        for i in order:
            p = self.params[i]
            if off - start > 0 and (off - start + p.numel()) * p.element_size() > bucket_bytes:
                self.buckets.append((start, off))
                start = off
            self.views[i] = self._view_like(p, off)
            self.bucket_of[i] = len(self.buckets)
            off += p.numel()
        self.buckets.append((start, off))
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.touched = [False] * len(self.params)
        self._expected = [0] * len(self.buckets)
        self._pending = [0] * len(self.buckets)
        self._next = len(self.buckets)                       # nothing to launch until prepare(sync=True)
        self._works = []
        self._sync = False
        self._callback_queued = False
        self._absent = None
        for i, p in enumerate(self.params):
            p.grad = self.views[i]
            p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _view_like(self, p, off):
        """A view of the flat buffer with the shape AND strides of `p` (conv weights are kept channels-last, MotionNet.channels_last_;
        the fused optimizer kernels require gradients in the layout of their parameters)."""
        try:
            if p.is_contiguous():
                return self.flat[off:off + p.numel()].view_as(p)
            if torch.ops.aten.is_non_overlapping_and_dense(p):
                return self.flat.as_strided(p.shape, p.stride(), off)
        except (RuntimeError, AttributeError):
            pass
        return self.flat[off:off + p.numel()].view_as(p)

    # ------------------------------------------------------------------------------------------------
    def _make_hook(self, i):
        def hook(param):
            g = param.grad
            if g is None or g.data_ptr() != self.views[i].data_ptr():     # somebody replaced the view (zero_grad(set_to_none)):
                if g is not None:                                          # fold what autograd wrote back into the buffer
                    self.views[i].add_(g)
                param.grad = self.views[i]
            self.touched[i] = True
            if self._sync:
                if not self._callback_queued:                # the rest of the buckets go out when this backward ends
                    torch.autograd.Variable._execution_engine.queue_callback(self._launch_all)
                    self._callback_queued = True
                b = self.bucket_of[i]
                self._pending[b] -= 1
                if self._pending[b] == 0:
                    self._launch_ready()
        return hook

    def _launch(self, b):
        s, e = self.buckets[b]
        if world_size() > 1:
            avg = dist.get_backend() == 'nccl'
            work = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, async_op=True)
            self._works.append((work, b, avg))

    def _launch_ready(self):
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            self._launch(self._next)
            self._next += 1

    def _launch_all(self):
        while self._next < len(self.buckets):
            self._launch(self._next)
            self._next += 1

    # ------------------------------------------------------------------------------------------------
    def zero(self):
        """Start of an accumulation window: one memset; every .grad is (again) its view."""
        self.flat.zero_()
        for i, p in enumerate(self.params):
            p.grad = self.views[i]
        self.touched = [False] * len(self.params)

    def prepare(self, loss, sync=True):
        """Before `loss.backward()`.  sync=False: accumulate only (a micro-step that is not the window's last)."""
        self._sync = bool(sync)
        self._callback_queued = False
        self._works = []
        if not self._sync:
            self._next = len(self.buckets)
            return
        self._next = 0
        reach = _reachable_parameters(loss) if (loss is not None and world_size() > 1) else None
        self._pending = [0] * len(self.buckets)
        for i, p in enumerate(self.params):
            if reach is None or id(p) in reach:
                self._pending[self.bucket_of[i]] += 1
        self._launch_ready()                                  # leading buckets nobody will write (skipped branches)

    def flush(self):
        """A rank whose forward / backward raised still issues every collective of the step (in order), so the others do not hang."""
        if self._sync:
            self._launch_all()

    def finish(self):
        """After backward: every bucket is out; the current stream waits for them (no host block with RCCL)."""
        if not self._sync:
            return
        self._launch_all()
        ws = world_size()
        for work, b, averaged in self._works:
            work.wait()
            if not averaged:
                s, e = self.buckets[b]
                self.flat[s:e].div_(ws)
        self._works = []
        self._sync = False

    def agree(self, ok=True, check_finite=False):
        """True iff every rank reports ok (and, with check_finite, every reduced gradient is finite: validate_gradient of
        toolbox/utils.py:147-157 on the buffer all ranks share); also settles which parameters received a gradient on ANY rank.
        One MIN all-reduce of 1 + n_params int32 and ONE read-back: the step's agreement point."""
        flags = [1 if ok else 0] + [0 if t else 1 for t in self.touched]          # MIN(absent) == 0 <=> somebody has it
        if world_size() > 1 or check_finite:
            t = torch.tensor(flags, dtype=torch.int32).to(self.flat.device)
            if check_finite:
                t[0] = torch.minimum(t[0], torch.isfinite(self.flat).all().to(torch.int32))
            if world_size() > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
            flags = t.cpu().tolist()
        self._absent = [bool(f) for f in flags[1:]]
        return bool(flags[0])

    class _Sparse(object):
        def __init__(self, reducer):
            self.r = reducer

        def __enter__(self):
            for p, absent in zip(self.r.params, self.r._absent or []):
                if absent:
                    p.grad = None                             # as in a single process: Adam / clip skip parameters without a gradient

        def __exit__(self, *exc):
            for i, p in enumerate(self.r.params):
                p.grad = self.r.views[i]
            return False

    def sparse_grads(self):
        """Context for clip + optimizer.step(): parameters no rank produced a gradient for have `.grad = None` inside."""
        return BucketedGradReducer._Sparse(self)


class DataParallelStep(object):
    """The per-batch body and the every-`iter_size` block of the reference's training loop (libs/trainer.py:165-196, 214-237) for
    N ranks: forward, `loss / iter_size` backward into the flat buffer (all-reduce overlapped on the window's last micro-step),
    then -- once per window -- agreement, non-finite check (toolbox/utils.py:147-157), clip, optimizer step, zero.
    Exceptions inside forward / loss / backward are caught like the reference does (the step is then skipped on EVERY rank)."""

    def __init__(self, model, optimizer, loss_fn, iter_size=1, grad_clip=1.0, check_finite=True, catch=True, reducer=None):
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.iter_size, self.grad_clip, self.check_finite, self.catch = int(iter_size), grad_clip, check_finite, catch
        self.reducer = reducer if reducer is not None else BucketedGradReducer(model.parameters())
        self.micro = 0
        self.ok = True
        self.skipped = 0
        self.last_error = None

    def __call__(self, inp, after_forward=None):
        """One micro-step on `inp`.  Returns the loss stats (None when this rank's forward failed)."""
        r = self.reducer
        if self.micro == 0:
            r.zero()
            self.ok = True
        last = self.micro == self.iter_size - 1
        stats, prepared = None, False
        try:
            out = self.model(inp)
            if after_forward is not None:
                after_forward()
            stats = self.loss_fn(out, inp)
            loss = stats['loss'] / self.iter_size if self.iter_size > 1 else stats['loss']
            r.prepare(loss, sync=last)
            prepared = True
            loss.backward()
        except Exception as e:                                # noqa: BLE001 -- libs/trainer.py:234-235
            if not self.catch:
                raise
            self.ok, self.last_error = False, e
            if last:
                if not prepared:
                    r.prepare(None, sync=True)
                r.flush()                                     # the other ranks are waiting in these collectives
        self.micro += 1
        if last:
            self.micro = 0
            r.finish()
            if r.agree(self.ok, check_finite=self.check_finite):
                with r.sparse_grads():
                    if self.grad_clip is not None:
                        torch.nn.utils.clip_grad_norm_(self.reducer.params, self.grad_clip)
                    self.optimizer.step()
            else:
                self.skipped += 1
        return stats


class FlatGradAllReduce(object):
    """One blocking all-reduce of the whole flat buffer after backward (round 1's path; kept as the simple reference the bucketed
    reducer is tested against)."""

    def __init__(self, params, dtype=torch.float32):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(self.numel, dtype=dtype, device=dev)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def __call__(self):
        ws = world_size()
        if ws == 1:
            return
        self.flat.zero_()
        have = [(p, v) for p, v in zip(self.params, self.views) if p.grad is not None]
        if have:
            torch._foreach_copy_([v for _, v in have], [p.grad for p, _ in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(ws)
        if have:
            torch._foreach_copy_([p.grad for p, _ in have], [v for _, v in have])
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()


def all_ok(ok, device):
    """True iff every rank reports ok (1-element MIN all-reduce)."""
    if world_size() == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def max_over_ranks(value, device):
    if world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized():
        dist.barrier()


def per_rank_library_cache(rank, world):
    """MIOpen's user find-db / kernel cache is a set of sqlite files that N ranks would write concurrently: give every rank its
    own copy (seeded from the shipped one) before the first convolution runs.  No-op for one rank."""
    if world <= 1:
        return
    import shutil
    import tempfile
    base = os.environ.get('MIOPEN_USER_DB_PATH')
    dst = os.path.join(tempfile.gettempdir(), 'pcacc_miopen_rank%d_%d' % (rank, os.getpid()))
    try:
        if base and os.path.isdir(base):
            shutil.copytree(base, dst, dirs_exist_ok=True)
        else:
            os.makedirs(dst, exist_ok=True)
        os.environ['MIOPEN_USER_DB_PATH'] = dst
        os.environ['MIOPEN_CUSTOM_CACHE_DIR'] = dst
    except OSError:
        pass
