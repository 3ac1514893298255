"""Host-side pieces that replace a third-party call of the reference, pinned against that call (CPU)."""
import numpy as np
import torch

from pcaccumulation_amd.tpointnet import mat2quat


def test_mat2quat_is_scipy():
    """models/tpointnet.py:62-66 calls scipy's Rotation.from_matrix(...).as_quat() on the host; the device restatement must
    give the same quaternion (same branch, same sign) for random rotations and the 180-degree / identity edge cases."""
    from scipy.spatial.transform import Rotation as R
    mats = np.concatenate([R.random(3000, random_state=7).as_matrix(),
                           R.from_euler('z', [179.999, 180, -180, 0, 1e-9, 90, -90], degrees=True).as_matrix(),
                           R.from_euler('xyz', [[180, 0, 0], [0, 180, 0], [179.9, 0.1, 0], [0, 0, 0]], degrees=True).as_matrix()])
    want = R.from_matrix(mats).as_quat()
    got = mat2quat(torch.from_numpy(mats)).numpy()
    assert got.dtype == np.float64 and np.abs(got - want).max() < 1e-15
    got32 = mat2quat(torch.from_numpy(mats.astype(np.float32))).numpy()            # float32 poses in, float64 out
    want32 = R.from_matrix(mats.astype(np.float32).astype(np.float64)).as_quat()
    # scipy re-orthogonalises matrices that are only orthogonal to float32 precision: agreement to that precision
    assert np.abs(got32 - want32).max() < 1e-7


def test_lazy_results_behave_like_the_reference_dict():
    """pcaccumulation_amd/lazy.py: scalar results are copied to the host asynchronously and become the reference's Python numbers
    on first read -- through every way a trainer reads a dict -- while raw() carries an entry along unread."""
    from pcaccumulation_amd.lazy import HostCopy, LazyDict, LazyValue, lazy_scalars, raw
    a, b = lazy_scalars([torch.tensor(1.5), torch.tensor(2.5, dtype=torch.float64)])
    d = LazyDict(loss=torch.tensor(3.0), rot=a, trans=b)
    assert isinstance(d.raw('rot'), LazyValue) and isinstance(raw(d, 'rot'), LazyValue)
    other = LazyDict(rot=d.raw('rot'))                                        # carried along, still unread
    assert isinstance(dict.__getitem__(other, 'rot'), LazyValue)
    assert d['rot'] == 1.5 and isinstance(d['rot'], float) and not isinstance(dict.__getitem__(d, 'rot'), LazyValue)
    assert other['rot'] == 1.5
    assert d.get('trans') == 2.5 and d.get('missing', 7) == 7
    copy = HostCopy(torch.arange(8, dtype=torch.float64))
    m = LazyDict(metric=LazyValue(copy, 0, 8, lambda v: {'intersection': v[:2], 'union': v[2:4]}), plain=4)
    items = dict(m.items())
    assert items['plain'] == 4 and np.array_equal(items['metric']['union'], [2.0, 3.0])
    assert list(LazyDict(x=LazyValue(copy, 1, 2, lambda v: float(v[0]))).values()) == [1.0]
    p = LazyDict(x=LazyValue(copy, 3, 4, lambda v: float(v[0])))
    assert p.pop('x') == 3.0 and 'x' not in p and p.pop('x', None) is None
    assert raw({'k': 1}, 'k') == 1                                            # plain dicts pass through


def test_amax_tags_follow_views_and_versions():
    """ops.amax_tag (the fp32x3 scale bookkeeping): a tag rides on the tensor object, a view inherits its base's tag (autograd hands data
    gradients on as permuted views), an in-place write to either invalidates it, and dense() keeps it across a contiguous copy."""
    from pcaccumulation_amd import ops
    x = torch.randn(4, 6, 8)
    parts = torch.zeros(256)
    parts[3] = float(x.abs().max())
    assert ops.amax_tag(x) is None
    ops.set_amax_tag(x, parts)
    assert ops.amax_tag(x) is parts
    v = x.permute(0, 2, 1)
    assert v._base is x and ops.amax_tag(v) is parts                      # a re-arrangement of the same elements
    assert ops.amax_tag(x[:, :3]) is parts                                # a slice: the bound still holds
    d = ops.dense(v)
    assert d.is_contiguous() and d is not v and ops.amax_tag(d) is parts  # the tag survives the copy
    assert ops.dense(x) is x
    y = torch.randn(4, 6, 8)
    assert ops.amax_tag(ops.carry_amax(x, y)) is parts
    x.mul_(2.0)                                                           # a new version: neither the tensor nor its views may use the old bound
    assert ops.amax_tag(x) is None and ops.amax_tag(v) is None and ops.amax_tag(x[:, :3]) is None
    assert ops.amax_tag(d) is parts                                       # the copy made before the write keeps its own
    m = ops.merge_amax(torch.randn(2), x, y)
    assert ops.amax_tag(m) is None                                        # a source without a tag: nothing to merge


def test_prepared_weight_copies_are_keyed_on_the_weight_epoch_too():
    """ops.prepared_conv_weights* key their copies on (object, version, address, weight epoch): torch's fused optimizers do not increment
    the version counter, so the epoch advances on every train() / eval() switch, after every step of a watched optimizer (MotionNet.watch_optimizer;
    DataParallelStep watches its own), and -- without a watched optimizer -- at every forward that runs with gradients enabled."""
    import torch
    from pcaccumulation_amd import ops
    from pcaccumulation_amd.config import default_config
    from pcaccumulation_amd.motionnet import MotionNet
    w = torch.nn.Parameter(torch.randn(4, 4, 3, 3))
    k0 = ops._weight_key(w)
    ops.weights_may_have_changed()
    k1 = ops._weight_key(w)
    assert k0 != k1 and k0[:2] == k1[:2]
    with torch.no_grad():
        w.add_(1.0)
    assert ops._weight_key(w)[0] != k1[0]                       # ordinary in-place writes still show in the version
    model = MotionNet(default_config('waymo', 'train', n_sweeps=3, xy_range=8))
    e = ops._WEIGHT_EPOCH
    model.eval()
    assert ops._WEIGHT_EPOCH > e
    e = ops._WEIGHT_EPOCH
    model.train()
    assert ops._WEIGHT_EPOCH > e
    # invalidation at the writer: a watched optimizer's step advances the epoch -- also for a fused optimizer, which leaves the
    # version counters alone -- and MotionNet.forward then no longer invalidates by itself (accumulation micro-steps share one preparation)
    assert not model._optimizer_watched
    opt = model.watch_optimizer(torch.optim.SGD(model.parameters(), lr=0.0))
    assert model._optimizer_watched
    for p_ in model.parameters():
        p_.grad = torch.zeros_like(p_)
    e = ops._WEIGHT_EPOCH
    opt.step()
    assert ops._WEIGHT_EPOCH == e + 1
    # ... and so does ANY other optimizer of the process, created later (resume, a second parameter group) and never shown to the model:
    # the hook is torch's process-wide one, registered once -- a second watch does not stack a second increment
    model.watch_optimizer(opt)
    ops.watch_all_optimizers()
    opt2 = torch.optim.Adam(model.parameters(), lr=0.0)
    e = ops._WEIGHT_EPOCH
    opt2.step()
    assert ops._WEIGHT_EPOCH == e + 1
    # an entry goes with its weight (no device copies of dead parameters until the 4096-entry sweep)
    import gc
    cache = {}
    w2 = torch.nn.Parameter(torch.randn(2, 2, 3, 3))
    cache[id(w2)] = (ops._weight_ref(cache, id(w2), w2), ops._weight_key(w2), None, None)
    assert len(cache) == 1
    del w2
    gc.collect()
    assert len(cache) == 0



def test_no_module_level_name_is_defined_twice():
    """A second `def` of the same name silently replaces the first for every earlier caller (round 4: a new helper in native.py shadowed the pool
    tail's `_pixel_pitch`)."""
    import ast
    import collections
    import os
    import pcaccumulation_amd
    root = os.path.dirname(pcaccumulation_amd.__file__)
    for name in sorted(os.listdir(root)):
        if not name.endswith('.py'):
            continue
        tree = ast.parse(open(os.path.join(root, name)).read())
        counts = collections.Counter(n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)))
        assert not [k for k, v in counts.items() if v > 1], name


def test_l3_domain_binding_is_a_subset_and_restorable():
    """distributed.bind_to_l3_domain: one cache domain per local rank out of the CPUs the process may use; the returned set restores the rest."""
    import os
    from pcaccumulation_amd import distributed as pdist
    before = os.sched_getaffinity(0)
    doms = pdist.l3_domains()
    assert all(set(d) <= before for d in doms) and len({c for d in doms for c in d}) == sum(len(d) for d in doms)
    try:
        prev = pdist.bind_to_l3_domain(1, 2)
        if len(doms) < 2:
            assert prev is None and os.sched_getaffinity(0) == before
        else:
            assert prev == before and os.sched_getaffinity(0) == set(doms[len(doms) // 2])
            os.sched_setaffinity(0, before)
            pdist.bind_to_l3_domain(0, 2)
            assert os.sched_getaffinity(0) == set(doms[0])
    finally:
        os.sched_setaffinity(0, before)


def test_tolerances_are_the_frozen_ones():
    """the parity bounds are closed since round 5.  Loosening one takes an edit HERE, in the table there, and a failing-run log under
    profiles/ -- not a quiet change beside the assertion that failed ("the direction is always looser")."""
    import importlib
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    cp = importlib.import_module('test_config_parity')
    tt = importlib.import_module('test_train_trajectory')
    cs = importlib.import_module('test_conv_split')
    ms = importlib.import_module('test_mlp_split')
    assert cp.GRAD_TOL == {'fp32': (3e-2, 2.5e-2), 'fp32x3': (3e-2, 2.5e-2), 'mixed': (3.5e-2, 2.5e-2), 'mixed2': (3.5e-2, 2.5e-2)}     # mixed2 [r6]: a new mode at mixed's bounds
    assert cp.GRAD_TOL_LIDAR == (6e-2, 2.5e-2)
    assert cp.BF16_TOL == dict(ego=1.5, iou=5e-2, epe=1.5)
    assert tt.ENVELOPE == 10.0
    assert tt.TOL['fp32'] == tt.TOL['fp32x3'] == dict(loss_tol=1e-3, grad_cos=0.999, grad_rel=1e-2, upd_cos=0.98)
    # [r6] term_floor_step2 = 3e-3: the one bound opened in round 6, for the mode with a bf16 backward only (cause, both measured values and the failing
    # run: tests/test_train_trajectory.py:check, DESIGN.md section 4a, profiles/r06_trajectory_mixed_c1_step2.txt)
    assert tt.TOL['mixed'] == dict(loss_tol=1e-3, grad_cos=0.999, grad_rel=3e-2, upd_cos=0.98, term_floor_step2=3e-3)
    assert tt.TOL['bf16'] == dict(loss_tol=0.15, grad_cos=0.9, grad_rel=3.0, upd_cos=0.5, term_tol=0.5, norm_tol=0.1, require_fb=False)
    assert cs.TOL == 3e-6 and ms.TOL == 3e-6


def test_device_sharing_is_decided_from_identity_and_survives_an_uninformative_id():
    """distributed.ranks_share_a_device gathers (host, hardware id, visibility mask, device index) per rank: two ranks on one GPU are 'shared' (the
    one-GPU test box), one rank per GPU is not -- also when the runtime reports ONE id for every GPU of the host (the ranks then differ by index or by
    mask), and never across hosts."""
    from pcaccumulation_amd.distributed import _idents_say_shared as shared
    assert shared([('h', 'a', '', 0), ('h', 'a', '', 0)])
    assert shared([('h', 'a', '0', 0), ('h', 'a', '0', 0)])
    assert shared([('h', 'a', '', 0), ('h', 'b', '', 1), ('h', 'a', '', 0)])
    assert not shared([('h', 'a', '', 0), ('h', 'b', '', 1)])
    assert not shared([('h', 'z', '', i) for i in range(8)])
    assert not shared([('h', 'z', str(i), 0) for i in range(8)])
    assert not shared([('h1', 'a', '', 0), ('h2', 'a', '', 0)])
    # cgroup / device-plugin isolation: every rank sees device index 0 under the same (or an empty) mask, but the hardware ids differ -- informative ids
    # alone decide (the fall-through to (mask, index) called this production layout 'shared' and cost it the two-stream step)
    assert not shared([('h', 'uuid-%d' % i, '', 0) for i in range(8)])
    assert not shared([('h', 'uuid-%d' % i, '0', 0) for i in range(8)])
    assert shared([('h', 'uuid-0', '', 0), ('h', 'uuid-1', '', 0), ('h', 'uuid-1', '', 0)])
    assert shared([('h1', 'a', '', 0), ('h2', 'b', '', 0), ('h2', 'b', '', 0)])


def test_bench_self_launch_decides_without_touching_a_gpu(monkeypatch):
    """bench.self_launch: None (= run the bench in this process) for one GPU, for a malformed --gpus, and inside a launch (WORLD_SIZE / RANK /
    TORCHELASTIC_RUN_ID set: this process IS a rank); only `--gpus N > 1` without a launcher starts a process tree (exercised on the GPU box,
    tests/test_bench_multirank.py::test_bench_starts_its_own_ranks)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)                                   # not __main__: nothing is launched by the import
    for k in ('WORLD_SIZE', 'RANK', 'TORCHELASTIC_RUN_ID'):
        monkeypatch.delenv(k, raising=False)
    assert bench.self_launch([]) is None
    assert bench.self_launch(['--gpus', '1', '--steps', '3']) is None
    assert bench.self_launch(['--gpus=1']) is None
    assert bench.self_launch(['--gpus', 'many']) is None
    monkeypatch.setenv('WORLD_SIZE', '2')
    assert bench.self_launch(['--gpus', '2']) is None
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.setenv('TORCHELASTIC_RUN_ID', 'x')
    assert bench.self_launch(['--gpus=8']) is None
    assert bench._descendants(os.getpid()) == [] or all(isinstance(p, int) for p in bench._descendants(os.getpid()))
