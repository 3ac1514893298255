"""TubeNet slot algebra (csrc/tube.hip, ops.tube_*) against the torch restatement of models/tpointnet.py:249-305 and
models/alignnet.py:257-263 in oracle/cpu_backend.py -- the restatement is what the CPU model-parity tests pin against the
reference-generated golden vectors (tests/golden/model_*.npz); here the HIP kernels are diffed against it op by op, forward and
backward, at slot counts below and above one workgroup pass and with the degenerate cases the formulas guard."""
import numpy as np
import pytest
import torch

from oracle import cpu_backend as cpu


def _rotations(rng, n):
    """Proper rotations covering all four branches of the matrix -> quaternion scheme (trace and each diagonal entry largest)."""
    q = rng.randn(n, 4)
    q[0::5] = [0.02, 0.01, -0.015, 1.0]                       # near identity: trace branch
    q[1::5, 3] *= 0.01                                       # half turns: a diagonal entry wins
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    r = np.stack([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (w * y + x * z),
                  2 * (w * z + x * y), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                  2 * (x * z - w * y), 2 * (w * x + y * z), w * w - x * x - y * y + z * z], 1).reshape(n, 3, 3)
    return r


def _case(seed, n_inst, T, n_pts, empty_slots=True):
    rng = np.random.RandomState(seed)
    S = n_inst * T
    slot = rng.randint(0, S, n_pts)
    if empty_slots and S > 4:
        slot[slot == 3] = 2                                   # a slot without points
        slot[slot == S - 1] = 0
    xyz = (rng.randn(n_pts, 3) * 3).astype(np.float32)
    centre = (rng.randn(S, 3) * 2).astype(np.float32)
    pose_vec = rng.randn(S, 7).astype(np.float32)
    pose_vec[:, :4] *= rng.uniform(0.2, 3.0, (S, 1)).astype(np.float32)
    rem = np.tile(np.eye(4, dtype=np.float32), (S, 1, 1))
    rem[:, :3, :3] = _rotations(rng, S)
    rem[:, :3, 3] = rng.randn(S, 3)
    total = np.tile(np.eye(4, dtype=np.float32), (S, 1, 1))
    total[:, :3, :3] = _rotations(rng, S)
    total[:, :3, 3] = rng.randn(S, 3)
    w = (rng.rand(S) * (rng.rand(S) > 0.2)).astype(np.float32)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(dt) if dt else torch.from_numpy(np.ascontiguousarray(a))
    return dict(slot=t(slot, torch.int32), xyz=t(xyz), centre=t(centre), pose_vec=t(pose_vec), rem=t(rem), total=t(total), w=t(w), T=T, S=S,
                n_inst=n_inst)


def _close(a, b, rtol=2e-5, atol=2e-6):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


CASES = [(0, 7, 5, 4000), (1, 300, 5, 20000), (2, 3, 1, 500), (3, 40, 10, 3000)]


@pytest.mark.gpu
@pytest.mark.parametrize('seed,n_inst,T,n_pts', CASES)
def test_tube_rows_code_and_pose_forward(seed, n_inst, T, n_pts):
    from pcaccumulation_amd import native
    c = _case(seed, n_inst, T, n_pts)
    g = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in c.items()}
    rows = native.tube_rows(g['xyz'], g['slot'], g['centre'], T)
    assert torch.equal(rows.cpu(), cpu.tube_rows(c['xyz'], c['slot'], c['centre'], T))
    rng = torch.Generator().manual_seed(seed)
    geo, motion, frame = torch.randn(n_inst, 128, generator=rng), torch.randn(n_inst, 128, generator=rng), torch.randn(c['S'], 128, generator=rng)
    code = native.tube_code(geo.cuda(), motion.cuda(), frame.cuda(), T)
    assert torch.equal(code.cpu(), cpu.tube_code(geo, motion, frame, T))
    gc = torch.randn(c['S'], 512, generator=rng)
    for got, want in zip(native.tube_code_backward(gc.cuda(), n_inst, T, 128), cpu.tube_code_backward(gc, n_inst, T, 128)):
        _close(got, want)
    for total in (None, 'total'):
        got = native.tube_pose_forward(g['pose_vec'], g['rem'], g[total] if total else None, g['centre'], g['w'], T)
        want = cpu.tube_pose_forward(c['pose_vec'], c['rem'], c[total] if total else None, c['centre'], c['w'], T)
        for a, b in zip(got, want):
            _close(a, b)
        assert got[5].dtype == torch.float64
        step = got[2].view(n_inst, T, 4, 4)
        assert torch.equal(step[:, 0].cpu(), torch.eye(4).expand(n_inst, 4, 4))      # frame 0 pinned


@pytest.mark.gpu
@pytest.mark.parametrize('seed,n_inst,T,n_pts', CASES)
def test_tube_pose_losses_and_gradients(seed, n_inst, T, n_pts):
    """ops.tube_pose on the GPU against the same autograd Function running on the restatement (the CPU backend)."""
    from pcaccumulation_amd import ops
    c = _case(seed, n_inst, T, n_pts)

    def run(dev, mp=None):
        mv = lambda v: v.to(dev)
        plan = ops.ScatterPlan(mv(c['slot']).long(), c['S'])
        rows = ops.tube_rows(mv(c['xyz']), plan, mv(c['centre']), T)
        pv = mv(c['pose_vec']).clone().requires_grad_(True)
        out = ops.tube_pose(pv, rows, plan, mv(c['rem']), mv(c['total']), mv(c['centre']), mv(c['w']), T)
        l1, l2, rot, trans = out[:4]
        assert not any(o.requires_grad for o in out[4:])
        grads = []
        for coef in ((1.0, 0.0, 0.0, 0.0), (0.0, 1.0, 0.0, 0.0), (0.0, 0.0, 1.0, 0.0), (0.0, 0.0, 0.0, 1.0), (0.7, 0.0, 1.3, 0.4)):
            obj = coef[0] * l1 + coef[1] * l2 + coef[2] * rot + coef[3] * trans
            grads.append(torch.autograd.grad(obj, pv, retain_graph=True)[0])
        return [o.detach().cpu() for o in out], [gr.cpu() for gr in grads]

    got_out, got_grads = run('cuda')
    mp = pytest.MonkeyPatch()
    try:
        cpu.install(mp)
        want_out, want_grads = run('cpu')
    finally:
        mp.undo()
    for a, b in zip(got_out, want_out):
        _close(a, b, rtol=5e-5, atol=5e-6)
    for a, b in zip(got_grads, want_grads):
        _close(a, b, rtol=2e-4, atol=2e-6)


@pytest.mark.gpu
def test_tube_pose_degenerate_inputs():
    """Estimated pose == ground truth (norms at zero: gradient 0, not NaN), an all-zero quaternion (F.normalize's eps clamp),
    all weights zero (the 1e-20 in the denominator)."""
    from pcaccumulation_amd import ops
    c = _case(5, 4, 3, 600, empty_slots=False)
    dev = 'cuda'
    plan = ops.ScatterPlan(c['slot'].to(dev).long(), c['S'])
    centre = torch.zeros(c['S'], 3, device=dev)
    rows = ops.tube_rows(c['xyz'].to(dev), plan, centre, c['T'])
    rem = c['rem'].to(dev)
    pv = torch.zeros(c['S'], 7)
    pv[:, 3] = 1.0                                            # identity pose
    pv[1] = 0.0                                               # |q| = 0: clamped
    rem_id = torch.eye(4).repeat(c['S'], 1, 1)
    pv = pv.to(dev).requires_grad_(True)
    out = ops.tube_pose(pv, rows, plan, rem_id.to(dev), None, centre, c['w'].to(dev), c['T'])
    (out[0] + out[1] + out[2] + out[3]).backward()
    assert torch.isfinite(pv.grad).all() and all(torch.isfinite(o).all() for o in out)
    keep = torch.ones(c['S'], dtype=torch.bool)
    keep[1] = False
    assert float(pv.grad[keep.to(dev)].abs().max()) == 0.0    # exact agreement: every norm sits at its kink, torch's gradient there is 0
    pv2 = c['pose_vec'].to(dev).requires_grad_(True)
    out = ops.tube_pose(pv2, rows, plan, rem, None, centre, torch.zeros(c['S'], device=dev), c['T'])
    assert all(float(o) == 0.0 for o in out[:4])
    (out[0] + out[2]).backward()
    assert float(pv2.grad.abs().max()) == 0.0


def test_tube_restatement_matches_reference_formulation():
    """The restatement's split (rows / code / pose / gap / finish) recombines to the reference's single expression: one slot table
    written out the long way, models/tpointnet.py:264-296."""
    from pcaccumulation_amd.tpointnet import batch_quat2mat, batch_mat2quat, evaluate_pose, reconstruct_sequence
    c = _case(7, 6, 4, 1500)
    T, S = c['T'], c['S']
    slot = c['slot'].long()
    inst, t_idx = slot // T, slot % T
    anchor = c['centre'].view(-1, T, 3)[:, 0]
    local = c['xyz'] - anchor[inst]
    pose_mat = batch_quat2mat(c['pose_vec'])
    gt_mat, gt_vec = batch_mat2quat(c['rem'].view(-1, T, 4, 4), anchor)
    mp = pytest.MonkeyPatch()
    try:
        cpu.install(mp)
        moved_est = reconstruct_sequence(local, t_idx, inst, pose_mat.view(-1, T, 4, 4), T)
        moved_gt = reconstruct_sequence(local, t_idx, inst, gt_mat.view(-1, T, 4, 4), T)
    finally:
        mp.undo()
    gap = moved_est - moved_gt
    cnt = torch.bincount(slot, minlength=S).float().clamp(min=1)
    mean = lambda v: torch.zeros(S).index_add_(0, slot, v) / cnt
    wsum = c['w'].sum() + 1e-20
    l1 = (mean(torch.norm(gap, p=2, dim=1)) * c['w']).sum() / wsum
    l2 = (mean(torch.norm(gap, p=1, dim=1)) * c['w']).sum() / wsum
    rot, trans = evaluate_pose(c['pose_vec'], gt_vec, c['w'])
    rows = cpu.tube_rows(c['xyz'], c['slot'], c['centre'], T)
    pose_c, gt_c, step, rem_out, total_out, loss_rt, ws = cpu.tube_pose_forward(c['pose_vec'], c['rem'], c['total'], c['centre'], c['w'], T)
    sums = torch.zeros(S, 4).index_add_(0, slot, cpu.tube_gap_forward(rows, c['slot'], pose_c, gt_c))
    l12 = cpu.tube_finish(sums, torch.bincount(slot, minlength=S).float(), c['w'], ws)
    assert np.allclose(l12.numpy(), [float(l1), float(l2)], rtol=1e-5)
    assert np.allclose(loss_rt.numpy(), [float(rot), float(trans)], rtol=1e-12)
    assert torch.allclose(torch.matmul(rem_out, step), c['rem'], atol=1e-5) and torch.allclose(total_out, torch.matmul(step, c['total']))


@pytest.mark.gpu
def test_inv4x4_and_update_gt_inst_motion():
    """pcacc_inv4x4 against torch.linalg.inv on rigid and on general (pivoting) matrices; update_gt_inst_motion evaluated for all
    samples at once against the reference's per-sample loop (models/alignnet.py:9-38)."""
    from pcaccumulation_amd import native
    from pcaccumulation_amd.alignnet import update_gt_inst_motion
    rng = np.random.RandomState(0)
    rigid = np.tile(np.eye(4, dtype=np.float32), (40, 1, 1))
    rigid[:, :3, :3] = _rotations(rng, 40)
    rigid[:, :3, 3] = rng.randn(40, 3) * 5
    general = rng.randn(25, 4, 4).astype(np.float32)
    general[::5, 0, 0] = 0.0                                  # forces a row exchange
    for m in (rigid, general):
        t = torch.from_numpy(m).cuda()
        got = native.inv4x4(t)
        want = torch.linalg.inv(t.double())
        # accuracy of an fp32 inverse scales with the matrix's condition number (one of the random matrices has inverse entries
        # of 3e3): the yardstick is the fp32 library inverse this call replaces, both measured against float64, matrix by matrix
        err = (got.double() - want).abs().amax(dim=(1, 2))
        err_lib = (torch.linalg.inv(t).double() - want).abs().amax(dim=(1, 2))
        scale = want.abs().amax(dim=(1, 2))
        assert bool((err <= 4 * err_lib + 1e-6 * scale).all()), (err / scale).max().item()
    assert float((native.inv4x4(torch.from_numpy(rigid).cuda()).double() - torch.linalg.inv(torch.from_numpy(rigid).cuda().double())).abs().max()) <= 2e-5
    B, T = 3, 5
    ego_gt = torch.from_numpy(rigid[:B * T].reshape(B, T, 4, 4)).cuda()
    ego_est = torch.from_numpy(rigid[B * T:2 * B * T].reshape(B, T, 4, 4)).cuda()
    motions = [torch.from_numpy(np.tile(rigid[30 + b][None, None], (k, T, 1, 1)) * 1.0) for b, k in enumerate((4, 0, 7))]
    got = update_gt_inst_motion(motions, ego_gt, ego_est)
    assert [g.shape[0] for g in got] == [4, 0, 7]
    for b, m in enumerate(motions):
        want = m.cuda().float().view(-1, T, 4, 4) @ ego_gt[b][None] @ torch.linalg.inv(ego_est[b])[None]
        assert got[b].shape == want.shape and torch.allclose(got[b], want, rtol=1e-5, atol=1e-4)
