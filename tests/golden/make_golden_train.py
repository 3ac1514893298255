#!/usr/bin/env python
"""Training-TRAJECTORY golden vectors: the REFERENCE (imported read-only from /root/reference on the CPU of this container, see
ref_harness.py) runs K optimizer steps at the cadence of libs/trainer.py:165-237 -- forward, FuseLoss, `loss / iter_size`, backward,
every iter_size-th micro-step validate_gradient -> clip_grad_norm_(1.0) -> Adam.step() -> zero_grad() -- and the fixture keeps what a
product step has to reproduce beyond one forward + backward: the loss of every micro-step, the foreground count of every micro-step
(a flipped decision changes the key-point draw: the test names it instead of reporting a loss mismatch), SAMPLED GRADIENT ENTRIES
(not norms) of the first optimizer step, per-parameter gradient norms of that step, and sampled weights after the first and the last
step.  torch.optim.Adam(lr, weight_decay) exactly as toolbox/config.py:21-22 builds it.

    python tests/golden/make_golden_train.py [tiny_i1 tiny_i2 c1_i1 c1_i2]

Inputs are regenerated from seeds by pcaccumulation_amd.synthetic; weights are synthetic.fill_state_dict_ plus two stored head-bias
offsets (probed in train mode on the first micro-batch).  Nothing of the reference's source is stored.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

rh.install()
from make_golden import save  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.synthetic import make_sequence, attach_voxels, fill_state_dict_  # noqa: E402

K_STEPS = 5
SAMPLES_PER_PARAM = 12
PIN_MARGIN = 30.0

# name: (cfg kwargs, T, points per frame, scenes per micro-step, iter_size, (unused), first scene seed, forward seed)
TRAJ = {
    'tiny_i1': (dict(n_sweeps=3, xy_range=8), 3, 1500, 2, 1, 'model_tiny_val', 700, 4100),
    'tiny_i2': (dict(n_sweeps=3, xy_range=8), 3, 1500, 2, 2, 'model_tiny_val', 700, 4100),
    'c1_i1': (dict(n_sweeps=5), 5, 20000, 1, 1, 'model_waymo_val', 800, 4200),
    'c1_i2': (dict(n_sweeps=5), 5, 20000, 1, 2, 'model_waymo_val', 800, 4200),
}


def sample_indices(named_shapes):
    """The flat indices the fixture samples per parameter: a fixed RandomState stream over the parameters in state_dict order
    (tests rebuild the same list from the product's parameter shapes)."""
    rng = np.random.RandomState(20260)
    out = []
    for name, numel in named_shapes:
        k = min(numel, SAMPLES_PER_PARAM)
        out.append(np.sort(rng.choice(numel, size=k, replace=False)).astype(np.int64))
    return out


def run_once(name, weight_noise=0.0):
    from models.motionnet import MotionNet
    from libs.loss import FuseLoss
    from toolbox.utils import validate_gradient
    kw, T, ppf, n_scenes, iter_size, tweak_fixture, seed0, fwd_seed0 = TRAJ[name]
    cfg = default_config('waymo', 'train', **kw)
    t0 = time.time()
    # the two head-bias offsets (make_golden_configs._probe, train() mode, first micro-batch).  The foreground one is then RAISED by PIN_MARGIN
    # above the largest logit difference: every pillar is predicted background by a wide margin for the whole trajectory.  With the boundary
    # inside the bulk of the logits (first version of this fixture) two correct fp32 implementations agree to 1e-5 on the first two
    # micro-steps and part ways at the third: an update moves a few pillars across the boundary on one side only, a frame's background
    # count changes, torch.randperm(n) (models/egomotion.py:157) draws another key-point set and the loss moves by 1e-2 -- the fixture
    # would test luck, not the loop.  Pinned, every discrete structure of the step is a function of the data alone; the single-step
    # fixtures (model_c*.npz) cover decisions at a boundary.
    from make_golden_configs import _probe
    tweaks, gap = _probe(cfg, [seed0 + j for j in range(n_scenes)], T, ppf, fwd_seed0, True)
    tweaks['semseg_head.seg_head.3.bias'] = tweaks['semseg_head.seg_head.3.bias'] + np.array([PIN_MARGIN, 0.0], np.float32)
    vox = rh.voxeliser(cfg)
    model = MotionNet(cfg)
    fill_state_dict_(model)
    with torch.no_grad():
        sd = model.state_dict()
        for k, v in tweaks.items():
            sd[k] += torch.from_numpy(v)
        if weight_noise:
            gen_ = torch.Generator().manual_seed(99)
            for p_ in model.parameters():
                p_.mul_(1.0 + weight_noise * torch.randn(p_.shape, generator=gen_))
    cfg_loss = dict(cfg['loss'], save_dir='/tmp', min_p_cluster=15)
    try:
        loss_fn = FuseLoss(cfg_loss)
    except Exception:
        import libs.loss as L
        L.ClusterEvaluation = lambda c: None
        loss_fn = FuseLoss(cfg_loss)
    opt = torch.optim.Adam(model.parameters(), lr=cfg['Adam']['learning_rate'], weight_decay=cfg['Adam']['weight_decay'])
    clip = cfg['train']['grad_clip']
    names = [k for k, _ in model.named_parameters()]
    idx = sample_indices([(k, p.numel()) for k, p in model.named_parameters()])
    model.train()
    opt.zero_grad()
    losses, terms, fb_sums, valid, total_norms = [], [], [], [], []
    grad_samples = grad_norms = w_after_1 = None
    term_keys = ('fb_loss', 'mos_loss', 'offset_loss', 'obj_loss', 'perm_loss', 'ego_l1_loss')
    micro = 0
    for step in range(K_STEPS):
        for it in range(iter_size):
            seeds = [seed0 + micro * n_scenes + j for j in range(n_scenes)]
            inp = rh.collate([attach_voxels(make_sequence(s, T, ppf, cfg), vox) for s in seeds])
            torch.manual_seed(fwd_seed0 + micro)
            out = model(inp)
            stats = loss_fn(out, inp)
            (stats['loss'] / iter_size).backward()
            losses.append(float(stats['loss'].detach()))
            terms.append([float(stats[k].detach()) if torch.is_tensor(stats.get(k)) else float(stats.get(k, 0.0)) for k in term_keys])
            fb_sums.append(int(out['fb_est_per_points'].sum()))
            assert fb_sums[-1] == 0, 'PIN_MARGIN too small: a pillar was predicted foreground'
            micro += 1
        ok = validate_gradient(model)
        valid.append(bool(ok))
        if step == 0:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in model.parameters()]
            grad_samples = np.concatenate([g_.reshape(-1)[torch.from_numpy(i)].numpy() for g_, i in zip(grads, idx)])
            grad_norms = np.array([float(g_.norm()) for g_ in grads])
        if ok:
            total_norms.append(float(torch.nn.utils.clip_grad_norm_(model.parameters(), clip)))
            opt.step()
        else:
            total_norms.append(float('nan'))
        opt.zero_grad()
        if step == 0:
            w_after_1 = np.concatenate([p.detach().reshape(-1)[torch.from_numpy(i)].numpy() for p, i in zip(model.parameters(), idx)])
        print('%s step %d: loss %s fb %s |g| %.4f  (%.0f s)' % (name, step, ['%.5f' % l for l in losses[-iter_size:]], fb_sums[-iter_size:], total_norms[-1],
                                                               time.time() - t0), flush=True)
    w_after_k = np.concatenate([p.detach().reshape(-1)[torch.from_numpy(i)].numpy() for p, i in zip(model.parameters(), idx)])
    return dict(config=name, cfg_kwargs=np.array(sorted(kw.items()), dtype=object).astype(str), n_frames=T, pts_per_frame=ppf,
         scenes_per_micro_step=n_scenes, iter_size=iter_size, k_steps=K_STEPS, seed0=seed0, fwd_seed0=fwd_seed0,
         lr=cfg['Adam']['learning_rate'], weight_decay=cfg['Adam']['weight_decay'], grad_clip=clip,
         tweak_keys=np.array(list(tweaks.keys())), tweak_vals=np.stack(list(tweaks.values())), fb_gap=gap,
         param_names=np.array(names), sample_counts=np.array([len(i) for i in idx]), sample_idx=np.concatenate(idx),
         loss=np.array(losses), term_keys=np.array(term_keys), terms=np.array(terms), fb_est_sum=np.array(fb_sums), gradient_valid=np.array(valid),
         total_grad_norm=np.array(total_norms), grad_samples_step1=grad_samples, grad_norms_step1=grad_norms,
         weights_after_step1=w_after_1, weights_after_last=w_after_k,
         bn_running_mean=model.semseg_head.seg_head[1].running_mean.numpy())


REPEATS = ((8, 0.0), (1, 0.0), (8, 1e-6))          # (threads, relative weight noise)


def gen(name):
    """The first entry of REPEATS is the fixture; the others give the reference's OWN envelope: the same trajectory with another thread count
    (reductions summed in another order) and with every weight multiplied by 1 + 1e-6 N(0,1) -- a few ulps, the size of the difference
    between two correct fp32 convolution implementations (another summation order moves a 288-term dot product by ~1e-6).  The reference
    against itself agrees to 1e-7 on the first loss and drifts apart by a factor of ~5 per optimizer step (tiny: 7e-8, 5e-6..4e-5, 3e-4,
    4e-4, 1.5e-3..2.4e-3): gradients of this model are 100-4000 x more sensitive than its forward maps (the ego terms go
    through Sinkhorn + SVD on soft correspondences) and Adam feeds them back.  The tests hold the product to that envelope, not to a fixed
    1e-3 at step 5 that the reference itself does not meet."""
    runs = []
    for nt, noise in REPEATS:
        torch.set_num_threads(nt)
        runs.append(run_once(name, noise))
    torch.set_num_threads(REPEATS[0][0])
    d = runs[0]
    d['repeat_threads'] = np.array([r[0] for r in REPEATS])
    d['repeat_weight_noise'] = np.array([r[1] for r in REPEATS])
    d['loss_runs'] = np.stack([r['loss'] for r in runs])
    d['terms_runs'] = np.stack([r['terms'] for r in runs])
    d['grad_samples_runs'] = np.stack([r['grad_samples_step1'] for r in runs])
    d['weights_after_last_runs'] = np.stack([r['weights_after_last'] for r in runs])
    d['total_grad_norm_runs'] = np.stack([r['total_grad_norm'] for r in runs])
    save('train_%s' % name, **d)


if __name__ == '__main__':
    for name in (sys.argv[1:] or list(TRAJ)):
        gen(name)
