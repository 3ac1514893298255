"""Whole-model golden vectors: the reference MotionNet (+ FuseLoss) run on synthetic sequences with
closed-form weights.  Called from make_golden.py (this container only)."""
import numpy as np
import torch

import ref_harness as rh
from pcaccumulation_amd.config import default_config
from pcaccumulation_amd.synthetic import make_sequence, attach_voxels, fill_state_dict_


def _tweak_biases(model, inp, seed):
    """Shift the two 2-class heads so that both classes occur (closed-form weights alone give ~100 % foreground,
    which would leave the ego-motion head without background pillars).  Returns {key: additive offset}."""
    tweaks = {}
    model.eval()
    with torch.no_grad():
        torch.manual_seed(seed)
        try:
            out = model(inp)
            fs = out['fb_seg_est']
            d = (fs[:, :, 1] - fs[:, :, 0])[out['occ_map'][:, :, 0] > 0]
        except Exception:                                   # no background pillar at all: probe the head directly
            d = torch.tensor([20.0])
        med = float(torch.quantile(d, 0.6))
        tweaks['semseg_head.seg_head.3.bias'] = np.array([med, 0.0], np.float32)
        model.semseg_head.seg_head[3].bias += torch.tensor([med, 0.0])
        torch.manual_seed(seed)
        out = model(inp)
        m = out['mos_est']
        fb = torch.logical_or(inp['fb_labels'][:, 0] == 1, out['fb_est_per_points'][:, 0] == 1)
        med2 = float(torch.median((m[:, 1] - m[:, 0])[fb]))
        tweaks['motionhead.mos_seg.seg_head.3.bias'] = np.array([med2, 0.0], np.float32)
        model.motionhead.mos_seg.seg_head[3].bias += torch.tensor([med2, 0.0])
    return tweaks


def _epe(out, inp, n_frames):
    """libs/tester.py:58-77 on batch element 0."""
    from toolbox.register_utils import ego_motion_compensation, reconstruct_sequence
    sel = inp['time_indice'][:, 0] == 0
    x = inp['input_points'][sel].float()          # tester path: f32 points with the f32 ego motion (libs/tester.py:60-67)
    t = inp['time_indice'][sel, 1].long()
    ego = inp['ego_motion_gt'].float()[0]
    comp = ego_motion_compensation(x, t, ego)
    rec_gt = reconstruct_sequence(comp, t, inp['inst_labels'][sel, 0], inp['inst_motion_gt'][0].float(), n_frames)
    err = torch.norm((out['rec_est'][sel] - x) - (rec_gt - x), p=2, dim=1)
    return err[t > 0]


def _run(cfg, seeds, n_frames, ppf, mode, fwd_seed, train, tweaks=None, points='uniform'):
    from models.motionnet import MotionNet
    from libs.loss import FuseLoss
    cfg = dict(cfg)
    cfg['misc'] = dict(cfg['misc'], mode=mode)
    vox = rh.voxeliser(cfg)
    inp = rh.collate([attach_voxels(make_sequence(s, n_frames, ppf, cfg, mode=points), vox) for s in seeds])
    model = MotionNet(cfg)
    fill_state_dict_(model)
    if tweaks is None:
        tweaks = _tweak_biases(model, inp, fwd_seed)
    else:
        with torch.no_grad():
            sd = model.state_dict()
            for k, v in tweaks.items():
                sd[k] += torch.from_numpy(v)
    cfg_loss = dict(cfg['loss'], save_dir='/tmp', min_p_cluster=15)
    loss_fn = None
    try:
        loss_fn = FuseLoss(cfg_loss)
    except Exception as e:                                  # ClusterEvaluation wants extra keys; build without it
        import libs.loss as L
        L.ClusterEvaluation = lambda c: None
        loss_fn = FuseLoss(cfg_loss)
    model.train(train)
    torch.manual_seed(fwd_seed)
    if train:
        out = model(inp)
        stats = loss_fn(out, inp)
        stats['loss'].backward()
    else:
        with torch.no_grad():
            out = model(inp)
            stats = loss_fn(out, inp)
    return model, inp, out, stats, tweaks


def _common(out, stats, inp, n_frames):
    d = {
        'ego_motion_est': out['ego_motion_est'].detach().numpy(), 'ego_motion_gt': out['ego_motion_gt'].detach().numpy(),
        'ego_rot_error': out['ego_rot_error'], 'ego_trans_error': out['ego_trans_error'],
        'ego_l1_loss': float(out['ego_l1_loss']), 'ego_l2_loss': float(out['ego_l2_loss']),
        'inst_l2_error': out['inst_l2_error'], 'dynamic_inst_l2_error': out['dynamic_inst_l2_error'],
        'loss': float(stats['loss']), 'fb_loss': float(stats['fb_loss']), 'mos_loss': float(stats['mos_loss']),
        'offset_loss': float(stats['offset_loss']), 'obj_loss': float(stats['obj_loss']), 'perm_loss': float(stats['perm_loss']),
        'offset_l2_error': stats['offset_l2_error'],
    }
    for name in ('mos_metric', 'fb_metric'):
        for k, v in stats[name].items():
            d['%s_%s' % (name, k)] = v
    i, u = stats['mos_metric']['intersection'], stats['mos_metric']['union']
    d['mos_iou'] = float((i / (u + 1e-20)).mean())
    epe = _epe(out, inp, n_frames)
    d['epe_mean'] = float(epe.mean())
    return d, epe


def gen_model(save, sha):
    from models.motionnet import MotionNet

    # ---- state_dict contract -----------------------------------------------------------------------
    cfg_w = default_config('waymo', 'val')
    rh.install()
    sd = MotionNet(cfg_w).state_dict()
    save('state_dict_keys', keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
         dtypes=np.array([str(v.dtype) for v in sd.values()]))

    # ---- tiny scene, eval mode ('val'), B=2, T=3, 64x64 grid ------------------------------------------
    cfg_s = default_config('waymo', 'val', n_sweeps=3, xy_range=8)
    model, inp, out, stats, tweaks = _run(cfg_s, (20, 21), 3, 1500, 'val', 123, train=False)
    d, epe = _common(out, stats, inp, 3)
    save('model_tiny_val', seeds=np.array([20, 21]), n_frames=3, pts_per_frame=1500, fwd_seed=123,
         tweak_keys=np.array(list(tweaks.keys())), tweak_vals=np.stack(list(tweaks.values())),
         fb_seg_est=out['fb_seg_est'].numpy(), fb_est_per_points=out['fb_est_per_points'].numpy(),
         fb_seg_gt=out['fb_seg_gt'].numpy(), occ_map=out['occ_map'].numpy(),
         transformed_points=out['transformed_points'].numpy(), mos_est=out['mos_est'].numpy(),
         offset_est=out['offset_est'].numpy(), rec_est=out['rec_est'].numpy(),
         perm_rowsum=np.stack([p.sum(2)[0].numpy() for p in out['perm_matrix']]),
         inst_pose_est=out['inst_pose_est'].numpy(), inst_labels_adjusted=out['inst_labels_adjusted'].numpy(),
         epe=epe.numpy(), **d)
    print('tiny val: fg ratio %.3f, mos1 ratio %.3f, rot err %.3f' % (
        float(out['fb_est_per_points'].float().mean()), float(out['mos_est'].argmax(1).float().mean()), d['ego_rot_error']))

    # ---- tiny scene, train mode (BatchNorm batch statistics), forward + FuseLoss + backward ---------------
    model, inp, out, stats, _ = _run(cfg_s, (20, 21), 3, 1500, 'train', 123, train=True, tweaks=tweaks)
    d, epe = _common(out, stats, inp, 3)
    names, norms, samples = [], [], []
    for k, p in model.named_parameters():
        names.append(k)
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        norms.append(float(g.norm()))
        samples.append(g.reshape(-1)[:4].numpy().copy() if g.numel() >= 4 else np.resize(g.reshape(-1).numpy(), 4))
    save('model_tiny_train', seeds=np.array([20, 21]), n_frames=3, pts_per_frame=1500, fwd_seed=123,
         tweak_keys=np.array(list(tweaks.keys())), tweak_vals=np.stack(list(tweaks.values())),
         mos_est=out['mos_est'].detach().numpy(), offset_est=out['offset_est'].detach().numpy(),
         rec_est=out['rec_est'].detach().numpy(), fb_est_per_points=out['fb_est_per_points'].numpy(),
         grad_names=np.array(names), grad_norms=np.array(norms), grad_samples=np.stack(samples),
         bn_running_mean=model.semseg_head.seg_head[1].running_mean.numpy(), epe=epe.detach().numpy(), **d)
    print('tiny train: loss %.4f' % d['loss'])

    # ---- Waymo geometry (288x288, T=5), 20k points per frame, eval: config c1 -------------------------------
    model, inp, out, stats, tweaks_w = _run(cfg_w, (30,), 5, 20000, 'val', 321, train=False)
    d, epe = _common(out, stats, inp, 5)
    idx = np.linspace(0, inp['input_points'].shape[0] - 1, 2048).astype(np.int64)
    save('model_waymo_val', seeds=np.array([30]), n_frames=5, pts_per_frame=20000, fwd_seed=321,
         tweak_keys=np.array(list(tweaks_w.keys())), tweak_vals=np.stack(list(tweaks_w.values())),
         sample_idx=idx, mos_est=out['mos_est'].numpy()[idx], offset_est=out['offset_est'].numpy()[idx],
         rec_est=out['rec_est'].numpy()[idx], transformed_points=out['transformed_points'].numpy()[idx],
         fb_est_per_points=out['fb_est_per_points'].numpy()[idx], fb_est_sum=int(out['fb_est_per_points'].sum()),
         fb_seg_est_sample=out['fb_seg_est'].numpy()[0, :, :, ::8, ::8], num_voxels=inp['num_voxels'].numpy(),
         coordinates_sha=sha(inp['coordinates'].numpy()), p2v_sha=sha(inp['point_to_voxel_map'].numpy()),
         epe_sample=epe.numpy()[::64], **d)
    print('waymo val: M=%d, fg ratio %.3f, rot err %.3f, epe %.4f' % (
        int(inp['num_voxels'][0]), float(out['fb_est_per_points'].float().mean()), d['ego_rot_error'], d['epe_mean']))


OFFSET_SCALE, MOS_SHIFT = 0.01, 6.0


def gen_model_test(save):
    """misc.mode='test' (B=1, as libs/tester.py runs it): estimated foreground only, DBSCAN instances instead of the
    ground-truth ones, identity instance-motion 'ground truth' (models/alignnet.py:190-192)."""
    import numpy as np
    g = np.load(__file__.replace('make_golden_model.py', 'model_tiny_val.npz'))
    tweaks = {str(k): v for k, v in zip(g['tweak_keys'], g['tweak_vals'])}
    cfg_s = default_config('waymo', 'val', n_sweeps=3, xy_range=8)
    from models.motionnet import MotionNet
    cfg = dict(cfg_s)
    cfg['misc'] = dict(cfg['misc'], mode='test')
    vox = rh.voxeliser(cfg)
    inp = rh.collate([attach_voxels(make_sequence(20, 3, 1500, cfg), vox)])
    model = MotionNet(cfg)
    fill_state_dict_(model)
    with torch.no_grad():
        sd = model.state_dict()
        for k, v in tweaks.items():
            sd[k] += torch.from_numpy(v)
        # closed-form weights give offsets of metres and scattered 'moving' points; damp the offset head and push the
        # motion head towards 'moving' so that DBSCAN finds instances (the test applies the same two edits)
        sd['motionhead.offset_head.seg_head.3.weight'] *= OFFSET_SCALE
        sd['motionhead.offset_head.seg_head.3.bias'] *= OFFSET_SCALE
        sd['motionhead.mos_seg.seg_head.3.bias'] += torch.tensor([0.0, MOS_SHIFT])
    model.eval()
    torch.manual_seed(123)
    with torch.no_grad():
        out = model(inp)
    lab = out['inst_labels_est'].numpy()
    print('tiny test: moving %.3f, instances %d, labelled points %d, inst_l2 %.4f' % (
        float(out['mos_est'].argmax(1).float().mean()), int(lab.max()), int((lab > 0).sum()), out.get('inst_l2_error', -1)))
    save('model_tiny_test', seeds=np.array([20]), n_frames=3, pts_per_frame=1500, fwd_seed=123,
         tweak_keys=np.array(list(tweaks.keys())), tweak_vals=np.stack(list(tweaks.values())),
         offset_scale=OFFSET_SCALE, mos_shift=MOS_SHIFT, inst_labels_est=lab, mos_est=out['mos_est'].numpy(), offset_est=out['offset_est'].numpy(),
         rec_est=out['rec_est'].numpy(), transformed_points=out['transformed_points'].numpy(),
         fb_est_per_points=out['fb_est_per_points'].numpy(), ego_motion_est=out['ego_motion_est'].numpy(),
         inst_pose_est=out['inst_pose_est'].numpy() if 'inst_pose_est' in out else np.zeros((0, 4, 4), np.float32),
         inst_l2_error=out.get('inst_l2_error', -1.0), dynamic_inst_l2_error=out.get('dynamic_inst_l2_error', -1.0),
         ego_rot_error=out['ego_rot_error'], ego_trans_error=out['ego_trans_error'])
