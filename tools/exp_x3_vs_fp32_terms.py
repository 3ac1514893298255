"""fp32x3 against fp32 on a golden train config, loss term by loss term: relative L2 difference of the gradient TENSORS of every early
parameter (pillar encoder, U-Net, heads) for fb_loss alone (a plain path: convolutions, BatchNorm2d, max-pools), for the ego terms alone
(through Sinkhorn / SVD on random-weight features) and of the forward maps.  Separates kernel error from amplification."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_config_parity as tcp  # noqa: E402


def run(g, mode):
    keep = {}
    orig = tcp.MotionNet.__init__

    def patched(self, cfg):
        orig(self, cfg)
        self.unet.register_forward_hook(lambda m, i, o: keep.__setitem__('bev_feats', o.detach().clone()))
        self.ego_feats_head.register_forward_hook(lambda m, i, o: keep.__setitem__('geo_feats', o.detach().clone()))
    tcp.MotionNet.__init__ = patched
    try:
        # _run calls backward on the total loss; we want per-term gradients: re-run the pieces by hand
        import torch as _t
        dev = _t.device('cuda:0')
        T, ppf, mode_s = int(g['n_frames']), int(g['pts_per_frame']), str(g['mode'])
        cfg = tcp.default_config(str(g['dataset']), mode_s, n_sweeps=T)
        cfg['misc']['compute_dtype'] = mode
        inp = tcp.make_batch(cfg, [int(s) for s in g['seeds']], T, ppf)
        model = tcp.MotionNet(cfg)
        tcp.fill_state_dict_(model)
        with _t.no_grad():
            sd = model.state_dict()
            for k, v in zip(g['tweak_keys'], g['tweak_vals']):
                sd[str(k)] += _t.from_numpy(v)
        model = model.to(dev).train(True).channels_last_()
        inp = {k: (v.to(dev) if _t.is_tensor(v) else v) for k, v in inp.items()}
        loss_fn = tcp.FuseLoss(cfg['loss'])
        _t.manual_seed(int(g['fwd_seed']))
        out = model(inp)
        stats = loss_fn(out, inp)
    finally:
        tcp.MotionNet.__init__ = orig
    early = [(n, p) for n, p in model.named_parameters() if n.startswith(('pillar_encoder', 'unet', 'semseg_head', 'ego_feats_head', 'ego_motion_head'))]
    grads = {}
    for term in ('fb_loss', 'ego', 'perm_loss'):
        t = stats['ego_l1_loss'] + stats['ego_l2_loss'] if term == 'ego' else stats[term]
        if not torch.is_tensor(t) or not t.requires_grad:
            continue
        gs = torch.autograd.grad(t, [p for _, p in early], retain_graph=True, allow_unused=True)
        grads[term] = {n: (gg.detach().clone() if gg is not None else None) for (n, _), gg in zip(early, gs)}
    keep['fb_seg'] = out['fb_seg_est'].detach().clone()
    return keep, grads


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'c5'
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_%s.npz' % name), allow_pickle=False)
    ka, ga = run(g, 'fp32')
    kb, gb = run(g, 'fp32x3')
    kc, gc = run(g, 'fp32')
    rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
    for k in ka:
        print('forward %-10s x3 vs fp32: rel L2 %.2e   max-rel %.2e   (fp32 vs fp32 again: %.2e)' %
              (k, rel(kb[k], ka[k]), float((kb[k] - ka[k]).abs().max() / ka[k].abs().max()), rel(kc[k], ka[k])))
    for term in ga:
        rows = [(n, rel(gb[term][n], ga[term][n]), rel(gc[term][n], ga[term][n])) for n in ga[term] if ga[term][n] is not None]
        rows.sort(key=lambda r: -r[1])
        print('%s: gradient tensors x3 vs fp32: max %.2e median %.2e | fp32 vs fp32: max %.2e median %.2e | worst %s' %
              (term, rows[0][1], float(np.median([r[1] for r in rows])), max(r[2] for r in rows), float(np.median([r[2] for r in rows])),
               [(r[0], '%.1e' % r[1]) for r in rows[:4]]), flush=True)


if __name__ == '__main__':
    main()
