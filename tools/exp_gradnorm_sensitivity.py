"""Why do the STPN gradient norms of the c3 golden config move by several per cent between implementations?  Runs the c3 train step in
fp32 (library convolutions), fp32x3 (split-bf16 kernels) and fp32 with the U-Net output perturbed by 1e-6 / 1e-5 relative noise, and prints
every parameter's gradient-norm deviation from the reference's golden value.  If the perturbed fp32 run scatters like the fp32x3 run, the
deviation is the max-over-frames / max-pool routing among near-tied cells (tests/test_model_parity.py:119-121), not a kernel error.
Also: the split kernels against the fp32 library at the full c3 layer sizes."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_config_parity as tcp  # noqa: E402
from pcaccumulation_amd import native, ops  # noqa: E402


def big_kernel_check():
    dev = torch.device('cuda:0')
    for (n, t, h, w, ci, co, kt) in [(20, 5, 288, 288, 32, 32, 3), (20, 1, 288, 288, 64, 64, 1), (20, 1, 144, 144, 64, 128, 1), (20, 1, 36, 36, 512, 256, 1)]:
        g = torch.Generator(device='cpu').manual_seed(1)
        x = torch.randn(n, h, w, ci, generator=g).to(dev)
        gy = torch.randn(n, h, w, co, generator=g).to(dev)
        shape = (co, ci, 3, 3, 3) if kt == 3 else (co, ci, 3, 3)
        wt = (torch.randn(*shape, generator=g) / (3 * (ci * kt) ** 0.5)).to(dev)
        b = torch.randn(co, generator=g).to(dev)
        wf, wb = native.conv3x3_split_prepare_weights(wt)
        y = native.conv3x3_split(x, wf, b, t, True)
        gx = native.conv3x3_split(gy, wb, None, t, False, mask=y)
        xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if kt == 3:
            x5 = xr.view(n // t, t, h, w, ci).permute(0, 4, 1, 2, 3)
            yr = F.conv3d(x5, wr, br, padding=1).permute(0, 2, 3, 4, 1).reshape(n, h, w, co)
            parts = [native.conv3x3_wgrad_split(gy, x, t, dt, mask=y) for dt in (-1, 0, 1)]
            gw = torch.stack([p[0].view(co, 3, 3, ci) for p in parts], dim=1).permute(0, 4, 1, 2, 3)
            gb = parts[1][1]
        else:
            yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, padding=1).permute(0, 2, 3, 1)
            gw, gb = native.conv3x3_wgrad_split(gy, x, mask=y)
            gw = gw.view(co, 3, 3, ci).permute(0, 3, 1, 2)
        (yr * (y > 0)).backward(gy)
        rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
        print('big check', (n, t, h, w, ci, co, kt), 'y %.2e gx %.2e gw %.2e gb %.2e' % (rel(y, torch.relu(yr)), rel(gx, xr.grad), rel(gw, wr.grad), rel(gb, br.grad)), flush=True)


def run(g, mode, noise=0.0):
    hooks = []
    orig = tcp.MotionNet.__init__

    def patched(self, cfg):
        orig(self, cfg)
        if noise:
            def hook(mod, inp, out):
                gen = torch.Generator(device=out.device).manual_seed(5)
                return out * (1 + noise * torch.randn(out.shape, device=out.device, generator=gen))
            hooks.append(self.unet.register_forward_hook(hook))
    tcp.MotionNet.__init__ = patched
    try:
        model, inp, out, stats, T = tcp._run(g, mode)
    finally:
        tcp.MotionNet.__init__ = orig
    got, ref = tcp._metrics(g, inp, out, stats, T)
    grads = dict(model.named_parameters())
    norms = {n: (float(grads[n].grad.norm()) if grads[n].grad is not None else 0.0) for n in grads}
    terms = {k: float(stats[k]) for k in ('fb_loss', 'mos_loss', 'offset_loss', 'obj_loss', 'perm_loss', 'ego_l1_loss', 'ego_l2_loss') if k in stats}
    return got, ref, norms, float(stats['loss']), terms


def main():
    if os.environ.get('BIG'):
        big_kernel_check()
    name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_%s.npz' % name), allow_pickle=False)
    names = [str(n) for n in g['grad_names']]
    refn = dict(zip(names, [float(v) for v in g['grad_norms']]))
    loose = ('motionhead.init_conv', 'motionhead.down_convs', 'motionhead.up_convs')
    res = {}
    for tag, mode, noise in (('fp32', 'fp32', 0.0), ('fp32x3', 'fp32x3', 0.0), ('fp32+1e-6', 'fp32', 1e-6), ('fp32+3e-6', 'fp32', 3e-6), ('fp32+1e-5', 'fp32', 1e-5)):
        got, ref, norms, loss, terms = run(g, mode, noise)
        dev = {n: abs(norms[n] - refn[n]) / max(abs(refn[n]), 1e-2) for n in names}
        lo = [dev[n] for n in names if n.startswith(loose)]
        ti = [dev[n] for n in names if not n.startswith(loose)]
        worst = sorted(names, key=lambda n: -dev[n])[:5]
        print('%-12s loss %.6f (ref %.6f)  metrics %s' % (tag, loss, float(g['loss']), {k: '%.2e' % abs(got[k] - ref[k]) for k in got}))
        print('   STPN-backbone params: max %.3f median %.4f | others: max %.4f median %.5f | worst %s' %
              (max(lo), float(np.median(lo)), max(ti), float(np.median(ti)), [(n, '%.3f' % dev[n]) for n in worst]), flush=True)
        res[tag] = norms
        print('   loss terms - ref:', {k: '%.2e' % (terms[k] - float(g[k])) for k in terms if k in g.files})
        print('   > 2 %:', [(n, '%.3f' % dev[n]) for n in names if dev[n] > 0.02], flush=True)
    a, b = res['fp32'], res['fp32x3']
    d = {n: abs(a[n] - b[n]) / max(abs(a[n]), 1e-2) for n in names}
    print('fp32x3 vs fp32: STPN-backbone max %.3f, others max %.4f' % (max(d[n] for n in names if n.startswith(loose)), max(d[n] for n in names if not n.startswith(loose))))


if __name__ == '__main__':
    main()
