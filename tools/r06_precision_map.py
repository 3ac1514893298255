#!/usr/bin/env python
"""Per-stage precision map of the matched-accuracy forward: which stages of MotionNet.forward need all three terms of the fp32x3
product (w_hi x_lo + w_lo x_hi + w_hi x_hi on fp16 halves, csrc/conv_split.hip) to keep mos_iou / ego errors / EPE within north_star's 1e-3 of the reference?

For every stage (pillar encoder, U-Net encoder, U-Net decoder, fg/bg head, ego feature head, STPN temporal stack, STPN U-Net, point heads, TubeNet) and every
reduced product -- x2a (activations without their lo half: two MFMAs), x2w (weights without theirs: two MFMAs), x1 (one fp16 term), bf16 (one term, both
operands rounded to bf16's 8 bits) -- the six whole-model fixtures of tests/test_config_parity.py are evaluated with ONLY that stage reduced, and the four
metrics compared with the reference's golden values; then the same on a model TRAINED for 150 steps (32 held-out scenes, validation-set level differences
against the all-x3 forward).  Runs on the -DPCACC_X3_EXPERIMENT build of the library (emulation at full cost: accuracy only).

    PCACC_LIB=build/x3exp/libpcacc_hip.so PCACC_BATCH_PREPARE=0 python tools/r06_precision_map.py > gpurun_out/r06_precision_map.txt"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pcaccumulation_amd import native, ops  # noqa: E402
from pcaccumulation_amd.config import default_config  # noqa: E402
from pcaccumulation_amd.loss import FuseLoss  # noqa: E402
from pcaccumulation_amd.motionnet import MotionNet  # noqa: E402
from pcaccumulation_amd.synthetic import fill_state_dict_  # noqa: E402

STAGES = ['pillar_encoder', 'unet_enc', 'unet_dec', 'fb_head', 'ego_head', 'stpn_temporal', 'stpn_unet', 'point_heads', 'tubenet']
MODES = {'x2a': 1, 'x2w': 2, 'x1': 3, 'bf16': 15}
KEYS = ('ego_rot_error', 'ego_trans_error', 'mos_iou', 'epe_mean')


def main():
    import test_config_parity as cp
    from helpers import make_batch
    dev = torch.device('cuda:0')
    only = os.environ.get('PCACC_MAP_CONFIGS')
    configs = only.split(',') if only else cp.CONFIGS
    combos = [('all', 'x3', {})] + [(st, md, {st: w}) for st in STAGES for md, w in MODES.items()] + \
             [('all', md, {st: w for st in STAGES}) for md, w in MODES.items()]
    table = {(st, md): {} for st, md, _ in combos}
    for name in configs:
        g = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_%s.npz' % name), allow_pickle=False)
        T, ppf, mode = int(g['n_frames']), int(g['pts_per_frame']), str(g['mode'])
        cfg = default_config(str(g['dataset']), mode, n_sweeps=T)
        cfg['misc']['compute_dtype'] = 'mixed'
        inp = make_batch(cfg, [int(s) for s in g['seeds']], T, ppf, mode=str(g['points']) if 'points' in g.files else 'uniform')
        model = MotionNet(cfg)
        fill_state_dict_(model)
        with torch.no_grad():
            sd = model.state_dict()
            for k, v in zip(g['tweak_keys'], g['tweak_vals']):
                sd[str(k)] += torch.from_numpy(v)
        model = model.to(dev).train(mode == 'train').channels_last_()
        inp = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
        loss_fn = FuseLoss(cfg['loss'])
        bn_state = {k: v.clone() for k, v in model.state_dict().items() if 'running_' in k or 'num_batches' in k}
        for st, md, words in combos:
            ops.set_stage_words(words)
            ops.weights_may_have_changed()
            model.load_state_dict(bn_state, strict=False)             # a training-mode forward moves the running statistics: every combination starts alike
            torch.manual_seed(int(g['fwd_seed']))
            with torch.no_grad():
                out = model(inp)
                stats = loss_fn(out, inp)
            got, ref = cp._metrics(g, inp, out, stats, T)
            table[(st, md)][name] = {k: got[k] - ref[k] for k in KEYS}
            torch.cuda.synchronize()
        del model, inp
        torch.cuda.empty_cache()
        print('# fixture %s done' % name, file=sys.stderr, flush=True)
    ops.set_stage_words(None)

    print('# Precision map, round 6: every stage reduced ALONE, every other stage at x3.  Entries: the largest |metric - reference golden| over (rotation error in')
    print('# degrees, translation error in m, mos_iou, EPE in m) of the fixture; bound 1e-3.  `ok` = every fixture within the bound.')
    print('%-15s %-5s ' % ('stage', 'mode') + ' '.join('%10s' % c for c in configs) + '   verdict')
    passing = {}
    for st, md, _ in combos:
        worst = [max(abs(v) for v in table[(st, md)][c].values()) for c in configs]
        ok = all(w <= 1e-3 for w in worst)
        passing[(st, md)] = ok
        print('%-15s %-5s ' % (st, md) + ' '.join('%10.2e' % w for w in worst) + '   ' + ('ok' if ok else 'FAILS 1e-3'))
    print()
    print('# per metric, the worst fixture')
    for st, md, _ in combos:
        row = {k: max((abs(table[(st, md)][c][k]), c) for c in configs) for k in KEYS}
        print('%-15s %-5s ' % (st, md) + '  '.join('%s %.2e (%s)' % (k.replace('ego_', '').replace('_error', ''), v[0], v[1]) for k, v in row.items()))

    if os.environ.get('PCACC_MAP_TRAINED', '1') != '0':
        # the trained-weight fixture of tests/test_config_parity.py::test_gpu_bf16_against_fp32_on_trained_weights, evaluated in the fp32x3 forward
        cfg, model = cp._trained_tiny_model()
        model.eval()
        cfgx = default_config('waymo', 'train', n_sweeps=3, xy_range=8)
        cfgx['misc']['compute_dtype'] = 'fp32x3'
        mx = MotionNet(cfgx).to(dev).channels_last_()
        mx.load_state_dict(model.state_dict())
        mx.eval()
        loss_fn = FuseLoss(cfg['loss'])
        batches = []
        for seed in range(5000, 5032):
            b = make_batch(cfg, [seed], 3, 1500)
            batches.append((seed, {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}))
        res = {}
        for st, md, words in combos:
            ops.set_stage_words(words)
            ops.weights_may_have_changed()
            rows = []
            for seed, b in batches:
                torch.manual_seed(seed)
                with torch.no_grad():
                    out = mx(b)
                    stats = loss_fn(out, b)
                got, _ = cp._metrics({k: 0.0 for k in KEYS}, b, out, stats, 3)
                got['mos_i'], got['mos_u'] = stats['mos_metric']['intersection'], stats['mos_metric']['union']
                rows.append(got)
            res[(st, md)] = rows
        ops.set_stage_words(None)
        base = res[('all', 'x3')]
        agg = lambda rows: float((sum(r['mos_i'] for r in rows) / (sum(r['mos_u'] for r in rows) + 1e-20)).mean())
        print()
        print('# trained weights (150 Adam steps, 32 held-out scenes; NOTE: tiny scenes -- most row layers and some convolutions fall below the split kernels\' size')
        print('# thresholds and run exact fp32 there): validation-set level |difference| to the all-x3 forward: rotation (deg), translation (m), mos_iou (summed counters), EPE (m)')
        for st, md, _ in combos[1:]:
            rows = res[(st, md)]
            d = [abs(float(np.mean([r[k] for r in rows])) - float(np.mean([r[k] for r in base]))) for k in ('ego_rot_error', 'ego_trans_error')]
            d.append(abs(agg(rows) - agg(base)))
            d.append(abs(float(np.mean([r['epe_mean'] for r in rows])) - float(np.mean([r['epe_mean'] for r in base]))))
            print('%-15s %-5s ' % (st, md) + ' '.join('%10.2e' % v for v in d) + '   ' + ('ok' if max(d) <= 1e-3 else 'FAILS 1e-3'))
    print()
    print(json.dumps({'%s/%s' % k: v for k, v in passing.items()}))


if __name__ == '__main__':
    main()
