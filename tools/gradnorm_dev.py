"""Per-parameter deviation of the gradient norms of the c3 / c5 train fixtures from the reference golden values (fp32 and fp32x3); PCACC_EGO_FUSED=0/1\nselects the formulation of the ego matching stage.  Development aid."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_config_parity as T
from conftest import *  # noqa
def golden(name):
    return np.load(os.path.join(ROOT, 'tests', 'golden', '%s.npz' % name), allow_pickle=True)
for cfgname in (sys.argv[1:] or ('c3', 'c5')):
    g = golden('model_' + cfgname)
    for mode in os.environ.get('PCACC_MODES', 'fp32,fp32x3,mixed').split(','):
        model, inp, out, stats, Tn = T._run(g, mode)
        grads = dict(model.named_parameters())
        devs = []
        for n, ref in zip([str(x) for x in g['grad_names']], g['grad_norms']):
            got = float(grads[n].grad.norm()) if grads[n].grad is not None else 0.0
            devs.append((abs(got - ref) / max(abs(ref), 1e-2), n))
        devs.sort(reverse=True)
        print(cfgname, mode, 'EGO_FUSED=%s' % os.environ.get('PCACC_EGO_FUSED', '1'), ' '.join('%s %.3f' % (n[-28:], d) for d, n in devs[:8]))
