"""Golden vectors for the results aggregation: the reference's own collect_results (toolbox/evaluation.py:20-98) run on three
synthetic flow_error.npz scenes (tests/helpers.py:flow_error_scenes).  The reference moves every array with `.cuda()`; in this
CPU-only container that call is replaced by the identity for the duration of the run (a harness stand-in, like those of
ref_harness.py).  Stored: the meters of static_stats.pkl (avg / sum / count per leaf), scene_stats.pkl, dynamic_dict.pth.
Run: python tests/golden/make_golden_collect.py"""
import json
import os
import pickle
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_harness  # noqa: E402
from helpers import flow_error_scenes  # noqa: E402


def _flat(meters, prefix=''):
    out = {}
    for k, v in meters.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + '/'))
        else:
            out[prefix + k] = [float(v.avg), float(v.sum), int(v.count)]
    return out


def main():
    ref_harness.install()
    from toolbox.evaluation import collect_results
    torch.Tensor.cuda = lambda self, *a, **k: self
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, 'results'), os.path.join(tmp, 'metrics')
        flow_error_scenes(src)
        collect_results(src, dst, 'waymo')
        static = pickle.load(open(os.path.join(dst, 'static_stats.pkl'), 'rb'))
        scene = pickle.load(open(os.path.join(dst, 'scene_stats.pkl'), 'rb'))
        dyn = torch.load(os.path.join(dst, 'dynamic_dict.pth'))
    # the reference concatenates the sampled dynamic errors in glob order; store them per multiset (sorted) as well
    np.savez_compressed(os.path.join(HERE, 'collect.npz'), static=json.dumps(_flat(static)), scene=json.dumps(scene),
                        dyn_rel_sorted=np.sort(dyn['relative_error'].numpy()), dyn_epe_sorted=np.sort(dyn['epe_per_point'].numpy()))
    print(json.dumps(_flat(static), indent=0)[:600])


if __name__ == '__main__':
    main()
