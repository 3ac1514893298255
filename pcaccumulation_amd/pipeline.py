"""Device-side data step in front of MotionNet: prep_input's voxelisation (libs/dataset.py:183-199) and
collate_fn (libs/dataloader.py:7-40) for samples that already live in HBM.

Produces exactly the input_dict layout the reference's collate_fn produces (same keys, shapes and dtypes,
see SURVEY.md 8b) so the model boundary does not change; only the voxeliser runs in the HIP kernel instead
of numba on a DataLoader worker.
"""
import os
import time

import torch

from . import native
from .voxel_generator import Voxelization


def sample_to_device(sample, device):
    """numpy sample dict (pcaccumulation_amd.synthetic.make_sequence) -> torch tensors on `device`."""
    # everything, including the per-instance motion table (the reference's collate keeps that one as a host list and every
    # consumer moves it with .to(device), models/alignnet.py:22, libs/loss.py:222: a blocking copy per sample per step)
    return {k: torch.from_numpy(v).to(device) for k, v in sample.items()}


def _host_wait(event):
    """Wait on the host for `event` (the voxel counts have reached pinned memory).  PCACC_PREFETCH_WAIT=poll: query the event in a sleep loop
    instead of blocking inside the runtime's event wait (experiment for the two-ranks-on-one-GPU collapse of DESIGN.md section 6)."""
    if os.environ.get('PCACC_PREFETCH_WAIT') == 'poll':
        while not event.query():
            time.sleep(20e-6)
        return
    event.synchronize()


class _Pending(object):
    """A batch whose voxelisation has been launched: everything that does not depend on the voxel counts is already queued."""
    __slots__ = ('samples', 'launched', 'static', 'tis', 'counts', 'event', 'stream', 'device', 'ready', 'result', 'batched')


class DeviceBatcher(object):
    """`batcher(samples)` = voxelise + collate now.  `start(samples)` / `finish(pending)` split the same work so that a training
    loop can queue the voxelisation of batch i+1 on a side stream while batch i trains (the DataLoader workers' role in the
    reference, libs/dataset.py:183-199): the voxel counts come back through pinned memory and an event, so finish() does not
    drain the compute stream -- the first of the three host stalls of a step otherwise."""

    def __init__(self, cfg):
        self.voxeliser = Voxelization(cfg['voxel_generator'])
        self._side = None

    def _launch(self, p):
        vox, samples, dev = self.voxeliser, p.samples, p.device
        p.batched = None
        if dev.type == 'cuda' and len(samples) <= 16 and os.environ.get('PCACC_BATCHED_COLLATE', '1') != '0' \
                and all(s['input_points'].shape[0] > 0 for s in samples):
            # one set of launches for the whole batch: copy into the collated layout + first-touch tables (one per sample) + ranks that ARE the
            # collated pillar ids (pcacc_collate_voxelize); ~7 launches instead of ~70
            b = native.collate_voxelize(samples, vox.voxel_size.tolist(), vox.point_cloud_range.tolist(), vox.grid_size.tolist(), vox.n_sweeps)
            p.batched = b
            p.launched = [(b['coords'], b['point_to_voxel_map'], b['num_voxels'])]
            p.static = {k: b[k] for k in ('input_points', 'time_indice', 'sd_labels', 'inst_labels', 'fb_labels') if k in b}
            p.static['num_points'] = torch.cat([s['num_points'] for s in samples], dim=0)
            p.static['ego_motion_gt'] = torch.stack([s['ego_motion_gt'] for s in samples], 0)
            p.static = {k: p.static[k] for k in ('input_points', 'num_points', 'time_indice', 'sd_labels', 'inst_labels', 'fb_labels', 'ego_motion_gt')
                        if k in p.static}
            return b['num_voxels']
        p.launched = []
        for s in samples:                                                # launch every voxelisation, then ONE read-back
            pts4 = torch.cat((s['input_points'].float(), s['time_indice'].float()), dim=1)
            p.launched.append(vox.voxelize_launch(pts4))
        counts = torch.cat([l[2] for l in p.launched])
        p.tis = [torch.cat((torch.full((s['time_indice'].shape[0], 1), float(b), dtype=torch.float64, device=dev),
                            s['time_indice'].double()), dim=1) for b, s in enumerate(samples)]
        cat = lambda key: torch.cat([s[key] for s in samples], dim=0)
        p.static = {'input_points': cat('input_points'), 'num_points': cat('num_points'), 'time_indice': torch.cat(p.tis, 0),
                    'sd_labels': cat('sd_labels'), 'inst_labels': cat('inst_labels'), 'fb_labels': cat('fb_labels'),
                    'ego_motion_gt': torch.stack([s['ego_motion_gt'] for s in samples], 0)}
        return counts

    def start(self, samples, side_stream=False):
        p = _Pending()
        p.samples, p.device = samples, samples[0]['input_points'].device
        p.stream = p.event = p.ready = p.result = None
        if side_stream and p.device.type == 'cuda':
            if self._side is None:
                self._side = torch.cuda.Stream(device=p.device)
            p.stream = self._side
            # the side stream starts where the compute stream stands NOW in queue order (the host may be many launches ahead of
            # the GPU): the copies then overlap what is queued after this point, not the kernels queued before it
            p.stream.wait_event(torch.cuda.current_stream(p.device).record_event())
            with torch.cuda.stream(p.stream):
                counts = self._launch(p)
                p.counts = torch.empty(counts.shape, dtype=counts.dtype, pin_memory=True)
                p.counts.copy_(counts, non_blocking=True)
                p.event = torch.cuda.Event()
                p.event.record(p.stream)
        else:
            p.counts = self._launch(p)
        return p

    def _collate(self, p, counts):
        """The batch dict of the reference's collate_fn from the launched voxelisations and their pillar counts."""
        vox, samples, dev = self.voxeliser, p.samples, p.device
        if getattr(p, 'batched', None) is not None:
            n_vox = [int(c) for c in counts]
            grid = torch.tensor(list(vox.grid_size) + [vox.n_sweeps], dtype=torch.int64)
            out = dict(p.static)
            out.update({
                'inst_motion_gt': [s['inst_motion_gt'] for s in samples],
                'coordinates': p.batched['coords'][:sum(n_vox)], 'num_voxels': native.upload_small(n_vox, torch.int64, dev),
                'shape': native.upload_small(grid[None].repeat(len(samples), 1), torch.int64, dev),
                'point_to_voxel_map': p.batched['point_to_voxel_map'],
            })
            return out
        coords, p2vs, n_vox = [], [], []
        offset = 0
        for b, s in enumerate(samples):
            m = int(counts[b])
            c, p2v = p.launched[b][0][:m], p.launched[b][1]
            bcol = torch.full((m, 1), float(b), dtype=torch.float64, device=dev)
            coords.append(torch.cat((bcol, c.double()), dim=1))
            p2vs.append((p2v + offset)[:, None])
            n_vox.append(m)
            offset += m
        grid = torch.tensor(list(vox.grid_size) + [vox.n_sweeps], dtype=torch.int64)
        out = dict(p.static)
        out.update({
            'inst_motion_gt': [s['inst_motion_gt'] for s in samples],
            'coordinates': torch.cat(coords, 0), 'num_voxels': native.upload_small(n_vox, torch.int64, dev),
            'shape': native.upload_small(grid[None].repeat(len(samples), 1), torch.int64, dev), 'point_to_voxel_map': torch.cat(p2vs, 0),
        })
        return out

    def finish_early(self, p, prepare=None):
        """Collate the pending batch -- and run `prepare(batch)` (MotionNet.prepare_inputs: pillar index, CSR, per-pillar means,
        point features -- everything the forward derives from the batch alone) -- on the side stream the voxelisation ran on, while
        the current step is still on the GPU.  Call it once the host has issued the step's launches (the pillar counts have long
        arrived by then); finish() then only joins the streams."""
        if p.event is None or p.result is not None:
            return
        _host_wait(p.event)
        with torch.cuda.stream(p.stream):
            out = self._collate(p, p.counts.tolist())
            if prepare is not None:
                out['_prepared'] = prepare(out)
            p.ready = torch.cuda.Event()
            p.ready.record(p.stream)
        p.result = out

    def finish(self, p):
        dev = p.device
        if p.result is not None:                                         # collated (and prepared) ahead of time on the side stream
            from .motionnet import share_with_stream
            main = torch.cuda.current_stream(dev)
            main.wait_event(p.ready)
            share_with_stream(main, p.result, [x for l in p.launched for x in l[:2]])     # allocated there, consumed here
            return p.result
        if p.event is not None:
            _host_wait(p.event)                                          # long past when the batch was started a step ago
            main = torch.cuda.current_stream(dev)
            main.wait_event(p.event)
            for t in [x for l in p.launched for x in l[:2]] + list(p.static.values()):
                t.record_stream(main)                                    # allocated on the side stream, consumed on this one
            counts = p.counts.tolist()
        else:
            counts = p.counts.cpu().tolist()
        return self._collate(p, counts)

    def __call__(self, samples):
        return self.finish(self.start(samples))
