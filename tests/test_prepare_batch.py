"""ops: every prepared weight form of a step in one launch (pcacc_prepare_weights_batch) is bit-identical to the per-weight entry points it
replaces, follows the weights through an optimizer step, and leaves weights it has not seen to the per-weight path."""
import pytest
import torch

from pcaccumulation_amd import native, ops

pytestmark = pytest.mark.gpu


def _weights(dev):
    g = torch.Generator().manual_seed(5)
    mk = lambda *shape: torch.nn.Parameter((torch.randn(*shape, generator=g) * 0.1).to(dev))
    w2 = mk(64, 32, 3, 3)
    w2cl = torch.nn.Parameter(mk(96, 64, 3, 3).detach().contiguous(memory_format=torch.channels_last))
    w3 = mk(32, 32, 3, 3, 3)
    wu = mk(64, 32, 2, 2)
    big = mk(256, 256, 3, 3)
    return w2, w2cl, w3, wu, big


def _same(a, b):
    if isinstance(a, tuple):
        return all(_same(x, y) for x, y in zip(a, b))
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    return torch.equal(a, b) if a.dtype == torch.float32 else torch.equal(a.view(torch.int16), b.view(torch.int16))


def test_batch_equals_single_and_follows_the_weights():
    if not ops._BATCH_ON:
        pytest.skip('PCACC_BATCH_PREPARE=0')
    dev = torch.device('cuda:0')
    w2, w2cl, w3, wu, big = _weights(dev)
    asks = [(ops.prepared_conv_weights_split, w2), (ops.prepared_conv_weights_split, w2cl), (ops.prepared_conv_weights_split, w3),
            (ops.prepared_conv_weights_split, big), (ops.prepared_upconv_weights_split, wu), (ops.prepared_upconv_weights_bf16, wu),
            (ops.prepared_conv_weights, w2),
            (ops.prepared_conv_weights, w2cl), (ops.prepared_conv_weights, w3), (ops.prepared_conv_weights, big)]
    for fn, w in asks:                                          # first sight: the per-weight path, and the batch learns the weight
        fn(w)
    singles = {0: native.conv3x3_split_prepare_weights, 1: native.upconv2x2_split_prepare_weights, 2: native.conv3x3_prepare_weights_pair,
               3: native.upconv2x2_bf16_prepare_weights}
    kind_of = {ops.prepared_conv_weights_split: 0, ops.prepared_upconv_weights_split: 1, ops.prepared_conv_weights: 2,
               ops.prepared_upconv_weights_bf16: 3}
    for step in range(3):
        with torch.no_grad():
            for w in (w2, w2cl, w3, wu, big):                   # what a fused optimizer does: new values, same version counter
                w.add_(torch.randn_like(w) * 0.01)
                w._version  # noqa: B018
        ops.weights_may_have_changed()
        state_before = ops._BATCH_STATE.get(0, {}).get('epoch')
        got = [fn(w) for fn, w in asks]
        assert ops._BATCH_STATE[0]['epoch'] == ops._WEIGHT_EPOCH != state_before
        for (fn, w), (fwd, bwd) in zip(asks, got):
            ref_f, ref_b = singles[kind_of[fn]](w.detach())
            assert _same(fwd, ref_f) and _same(bwd, ref_b), (step, fn.__name__, tuple(w.shape))
        if step:                                                # buffers stay allocated: same storage as the step before
            assert all(_ptr(g) == p for g, p in zip(got, ptrs))
        ptrs = [_ptr(g) for g in got]
    # a weight the batch has not seen takes the per-weight path and is then known
    fresh = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=dev))
    ops.weights_may_have_changed()
    f1 = ops.prepared_conv_weights_split(fresh)
    assert _same(f1[0], native.conv3x3_split_prepare_weights(fresh.detach())[0])
    assert (id(fresh), 0) in ops._BATCH_SEEN
    del fresh
    import gc
    gc.collect()
    assert all(ref() is not None for ref in ops._BATCH_SEEN.values())


def _ptr(forms):
    f = forms[0]
    return (f[0] if isinstance(f, tuple) else f).data_ptr()


def test_version_change_inside_an_epoch_is_seen():
    dev = torch.device('cuda:0')
    w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=dev))
    ops.prepared_conv_weights_split(w)
    ops.weights_may_have_changed()
    ops.prepared_conv_weights_split(w)
    with torch.no_grad():
        w.mul_(2.0)                                             # bumps the version counter, no epoch change
    fwd, _ = ops.prepared_conv_weights_split(w)
    ref, _ = native.conv3x3_split_prepare_weights(w.detach())
    assert torch.equal(fwd[0].view(torch.int16), ref[0].view(torch.int16)) and torch.equal(fwd[1], ref[1])
