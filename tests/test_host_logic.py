"""CPU: host-side logic of the wrappers that needs no kernel (index bookkeeping, the raw-slot hand-over rules)."""
import torch


def test_time_index_normalisation_makes_no_copy_of_a_ready_index():
    """`normalise_time_index`: int64 + contiguous, the tensor itself when it already is (no clone, no read-back --
    the kernels validate the indices on the device, `_lib.IndexStatus`), with or without a version counter."""
    from waldo_amd import functional as WF
    ts2 = torch.arange(4).view(1, 1, 4)
    assert WF.normalise_time_index(ts2) is ts2
    exp = torch.arange(2).view(1, 2, 1).expand(3, 2, 4)   # the expanded view of synthesizer.py:438
    n = WF.normalise_time_index(exp)
    assert n.is_contiguous() and n.dtype == torch.int64 and torch.equal(n, exp)
    assert WF.normalise_time_index(torch.arange(3, dtype=torch.int32)).dtype == torch.int64
    with torch.inference_mode():
        ts = torch.arange(6).view(1, 2, 3)
        assert WF.normalise_time_index(ts) is ts


def test_an_index_made_by_arange_index_is_known_to_be_the_identity_until_written_to():
    """`time_gather` hands out the clip itself for `x[:, ts]` with ts = 0 .. n-1 (LVD ctx_mode "prev") only for an
    index the CALLER built with `arange_index` (a host-side fact: no device read, also during HIP-graph capture);
    the mark is tied to the tensor object and its version counter."""
    import copy
    from waldo_amd import functional as WF
    ts = WF.arange_index(5)
    assert ts.dtype == torch.int64 and ts.tolist() == [0, 1, 2, 3, 4]
    assert WF._is_arange_index(ts) and WF.normalise_time_index(ts) is ts
    assert not WF._is_arange_index(torch.arange(5))           # a plain tensor: never
    assert not WF._is_arange_index(ts[:3]) and not WF._is_arange_index(ts + 0) and not WF._is_arange_index(ts.clone())
    assert WF._is_arange_index(copy.deepcopy(ts))
    ts[0] = 3                                                 # a write bumps the version
    assert not WF._is_arange_index(ts) and not WF._is_arange_index(copy.deepcopy(ts))
    with torch.inference_mode():                              # no version counter to vouch for it: a plain index
        assert not WF._is_arange_index(WF.arange_index(4))


def test_raw_slots_are_handed_over_explicitly():
    """`flow_ctx_warp_into_raw` returns its `RawSlots` beside the alpha view and `frame_warp_fuse_raw` takes them as
    an argument: nothing rides on a tensor attribute, `frame_warp_fuse` never inspects its alpha argument."""
    import inspect
    from waldo_amd import functional as WF
    assert list(inspect.signature(WF.frame_warp_fuse_raw).parameters)[:4] == ["input", "flow", "slots", "ctx_ts"]
    slots = WF.RawSlots(torch.zeros(1, 2, 3, 5, 4, 4), torch.zeros(1, 3, 2, 4, 4), 2, False)
    assert slots.channels == 2 and not slots.include_self
    assert "_waldo_raw" not in inspect.getsource(WF)


def test_zero_filled_gradients_come_out_of_one_buffer():
    """`_zeros_like_each`: the small tensors a backward kernel accumulates into share ONE zero-filled buffer (one fill
    launch); every view starts on a 256-byte boundary and keeps its tensor's shape; None stays None."""
    from waldo_amd.functional import _zeros_like_each
    a, b, c = torch.ones(3, 5), torch.ones(7), torch.ones(2, 2, 2)
    big = torch.ones(1 << 18)                            # 1 MB: a buffer of its own (it would pin the others)
    za, none, zb, zc, zbig = _zeros_like_each(a, None, b, c, big)
    assert zbig.shape == big.shape and float(zbig.sum()) == 0.0
    assert zbig.untyped_storage().data_ptr() != za.untyped_storage().data_ptr()
    assert zbig.untyped_storage().nbytes() == big.numel() * 4
    assert none is None and za.shape == a.shape and zb.shape == b.shape and zc.shape == c.shape
    assert float(za.sum() + zb.sum() + zc.sum()) == 0.0
    base = za.untyped_storage().data_ptr()
    assert zb.untyped_storage().data_ptr() == base and zc.untyped_storage().data_ptr() == base
    assert [t.storage_offset() % 64 for t in (za, zb, zc)] == [0, 0, 0]
    za.add_(1.0)                                         # the views do not overlap
    assert float(zb.sum()) == 0.0 and float(zc.sum()) == 0.0
    assert _zeros_like_each(None, None) == [None, None]
    (only,) = _zeros_like_each(a)
    assert only.shape == a.shape and float(only.sum()) == 0.0


def test_device_guard_is_free_without_a_device_index():
    """`_lib.on_device`: a no-op guard for a device without an index (and, on a GPU box, for the current device)."""
    from waldo_amd import _lib
    with _lib.on_device(torch.device("cpu")):
        pass
    t = torch.zeros(3)
    assert _lib.ptr(None) is None and _lib.ptr(t) == t.data_ptr()
