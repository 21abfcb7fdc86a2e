"""CPU: host-side logic of the wrappers that needs no kernel (index bookkeeping, the raw-slot hand-over rules)."""
import torch


def test_private_index_copy_is_made_once_under_inference_mode():
    """under torch.inference_mode() an index tensor has no version counter, so the wrappers work on a
    private copy -- ONE copy per decode: normalising the copy again (decode_output, _flow_common and input_to_output
    each do) hands the same object back instead of cloning and reading it back again."""
    from waldo_amd import functional as WF
    with torch.inference_mode():
        ts = torch.arange(6).view(1, 2, 3)
        first = WF.normalise_time_index(ts)
        assert first is not ts and torch.equal(first, ts)
        again = WF.normalise_time_index(first)
        assert again is first
        assert WF.normalise_time_index(again) is first
    # with a version counter nothing is copied at all
    ts2 = torch.arange(4).view(1, 1, 4)
    assert WF.normalise_time_index(ts2) is ts2


def test_raw_slots_short_way_needs_a_version_counter_or_a_vouching_caller():
    """RawSlots.still_describes cannot see a write to an inference tensor (no version counter): such a view
    takes the short way through frame_warp_fuse only when the caller that kept it in its hands vouches for it."""
    from waldo_amd.functional import RawSlots
    raw = torch.zeros(1, 2, 3, 5, 4, 4)
    view = raw[:, :, :3, 2:].permute(0, 2, 1, 3, 4, 5)
    slots = RawSlots(raw, torch.zeros(1, 3, 2, 4, 4), 2, False, view)
    assert slots.still_describes(view)
    view.add_(1.0)  # a write bumps the version: the score sums would be stale
    assert not slots.still_describes(view)
    with torch.inference_mode():
        raw_i = torch.zeros(1, 2, 3, 5, 4, 4)
        view_i = raw_i[:, :, :3, 2:].permute(0, 2, 1, 3, 4, 5)
        slots_i = RawSlots(raw_i, torch.zeros(1, 3, 2, 4, 4), 2, False, view_i)
        assert not slots_i.still_describes(view_i)  # nobody vouches: the copying kernel
        slots_i.vouched = True
        assert slots_i.still_describes(view_i)
        assert not slots_i.still_describes(view_i[:, 1:])  # another view


def test_index_bookkeeping_knows_an_identity_index_and_forgets_it_after_a_write():
    """`time_gather` hands out the clip itself for `x[:, ts]` with ts = 0 .. n-1 (LVD ctx_mode "prev"): the wrappers
    learn that from ONE read per (tensor, version) -- `_index_info` -- and `_known_arange` only ever answers from that
    read (it is asked during HIP-graph capture, where nothing may be read back)."""
    from waldo_amd import functional as WF
    ts = torch.arange(5)
    assert not WF._known_arange(ts)                      # never read: not known
    assert WF._index_info(ts) == (0, 4, True) and WF._known_arange(ts)
    assert WF._index_range(ts) == (0, 4)
    ts[0] = 3                                            # a write bumps the version: read again, no longer the identity
    assert not WF._known_arange(ts)
    assert WF._index_info(ts) == (1, 4, False) and not WF._known_arange(ts)
    assert WF._index_info(torch.arange(2, 6)) == (2, 5, False)            # a run that does not start at 0
    assert WF._index_info(torch.arange(6).view(1, 2, 3)) == (0, 5, False)  # only 1-D indices select whole frames
    with torch.inference_mode():
        ti = WF.normalise_time_index(torch.arange(4))    # private copy, read once
        assert WF._known_arange(ti) and WF._index_info(ti) == (0, 3, True)


def test_zero_filled_gradients_come_out_of_one_buffer():
    """`_zeros_like_each`: the small tensors a backward kernel accumulates into share ONE zero-filled buffer (one fill
    launch); every view starts on a 256-byte boundary and keeps its tensor's shape; None stays None."""
    from waldo_amd.functional import _zeros_like_each
    a, b, c = torch.ones(3, 5), torch.ones(7), torch.ones(2, 2, 2)
    za, none, zb, zc = _zeros_like_each(a, None, b, c)
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
