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
