"""GPU: the RCCL entry points of the C ABI (m3_ep_unique_id / m3_ep_init / m3_ep_exchange_counts / m3_ep_dispatch /
m3_ep_return / m3_ep_destroy, csrc/ep_rccl.hip) on a ONE-rank communicator - the build box has one GPU and RCCL refuses two
ranks on one device, so what is checked here is the plumbing: librccl opened on demand, communicator life cycle, grouped
send / recv pairs on a side stream with the event hand-back, error paths.  The multi-rank exchange LOGIC (splits, regroup,
chunks) is covered over gloo in tests/test_ep_gloo.py and tests/test_ep_engine_gpu.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_native_exchange_one_rank_round_trip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from m3vit_amd import _lib
    from m3vit_amd.ep_native import NativeExchange
    ex = NativeExchange(rank=0, world=1)
    try:
        counts = torch.tensor([5, 0, 7, 3], dtype=torch.int64, device="cuda")
        assert torch.equal(ex.exchange_counts(counts), counts)                  # one rank: what I route to my experts is what I get
        x = torch.randn(15, 64, device="cuda").half()
        out = torch.full_like(x, float("nan"))
        ex.dispatch_async(out, x, [15], [15]).wait()
        back = torch.full_like(x, float("nan"))
        ex.return_async(back, out, [15], [15]).wait()
        torch.cuda.synchronize()
        assert torch.equal(out, x) and torch.equal(back, x)
        empty = torch.empty(0, 64, device="cuda").half()
        ex.dispatch_async(torch.empty_like(empty), empty, [0], [0]).wait()      # an exchange without rows is a no-op
        # a second communicator beside the first; handles are independent
        ex2 = NativeExchange(rank=0, world=1)
        assert ex2.handle != ex.handle
        ex2.close()
        with pytest.raises(_lib.M3Error):
            _lib.check(_lib.lib().m3_ep_destroy(ex2.handle if ex2.handle >= 0 else 15), "m3_ep_destroy")
    finally:
        ex.close()
    with pytest.raises(AssertionError):
        NativeExchange(rank=0, world=1).dispatch_async(torch.empty(3, 4, device="cuda"), torch.empty(2, 4, device="cuda"), [3], [3])
