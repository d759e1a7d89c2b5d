"""Host side of the peer-store communicator (ominix-mlx_amd/comm.py PeerComm) without a GPU: a rank whose inbox cannot be created
must not leave its peers inside the handle exchange -- it takes part with an error record and EVERY rank raises."""
import pytest


def test_failed_inbox_creation_raises_on_every_rank(omx):
    if omx.device_count() > 0:
        pytest.skip("GPU present: creation succeeds (tests/test_gpu_qwen3.py covers the working path)")
    from ominix_mlx_amd import comm
    seen = []

    def gather(b):                       # a 2-rank exchange in which the peer's creation worked (64-byte handle)
        seen.append(b)
        return [b, bytes(64)]

    with pytest.raises(RuntimeError, match="inbox creation failed on rank 0"):
        comm.PeerComm(gather, 0, 2)
    assert len(seen) == 1 and seen[0].startswith(b"!")      # the failing rank still took part in the exchange


def test_peer_failure_is_reported_on_a_healthy_rank(omx, monkeypatch):
    """The healthy rank's view: its own creation is stubbed to succeed, the peer's record is an error."""
    from ominix_mlx_amd import comm, lib
    import ctypes

    class FakeLib:
        def __getattr__(self, name):
            real = getattr(lib, name)
            if name == "omx_peer_comm_create":
                def create(out, rank, world, a, b):
                    return 0
                create.restype, create.argtypes = None, None
                return create
            if name == "omx_peer_comm_handle":
                def handle(h, buf):
                    ctypes.memmove(buf, b"\\x01" * 64, 64)
                    return 0
                return handle
            if name == "omx_peer_comm_destroy":
                return lambda h: 0
            return real

    import ominix_mlx_amd
    monkeypatch.setattr(ominix_mlx_amd, "lib", FakeLib())
    with pytest.raises(RuntimeError, match="rank 1: no device"):
        comm.PeerComm(lambda b: [b, b"!no device"], 0, 2)
