"""Host side of the peer-store communicator (ominix-mlx_amd/comm.py PeerComm) without a GPU: a rank whose inbox cannot be created
must not leave its peers inside the handle exchange -- it takes part with an error record and EVERY rank raises."""
import pytest


def test_failed_inbox_creation_raises_on_every_rank(omx):
    if omx.device_count() > 0:
        pytest.skip("GPU present: creation succeeds (tests/test_gpu_qwen3.py covers the working path)")
    from ominix_mlx_amd import comm
    seen = []

    def gather(b):                       # a 2-rank exchange in which the peer's creation worked (status 0 + 64-byte handle)
        seen.append(b)
        return [b, b"\x00" + bytes(64)]

    with pytest.raises(RuntimeError, match="inbox creation failed on rank 0"):
        comm.PeerComm(gather, 0, 2)
    assert len(seen) == 1 and seen[0][:1] == b"\x01" and len(seen[0]) == 65      # the failing rank still took part: status 1 + message


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
        comm.PeerComm(lambda b: [b, b"\x01" + b"no device".ljust(64)], 0, 2)


def test_error_text_of_any_length_is_never_taken_for_a_handle(omx):
    """the exchange used to tell an error from a handle by `len(b) != 64` -- an error of exactly 63 characters after the
    marker passed for a handle.  Records are fixed-size now (status byte + 64 payload bytes)."""
    if omx.device_count() > 0:
        pytest.skip("GPU present: creation succeeds")
    from ominix_mlx_amd import comm
    err63 = b"\x01" + (b"e" * 63).ljust(64)
    with pytest.raises(RuntimeError, match="inbox creation failed on rank 0.*rank 1"):
        comm.PeerComm(lambda b: [b, err63], 0, 2)
