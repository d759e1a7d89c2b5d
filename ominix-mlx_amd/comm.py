"""RCCL plumbing for the multi-GPU paths (one process per GPU).

torch ships librccl.so and `torch.distributed` (backend "nccl" == RCCL) is only used to bootstrap:
rank 0's ncclUniqueId is broadcast through it, every rank then creates a raw communicator that the
C++ engines (`omx_qwen3_set_comm`, `omx_klein_set_comm`) and the expert-parallel exchange below drive
on their own HIP streams.  Nothing here touches the data path of a single-GPU run."""
from __future__ import annotations

import ctypes
import os

NCCL_UINT8, NCCL_INT32, NCCL_UINT32, NCCL_FLOAT32, NCCL_BFLOAT16 = 1, 2, 3, 7, 9


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def load_rccl():
    import torch
    return ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), mode=ctypes.RTLD_GLOBAL)


def rccl_comm(dist, rank: int, world: int):
    """-> (librccl handle, ncclComm_t as int, address of ncclAllReduce).  `dist` is an initialised
    torch.distributed module (or None for a 1-rank communicator)."""
    import torch
    lib = load_rccl()
    uid = UniqueId()
    if rank == 0:
        lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
        assert lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).cuda()
    if dist is not None:
        dist.broadcast(t, 0)
    ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
    comm = ctypes.c_void_p()
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rc = lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank)
    if rc != 0:
        raise RuntimeError(f"ncclCommInitRank failed with {rc}")
    # one eager collective now, so that the communicator's lazy set-up (channels, peer mappings) is
    # finished before an engine first meets it inside a stream capture
    warm = torch.ones(64, dtype=torch.float32, device="cuda")
    lib.ncclAllReduce.restype = ctypes.c_int
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p]
    rc = lib.ncclAllReduce(warm.data_ptr(), warm.data_ptr(), 64, NCCL_FLOAT32, 0, comm, None)
    torch.cuda.synchronize()
    if rc != 0 or float(warm[0].item()) != float(world):
        raise RuntimeError(f"RCCL warm-up all-reduce failed (rc {rc}, got {float(warm[0].item())}, want {world})")
    fn = ctypes.cast(lib.ncclAllReduce, ctypes.c_void_p).value
    return lib, comm.value, fn


PEER_SIGNATURES = {
    "omx_peer_comm_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "omx_peer_comm_handle": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "omx_peer_comm_connect": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "omx_peer_allreduce": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "omx_peer_allreduce_fn": (ctypes.c_void_p, []),
    "omx_peer_comm_stage_bytes": (ctypes.c_size_t, [ctypes.c_void_p]),
    "omx_peer_comm_counts": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]),
    "omx_peer_moe_combine": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "omx_peer_comm_device": (ctypes.c_void_p, [ctypes.c_void_p]),
    "omx_peer_comm_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]),
    "omx_peer_comm_set_scope": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    "omx_peer_comm_scope": (ctypes.c_int, [ctypes.c_void_p]),
    "omx_peer_device_id": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "omx_peer_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
}


def lib_error() -> str:
    from . import lib
    lib.omx_last_error.restype = ctypes.c_char_p
    m = lib.omx_last_error()
    return m.decode(errors="replace") if m else ""


class PeerComm:
    """One-shot all-reduce over xGMI peer stores (csrc/peer_allreduce.hip) for the small reductions of the tensor-parallel decode step;
    larger calls go to `rccl` = (lib, comm, fn) of rccl_comm() when given.  `all_gather_bytes(b: bytes) -> list[bytes]` is the host-side
    bootstrap (torch.distributed on any backend): the 64-byte IPC handles of the ranks' inboxes are exchanged through it once.

    .comm / .fn are what `Model.set_comm` takes.  `self_test()` reduces seeded vectors and compares with the rank-ordered sum every
    rank can compute locally -- call it before trusting the fabric (bounded waits: a failure raises, it does not hang)."""

    def __init__(self, all_gather_bytes, rank: int, world: int, rccl=None):
        from . import lib
        for name, (res, args) in PEER_SIGNATURES.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        self.rank, self.world, self._rccl = rank, world, rccl
        # every step that can fail on ONE rank is followed by an exchange, so that all ranks raise together instead of one of
        # them leaving the others inside a collective
        h, err = ctypes.c_void_p(), b""
        mine = ctypes.create_string_buffer(64)
        if lib.omx_peer_comm_create(ctypes.byref(h), rank, world, rccl[1] if rccl else None, rccl[2] if rccl else None) != 0 or \
                lib.omx_peer_comm_handle(h, mine) != 0:
            err = (lib_error() or "omx_peer_comm_create failed").encode()
        self._h = h if not err else None
        # a fixed-size record per rank -- 1 status byte + 64 payload bytes -- so that an error text can never be mistaken for a
        # handle (an error of exactly 63 bytes used to be): status 0 = the 64-byte IPC handle, 1 = an error message
        rec = (b"\x01" + err[:64].ljust(64, b" ")) if err else (b"\x00" + mine.raw)
        recs = all_gather_bytes(rec)
        bad = [f"rank {r}: {b[1:].decode(errors='replace').strip()}" for r, b in enumerate(recs) if len(b) != 65 or b[:1] != b"\x00"]
        handles = [b[1:] for b in recs]
        if bad:
            self.close()
            raise RuntimeError("peer all-reduce: inbox creation failed on " + "; ".join(bad))
        err = b""
        if world > 1 and lib.omx_peer_comm_connect(h, b"".join(handles)) != 0:
            err = (lib_error() or "omx_peer_comm_connect failed").encode()
        bad = [f"rank {r}: {b.decode(errors='replace')}" for r, b in enumerate(all_gather_bytes(err)) if b]
        if bad:
            self.close()
            raise RuntimeError("peer all-reduce: mapping the peers' inboxes failed on " + "; ".join(bad))
        self.comm = h.value
        self.fn = lib.omx_peer_allreduce_fn()
        # hand-off scope of the two-shot / exchange path (csrc/peer_allreduce.hip large_publish / large_wait): system-scope release /
        # acquire unless EVERY rank sits on the same device (the one-GPU pre-flight, where a "peer" stage is local HBM and the
        # agent-scope form cannot fail) or OMX_PEER_SCOPE says otherwise.  Ranks on different GPUs never get the agent form by default:
        # it is unproven across xGMI.
        dev = ctypes.create_string_buffer(64)
        ids = all_gather_bytes(dev.value if lib.omx_peer_device_id(dev, 64) == 0 else b"?" + str(rank).encode())
        self.same_device = world > 1 and len(set(ids)) == 1 and not ids[0].startswith(b"?")
        if "OMX_PEER_SCOPE" not in os.environ:
            # (checked: a failed upload of the device table would leave the kernels on the old scope while the host field reports the new one)
            if lib.omx_peer_comm_set_scope(h, 0 if self.same_device else 1) != 0:
                from . import OmxError
                raise OmxError("peer communicator: setting the memory scope failed: " + lib_error())
        self.scope = "system" if lib.omx_peer_comm_scope(h) == 1 else "agent"
        self._gather = all_gather_bytes

    def counts(self) -> dict:
        """launches issued through this communicator by path (host-side counters)."""
        from . import check, lib
        v = (ctypes.c_ulonglong * 4)()
        check(lib.omx_peer_comm_counts(self._h, v))
        return {"one_shot": int(v[0]), "two_shot": int(v[1]), "moe_combine": int(v[2]), "rccl": int(v[3])}

    def aborted(self) -> bool:
        from . import check, lib
        v = ctypes.c_uint(0)
        check(lib.omx_peer_comm_status(self._h, ctypes.byref(v)))
        return bool(v.value)

    def allreduce_f32(self, t):
        """in place on a device Tensor of float32 (tests, self-test)."""
        from . import lib
        rc = lib.omx_peer_allreduce(t.ptr, t.ptr, t.size, NCCL_FLOAT32, 0, self._h, None)
        if rc != 0:
            raise RuntimeError("omx_peer_allreduce failed (unsupported call and no RCCL communicator behind it)")
        return t

    def self_test(self, rounds: int = 6, n: int = 4096, large: bool = True):
        """Every rank derives ALL ranks' inputs from (round, rank) seeds, so the expected rank-ordered f32 sum is known locally.
        One-shot path: `rounds` reductions of n floats.  With the two-shot / exchange path on (stage_bytes > 0) and `large`: seeded
        1 MB .. 32 MB messages in f32 and bf16 (chunked when the stage is smaller), each with one deliberately LATE rank and a
        streaming kernel queued right behind the call, and one round of the MoE combine kernel -- all before the first real use of those
        kernels.  The verdict is agreed over the bootstrap group: every rank raises when any rank failed."""
        import numpy as np
        from .ops import Tensor, synchronize
        err = ""
        try:
            for it in range(rounds):
                parts = [np.random.default_rng(1000 * it + r).standard_normal(n).astype(np.float32) for r in range(self.world)]
                want = parts[0].copy()
                for r in range(1, self.world):
                    want = want + parts[r]
                t = Tensor.from_numpy(parts[self.rank], "f32")
                self.allreduce_f32(t)
                synchronize()
                if self.aborted():
                    raise RuntimeError(f"peer all-reduce: rank {self.rank} gave up waiting for a peer in round {it}")
                got = t.numpy()
                if not np.array_equal(got, want):
                    raise RuntimeError(f"peer all-reduce: rank {self.rank} round {it}: {int((got != want).sum())} of {n} sums differ from the rank-ordered sum")
            if large and self.world > 1 and self.stage_bytes() > 0:
                self._self_test_large()
        except Exception as e:   # noqa: BLE001
            err = str(e) or type(e).__name__
        bad = [f"rank {r}: {b.decode(errors='replace')}" for r, b in enumerate(self._gather(err.encode()[:200])) if b]
        if bad:
            raise RuntimeError("peer communicator self-test failed on " + "; ".join(bad))
        return True

    def stage_bytes(self) -> int:
        from . import lib
        return int(lib.omx_peer_comm_stage_bytes(self._h))

    def _self_test_large(self):
        import time
        import numpy as np
        from . import lib
        from .ops import Tensor, fill_uniform, synchronize

        def bf16_round(x):
            u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
            return ((u + ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)) & np.uint64(0xFFFF0000)).astype(np.uint32).view(np.float32)

        stage = self.stage_bytes()
        # (elements, dtype): 1 MB f32, 4 MB bf16 with a ragged tail, 32 MB f32; and one message LARGER than the stage (chunked) when that is
        # affordable (a 64 MB default stage would need > 64 MB messages: covered by tests that shrink the stage with OMX_PEER_STAGE_MB)
        cases = [(262144, "f32"), (2 * 1024 * 1024 + 8, "bf16"), (8 * 1024 * 1024, "f32")]
        if stage <= (16 << 20):
            cases.append((stage // 4 + stage // 8 + 4, "f32"))
            cases.append((stage // 2 + 4096 + 8, "bf16"))
        for it, (n, dt) in enumerate(cases):
            parts = [np.random.default_rng(7000 + 16 * it + r).standard_normal(n).astype(np.float32) for r in range(self.world)]
            if dt == "bf16":
                parts = [bf16_round(p) for p in parts]
            want = parts[0].copy()
            for p in parts[1:]:
                want = want + p
            if dt == "bf16":
                want = bf16_round(want)
            buf = Tensor.from_numpy(parts[self.rank], dt)
            synchronize()
            if self.rank == it % self.world:
                time.sleep(0.05)                                   # the deliberately late rank: its peers poll meanwhile
            rc = lib.omx_peer_allreduce(buf.ptr, buf.ptr, n, NCCL_BFLOAT16 if dt == "bf16" else NCCL_FLOAT32, 0, self._h, None)
            fill_uniform((64 * 1024 * 1024,), 99 + it, 1.0)        # a streaming kernel queued right behind it (HBM busy while peers still read)
            synchronize()
            if rc != 0 or self.aborted():
                raise RuntimeError(f"two-shot all-reduce of {n} {dt}: rc {rc}, aborted {self.aborted()}")
            got = buf.numpy().astype(np.float32).ravel()
            if not np.array_equal(got, want):
                raise RuntimeError(f"two-shot all-reduce of {n} {dt} ({self.scope} scope): {int((got != want).sum())} sums differ from the rank-ordered sum")
        # one round of the expert-parallel combine (peer_moe_combine_kernel): T tokens x top-2 over 2 * world experts, every rank holds ALL slot
        # rows (seeded) and contributes those of ITS experts; expected rows: bf16(resid + bf16(sum in slot order of bf16(y_j * score_j)))
        T_, hid, k = 96 * self.world + 5, 512, 2
        rng = np.random.default_rng(4242)
        y = bf16_round(rng.standard_normal((T_ * k, hid)).astype(np.float32))
        sc = bf16_round(rng.uniform(0.1, 0.9, T_ * k).astype(np.float32))
        resid = bf16_round(rng.standard_normal((T_, hid)).astype(np.float32))
        inds = rng.integers(0, 2 * self.world, T_ * k).astype(np.uint32)
        prod = bf16_round(y * sc[:, None]).reshape(T_, k, hid)
        acc = prod[:, 0].copy()
        for j in range(1, k):
            acc = acc + prod[:, j]
        want = bf16_round(resid + bf16_round(acc))

        class Slots(ctypes.Structure):
            _fields_ = [("y", ctypes.c_void_p), ("pos_of_slot", ctypes.c_void_p), ("inds", ctypes.c_void_p), ("scores", ctypes.c_void_p)]

        ty, tp, ti, ts = (Tensor.from_numpy(y, "bf16"), Tensor.from_numpy(np.arange(T_ * k, dtype=np.uint32), "u32"), Tensor.from_numpy(inds, "u32"),
                          Tensor.from_numpy(sc, "bf16"))
        tr, out = Tensor.from_numpy(resid, "bf16"), Tensor((T_, hid), "bf16")
        sl = Slots(ty.ptr, tp.ptr, ti.ptr, ts.ptr)
        synchronize()
        if self.rank == self.world - 1:
            time.sleep(0.05)
        rc = lib.omx_peer_moe_combine(out.ptr, tr.ptr, ctypes.byref(sl), T_, hid, k, 2 * self.rank, 2, self._h, None)
        synchronize()
        if rc != 0 or self.aborted():
            raise RuntimeError(f"MoE combine exchange: rc {rc}, aborted {self.aborted()}")
        got = out.numpy().astype(np.float32)
        if not np.array_equal(got, want):
            raise RuntimeError(f"MoE combine exchange ({self.scope} scope): {int((got != want).any(axis=1).sum())} of {T_} rows differ from the slot-ordered sum")

    def close(self):
        from . import lib
        if self._h:
            lib.omx_peer_comm_destroy(self._h)
            self._h = None


def torch_all_gather_bytes(dist):
    """bootstrap for PeerComm over an initialised torch.distributed (gloo or nccl)."""
    def gather(b: bytes):
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, b)
        return out
    return gather


class RcclExchange:
    """all-to-all(v) of device rows as one ncclGroup of point-to-point sends/receives: xGMI is a
    point-to-point fabric, so the exchange is exactly one transfer per peer pair and direction."""

    def __init__(self, lib, comm: int, rank: int, world: int, stream: int = 0):
        self.lib, self.comm, self.rank, self.world, self.stream = lib, ctypes.c_void_p(comm), rank, world, ctypes.c_void_p(stream)
        for f in (lib.ncclSend, lib.ncclRecv):
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib.ncclGroupStart.restype = lib.ncclGroupEnd.restype = ctypes.c_int

    def counts(self, send_counts):
        """all-to-all of one int32 per peer (host lists in, host list out)."""
        import numpy as np
        from .ops import Tensor, synchronize
        s = Tensor.from_numpy(np.asarray(send_counts, np.uint32), "u32")
        r = Tensor((self.world,), "u32")
        self._group([(s.ptr + 4 * p, 1, p) for p in range(self.world)], [(r.ptr + 4 * p, 1, p) for p in range(self.world)], NCCL_UINT32)
        synchronize()
        return [int(v) for v in r.numpy()]

    def rows(self, send, send_counts, recv_counts, row_bytes: int):
        """send: device Tensor whose rows are grouped by destination rank -> device Tensor of received rows."""
        from .ops import Tensor
        recv = Tensor((max(1, sum(recv_counts)) * row_bytes,), "u8")
        so = ro = 0
        sends, recvs = [], []
        for p in range(self.world):
            if send_counts[p]:
                sends.append((send.ptr + so * row_bytes, send_counts[p] * row_bytes, p))
            if recv_counts[p]:
                recvs.append((recv.ptr + ro * row_bytes, recv_counts[p] * row_bytes, p))
            so += send_counts[p]
            ro += recv_counts[p]
        self._group(sends, recvs, NCCL_UINT8)
        return recv

    def _group(self, sends, recvs, dtype):
        lib = self.lib
        assert lib.ncclGroupStart() == 0
        for ptr, n, peer in sends:
            rc = lib.ncclSend(ctypes.c_void_p(ptr), n, dtype, peer, self.comm, self.stream)
            assert rc == 0, f"ncclSend -> {rc}"
        for ptr, n, peer in recvs:
            rc = lib.ncclRecv(ctypes.c_void_p(ptr), n, dtype, peer, self.comm, self.stream)
            assert rc == 0, f"ncclRecv -> {rc}"
        assert lib.ncclGroupEnd() == 0


LOOPBACK_SIGNATURES = {
    "omx_loopback_create": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_size_t]),
    "omx_loopback_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "omx_loopback_rank_comm": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int]),
    "omx_loopback_abort": (ctypes.c_int, [ctypes.c_void_p]),
}


class LoopbackGroup:
    """In-process stand-in for a communicator (csrc/loopback_comm.hip): `world` engines driven by `world` host
    threads of this process all-reduce through it -- tensor parallelism with real shards on one GPU (tests)."""

    def __init__(self, world: int, max_bytes: int):
        from . import check, lib
        for name, (res, args) in LOOPBACK_SIGNATURES.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        self._lib, self.world = lib, world
        self._h = ctypes.c_void_p()
        check(lib.omx_loopback_create(ctypes.byref(self._h), world, max_bytes))
        self.allreduce_fn = ctypes.cast(lib.omx_loopback_allreduce, ctypes.c_void_p).value

    def rank_comm(self, rank: int) -> int:
        return self._lib.omx_loopback_rank_comm(self._h, rank)

    def abort(self) -> None:
        self._lib.omx_loopback_abort(self._h)

    def close(self) -> None:
        if self._h.value:
            self._lib.omx_loopback_destroy(self._h)
            self._h = ctypes.c_void_p()


def run_ranks(world: int, fn, group: "LoopbackGroup" = None):
    """fn(rank) on `world` host threads (ctypes releases the GIL inside library calls); re-raises the first error
    after releasing the peers of a failed rank."""
    import threading
    results, errors = [None] * world, [None] * world

    def body(r):
        try:
            results[r] = fn(r)
        except BaseException as e:   # noqa: BLE001
            errors[r] = e
            if group is not None:
                group.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return results
