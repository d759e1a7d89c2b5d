"""Per-call decode timing of a TP = 2 Qwen3-8B engine pair, one process per rank on ONE GPU (gloo bootstrap, peer communicator):
which decode calls are slow?  usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/two_rank_windows.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMX_ATTN_OPROJ", "0")
os.environ.setdefault("OMX_PEER_FUSED", "0")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
import bench
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import comm, engine
MODE = os.environ.get("TRW_MODE", "")
if "model_first" in MODE:
    m = engine.Model(max_context=int(os.environ.get('TRW_CTX', '4096')), tp_rank=rank, tp_size=world, **bench.QWEN3_8B)
pc = comm.PeerComm(comm.torch_all_gather_bytes(dist), rank, world)
pc.self_test()
if "cuda_tensor" in MODE:      # bench.py's peer_comm(): a torch CUDA tensor reduced over gloo
    ok = torch.tensor([1], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
if "model_first" not in MODE:
    m = engine.Model(max_context=int(os.environ.get('TRW_CTX', '4096')), tp_rank=rank, tp_size=world, **bench.QWEN3_8B)
m.set_comm(pc.comm, pc.fn)
m.synth_weights()
ids = bench.prompt_ids(2048, bench.QWEN3_8B["vocab_size"])
m.prefill(ids)
if len(sys.argv) > 1 and sys.argv[1] == "reset":      # bench.py's sequence: the same prompt once more on the emptied cache
    m.reset()
    dist.barrier()
    m.prefill(ids)
for n in ((4, 32, 32, 32) if os.environ.get('TRW_CTX') else (4, 32, 32, 8, 32, 64, 32, 16, 16)):
    dist.barrier()
    if "torch_sync" in MODE:
        torch.cuda.synchronize()
    omx.check(omx.lib.omx_synchronize(m.stream()))
    t0 = time.perf_counter()
    m.decode(n)
    omx.check(omx.lib.omx_synchronize(m.stream()))
    dt = time.perf_counter() - t0
    if rank == 0:
        print(f"decode({n:3d}) at offset {m.offset():5d}: {dt * 1e3 / n:7.3f} ms / step  ({m.decode_path()}, device {m.last_decode_ms() / n:.3f} ms / step)", flush=True)
dist.barrier()
m.close(); pc.close()
dist.destroy_process_group()
