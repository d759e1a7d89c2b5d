"""Pre-flight for `bench.py --gpus N`: the tensor-parallel engine at the REAL Qwen3-8B shard shapes (2 layers of them) for N = 2, 4, 8
on ONE GPU -- N engine instances on N host threads through the in-process communicator (csrc/loopback_comm.hip).  Every rank must
emit the same tokens; they are printed next to the single-GPU engine's (summation order differs: later tokens may part at near-ties).
usage: python tools/tp_shapes_check.py [worlds ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402

omx = omx_import.load_package()
from ominix_mlx_amd import comm, engine  # noqa: E402

cfg = dict(bench.QWEN3_8B)
cfg["num_hidden_layers"] = 2
prompt = bench.prompt_ids(192, cfg["vocab_size"])
N_NEW = 12


def run_world(world):
    if world == 1:
        m = engine.Model(max_context=512, **cfg)
        m.synth_weights()
        toks = [int(m.prefill(prompt))] + [int(t) for t in m.decode(N_NEW)]
        m.close()
        return [toks]
    group = comm.LoopbackGroup(world, 1 << 24)
    models = []
    for r in range(world):
        m = engine.Model(max_context=512, tp_rank=r, tp_size=world, **cfg)
        m.synth_weights()
        m.set_comm(group.rank_comm(r), group.allreduce_fn)
        models.append(m)

    def run(r):
        return [int(models[r].prefill(prompt))] + [int(t) for t in models[r].decode(N_NEW)]

    outs = comm.run_ranks(world, run, group)
    for m in models:
        m.close()
    return outs


ref = run_world(1)[0]
print("tp1 ", ref)
bad = 0
for w in ([int(a) for a in sys.argv[1:]] or (2, 4, 8)):
    outs = run_world(w)
    same = all(o == outs[0] for o in outs)
    agree = next((i for i, (a, b) in enumerate(zip(outs[0], ref)) if a != b), len(ref))
    print(f"tp{w} ", outs[0], "| ranks agree:", same, f"| first {agree} of {len(ref)} tokens equal tp1")
    bad += 0 if same else 1
sys.exit(1 if bad else 0)
