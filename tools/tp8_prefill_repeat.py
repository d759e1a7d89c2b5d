"""Does a tensor-parallel rank's batched prompt pass repeat itself?  `world` ranks of Qwen3-8B as host threads on ONE GPU (loopback communicator),
the 2 048-token prompt three times on an emptied cache: the first token and the last-row logits must not change.  usage: [world] (default 8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import comm, engine
import bench

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = dict(bench.MODELS["qwen3-8b"])
prompt = bench.prompt_ids(2048, cfg["vocab_size"])
group = comm.LoopbackGroup(world, 2048 * cfg["hidden_size"] * 4)
models = []
for r in range(world):
    m = engine.Model(max_context=2048 + 64, tp_rank=r, tp_size=world, **cfg)
    m.synth_weights()
    m.set_comm(group.rank_comm(r), group.allreduce_fn)
    models.append(m)

def run(r):
    m = models[r]
    out = []
    for rep in range(3):
        m.reset()
        tok = int(m.prefill(prompt))
        out.append((tok, m.last_logits().copy()))
    return out

outs = comm.run_ranks(world, run, group)
for rep in range(3):
    toks = [outs[r][rep][0] for r in range(world)]
    same = all(np.array_equal(outs[r][rep][1], outs[r][0][1]) for r in range(world))
    print(f"pass {rep}: tokens {sorted(set(toks))} logits equal to pass 0 on every rank: {same}", flush=True)
for m in models:
    m.close()
group.close()
