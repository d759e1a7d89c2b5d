import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
import bench, omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
cfg = dict(bench.QWEN3_8B); cfg["quantization"] = {"bits": 4, "group_size": 64}
ids = bench.prompt_ids(2048, cfg["vocab_size"])
m = engine.Model(max_context=2048 + 64, **cfg); m.synth_weights()
first = int(m.prefill(ids)); lg = m.last_logits()
r = m.per_op_route_forced(ids, [first], [0, 1])
d = np.abs(r["logits"][0] - lg)
top2 = np.sort(lg)[-2:]
print("engine first", first, "route first", int(r["tokens"][0]), "max|dlogit|", float(d.max()), "absmax", float(np.abs(lg).max()), "engine margin", float(top2[1] - top2[0]),
      "route logit at engine token", float(r["logits"][0][first]), "route max", float(r["logits"][0].max()))
tok1 = int(m.decode(1)[0]); lg1 = m.last_logits()
print("step 1: engine", tok1, "route", int(r["tokens"][1]), "max|dlogit|", float(np.abs(r["logits"][1] - lg1).max()))
