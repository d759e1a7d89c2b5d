"""Decode step time against the decode attention's split count (OMX_ATTN_STEP_SPLITS; default 256 / Hkv = 32 at Qwen3-8B): one engine per value
(the plan is made when the graph is built), 2 048-token prompt, 3 x 64 tokens timed by the host around decode()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
cfg = dict(bench.QWEN3_8B)
prompt = bench.prompt_ids(2048, cfg["vocab_size"])
vals = [int(v) for v in sys.argv[1:]] or [8, 16, 24, 32, 40, 48]
for rnd in range(2):
    for v in vals:
        os.environ["OMX_ATTN_STEP_SPLITS"] = str(v)
        m = engine.Model(max_context=2048 + 512, **cfg)
        m.synth_weights()
        first = m.prefill(prompt)
        m.decode(16)
        best = 1e9
        for rep in range(3):
            omx.ops.synchronize(); t0 = time.perf_counter()
            toks = m.decode(64)
            omx.ops.synchronize(); best = min(best, (time.perf_counter() - t0) / 64)
        print(f"splits {v:3d}: {best * 1e3:.4f} ms / token  {1 / best:7.1f} tok/s   first {int(first)} path {m.decode_path()}", flush=True)
        m.close()
