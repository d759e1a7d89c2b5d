"""A/B of the 4-bit decode step with the O projection in the attention launch (OMX_ATTN_OPROJ=1, default) and as its own packed GEMV
(=0): Qwen3-8B shapes as an MLX 4-bit group-64 checkpoint, 2048-token prompt, interleaved timed rounds, tokens compared.
Any other launch-time switch can be compared the same way: OMX_AB_VAR=<environment variable> OMX_AB_VALUES=<a>,<b> (e.g. OMX_QGEMV_DEPTH 2,6).
usage: python tools/q4_step_ab.py [prompt] [steps] [rounds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

CFG = dict(hidden_size=4096, num_hidden_layers=36, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8, head_dim=128,
           vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)


def main():
    n_prompt = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rng = np.random.default_rng(0)
    prompt = rng.integers(0, CFG["vocab_size"], n_prompt).astype(np.uint32)
    var = os.environ.get("OMX_AB_VAR", "OMX_ATTN_OPROJ")
    va, vb = os.environ.get("OMX_AB_VALUES", "0,1").split(",")
    models, toks = {}, {}
    for mode in ("0", "1"):
        os.environ[var] = (va, vb)[int(mode)]
        m = engine.Model(max_context=n_prompt + 16 + steps * rounds + 16, quantization={"bits": 4, "group_size": 64}, **CFG)
        m.synth_weights()
        toks[mode] = [int(m.prefill(prompt))] + [int(t) for t in m.decode(16)]
        models[mode] = m
    print("tokens equal:", toks["0"] == toks["1"], toks["1"][:6])
    best = {"0": 0.0, "1": 0.0}
    for r in range(rounds):
        for mode in ("0", "1"):
            os.environ[var] = (va, vb)[int(mode)]
            m = models[mode]
            t0 = time.perf_counter()
            out = m.decode(steps)
            omx.check(omx.lib.omx_synchronize(m.stream()))
            dt = time.perf_counter() - t0
            toks[mode] += [int(t) for t in out]
            best[mode] = max(best[mode], steps / dt)
            print(f"round {r} {var}={(va, vb)[int(mode)]}: {steps / dt:.1f} tok/s")
    print("all tokens equal:", toks["0"] == toks["1"])
    print(f"best: {var}={va} {best['0']:.1f} tok/s, {var}={vb} {best['1']:.1f} tok/s")


if __name__ == "__main__":
    main()
