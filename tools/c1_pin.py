"""BASELINE configs[0] at its real shapes: Qwen3-0.6B -- 28 layers, 1024 hidden, 16 / 8 heads of 128, 3072 FFN, tied
151 936-entry vocabulary -- 128-token synthetic prompt, greedy, 32 new tokens, on the numpy oracle with the synthetic weights the engine
generates on the device.  Writes tests/golden/qwen3_0p6b_c1_pin.npz: the 32 token ids, per-step top-1 / top-2 margin, largest |logit|
and top-8 (index, logit) pairs.  tests/test_gpu_fullsize_pin.py replays it on the engine.
Runs in the build container (a few minutes, ~8 GB):   python tools/c1_pin.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_qwen3 as rq, synth  # noqa: E402


def main():
    cfg = rq.Qwen3Config.qwen3_0_6b()
    t0 = time.time()
    weights = rq.synth_weights(cfg)
    print(f"weights: {time.time() - t0:.0f} s", flush=True)
    prompt = synth.prompt_ids(128, cfg.vocab_size)
    t0 = time.time()
    tokens, logits = rq.Qwen3Oracle(cfg, weights).generate(prompt, 32, return_logits=True)
    print(f"oracle generate: {time.time() - t0:.0f} s", flush=True)
    srt = np.sort(logits, axis=1)
    order = np.argsort(-logits, axis=1, kind="stable")[:, :8]                       # every step's top-8
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "qwen3_0p6b_c1_pin.npz"), prompt=prompt, tokens=tokens.astype(np.uint32),
                        margin=(srt[:, -1] - srt[:, -2]).astype(np.float32), max_abs=np.abs(logits).max(axis=1).astype(np.float32),
                        top_idx=order.astype(np.uint32), top_val=np.take_along_axis(logits, order, axis=1).astype(np.float32))
    print("tokens", tokens.tolist())
    print("margins", np.round(srt[:, -1] - srt[:, -2], 3).tolist())


if __name__ == "__main__":
    main()
