"""Oracle value for FLUX.2-klein at its REAL widths (VERDICT r4 "Next" 4b: until round 5 the oracle comparisons ran KleinParams.tiny(),
the real-width test held properties only): hidden 3072, 24 heads of 128, MLP 9216, text width 7680 (klein_model.rs:182-196 defaults) with
ONE double and ONE single block, 128 text tokens + a 16 x 32 latent grid (640 tokens), through oracle/ref_klein.py (numpy, float64
accumulation, one rounding per op output where the reference rounds) on the synthetic weights the device generator builds by name.
Writes tests/golden/klein_fullwidth_pin.npz: the velocity [512, 128] and the largest |value|; tests/test_gpu_klein.py replays it on the
engine (inputs are seeded in the test: not stored).  Build container, a few minutes:   python tools/klein_fullwidth_pin.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_core as rc, ref_klein as rk  # noqa: E402

S_TXT, GRID, TIMESTEP, SEED = 128, (16, 32), 600.0, 11
# round 6 (VERDICT r5 "Next" 6b): `python tools/klein_fullwidth_pin.py flux` runs the SAME blocks at the FLUX 1024^2 sequence -- 512 text
# tokens + a 64 x 64 latent grid = 4 608 tokens, the shape the four-wave flash kernel and the full-chip GEMM tiles run at in the benchmark --
# and keeps every 8th row of the velocity (-> klein_fullwidth_pin_s4608.npz; ~10 minutes and ~12 GB on 8 cores: the float64 scores of one
# attention call are 24 x 4608 x 4608 x 8 bytes)
FLUX = len(sys.argv) > 1 and sys.argv[1] == "flux"
if FLUX:
    S_TXT, GRID = 512, (64, 64)
ROW_STEP = 8 if FLUX else 1


def inputs(p):
    g = np.random.default_rng(SEED)
    latent = rc.bf16_round(g.standard_normal((GRID[0] * GRID[1], p.in_channels)).astype(np.float32))
    txt = rc.bf16_round(g.standard_normal((S_TXT, p.txt_embed_dim)).astype(np.float32))
    return latent, txt


def main():
    p = rk.KleinParams(depth=1, depth_single=1)
    t0 = time.time()
    weights = rk.synth_weights(p)
    print(f"weights: {time.time() - t0:.0f} s ({sum(w.size for w in weights.values()) / 1e6:.0f} M values)", flush=True)
    latent, txt = inputs(p)
    cos, sin = rk.compute_rope(np.concatenate([rk.create_txt_ids(S_TXT), rk.create_img_ids(*GRID)], 0))
    t0 = time.time()
    ref = rk.KleinOracle(p, weights).forward_with_rope(latent, txt, TIMESTEP, cos, sin)
    print(f"oracle forward: {time.time() - t0:.0f} s, max |v| {np.abs(ref).max():.4f}, std {ref.std():.4f}", flush=True)
    out = os.path.join(ROOT, "tests", "golden", "klein_fullwidth_pin_s4608.npz" if FLUX else "klein_fullwidth_pin.npz")
    np.savez_compressed(out, velocity=ref[::ROW_STEP].astype(np.float32), row_step=ROW_STEP, max_abs=np.float32(np.abs(ref).max()), s_txt=S_TXT,
                        grid=np.asarray(GRID), timestep=np.float32(TIMESTEP), seed=SEED)
    print("->", out)


if __name__ == "__main__":
    main()
