"""Generates the committed golden fixtures (tests/golden/*.npz, *.json).

The reference cannot be built or imported in this environment (Rust toolchain absent, MLX core not
vendored -- DESIGN.md section 2), so fixtures are of two kinds:
  * kats.json        -- numbers COPIED from the reference's own tests (file:line cited per entry): the seeded
                        input statistics and expected outputs this repo's oracle must reproduce;
  * *.npz            -- inputs + outputs of the pinned oracle on seeded inputs for the rows the reference has
                        no value test for (SDPA, KV cache, MoE, decode, mel frontend, DiT); they freeze the
                        oracle (regression) and give the GPU tests device-independent vectors.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_audio as ra, ref_core as rc, ref_klein as rk, ref_moe as rm, ref_qwen3 as rq, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).uniform(-1, 1, size=shape) * scale).astype(np.float32)


def main():
    kats = {
        "rope_seed71": {"ref": "mlx-rs/src/fast.rs:232-250", "input_mean": 0.5082664489746094, "input_sum": 130.1162109375,
                        "mean": 0.45625377, "sum": 116.800964, "args": {"dims": 8, "traditional": False, "base": 10000.0}},
        "rms_norm_seed103": {"ref": "mlx-rs/src/fast.rs:254-273", "mean": 0.87293875, "sum": 223.47232, "eps": 1e-5},
        "layer_norm_seed635": {"ref": "mlx-rs/src/fast.rs:277-298", "mean_col0": 0.29099038, "sum_col0": 4.655846, "eps": 1e-5},
        "silu_seed22": {"ref": "mlx-rs/src/nn/activation.rs:1291-1320", "input_mean": 0.5029706, "input_sum": 128.76047,
                        "mean": 0.33197093, "sum": 84.98456},
        "softmax_seed853": {"ref": "mlx-rs/src/nn/activation.rs:1156-1180", "input_mean": 0.5143963, "input_sum": 131.68546,
                            "mean": 0.062499996, "sum": 15.999999},
        "linear_seed744": {"ref": "mlx-rs/src/nn/linear.rs:224-252", "input_mean": 0.50868857, "input_sum": 130.22427,
                           "mean": 0.10419309, "sum": 8.335447},
        "matmul": {"ref": "mlx-rs/src/ops/arithmetic.rs:1921-1937", "a": [[1, 2], [3, 4]], "b": [[-5, 37.5, 4], [7, 1, 0]],
                   "out": [9, 39.5, 4, 13, 116.5, 12]},
        "euler_step": {"ref": "flux-klein-mlx/src/sampler.rs:390-407", "latent": [1.0, 2.0], "v": [0.5, 0.5], "t": [1.0, 0.75],
                       "out": [0.875, 1.875]},
        "stft_rel_l2_bound": {"ref": "funasr-mlx/examples/validate_correctness.rs:284-287", "bound": 1e-5},
    }
    json.dump(kats, open(os.path.join(OUT, "kats.json"), "w"), indent=1)

    # SDPA decode + prefill (bf16 grid values stored as float32)
    q = rc.bf16_round(rand((1, 8, 1, 64), 1)); k = rc.bf16_round(rand((1, 2, 150, 64), 2)); v = rc.bf16_round(rand((1, 2, 150, 64), 3))
    qp = rc.bf16_round(rand((1, 4, 40, 64), 4)); kp = rc.bf16_round(rand((1, 2, 72, 64), 5)); vp = rc.bf16_round(rand((1, 2, 72, 64), 6))
    np.savez_compressed(os.path.join(OUT, "sdpa.npz"), q=q, k=k, v=v, scale=np.float32(0.125),
                        out_decode=rc.scaled_dot_product_attention(q, k, v, 0.125, None, "bf16"),
                        qp=qp, kp=kp, vp=vp, mask=rc.create_causal_mask(40, 32),
                        out_prefill=rc.scaled_dot_product_attention(qp, kp, vp, 0.125, rc.create_causal_mask(40, 32), "bf16"))

    # tiny Qwen3: 128-token synthetic prompt, 12 greedy tokens, first/last-step logits
    cfg = rq.Qwen3Config(512, 2, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False)
    oracle = rq.Qwen3Oracle(cfg, rq.synth_weights(cfg))
    prompt = synth.prompt_ids(128, cfg.vocab_size)
    toks, logits = oracle.generate(prompt, 12, return_logits=True)
    np.savez_compressed(os.path.join(OUT, "qwen3_tiny.npz"), config=np.array([512, 2, 1536, 8, 4, 64, 2048]), prompt=prompt,
                        tokens=toks, logits_first=logits[0], logits_last=logits[-1], margins=rc.argmax_margin(logits))

    # MoE block (Mixtral routing), 5 tokens
    E, h, I, kk = 8, 512, 1024, 2
    gw = rc.bf16_round(rand((E, h), 60, 0.5)); wg = rc.bf16_round(rand((E, I, h), 61, 0.05))
    wu = rc.bf16_round(rand((E, I, h), 62, 0.05)); wd = rc.bf16_round(rand((E, h, I), 63, 0.05))
    x = rc.bf16_round(rand((5, h), 64))
    out, inds, scores = rm.moe_block(x, gw, wg, wu, wd, kk, "mixtral")
    np.savez_compressed(os.path.join(OUT, "moe.npz"), x=x, out=out, inds=inds.astype(np.uint32), scores=scores, seeds=np.array([60, 61, 62, 63]))

    # mel frontend: 0.5 s of the reference's speech-like generator
    sig = ra.signals(16000, 0.5)["speech_like"]
    r = ra.mel_frontend(sig)
    np.savez_compressed(os.path.join(OUT, "mel.npz"), audio=sig, logmel=r["logmel"], feats=r["feats"])

    # klein tiny forward
    p = rk.KleinParams.tiny()
    ko = rk.KleinOracle(p, rk.synth_weights(p))
    g = np.random.default_rng(7)
    lat = rc.bf16_round(g.standard_normal((24, 128)).astype(np.float32)); txt = rc.bf16_round(g.standard_normal((16, 512)).astype(np.float32))
    cos, sin = rk.compute_rope(np.concatenate([rk.create_txt_ids(16), rk.create_img_ids(4, 6)], 0))
    np.savez_compressed(os.path.join(OUT, "klein_tiny.npz"), latent=lat, txt=txt, timestep=np.float32(750.0),
                        out=ko.forward_with_rope(lat, txt, 750.0, cos, sin).astype(np.float32))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith((".npz", ".json"))))


if __name__ == "__main__":
    main()
