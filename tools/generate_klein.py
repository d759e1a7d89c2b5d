"""flux-klein-mlx/examples/generate_klein.rs end to end on the MI355X build, with synthetic weights of the real shapes
(no checkpoints in this environment): Qwen3-4B text encoder (taps 8/17/26 -> 7680) -> FLUX.2-klein DiT, official
schedule, Euler steps -> latent unpack -> VAE decoder -> PPM.   python tools/generate_klein.py [size] [steps] [out.ppm]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine, flux_pipeline, klein, vae

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
out_path = sys.argv[3] if len(sys.argv) > 3 else None
T = omx.ops.Tensor
t0 = time.perf_counter()
enc = engine.Model(hidden_size=2560, num_hidden_layers=36, intermediate_size=9728, num_attention_heads=32, num_key_value_heads=8,
                   head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True, max_context=512)
enc.synth_weights()
dit = klein.FluxKlein()
dit.synth_weights()
dec = vae.VaeDecoder()
dec.load_weights(vae.random_decoder_weights(1))
omx.ops.synchronize()
load_s = time.perf_counter() - t0

ids = np.ones(512, np.uint32)                       # generate_klein.rs:372-378: dummy tokens, full attention mask
mask = np.ones(512, np.uint8)
t = time.perf_counter(); txt = enc.encode(ids, mask); omx.ops.synchronize(); enc_ms = (time.perf_counter() - t) * 1e3
step_ms = []
t = time.perf_counter()
latent = flux_pipeline.denoise(dit, txt, size, size, steps, seed=0, on_step=lambda i, a, b, ms: step_ms.append(round(ms, 2)))
den_ms = (time.perf_counter() - t) * 1e3
ph = size // 16
z = flux_pipeline.unpack_latents(latent, ph, ph)
t = time.perf_counter(); img = dec.decode(T.from_numpy(z.astype(np.float32))); vae_ms = (time.perf_counter() - t) * 1e3
rgb = vae.to_rgb8(img.numpy())
if out_path:
    vae.write_ppm(out_path, rgb)
print(json.dumps({"image": f"{size}x{size}", "steps": steps, "model_setup_s": round(load_s, 2), "text_encoder_ms": round(enc_ms, 2),
                  "text_encoder_device_ms": round(enc.last_prefill_ms(), 2), "dit_step_ms": step_ms, "denoise_wall_ms": round(den_ms, 2),
                  "vae_ms": round(vae_ms, 2), "vae_device_ms": round(dec.last_ms(), 2),
                  "end_to_end_ms": round(enc_ms + den_ms + vae_ms, 2), "rgb_shape": list(rgb.shape),
                  "finite": bool(np.isfinite(latent).all())}), flush=True)
