"""BASELINE config 4: Paraformer-large, 30 s of 16 kHz audio -> mel/STFT + LFR + CMVN -> 50-layer SAN-M encoder ->
CIF -> 16-layer decoder -> token ids, on one MI355X (synthetic weights with the reference's checkpoint keys).
Prints the wall time per stage and the real-time factor."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import audio, paraformer

cfg = dict(paraformer.DEFAULT_CONFIG)
w = paraformer.random_checkpoint(cfg, 3)
DT = sys.argv[1] if len(sys.argv) > 1 else "f32"      # "f32" = the reference's arithmetic (default), "bf16"
m = paraformer.Paraformer(w, cfg, dtype=DT)
sr, secs = 16000, 30
g = np.random.default_rng(0)
t = np.arange(sr * secs) / sr
wave = (0.3 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.03 * g.standard_normal(t.size)).astype(np.float32)
fe = audio.MelFrontend()
T = omx.ops.Tensor
wave_d = T.from_numpy(wave, "f32")      # audio resident in HBM when the timed region starts


def run():
    t0 = time.perf_counter()
    mel = fe.forward(wave_d)
    mel = mel.view(mel.shape[1:])
    omx.ops.synchronize(); t1 = time.perf_counter()
    enc = m.encode(mel)
    omx.ops.synchronize(); t2 = time.perf_counter()
    emb, n, _ = m.predict(enc)
    omx.ops.synchronize(); t3 = time.perf_counter()
    tok = omx.ops.argmax(m.decode(emb, enc)).numpy() if n else np.zeros(0)
    t4 = time.perf_counter()
    return (t1 - t0, t2 - t1, t3 - t2, t4 - t3), mel.shape, n


run()
best = None
for _ in range(5):
    r = run()
    if best is None or sum(r[0]) < sum(best[0]):
        best = r
(ta, tb, tc, td), mshape, n = best
tot = ta + tb + tc + td
print(f"[{DT}] mel {mshape}: frontend {ta*1e3:.2f} ms | encoder {tb*1e3:.2f} ms | predictor+CIF {tc*1e3:.2f} ms | decoder ({n} tokens) {td*1e3:.2f} ms"
      f" | total {tot*1e3:.2f} ms  RTF {tot/secs:.5f}  ({secs/tot:.0f}x real time)")
