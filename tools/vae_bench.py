"""FLUX.2 VAE decode time at the real configuration (ch 128, mult 1/2/4/4, 32 latent channels), random weights."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import vae
T = omx.ops.Tensor
w = vae.random_decoder_weights(1)
dec = vae.VaeDecoder()
dec.load_weights(w)
for side in (512, 1024):
    h = side // 8
    z = T.from_numpy(np.random.default_rng(0).standard_normal((h, h, 32)).astype(np.float32))
    dec.decode(z)
    ms = []
    for _ in range(3):
        img = dec.decode(z)
        ms.append(dec.last_ms())
    x = img.numpy()
    print(json.dumps({"image": f"{side}x{side}", "ms": [round(m, 2) for m in ms], "finite": bool(np.isfinite(x).all()),
                      "absmax": float(np.abs(x).max())}), flush=True)
