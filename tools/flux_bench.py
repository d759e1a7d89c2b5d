"""FLUX.2-klein denoise-step timing on one MI355X (BASELINE config 5 at TP=1): full-size model
(3072 hidden, 24 heads, 5 double + 20 single blocks), synthetic weights/inputs, bf16."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import klein

def run(res=1024, s_txt=512, steps=4, warmup=1):
    g = res // 16
    s_img = g * g
    m = klein.FluxKlein()
    m.synth_weights()
    lat = omx.ops.fill_uniform((s_img, 128), 1, 1.7)
    txt = omx.ops.fill_uniform((s_txt, 7680), 2, 1.7)
    rc, rs = klein.compute_rope(klein.create_txt_ids(s_txt), klein.create_img_ids(g, g))
    for _ in range(warmup):
        m.forward_with_rope(lat, txt, 1000.0, rc, rs)
    ts = []
    for i in range(steps):
        t = 1.0 - i / steps
        m.forward_with_rope(lat, txt, t * 1000.0, rc, rs)
        ts.append(m.last_ms())
    S = s_txt + s_img
    h, mh = 3072, 9216
    lin = 2 * (s_img * 128 * h + s_txt * 7680 * h + 5 * 2 * 0 + 0)  # embedders
    lin += 5 * (2 * S * (4 * h * h) + 2 * S * (h * 2 * mh) + 2 * S * (mh * h))
    lin += 20 * (2 * S * h * (3 * h + 2 * mh) + 2 * S * (h + mh) * h)
    lin += 2 * s_img * h * 128
    attn = 25 * 4 * S * S * h
    flop = lin + attn
    ms = float(np.median(ts))
    out = {"workload": f"flux.2-klein {res}x{res} bf16, S_img={s_img}, S_txt={s_txt}", "sec_per_step": round(ms / 1e3, 5),
           "ms_all": [round(t, 2) for t in ts], "tflop_per_step": round(flop / 1e12, 2),
           "achieved_tflops": round(flop / ms / 1e9, 1), "mfma_frac_of_2500": round(flop / ms / 1e9 / 2500.0, 4)}
    m.close()
    return out

if __name__ == "__main__":
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    print(json.dumps(run(res)), flush=True)
