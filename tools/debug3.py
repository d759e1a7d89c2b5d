import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
from oracle import ref_core as rc, ref_qwen3 as rq, synth
lib = omx.lib
lib.omx_qwen3_debug_read.restype = ctypes.c_int
lib.omx_qwen3_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
def rd(m, name, n):
    raw = np.empty(n, np.uint16); omx.check(lib.omx_qwen3_debug_read(m._h, name.encode(), raw.ctypes.data, n)); return rc.from_bf16_bits(raw)
C = rq.Qwen3Config
cfg = C(512, 1, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False)
w = rq.synth_weights(cfg); o = rq.Qwen3Oracle(cfg, w)
prompt = synth.prompt_ids(2, cfg.vocab_size)
for cap in (256, 512):
    m = engine.Model(hidden_size=512, num_hidden_layers=1, intermediate_size=1536, num_attention_heads=8,
                     num_key_value_heads=4, head_dim=64, vocab_size=2048, max_context=cap)
    m.synth_weights()
    m.prefill(prompt)
    # oracle intermediates for token 1
    caches = [rc.KVCache()]
    h_all = w["model.embed_tokens.weight"][prompt][None]
    p = "model.layers.0."
    xn = rc.rms_norm(h_all, w[p+"input_layernorm.weight"], 1e-6, "bf16")
    q = rc.linear(xn, w[p+"self_attn.q_proj.weight"], None, "bf16"); k = rc.linear(xn, w[p+"self_attn.k_proj.weight"], None, "bf16"); v = rc.linear(xn, w[p+"self_attn.v_proj.weight"], None, "bf16")
    qkv_ref = np.concatenate([q[0,1], k[0,1], v[0,1]])
    qkv = rd(m, "qkv", qkv_ref.size)
    print("cap", cap, "qkv err", np.abs(qkv - qkv_ref).max())
    kk = k.reshape(1,2,4,64).transpose(0,2,1,3); kk = rc.rms_norm(kk, w[p+"self_attn.k_norm.weight"], 1e-6, "bf16"); kk = rc.rope(kk, 64, False, 1e6, 1.0, 0, "bf16")
    kc = rd(m, "k0", 4*cap*64).reshape(4, cap, 64)
    print("  kcache err tok0", np.abs(kc[:,0]-kk[0,:,0]).max(), "tok1", np.abs(kc[:,1]-kk[0,:,1]).max(), "rest nonzero", np.abs(kc[:,2:]).max())
    vv = v.reshape(1,2,4,64).transpose(0,2,1,3)
    vc = rd(m, "v0", 4*cap*64).reshape(4, cap, 64)
    print("  vcache err tok0", np.abs(vc[:,0]-vv[0,:,0]).max(), "tok1", np.abs(vc[:,1]-vv[0,:,1]).max())
    qq = q.reshape(1,2,8,64).transpose(0,2,1,3); qq = rc.rms_norm(qq, w[p+"self_attn.q_norm.weight"], 1e-6, "bf16"); qq = rc.rope(qq, 64, False, 1e6, 1.0, 0, "bf16")
    att = rc.scaled_dot_product_attention(qq[:,:,1:2], kk, vv, 64**-0.5, None, "bf16")[0,:,0].reshape(-1)
    ao = rd(m, "attn_out", 512)
    print("  attn_out err", np.abs(ao-att).max(), "per head", np.abs(ao-att).reshape(8,64).max(1))
