"""The plain-C port (oracle/c/omx_oracle.c: cpu_baseline implementation) must agree with the
numpy oracle (oracle/ref_qwen3.py) it is a twin of.  CPU only."""
import ctypes

import numpy as np

from oracle import c_oracle, ref_core as rc, ref_qwen3 as rq, synth


def test_c_fill_matches_numpy_twin():
    lib = c_oracle.load()
    out = np.empty(5000, np.uint16)
    lib.oracle_fill_uniform_bf16(c_oracle.ptr(out), out.size, 1234, np.float32(0.05), np.float32(1.0))
    ref = synth.uniform_pm((5000,), 1234, 0.05, 1.0, "bf16")
    np.testing.assert_array_equal(rc.from_bf16_bits(out), ref)


def test_c_layer_decode_matches_numpy_oracle():
    lib = c_oracle.load()
    cfg = rq.Qwen3Config(256, 1, 768, 4, 2, 64, 512, 1e-6, 1e6, False)
    w = rq.synth_weights(cfg)
    oracle = rq.Qwen3Oracle(cfg, w)
    cap, n_ctx = 64, 9
    # numpy oracle: run n_ctx tokens through block 0, one at a time (decode shape)
    caches = [rc.KVCache()]
    g = np.random.default_rng(0)
    hs = rc.bf16_round(g.standard_normal((n_ctx, cfg.hidden_size)).astype(np.float32))
    ref_out = []
    for t in range(n_ctx):
        ref_out.append(oracle.block(0, hs[t][None, None, :], None, caches[0])[0, 0])
    # C port
    bits = {k: rc.to_bf16_bits(v) for k, v in w.items()}
    p = "model.layers.0."
    kc = np.zeros((cfg.num_key_value_heads, cap, cfg.head_dim), np.uint16)
    vc = np.zeros_like(kc)
    L = c_oracle.Layer(*[c_oracle.ptr(np.ascontiguousarray(bits[p + n])) for n in (
        "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
        "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight", "self_attn.q_norm.weight",
        "self_attn.k_norm.weight", "input_layernorm.weight", "post_attention_layernorm.weight")],
        c_oracle.ptr(kc), c_oracle.ptr(vc))
    lc = c_oracle.LayerCfg(cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.num_key_value_heads,
                           cfg.head_dim, cap, cfg.rms_norm_eps, cfg.rope_theta, 1.0)
    scratch = np.zeros(lib.oracle_qwen3_scratch_elems(ctypes.byref(lc)), np.uint16)
    for t in range(n_ctx):
        h = rc.to_bf16_bits(hs[t]).copy()
        lib.oracle_qwen3_layer_decode(ctypes.byref(lc), ctypes.byref(L), c_oracle.ptr(h), t, c_oracle.ptr(scratch))
        got = rc.from_bf16_bits(h)
        # identical algorithm and rounding points; libm vs numpy exp/sin/cos may differ in the last
        # ulp of a double, which can flip a bf16 rounding on rare elements
        diff = np.abs(got - ref_out[t])
        assert (diff > 0).mean() < 0.01
        assert diff.max() <= 2 * 2.0 ** -7 * np.abs(ref_out[t]).max()
