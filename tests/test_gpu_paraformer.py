"""GPU parity of the Paraformer body pieces (a13) against oracle/ref_paraformer.py (a float64 restatement).
Two arithmetic modes (include/omx.h): "f32" is the reference's own -- f32 weights and activations (funasr-mlx/src/paraformer.rs),
exact-f32 matrix-core GEMMs -- and is held to 1e-4 of the largest reference value (VERDICT r1 "Next" #5); "bf16" (bf16 weights and
activations, fp32 accumulation) is compared at 2^-6 * max|ref|.  CIF is float32 on both sides (1e-5 relative)."""
import numpy as np
import pytest

from oracle import ref_core as rc, ref_paraformer as rp

pytestmark = pytest.mark.gpu


def _weights(in_dim, dim, ffn, k, seed):
    g = np.random.default_rng(seed)
    r = lambda *s, sc=1.0: rc.bf16_round((g.standard_normal(s) * sc).astype(np.float32))
    return {"norm1_w": r(in_dim, sc=0.1) + 1, "norm1_b": r(in_dim, sc=0.1), "qkv_w": r(3 * dim, in_dim, sc=0.05), "qkv_b": r(3 * dim, sc=0.1),
            "out_w": r(dim, dim, sc=0.05), "out_b": r(dim, sc=0.1), "fsmn_w": r(dim, k, sc=0.2), "norm2_w": r(dim, sc=0.1) + 1,
            "norm2_b": r(dim, sc=0.1), "ffn_up_w": r(ffn, dim, sc=0.05), "ffn_up_b": r(ffn, sc=0.1), "ffn_down_w": r(dim, ffn, sc=0.05),
            "ffn_down_b": r(dim, sc=0.1)}


TOL = {"f32": 1e-4, "bf16": 2.0 ** -6}


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("T,in_dim", [(101, 512), (200, 512), (501, 512), (600, 512), (77, 560)])      # 560: the first layer (no attention residual);
# 101 / 200 / 501 keys: the three widths of the one-launch f32 attention, 600: beyond it (GEMM + softmax + GEMM)
def test_sanm_encoder_layer_matches_oracle(omx, T, in_dim, dtype):
    from ominix_mlx_amd import paraformer
    dim, ffn, heads, k = 512, 2048, 4, 11
    w = _weights(in_dim, dim, ffn, k, 3)
    for key in ("norm1_w", "norm2_w"):
        w[key] = rc.bf16_round(w[key])
    x = np.random.default_rng(4).standard_normal((T, in_dim)).astype(np.float32)
    if dtype == "bf16":
        x = rc.bf16_round(x)
    else:                                                            # f32 mode: weights that are NOT bf16-representable
        w = {k_: (v * np.float32(1.0009765625)).astype(np.float32) for k_, v in w.items()}
    ref = rp.sanm_encoder_layer(x, w, heads)
    got = paraformer.SanmEncoderLayer(w, heads, k, dtype).forward(omx.ops.Tensor.from_numpy(x, dtype)).numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= TOL[dtype] * np.abs(ref).max()


def test_cif_fire_matches_oracle(omx):
    from ominix_mlx_amd import paraformer
    g = np.random.default_rng(5)
    B, T, H = 2, 501, 512
    hidden = g.standard_normal((B, T, H)).astype(np.float32)
    alphas = (g.random((B, T)) * 0.35).astype(np.float32)
    alphas[1, 300:] = 0.0                                             # ragged: the second item fires fewer tokens
    ref_frames, ref_counts = rp.cif_fire(hidden, alphas)
    T_ = omx.ops.Tensor
    frames, counts = paraformer.cif_fire(T_.from_numpy(hidden, "f32"), T_.from_numpy(alphas, "f32"))
    np.testing.assert_array_equal(counts, ref_counts)
    np.testing.assert_allclose(frames, ref_frames, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", ["no_fire_tail", "no_fire_no_tail", "exact_threshold", "long", "one_step"])
def test_cif_fire_edge_cases(omx, case):
    """The fire recurrence is replayed by one thread and the frames are summed by many (csrc/paraformer.hip): counts are exact,
    frames agree with the serial definition, for sequences that never fire, end without a tail frame, hit the threshold exactly,
    or are much longer than the model's 30 s window."""
    from ominix_mlx_amd import paraformer
    g = np.random.default_rng(11)
    B, T, H = {"no_fire_tail": (1, 40, 64), "no_fire_no_tail": (1, 40, 64), "exact_threshold": (2, 64, 128),
               "long": (1, 6000, 96), "one_step": (1, 1, 64)}[case]
    hidden = g.standard_normal((B, T, H)).astype(np.float32)
    if case == "no_fire_tail":
        alphas = np.full((B, T), 0.02, np.float32)              # sum 0.8: no fire, tail frame (> 0.45)
    elif case == "no_fire_no_tail":
        alphas = np.full((B, T), 0.01, np.float32)              # sum 0.4: nothing at all
    elif case == "exact_threshold":
        alphas = np.full((B, T), 0.25, np.float32)              # integrate reaches 1.0 exactly every 4th step, no remainder
        alphas[1, 10:] = 0.5
    elif case == "one_step":
        alphas = np.full((B, T), 0.9, np.float32)
    else:
        alphas = (g.random((B, T)) * 0.6).astype(np.float32)
    ref_frames, ref_counts = rp.cif_fire(hidden, alphas)
    T_ = omx.ops.Tensor
    frames, counts = paraformer.cif_fire(T_.from_numpy(hidden, "f32"), T_.from_numpy(alphas, "f32"))
    np.testing.assert_array_equal(counts, ref_counts)
    n = int(ref_counts.max()) if ref_counts.size else 0
    if n:
        np.testing.assert_allclose(frames[:, :n], ref_frames[:, :n], rtol=1e-5, atol=1e-5)


TINY = dict(n_mels=80, lfr_m=7, encoder_dim=512, encoder_layers=3, encoder_heads=4, encoder_ffn_dim=1024, decoder_dim=512,
            decoder_layers=2, decoder_heads=4, decoder_ffn_dim=1024, vocab_size=640, sanm_kernel_size=11, cif_l_order=1,
            cif_r_order=1, cif_threshold=1.0, cif_tail_threshold=0.45)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("N,Ts", [(1, 40), (23, 120), (140, 501)])
def test_decoder_layer_matches_oracle(omx, N, Ts, dtype):
    """ParaformerDecoderLayer::forward incl. cross-attention over the encoder output (Tq != Tk)."""
    import ctypes
    from ominix_mlx_amd import paraformer
    T = omx.ops.Tensor
    w = rp.synth_checkpoint(TINY, 7)
    p = rp._dec_params(w, "decoder.layers.1")
    g = np.random.default_rng(8)
    x, enc = g.standard_normal((N, 512)).astype(np.float32), g.standard_normal((Ts, 512)).astype(np.float32)
    if dtype == "bf16":
        x, enc = rc.bf16_round(x), rc.bf16_round(enc)
    ref = rp.decoder_layer(x, enc, p, 4)
    dev = {k: T.from_numpy(np.ascontiguousarray(v), dtype) for k, v in p.items()}
    ws = paraformer.DecoderLayerWeights(*[dev[k].ptr for k in paraformer._DEC_FIELDS])
    out, xd, ed = T((N, 512), dtype), T.from_numpy(x, dtype), T.from_numpy(enc, dtype)      # keep the device buffers alive over the async launch
    omx.check(omx.lib.omx_paraformer_decoder_layer(out.ptr, xd.ptr, ed.ptr, ctypes.byref(ws), N, Ts, 512, 512, 4, 1024, 11, out.dtype, None))
    got = out.numpy()
    assert np.abs(got - ref).max() <= TOL[dtype] * np.abs(ref).max()


@pytest.mark.parametrize("Tq,Tk,heads", [(1, 1, 1), (16, 16, 4), (17, 33, 2), (101, 101, 4), (215, 501, 4), (501, 501, 4), (64, 129, 4), (40, 257, 1), (128, 400, 4),
                                         (33, 512, 2), (20, 600, 4)])
def test_f32_attention_matches_the_explicit_form(omx, Tq, Tk, heads):
    """omx_paraformer_attention_f32 against numpy's explicit form (paraformer.rs:509-516): scores = q k^T / sqrt(128), softmax over keys, scores v,
    all float32.  q | k | v are read in place from one fused [T, 3 x heads x 128] projection when Tq == Tk (the encoder's layout), from separate
    buffers otherwise; covers the kernel's three widths (<= 128, <= 256, <= 512 keys), ragged tails, the key split (215 x 501 and
    128 x 400: four blocks per tile, 501 x 501: two -- the last to finish combines the shares) and the GEMM + softmax + GEMM path (600)."""
    from ominix_mlx_amd import paraformer  # noqa: F401  (registers the entry point's ctypes signature)
    T = omx.ops.Tensor
    g = np.random.default_rng(Tq * 1000 + Tk)
    D = heads * 128
    if Tq == Tk:
        qkv = g.standard_normal((Tq, 3 * D)).astype(np.float32)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        qkv_d = T.from_numpy(qkv, "f32")
        qp, kp, vp, ldq, ldkv = qkv_d.ptr, qkv_d.ptr + 4 * D, qkv_d.ptr + 8 * D, 3 * D, 3 * D
    else:
        q = g.standard_normal((Tq, D)).astype(np.float32)
        kv = g.standard_normal((Tk, 2 * D)).astype(np.float32)
        k, v = kv[:, :D], kv[:, D:]
        q_d, kv_d = T.from_numpy(q, "f32"), T.from_numpy(kv, "f32")
        qp, kp, vp, ldq, ldkv = q_d.ptr, kv_d.ptr, kv_d.ptr + 4 * D, D, 2 * D
    out = T((Tq, D), "f32")
    omx.check(omx.lib.omx_paraformer_attention_f32(out.ptr, qp, kp, vp, ldq, ldkv, D, Tq, Tk, heads, None))
    got = out.numpy()
    ref = np.empty((Tq, D), np.float32)
    for h in range(heads):
        sl = slice(128 * h, 128 * h + 128)
        sc = (q[:, sl].astype(np.float64) @ k[:, sl].astype(np.float64).T) / np.sqrt(128.0)
        p = np.exp(sc - sc.max(axis=1, keepdims=True))
        ref[:, sl] = (p / p.sum(axis=1, keepdims=True)) @ v[:, sl].astype(np.float64)
    assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) * np.sqrt(Tk)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("stacked_kv", [True, False])
@pytest.mark.parametrize("N,Ts", [(23, 120), (140, 501)])
def test_decoder_stack_matches_oracle_layer_chain(omx, N, Ts, stacked_kv, dtype):
    """omx_paraformer_decoder_stack = the oracle's decoder layers applied one after another (paraformer.rs:1144-1156).  With the layers'
    linear_k_v weights back to back in memory the encoder output is projected for all layers by one GEMM up front; with separate buffers
    (stacked_kv False) each layer projects its own -- same result either way, and the same as the per-layer entry point."""
    import ctypes
    from ominix_mlx_amd import paraformer
    T = omx.ops.Tensor
    cfg = dict(TINY, decoder_layers=3)
    w = rp.synth_checkpoint(cfg, 17)
    g = np.random.default_rng(18)
    x, enc = g.standard_normal((N, 512)).astype(np.float32), g.standard_normal((Ts, 512)).astype(np.float32)
    if dtype == "bf16":
        x, enc = rc.bf16_round(x), rc.bf16_round(enc)
    params = [rp._dec_params(w, f"decoder.layers.{i}") for i in range(3)]
    ref = x
    for p in params:
        ref = rp.decoder_layer(ref, enc, p, 4)
    keep, layers = [], []
    esz = 4 if dtype == "f32" else 2
    if stacked_kv:
        kvw = T.from_numpy(np.concatenate([p["kv_w"] for p in params], axis=0), dtype)
        kvb = T.from_numpy(np.concatenate([p["kv_b"] for p in params], axis=0), dtype)
        keep += [kvw, kvb]
    for i, p in enumerate(params):
        dev = {k: T.from_numpy(np.ascontiguousarray(v), dtype) for k, v in p.items()}
        keep.append(dev)
        ptrs = {k: dev[k].ptr for k in paraformer._DEC_FIELDS}
        if stacked_kv:
            ptrs["kv_w"] = kvw.ptr + i * 1024 * 512 * esz
            ptrs["kv_b"] = kvb.ptr + i * 1024 * esz
        layers.append(paraformer.DecoderLayerWeights(*[ptrs[k] for k in paraformer._DEC_FIELDS]))
    arr = (paraformer.DecoderLayerWeights * 3)(*layers)
    out, xd, ed = T((N, 512), dtype), T.from_numpy(x, dtype), T.from_numpy(enc, dtype)
    scratch = [T((N, 512), dtype) for _ in range(4)]
    kv_all = T((Ts, 3 * 1024), dtype)
    omx.check(omx.lib.omx_paraformer_decoder_stack(out.ptr, xd.ptr, ed.ptr, arr, 3, N, Ts, 512, 512, 4, 1024, 11, scratch[0].ptr, scratch[1].ptr,
                                                   scratch[2].ptr, scratch[3].ptr, kv_all.ptr, out.dtype, None))
    got = out.numpy()
    assert np.abs(got - ref).max() <= 3 * TOL[dtype] * np.abs(ref).max()          # three layers deep
    # the per-layer entry point, chained by hand, lands on the same values (same kernels; the projection's tile choice may differ)
    a, b = T.from_numpy(x, dtype), T((N, 512), dtype)
    for i in range(3):
        omx.check(omx.lib.omx_paraformer_decoder_layer(b.ptr, a.ptr, ed.ptr, ctypes.byref(layers[i]), N, Ts, 512, 512, 4, 1024, 11, b.dtype, None))
        a, b = b, a
    assert np.abs(got - a.numpy()).max() <= TOL[dtype] * np.abs(ref).max()
    assert keep


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_predictor_alphas_and_position_encoding_match_oracle(omx, dtype):
    from ominix_mlx_amd import paraformer
    T = omx.ops.Tensor
    w = rp.synth_checkpoint(TINY, 9)
    g = np.random.default_rng(10)
    mel = (g.standard_normal((57, 560)) * 0.5).astype(np.float32)
    h, mel_d = T((57, 560), dtype), T.from_numpy(mel, "f32")
    omx.check(omx.lib.omx_paraformer_embed(h.ptr, mel_d.ptr, 57, 560, h.dtype, None))
    ref_h = rp.encoder_embed(mel)
    assert np.abs(h.numpy() - ref_h).max() <= (2.0 ** -7 if dtype == "bf16" else 2e-6) * np.abs(ref_h).max()
    enc = g.standard_normal((57, 512)).astype(np.float32)
    if dtype == "bf16":
        enc = rc.bf16_round(enc)
    conv_w = np.ascontiguousarray(w["predictor.conv.weight"].transpose(0, 2, 1))
    alphas, hidden = T((57,), "f32"), T((57, 512), "f32")
    dev = [T.from_numpy(a, dtype) for a in (enc, conv_w, w["predictor.conv.bias"], w["predictor.output_proj.weight"], w["predictor.output_proj.bias"])]
    omx.check(omx.lib.omx_cif_alphas(alphas.ptr, hidden.ptr, *[d.ptr for d in dev], 57, 512, 3, dev[0].dtype, None))
    ref_a = rp.predictor_alphas(enc, conv_w, w["predictor.conv.bias"], w["predictor.output_proj.weight"], w["predictor.output_proj.bias"])
    np.testing.assert_array_equal(hidden.numpy(), enc)
    assert np.abs(alphas.numpy() - ref_a).max() <= (2.0 ** -7 if dtype == "bf16" else 1e-5)


def test_tiny_paraformer_end_to_end_f32_matches_oracle(omx):
    """The reference's arithmetic end to end (f32 weights, f32 activations): encoder output, alphas, token count, logits
    within 1e-4 of the float64 restatement's largest value, token ids equal wherever the top-2 margin exceeds twice that."""
    from ominix_mlx_amd import paraformer
    T = omx.ops.Tensor
    w = rp.synth_checkpoint(TINY, 11)
    mel = (np.random.default_rng(12).standard_normal((83, 560)) * 0.5).astype(np.float32)
    ref_tok, ref_logits, ref_enc, ref_alphas, ref_emb = rp.transcribe_from_mel(mel, w, TINY)
    m = paraformer.Paraformer(w, TINY)                               # dtype "f32" is the default
    assert m.dtype == "f32"
    enc = m.encode(T.from_numpy(mel, "f32"))
    assert enc.dtype == omx.FLOAT32 and np.abs(enc.numpy() - ref_enc).max() <= 1e-4 * np.abs(ref_enc).max()
    emb, n, alphas = m.predict(enc)
    assert np.abs(alphas.numpy()[0] - ref_alphas).max() <= 1e-4 and n == len(ref_tok)
    logits = m.decode(emb, enc).numpy()
    bound = 1e-4 * np.abs(ref_logits).max()
    assert np.abs(logits - ref_logits).max() <= bound
    tok, n2 = m.transcribe_from_mel(T.from_numpy(mel, "f32"))
    safe = rc.argmax_margin(ref_logits) > 2 * bound
    assert n2 == n and safe.mean() > 0.9
    np.testing.assert_array_equal(tok[safe], ref_tok[safe])


def test_random_shapes_of_the_f32_kernels(omx):
    """tools/fuzz_f32.py, 60 seeded cases per family: f32 linear over random aligned and ragged M / N / K (all three GEMM kernels and their tail
    forms) against float64, the f32 attention over random Tq / Tk / heads (every width, split and the fallback) against the explicit form."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_f32.py"), "60", "5"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "failures 0" in r.stdout


def test_one_model_transcribes_utterances_of_changing_length(omx):
    """The model keeps its scratch between calls (one growing buffer per role) and hands `transcribe_from_mel` views of it: utterances of
    83, 40, 600 (beyond the one-launch attention's 512 keys) and again 83 frames through ONE model give, each, exactly what a fresh model
    gives -- no stale rows from a longer predecessor, no buffer too small for a longer successor."""
    from ominix_mlx_amd import paraformer
    T = omx.ops.Tensor
    w = rp.synth_checkpoint(TINY, 11)
    g = np.random.default_rng(21)
    mels = [(g.standard_normal((n, 560)) * 0.5).astype(np.float32) for n in (83, 40, 600, 83)]
    kept = paraformer.Paraformer(w, TINY)
    for mel in mels:
        got, n = kept.transcribe_from_mel(T.from_numpy(mel, "f32"))
        fresh = paraformer.Paraformer(w, TINY)
        want, n_want = fresh.transcribe_from_mel(T.from_numpy(mel, "f32"))
        assert n == n_want
        np.testing.assert_array_equal(got, want)
        # and the staged entry points with caller-owned outputs agree with it
        enc = fresh.encode(T.from_numpy(mel, "f32"))
        emb, n2, _ = fresh.predict(enc)
        assert n2 == n
        if n:
            np.testing.assert_array_equal(omx.ops.argmax(fresh.decode(emb, enc)).numpy().astype(np.int32), want)


def test_tiny_paraformer_end_to_end_matches_oracle(omx):
    """Paraformer::transcribe_from_mel on a 3+2-layer model with the reference's checkpoint keys: encoder output,
    CIF token count, logits and token ids against the float64 restatement.  bf16 activations: encoder output
    within 2^-6 * max * sqrt(layers); CIF fires the same number of tokens when no integrate value sits within that
    error of the threshold; token ids equal wherever the oracle's top-2 margin exceeds twice the logit bound."""
    from ominix_mlx_amd import paraformer
    T = omx.ops.Tensor
    w = rp.synth_checkpoint(TINY, 11)
    mel = (np.random.default_rng(12).standard_normal((83, 560)) * 0.5).astype(np.float32)
    ref_tok, ref_logits, ref_enc, ref_alphas, ref_emb = rp.transcribe_from_mel(mel, w, TINY)
    m = paraformer.Paraformer(w, TINY, dtype="bf16")
    enc = m.encode(T.from_numpy(mel, "f32"))
    assert np.abs(enc.numpy() - ref_enc).max() <= 2.0 ** -6 * np.abs(ref_enc).max() * np.sqrt(TINY["encoder_layers"])
    emb, n, alphas = m.predict(enc)
    assert np.abs(alphas.numpy()[0] - ref_alphas).max() <= 2.0 ** -5
    assert n == len(ref_tok), f"CIF fired {n} tokens, oracle {len(ref_tok)}"
    logits = m.decode(emb, enc).numpy()
    bound = 2.0 ** -5 * np.abs(ref_logits).max() * np.sqrt(TINY["decoder_layers"] + 1)
    assert np.abs(logits - ref_logits).max() <= bound
    tok, n2 = m.transcribe_from_mel(T.from_numpy(mel, "f32"))
    assert n2 == n
    margins = rc.argmax_margin(ref_logits)
    safe = margins > 2 * bound
    np.testing.assert_array_equal(tok[safe], ref_tok[safe])
    with pytest.raises(KeyError, match="Missing weight"):
        paraformer.Paraformer({k: v for k, v in w.items() if k != "decoder.after_norm.bias"}, TINY)


def test_layers_match_the_torch_pins(omx):
    """The device layers against an implementation that shares nothing with this repository: tests/golden/torch_paraformer.npz holds the
    outputs of a torch (float64) restatement of SanmEncoderLayer / ParaformerDecoderLayer (tests/golden/make_torch_pins.py) on
    committed inputs, the weights come from the seeded generator; float32 mode, 1e-4 of the largest value like the oracle tests."""
    import ctypes
    import os
    from ominix_mlx_amd import paraformer
    T = omx.ops.Tensor
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "torch_paraformer.npz"))
    w = rp.synth_checkpoint(TINY, int(z["seed"]))
    for x, prefix, want in ((z["x"], "encoder.layers.0", z["enc_out"]), (z["x0"], "encoder.encoders0.0", z["enc0_out"])):
        got = paraformer.SanmEncoderLayer(rp._enc_params(w, prefix), 4, 11, "f32").forward(T.from_numpy(x, "f32")).numpy()
        assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max()
    p = rp._dec_params(w, "decoder.layers.1")
    dev = {k: T.from_numpy(np.ascontiguousarray(v), "f32") for k, v in p.items()}
    ws = paraformer.DecoderLayerWeights(*[dev[k].ptr for k in paraformer._DEC_FIELDS])
    N, Ts = z["xd"].shape[0], z["enc_out"].shape[0]
    out, xd, ed = T((N, 512), "f32"), T.from_numpy(z["xd"], "f32"), T.from_numpy(z["enc_out"], "f32")
    omx.check(omx.lib.omx_paraformer_decoder_layer(out.ptr, xd.ptr, ed.ptr, ctypes.byref(ws), N, Ts, 512, 512, 4, 1024, 11, out.dtype, None))
    assert np.abs(out.numpy() - z["dec_out"]).max() <= 1e-4 * np.abs(z["dec_out"]).max()
