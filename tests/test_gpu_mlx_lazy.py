"""GPU: the deferred op list behind the mlx-c ABI (csrc/mlxc_lazy.hpp, round 6) -- the ABI's lazy contract (mlx-c transforms.h:30,42;
mlx-rs/src/transforms/mod.rs:67-85) used the way qwen3-mlx's Generate::next uses it (model.rs:804-843).

What must hold:
  * the SAME values whether an op launches when it is called (lazy off), is recorded and launched as recorded (fuse off), or is rewritten
    onto the fused GEMV family (default) -- compared through the oracle where roundings may differ, bit for bit where they may not;
  * an intermediate a caller still HOLDS is never optimised away; one that nobody holds and only fused readers used may be;
  * an op that was never taught to defer sees all earlier recorded work (Arr::ptr() is the gate);
  * item() of an older result does not have to wait for newer queued work, and still returns the right value.
"""
import numpy as np
import pytest

from oracle import ref_core as rc, synth
from test_gpu_primitives import assert_bf16_close, rand

pytestmark = pytest.mark.gpu


@pytest.fixture()
def mx(omx):
    from ominix_mlx_amd import mlx_c
    mlx_c.lazy_mode(True, True)
    yield mlx_c
    mlx_c.lazy_mode(True, True)


def _mlp_inputs(hidden=512, inter=1536, seed=5):
    x = rc.bf16_round(rand((1, 1, hidden), seed))
    nw = rc.bf16_round(1 + 0.1 * rand((hidden,), seed + 1))
    wg = rc.bf16_round(rand((inter, hidden), seed + 2) * 0.05)
    wu = rc.bf16_round(rand((inter, hidden), seed + 3) * 0.05)
    wd = rc.bf16_round(rand((hidden, inter), seed + 4) * 0.05)
    return x, nw, wg, wu, wd


def _mlp(mx, X, NW, WG, WU, WD, hold=False):
    """qwen3-mlx's block tail: h + down(silu(gate(rms_norm(h))) * up(rms_norm(h)))  (model.rs:263-267, 330-339)"""
    hn = mx.rms_norm(X, NW, 1e-6)
    g = mx.matmul(hn, mx.transpose(WG))
    act = mx.multiply(mx.multiply(g, mx.sigmoid(g)), mx.matmul(hn, mx.transpose(WU)))
    out = mx.add(X, mx.matmul(act, mx.transpose(WD)))
    return (out, hn, g, act) if hold else (out,)


@pytest.mark.parametrize("lazy,fuse", [(False, False), (True, False), (True, True)])
def test_block_tail_matches_the_oracle_in_every_mode(mx, lazy, fuse):
    x, nw, wg, wu, wd = _mlp_inputs()
    mx.lazy_mode(lazy, fuse)
    X, NW, WG, WU, WD = (mx.Array.from_numpy(a) for a in (x, nw, wg, wu, wd))
    before = mx.lazy_stats()
    (out,) = _mlp(mx, X, NW, WG, WU, WD)
    got = out.numpy()
    after = mx.lazy_stats()
    hn = rc.rms_norm(x, nw, 1e-6, "bf16")
    g = rc.linear(hn, wg, None, "bf16")
    act = rc.multiply(rc.multiply(g, rc.sigmoid(g, "bf16"), "bf16"), rc.linear(hn, wu, None, "bf16"), "bf16")
    ref = rc.add(x, rc.linear(act, wd, None, "bf16"), "bf16")
    assert_bf16_close(got, ref, 2)
    fused = after["fused_launches"] - before["fused_launches"]
    if lazy and fuse:
        # nobody held hn, g, sigmoid(g), g * sigmoid(g), up(hn), down(act): two launches remain -- [norm + gate/up + SwiGLU], [down + residual]
        assert fused == 2 and after["launched_as_recorded"] == before["launched_as_recorded"]
    else:
        assert fused == 0


def test_the_three_modes_agree_bit_for_bit_on_the_block_tail(mx):
    x, nw, wg, wu, wd = _mlp_inputs(seed=11)
    outs = []
    for lazy, fuse in ((False, False), (True, False), (True, True)):
        mx.lazy_mode(lazy, fuse)
        arrs = [mx.Array.from_numpy(a) for a in (x, nw, wg, wu, wd)]
        outs.append(_mlp(mx, *arrs)[0].numpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    np.testing.assert_array_equal(outs[0], outs[2])      # the fused kernels keep the per-op rounding points


def test_an_intermediate_somebody_holds_is_computed(mx):
    x, nw, wg, wu, wd = _mlp_inputs(seed=21)
    mx.lazy_mode(False, False)
    eager = [a.numpy() for a in _mlp(mx, *[mx.Array.from_numpy(a) for a in (x, nw, wg, wu, wd)], hold=True)]
    mx.lazy_mode(True, True)
    arrs = [mx.Array.from_numpy(a) for a in (x, nw, wg, wu, wd)]
    before = mx.lazy_stats()
    held = _mlp(mx, *arrs, hold=True)        # out, rms_norm(x), gate(x), the activation: all still referenced when the list runs
    mx.eval(held[0])
    after = mx.lazy_stats()
    for got, want in zip(held, eager):
        np.testing.assert_array_equal(got.numpy(), want)
    # hn and g are visible: the norm is launched as recorded, gate / up stay plain products; only [down + residual] has an epilogue to take
    assert after["launched_as_recorded"] - before["launched_as_recorded"] >= 4


def test_an_op_that_never_learnt_to_defer_sees_the_recorded_work(mx):
    a = rc.bf16_round(rand((4, 64), 31))
    b = rc.bf16_round(rand((4, 64), 32))
    A, B = mx.Array.from_numpy(a), mx.Array.from_numpy(b)
    s = mx.add(A, B)                                   # recorded
    before = mx.lazy_stats()["flushes"]
    e = mx.exp(s)                                      # recorded too (unary)
    srt = mx.sort_axis(e, -1)                          # eager: must run behind both
    assert mx.lazy_stats()["flushes"] == before + 1
    ref = np.sort(rc.bf16_round(np.exp(rc.bf16_round(a + b).astype(np.float64))), axis=-1)
    assert_bf16_close(srt.numpy(), ref, 1)


def test_cache_update_in_place_and_reads_in_program_order(mx):
    """KVCache::update_and_fetch (cache.rs:140-193) twice before anything runs: the second update and the reads in between are recorded
    against the same buffer and must see each other in call order."""
    cache = mx.zeros([1, 2, 8, 64], mx.BFLOAT16)
    k0 = rc.bf16_round(rand((1, 2, 3, 64), 41))
    k1 = rc.bf16_round(rand((1, 2, 1, 64), 42))
    c1 = mx.slice_update(cache, mx.Array.from_numpy(k0), [0, 0, 0, 0], [1, 2, 3, 64])
    del cache
    seen1 = mx.multiply(mx.slice(c1, [0, 0, 0, 0], [1, 2, 3, 64]), mx.Array.from_numpy(np.full((1,), 2.0, np.float32), mx.BFLOAT16))
    c2 = mx.slice_update(c1, mx.Array.from_numpy(k1), [0, 0, 3, 0], [1, 2, 4, 64])
    del c1
    full = mx.slice(c2, [0, 0, 0, 0], [1, 2, 4, 64]).numpy()
    np.testing.assert_array_equal(full[:, :, :3], k0)
    np.testing.assert_array_equal(full[:, :, 3:], k1)
    np.testing.assert_array_equal(seen1.numpy(), rc.bf16_round(k0 * 2.0))


def test_item_of_an_older_result_with_newer_work_queued(mx):
    """Generate::next: async_eval(next), then the caller reads the PREVIOUS token."""
    w = rc.bf16_round(rand((2048, 512), 51) * 0.05)
    x0 = rc.bf16_round(rand((1, 1, 512), 52))
    W = mx.Array.from_numpy(w)
    toks, prev = [], None
    X = mx.Array.from_numpy(x0)
    want = []
    xr = x0
    for i in range(6):
        logits = mx.matmul(X, mx.transpose(W))
        y = mx.argmax_axis(logits, -1)
        mx.async_eval(y)
        lr = rc.linear(xr, w, None, "bf16")
        want.append(int(np.argmax(lr[0, 0])))
        if prev is not None:
            toks.append(int(prev.item()))
        prev = y
        # next input: the row of W the token picked (an embedding take on the device token, as compute_next does)
        X = mx.reshape(mx.take_axis(W, mx.reshape(y, [1, 1]), 0), [1, 1, 512])
        xr = w[want[-1]][None, None, :]
    toks.append(int(prev.item()))
    assert toks == want


def test_an_error_inside_the_deferred_list_surfaces_at_the_evaluation_point(mx, omx):
    q = mx.Array.from_numpy(rc.bf16_round(rand((1, 2, 1, 48), 61)))       # head_dim 48: no attention kernel takes it
    out = mx.scaled_dot_product_attention(q, q, q, 1.0)                    # recorded: shapes are consistent, the launch is what fails
    with pytest.raises(omx.OmxError):
        mx.eval(out)
    # the list is empty again and the layer keeps working
    a = mx.Array.from_numpy(np.ones((2, 2), np.float32), mx.FLOAT32)
    np.testing.assert_array_equal(mx.add(a, a).numpy(), np.full((2, 2), 2.0, np.float32))


def test_drop_in_route_fuses_and_keeps_its_tokens(omx, mx):
    """csrc/per_op_route.hip (qwen3-mlx forward + Generate::next, replayed through the ABI) in the three modes on one model: the same
    greedy tokens, and in the default mode the decode step collapses to <= 8 launches per layer -- [norm + q | k | v], [q / k norm + rope +
    cache writes], attention, [o + residual], [norm + gate / up + SwiGLU], [down + residual] -- plus embedding take and [norm + head + argmax]."""
    from ominix_mlx_amd import engine
    cfg = dict(hidden_size=512, num_hidden_layers=3, intermediate_size=1536, num_attention_heads=8, num_key_value_heads=4, head_dim=64,
               vocab_size=2048, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
    m = engine.Model(max_context=1024, **cfg)
    m.synth_weights()
    prompt = synth.prompt_ids(300, cfg["vocab_size"])
    n_new = 220                                            # crosses the cache's 512-token growth
    runs = {}
    for name, (lazy, fuse, worker) in {"eager": (False, False, False), "recorded": (True, False, False), "fused": (True, True, False),
                                       "fused, launch worker": (True, True, True)}.items():
        mx.lazy_mode(lazy, fuse, worker)
        s0 = mx.lazy_stats()
        runs[name] = ([int(t) for t in m.per_op_route(prompt, n_new)["tokens"]], s0, mx.lazy_stats())
    assert runs["eager"][0] == runs["recorded"][0] == runs["fused"][0] == runs["fused, launch worker"][0]
    _, s0, s1 = runs["fused"]
    passes = n_new + 2                                     # the prompt, then one decode pass per token and the one Generate keeps in flight
    launches = (s1["launched_as_recorded"] - s0["launched_as_recorded"]) + (s1["fused_launches"] - s0["fused_launches"])
    per_decode_pass = (launches - 40 * cfg["num_hidden_layers"]) / (passes - 1)        # (the prompt pass: at most 40 launches per layer)
    assert per_decode_pass <= 8 * cfg["num_hidden_layers"] + 4, per_decode_pass
    assert s1["fused_launches"] - s0["fused_launches"] >= 5 * cfg["num_hidden_layers"] * (passes - 1)
    m.close()


@pytest.mark.parametrize("bits", [4, 8])
def test_drop_in_route_on_a_quantized_checkpoint(omx, mx, bits):
    """The reference's flagship format (qwen3-mlx/README.md:102: MLX 4-bit): every Linear is quantized_matmul on a (weight, scales, biases)
    triplet, the embedding dequantises its rows (quantized.rs:120-164, 361-385).  The replay drives exactly those calls; the deferred list
    rewrites them onto the packed-GEMV family -- quant.hip, and for 4-bit K = 4096 matrices the matrix-core kernel on tiles it builds once
    per weight buffer (qgemv_mfma.hip) -- with the same prologues / epilogues as the bf16 idioms.  The three modes agree on every token on
    the VALU kernels (8-bit); where the matrix-core kernel steps in (4-bit, K a multiple of 1 024) its different accumulation order is
    allowed to move a near-tie, so that case is held through the streams' common prefix (the logits are held to the oracle at full size:
    test_gpu_fullsize_pin.py, test_gpu_quant.py)."""
    from ominix_mlx_amd import engine
    cfg = dict(hidden_size=1024, num_hidden_layers=2, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=4, head_dim=128,
               vocab_size=2048, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
    m = engine.Model(max_context=512, quantization={"bits": bits, "group_size": 64}, **cfg)
    m.synth_weights()
    prompt = synth.prompt_ids(100, cfg["vocab_size"])
    want = [int(m.prefill(prompt))] + [int(t) for t in m.decode(60)]
    runs = {}
    for name, (lazy, fuse) in {"eager": (False, False), "recorded": (True, False), "fused": (True, True)}.items():
        mx.lazy_mode(lazy, fuse)
        s0 = mx.lazy_stats()
        runs[name] = ([int(t) for t in m.per_op_route(prompt, 60)["tokens"]], s0, mx.lazy_stats())
    assert runs["eager"][0] == runs["recorded"][0]                            # the same launches, now or later
    if bits == 8:
        assert runs["fused"][0] == runs["eager"][0]                          # 8-bit: the VALU kernel in every mode, the same rounding points
        assert runs["fused"][0][:24] == want[:24]
    else:
        # 4-bit, K a multiple of 1 024: the fused plans (and the engine) multiply on the matrix cores -- another accumulation order than the
        # VALU kernel the unfused calls take, so a near-tie of the flat synthetic logits may move; the streams must share their start
        def common(a, b):
            n = 0
            while n < min(len(a), len(b)) and a[n] == b[n]:
                n += 1
            return n
        assert common(runs["fused"][0], runs["eager"][0]) >= 8, (runs["fused"][0][:12], runs["eager"][0][:12])
        assert common(runs["fused"][0], want) >= 8, (runs["fused"][0][:12], want[:12])
    _, s0, s1 = runs["fused"]
    assert s1["fused_launches"] - s0["fused_launches"] >= 5 * cfg["num_hidden_layers"] * 60
    m.close()


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
@pytest.mark.parametrize("shape", [(1, 5, 8), (3, 7), (2, 64), (4, 1000), (2048, 96)])
def test_elementwise_ops_on_the_vector_path_and_the_general_one(mx, dtype, shape):
    """Same-shape contiguous float operands take 16-byte-per-lane kernels (the residual adds and SwiGLU products of a prompt pass), everything
    else the general broadcast kernel: both against numpy, every op, sizes that are and are not whole vectors."""
    g = np.random.default_rng(sum(shape))
    a, b = g.standard_normal(shape).astype(np.float32), g.standard_normal(shape).astype(np.float32) + 3.0
    dt = mx.BFLOAT16 if dtype == "bf16" else mx.FLOAT32
    rnd = rc.bf16_round if dtype == "bf16" else (lambda v: np.asarray(v, np.float32))
    a, b = rnd(a), rnd(b)
    A, B = mx.Array.from_numpy(a, dt), mx.Array.from_numpy(b, dt)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    cases = {"negative": (mx.negative(A), -a64), "add": (mx.add(A, B), a64 + b64), "subtract": (mx.subtract(A, B), a64 - b64),
             "multiply": (mx.multiply(A, B), a64 * b64), "divide": (mx.divide(A, B), a64 / b64),
             "sigmoid": (mx.sigmoid(A), 1 / (1 + np.exp(-a64))), "exp": (mx.exp(A), np.exp(a64))}
    tol = 2.0 ** -8 if dtype == "bf16" else 2.0 ** -20
    for name, (got, ref) in cases.items():
        err = np.abs(got.numpy().astype(np.float64) - ref)
        assert (err <= tol * np.maximum(np.abs(ref), 1e-3) + 1e-7).all(), (name, float(err.max()))


def test_stacked_products_never_move_behind_a_reader(mx):
    """Two products of one activation are stacked into one launch at the position of the LAST one -- unless something recorded between
    them reads the first (here: sigmoid(gate) between gate and up, all results held so that the SwiGLU rewrite cannot take them)."""
    g = np.random.default_rng(3)
    x = rc.bf16_round(g.standard_normal((1, 64, 256)).astype(np.float32))
    wg = rc.bf16_round((g.standard_normal((512, 256)) * 0.05).astype(np.float32))
    wu = rc.bf16_round((g.standard_normal((512, 256)) * 0.05).astype(np.float32))
    outs = {}
    for fuse in (False, True):
        mx.lazy_mode(True, fuse)
        X, WG, WU = mx.Array.from_numpy(x), mx.Array.from_numpy(wg), mx.Array.from_numpy(wu)
        gate = mx.matmul(X, mx.transpose(WG))
        sg = mx.sigmoid(gate)
        up = mx.matmul(X, mx.transpose(WU))
        both = mx.matmul(X, mx.transpose(WG))          # ... and a pair that CAN be stacked with `up` (nothing reads it in between)
        mx.eval(gate, sg, up, both)
        outs[fuse] = [t.numpy() for t in (gate, sg, up, both)]
    for a, b in zip(outs[False], outs[True]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(outs[True][0], outs[True][3])


def test_random_programs_agree_bit_for_bit_in_all_three_modes(omx):
    """tools/fuzz_lazy.py, 60 seeded programs of 40 steps over the ops the decode / prompt idioms are made of (with held and dropped
    intermediates, cache writes and reads, item() and eval() in the middle): launched as called, launched as recorded and rewritten by the
    peephole pass give the same bits in every surviving array and the same item() values.  (400 programs x 60 steps were run when this was
    added: ~5 000 fused launches among 28 000 recorded ops, no divergence.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_lazy.py"), "60", "40", "7"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "bit-identical" in r.stdout and "fused_launches +0" not in r.stdout
