"""The engine at the REAL BASELINE size (Qwen3-8B shapes: 36 layers, 4096 hidden, 32 / 8 heads of 128, 12288 FFN, 151 936 vocabulary)
against the numpy oracle run once at that size (tools/full_size_pin.py -> tests/golden/qwen3_8b_fullsize_pin.npz: greedy token,
top-8 logits and top-1 / top-2 margin at every position of a 16-token synthetic prompt).  Both routes of the engine are held to it:
the batched matrix-core route (Model.verify: GEMMs + flash attention + [n, V] lm_head GEMM) at all 16 positions, and the decode step
(GEMV + attn_step + fused head) at the last one.  Bound: the engine's usual 2^-7 * max|logit| * sqrt(layers); tokens must be EQUAL
wherever the oracle's margin exceeds twice that."""
import os

import numpy as np
import pytest

import bench

pytestmark = pytest.mark.gpu


def _report(fixture, route, worst, max_logit, layers, rule, tokens=None):
    """One line per (fixture, route) for DESIGN.md section 2's table: the measured deviation as a multiple of SURVEY 8c's
    2^-7 * max|logit| next to the rule the test holds it to (written to gpurun_out/parity_r06.jsonl when that directory exists)."""
    import json
    unit = 2.0 ** -7 * max_logit
    row = {"fixture": fixture, "route": route, "layers": layers, "max_abs_logit": round(float(max_logit), 4), "worst_deviation": round(float(worst), 4),
           "multiple_of_survey_unit": round(float(worst) / unit, 2), "rule": rule, "rule_multiple": round({"sqrt(L)": np.sqrt(layers), "sqrt(2L)": np.sqrt(2 * layers)}[rule], 2)}
    if tokens is not None:
        row["greedy_tokens_equal"] = tokens
    print("parity:", json.dumps(row))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_r06.jsonl"), "a") as f:
            f.write(json.dumps(row) + "\n")


PIN = os.path.join(os.path.dirname(__file__), "golden", "qwen3_8b_fullsize_pin.npz")


@pytest.mark.skipif(not os.path.exists(PIN), reason="full-size pin fixture not generated")
def test_full_size_engine_matches_full_size_oracle(omx):
    from ominix_mlx_amd import engine
    pin = np.load(PIN)
    cfg = dict(bench.QWEN3_8B)
    prompt = pin["prompt"]
    n = prompt.size
    m = engine.Model(max_context=64, **cfg)
    m.synth_weights()
    bound = 2.0 ** -7 * float(pin["max_abs"].max()) * np.sqrt(cfg["num_hidden_layers"])
    # ---- batched route: every position in one pass ----
    got = m.verify(prompt)
    assert m.offset() == n
    worst = 0.0
    for i in range(n):
        lg = m.verify_logits(i)
        worst = max(worst, float(np.abs(lg[pin["top_idx"][i]] - pin["top_val"][i]).max()))
        if pin["margin"][i] > 2 * bound:
            assert got[i] == pin["greedy"][i], f"position {i}: token {got[i]} != oracle {pin['greedy'][i]} (margin {pin['margin'][i]:.3f})"
        else:   # (random weights give flat logits: most margins are below the bound; the oracle's choice must still be a near-maximum here)
            assert lg[pin["greedy"][i]] >= lg.max() - 2 * bound
    assert worst <= bound, f"batched route: top-8 logits off by {worst:.4f} (bound {bound:.4f})"
    _report("qwen3_8b_fullsize_pin (16 positions)", "engine, batched pass", worst, float(pin["max_abs"].max()), cfg["num_hidden_layers"], "sqrt(L)",
            f"{int((got == pin['greedy']).sum())} of {n}")
    # ---- decode step: forget the last prompt token and run it through the step ----
    m.trim(1, int(prompt[n - 1]))
    tok = int(m.decode(1)[0])
    lg = m.last_logits()
    assert float(np.abs(lg[pin["top_idx"][n - 1]] - pin["top_val"][n - 1]).max()) <= bound
    assert tok == pin["greedy"][n - 1] or pin["margin"][n - 1] <= 2 * bound
    agree = int((got == pin["greedy"]).sum())
    assert agree >= n // 2, f"only {agree} of {n} greedy tokens agree with the full-size oracle"
    m.close()


C1_PIN = os.path.join(os.path.dirname(__file__), "golden", "qwen3_0p6b_c1_pin.npz")


@pytest.mark.skipif(not os.path.exists(C1_PIN), reason="c1 pin fixture not generated")
def test_baseline_config_c1_at_real_shapes(omx):
    """BASELINE configs[0] with SURVEY.md 8d's protocol at the model's real shapes (Qwen3-0.6B: 28 layers, tied 151 936-entry
    vocabulary): 128-token synthetic prompt, greedy, 32 new tokens.  Random weights give flat logits (several oracle margins are one
    bf16 ulp), so the comparison is made on the LOGITS of every step with the oracle's token fed back (teacher forcing through
    `trim(0, token)`): top-8 values within the engine's bound at all 32 steps, and the engine's own greedy choice must be a
    near-maximum of the oracle wherever it differs."""
    from ominix_mlx_amd import engine
    pin = np.load(C1_PIN)
    cfg = dict(bench.QWEN3_0_6B)
    m = engine.Model(max_context=256, **cfg)
    m.synth_weights()
    want = pin["tokens"]
    bound = 2.0 ** -7 * float(pin["max_abs"].max()) * np.sqrt(cfg["num_hidden_layers"])
    tok = int(m.prefill(pin["prompt"]))
    worst, equal = 0.0, 0
    for i in range(32):
        lg = m.last_logits()
        worst = max(worst, float(np.abs(lg[pin["top_idx"][i]] - pin["top_val"][i]).max()))
        equal += int(tok == want[i])
        if tok != want[i]:
            assert pin["margin"][i] <= 2 * bound and lg[want[i]] >= lg.max() - 2 * bound, f"step {i}: {tok} vs oracle {want[i]}"
        if i < 31:
            m.trim(0, int(want[i]))            # continue on the ORACLE's sequence
            tok = int(m.decode(1)[0])
    assert worst <= bound, f"top-8 logits off by {worst:.4f} over 32 steps (bound {bound:.4f})"
    _report("qwen3_0p6b_c1_pin (128 + 32, oracle tokens forced)", "engine", worst, float(pin["max_abs"].max()), cfg["num_hidden_layers"], "sqrt(L)", f"{equal} of 32")
    assert equal >= 16, f"only {equal} of 32 greedy tokens equal the oracle's"
    m.close()


QWEN3_0_6B_UNTIED = dict(hidden_size=1024, num_hidden_layers=28, intermediate_size=3072, num_attention_heads=16, num_key_value_heads=8,
                         head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6)


def test_protocol_c2_iid_logits_with_the_oracle_tokens_forced(omx):
    """BASELINE configs[1] at Qwen3-8B's shapes on the PLAIN i.i.d. checkpoint (VERDICT r4 "Next" 4a): the C oracle ran the whole protocol --
    2 048-token prompt, 256 greedy tokens -- token by token (tools/protocol_pin.py c2 256 iid -> qwen3_c2_protocol_iid_pin.npz); here the
    prompt goes through ONE batched prefill and every later position through the decode step with the ORACLE's token forced
    (`trim(0, token)`), so both sides see the same 2 048 .. 2 304 tokens of context.  Compared: the top-8 LOGITS at steps 0, 1, 128, 255,
    256 within 2^-7 * max|logit| * sqrt(36) (~0.26: the engine's bf16 bound, not the peaked fixture's 3.8), the greedy token wherever the
    oracle's margin allows.  This is the oracle value at full shape beyond 16 tokens of context that the peaked fixture (a plumbing
    check: its tokens follow from the embedding and the head alone) cannot give."""
    from ominix_mlx_amd import engine
    path = os.path.join(os.path.dirname(__file__), "golden", "qwen3_c2_protocol_iid_pin.npz")
    if not os.path.exists(path):
        pytest.skip("qwen3_c2_protocol_iid_pin.npz not generated (tools/protocol_pin.py c2 256 iid)")
    pin = np.load(path)
    cfg = dict(bench.QWEN3_8B)
    n_prompt, want = int(pin["prompt_len"]), pin["tokens"]
    n_new = want.size - 1
    prompt = bench.prompt_ids(n_prompt, cfg["vocab_size"])
    m = engine.Model(max_context=n_prompt + n_new + 16, **cfg)
    m.synth_weights()
    bound = 2.0 ** -7 * float(pin["logit_absmax"]) * np.sqrt(cfg["num_hidden_layers"])
    pins = {int(s): i for i, s in enumerate(pin["pin_steps"])}
    # round 6 (VERDICT r5 "Next" 6a): the fixture pins every 16th step, a fixed 256-entry sample of the row beside its top-8, and keeps the
    # oracle's top-1 / top-2 margin of EVERY step -- so the token is asserted wherever the margin decides it, not at five steps only
    all_margins = pin["all_margins"] if "all_margins" in pin.files else None
    tok = int(m.prefill(prompt))
    equal, worst, worst_sub = 0, 0.0, 0.0
    for step in range(n_new + 1):
        equal += int(tok == int(want[step]))
        if all_margins is not None and float(all_margins[step]) > 2 * bound:
            assert tok == int(want[step]), f"step {step}: {tok} vs oracle {int(want[step])} at margin {float(all_margins[step]):.3f}"
        if step in pins:
            i = pins[step]
            lg = m.last_logits()
            err = float(np.abs(lg[pin["top_idx"][i]] - pin["top_val"][i]).max())
            worst = max(worst, err)
            assert err <= bound, f"step {step}: top-8 logits off by {err:.4f} (bound {bound:.4f})"
            if "sub_idx" in pin.files:
                sub = float(np.abs(lg[pin["sub_idx"]] - pin["sub_val"][i]).max())
                worst_sub = max(worst_sub, sub)
                assert sub <= bound, f"step {step}: sampled logits off by {sub:.4f} (bound {bound:.4f})"
            if float(pin["margins"][i]) > 2 * bound:
                assert tok == int(want[step]), f"step {step}: {tok} vs oracle {int(want[step])} at margin {float(pin['margins'][i]):.3f}"
            else:
                assert lg[int(want[step])] >= lg.max() - 2 * bound
        if step < n_new:
            m.trim(0, int(want[step]))           # continue on the ORACLE's sequence
            tok = int(m.decode(1)[0])
    print(f"c2 i.i.d. pin: worst top-8 logit deviation {worst:.4f} (bound {bound:.4f}); {equal} of {n_new + 1} greedy tokens equal the oracle's")
    _report(f"qwen3_c2_protocol_iid_pin (2 048 + 256, {len(pins)} pinned steps)", "engine", max(worst, worst_sub), float(pin["logit_absmax"]),
            cfg["num_hidden_layers"], "sqrt(L)", f"{equal} of {n_new + 1}")
    # flat synthetic logits: most margins are below the bound, where two bf16 pipelines may legitimately differ -- yet they agree on nine
    # tokens in ten (measured 235 of 257); an engine that agreed on a quarter (the old assert) would be broken
    assert equal >= int(0.8 * (n_new + 1)), f"only {equal} of {n_new + 1} greedy tokens equal the oracle's"
    m.close()


def test_protocol_c2_iid_logits_of_the_drop_in_route(omx):
    """The same fixture against the DROP-IN route (VERDICT r5 "Next" 3): qwen3-mlx's forward replayed call for call through the mlx-c ABI
    (csrc/per_op_route.hip) with the deferred list rewriting the decode idioms onto the fused GEMV family (csrc/mlxc_lazy.hpp) -- compared
    with the ORACLE, not with the engine: top-8 logits (and the fixed 256-entry sample where the fixture carries one) at every pinned step
    within the same bf16 bound, the oracle's tokens forced."""
    from ominix_mlx_amd import engine
    path = os.path.join(os.path.dirname(__file__), "golden", "qwen3_c2_protocol_iid_pin.npz")
    if not os.path.exists(path):
        pytest.skip("qwen3_c2_protocol_iid_pin.npz not generated (tools/protocol_pin.py c2 256 iid)")
    pin = np.load(path)
    cfg = dict(bench.QWEN3_8B)
    n_prompt, want = int(pin["prompt_len"]), pin["tokens"]
    prompt = bench.prompt_ids(n_prompt, cfg["vocab_size"])
    m = engine.Model(max_context=64, **cfg)          # (only its weights are used: the route keeps its own cache arrays)
    m.synth_weights()
    bound = 2.0 ** -7 * float(pin["logit_absmax"]) * np.sqrt(cfg["num_hidden_layers"])
    steps = [int(s) for s in pin["pin_steps"]]
    got = m.per_op_route_forced(prompt, want[:-1], steps)
    worst, worst_sub = 0.0, 0.0
    for i, step in enumerate(steps):
        lg = got["logits"][i]
        err = float(np.abs(lg[pin["top_idx"][i]] - pin["top_val"][i]).max())
        worst = max(worst, err)
        assert err <= bound, f"step {step}: top-8 logits of the route off by {err:.4f} (bound {bound:.4f})"
        if "sub_idx" in pin.files:
            worst_sub = max(worst_sub, float(np.abs(lg[pin["sub_idx"]] - pin["sub_val"][i]).max()))
        tok = int(got["tokens"][step])
        if float(pin["margins"][i]) > 2 * bound:
            assert tok == int(want[step]), f"step {step}: route {tok} vs oracle {int(want[step])} at margin {float(pin['margins'][i]):.3f}"
        else:
            assert lg[int(want[step])] >= lg.max() - 2 * bound
    assert worst_sub <= bound, f"sampled logits of the route off by {worst_sub:.4f} (bound {bound:.4f})"
    equal = int((got["tokens"].astype(np.int64) == want).sum())
    print(f"c2 i.i.d. pin, drop-in route: worst top-8 deviation {worst:.4f}, sample {worst_sub:.4f} (bound {bound:.4f}); {equal} of {want.size} greedy tokens equal")
    _report(f"qwen3_c2_protocol_iid_pin (2 048 + 256, {len(steps)} pinned steps)", "drop-in route (mlx-c ABI, deferred + fused)", max(worst, worst_sub),
            float(pin["logit_absmax"]), cfg["num_hidden_layers"], "sqrt(L)", f"{equal} of {want.size}")
    m.close()


def _protocol_pin(omx, which, cfg):
    """PLUMBING CHECK, not an arithmetic pin: on the peaked checkpoint the greedy successor of token t is t - 1 by construction
    of the embedding and the head alone -- an engine with its attention zeroed would pass; the arithmetic at these shapes is pinned by
    test_protocol_c2_iid_logits_with_the_oracle_tokens_forced and test_full_size_engine_matches_full_size_oracle.  What this holds: the
    protocol's plumbing (batched prefill of 2 048 tokens, 256 decode steps, context buckets, ring read-back) delivers the right tokens.
    One of BASELINE.json's decode protocols at the model's real shapes on the PEAKED synthetic checkpoint (embedding std 64,
    lm_head[v] = table[(v + 1) mod V]: top-1 margins ~20 resp. ~80 against a bound below 1), against the C oracle's token-by-token run
    (tools/protocol_pin.py): the prompt in ONE batched prefill, then every new token through the decode step -- all token ids EQUAL,
    top-8 logits within the engine's bound at the pinned steps."""
    from ominix_mlx_amd import engine
    path = os.path.join(os.path.dirname(__file__), "golden", f"qwen3_{which}_protocol_pin.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} not generated (tools/protocol_pin.py {which})")
    pin = np.load(path)
    n_prompt, want = int(pin["prompt_len"]), pin["tokens"]
    n_new = want.size - 1
    prompt = bench.prompt_ids(n_prompt, cfg["vocab_size"])
    m = engine.Model(max_context=n_prompt + n_new + 16, **cfg)
    m.synth_weights(peaked=True)
    bound = 2.0 ** -7 * float(pin["logit_absmax"]) * np.sqrt(cfg["num_hidden_layers"])
    assert float(pin["margins"].min()) > 4 * bound, "the fixture's margins must make token equality a fair demand"
    got, logits_at = [int(m.prefill(prompt))], {0: m.last_logits()}
    pins = {int(s): i for i, s in enumerate(pin["pin_steps"])}
    step = 0
    while step < n_new:                      # decode in runs that end on the pinned steps (their logits are read back)
        nxt = min([s for s in pins if s > step] + [n_new])
        got += [int(t) for t in m.decode(nxt - step)]
        step = nxt
        if step in pins:
            logits_at[step] = m.last_logits()
    np.testing.assert_array_equal(np.asarray(got, np.int64), want)
    assert got == [(int(prompt[-1]) - 1 - i) % cfg["vocab_size"] for i in range(len(got))]     # (what the peaked weights encode)
    for s, i in pins.items():
        err = float(np.abs(logits_at[s][pin["top_idx"][i]] - pin["top_val"][i]).max())
        assert err <= bound, f"step {s}: top-8 logits off by {err:.4f} (bound {bound:.4f})"
    m.close()


def test_protocol_c1_token_ids_equal_at_real_shapes(omx):
    """BASELINE configs[0] (qwen3-mlx 0.5B greedy decode, 128-token prompt; 32 new tokens) at Qwen3-0.6B's shapes."""
    _protocol_pin(omx, "c1", QWEN3_0_6B_UNTIED)


def test_protocol_c2_token_ids_equal_at_real_shapes(omx):
    """BASELINE configs[1] (qwen3-mlx 7B bf16, 2k prefill + 256 decode) at Qwen3-8B's shapes: 2 048-token prompt + 256 tokens."""
    _protocol_pin(omx, "c2", dict(bench.QWEN3_8B))


MIXTRAL_PIN = os.path.join(os.path.dirname(__file__), "golden", "mixtral_fullwidth_pin.npz")


@pytest.mark.skipif(not os.path.exists(MIXTRAL_PIN), reason="Mixtral full-width pin fixture not generated")
@pytest.mark.parametrize("fixture", ["mixtral_fullwidth_pin.npz", "mixtral_fullwidth_pin_6l.npz"])
def test_full_width_mixtral_matches_the_oracle_where_routing_cannot_flip(omx, monkeypatch, fixture):
    """Mixtral-8x7B's REAL widths (hidden 4096, 32 / 8 heads of 128, 8 experts of 4096 x 14336, top-2, vocabulary 32 000; depth cut to
    the fixture's n_layers so that the numpy oracle's f32 weights fit the build container) against the oracle, on a prompt grown so
    that at every (position, layer) the second and third router logit are further apart than twice the bf16 bound of an activation at
    that depth (tools/mixtral_pin.py): the two implementations must then choose the same experts, and the comparison is the usual one
    -- top-8 logits and a fixed 256-entry sample within 2^-7 max|logit| sqrt(2 layers) (see the bound below), tokens equal wherever the
    oracle's margin allows.  Two fixtures: 4 layers (round 4) and 6 layers (round 5, the deepest the build container's memory generates).
    Both routes: the batched pass (sorted grouped matrix-core GEMMs) at all positions, and the decode step (expert-selected GEMVs,
    weighted sum folded into the next launch) replayed position by position."""
    from ominix_mlx_amd import engine
    path = os.path.join(os.path.dirname(MIXTRAL_PIN), fixture)
    if not os.path.exists(path):
        pytest.skip(f"{fixture} not generated (tools/mixtral_pin.py)")
    pin = np.load(path)
    cfg = dict(bench.MIXTRAL_8X7B)
    cfg["num_hidden_layers"] = int(pin["n_layers"])
    prompt = pin["prompt"]
    n = prompt.size
    assert float(pin["route_min_rel_gap"]) > float(pin["route_safety"])
    m = engine.Model(max_context=64, **cfg)
    m.synth_weights()
    # The dense rule 2^-7 max|logit| sqrt(L) counts the bf16 roundings a dense block puts on the residual path: attention output, MLP output,
    # two residual sums.  A sparse-MoE block puts eight there -- attention output, TWO expert outputs, their two products with the routing
    # scores, the weighted sum, two residual sums (mixtral-mlx/src/model.rs:300-380) -- so independent roundings grow the deviation by
    # sqrt(8 / 4): bound = 2^-7 max|logit| sqrt(2 L).  (Round 4 used the dense rule at 4 layers and sat at 0.95 of it; at 6 layers the
    # engine is at 1.03 of the dense rule, 0.73 of this one -- the MoE engine tests use the looser 2 x dense.)
    bound = 2.0 ** -7 * float(pin["max_abs"].max()) * np.sqrt(2 * cfg["num_hidden_layers"])

    def check(lg, i, tok, what):
        worst = max(float(np.abs(lg[pin["top_idx"][i]] - pin["top_val"][i]).max()), float(np.abs(lg[pin["sub_idx"]] - pin["sub_val"][i]).max()))
        assert worst <= bound, f"{what}, position {i}: logits off by {worst:.4f} (bound {bound:.4f})"
        if pin["margin"][i] > 2 * bound:
            assert tok == pin["greedy"][i], f"{what}, position {i}: token {tok} != oracle {pin['greedy'][i]} (margin {pin['margin'][i]:.3f})"
        return worst

    got = m.verify(prompt)
    assert m.offset() == n
    worst_batched = max(check(m.verify_logits(i), i, int(got[i]), "batched route") for i in range(n))
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")
    m.reset()
    tok = int(m.prefill(prompt[:1]))
    worst_step = check(m.last_logits(), 0, tok, "decode step")
    for i in range(1, n):
        m.trim(0, int(prompt[i]))
        tok = int(m.decode(1)[0])
        worst_step = max(worst_step, check(m.last_logits(), i, tok, "decode step"))
    print(f"full-width Mixtral ({cfg['num_hidden_layers']} layers, {n} positions): worst logit deviation batched {worst_batched:.4f}, "
          f"step {worst_step:.4f}, bound {bound:.4f}")
    _report(f"{fixture} ({n} positions)", "engine, batched pass", worst_batched, float(pin["max_abs"].max()), cfg["num_hidden_layers"], "sqrt(2L)")
    _report(f"{fixture} ({n} positions)", "engine, decode step", worst_step, float(pin["max_abs"].max()), cfg["num_hidden_layers"], "sqrt(2L)")
    m.close()


def test_full_size_mixtral_routes_agree_when_routing_cannot_flip(omx, monkeypatch):
    """Mixtral-8x7B at its real shapes (32 layers, 8 experts of 4096 x 14336, 93 GB of bf16 weights generated on the device) has no
    oracle pin: with random weights top-2-of-8 routing flips under bf16 rounding.  The size-independent property
    that CAN be held at full size: with every expert selected (top-8 of 8: the selection cannot flip, the softmax weights, SwitchGLU and
    combine are all exercised) the batched route -- expert-selected GEMVs for 4 tokens, the sorted grouped GEMM for 16 -- and the
    decode-step route compute the same logits up to bf16 rounding."""
    from ominix_mlx_amd import engine
    cfg = dict(bench.MIXTRAL_8X7B)
    cfg["num_experts_per_tok"] = 8
    prompt = bench.prompt_ids(16, cfg["vocab_size"])
    m = engine.Model(max_context=64, **cfg)
    m.synth_weights()
    monkeypatch.setenv("OMX_PREFILL_SERIAL", "1")
    m.prefill(prompt[:1])
    step = [m.last_logits()]
    for i in range(1, 16):
        m.trim(0, int(prompt[i]))
        m.decode(1)
        step.append(m.last_logits())
    bound = 2.0 ** -7 * float(np.abs(step[0]).max()) * np.sqrt(cfg["num_hidden_layers"])
    for n in (4, 16):
        m.reset()
        m.verify(prompt[:n])
        worst = max(float(np.abs(m.verify_logits(i) - step[i]).max()) for i in range(n))
        assert worst <= 2 * bound, f"verify({n}) vs decode steps: {worst:.3f} over the whole vocabulary (bound {2 * bound:.3f})"
    m.close()
