"""Mixtral-8x7B at its REAL widths (hidden 4096, 32 / 8 heads of 128, 8 experts of 4096 x 14336, top-2, vocabulary 32 000) against the
numpy oracle (VERDICT r3 "Next" 7b) -- with a prompt on which top-2 routing CANNOT flip between two correct bf16 implementations.

With random weights the gap between the second and third router logit falls below the rounding noise of the hidden state at a few
percent of all (position, layer) pairs, so a free-running comparison at full size diverges for reasons that are nobody's bug
(DESIGN.md section 2).  Here the prompt is grown token by token: a candidate token is kept only if at EVERY layer the gap between its
second and third router logit exceeds 2 x 2^-6 x max|router logit| (twice the engine's own bound on a bf16 activation at this depth);
otherwise the caches are trimmed by one position and the next candidate is tried.  The depth is cut to NL layers (default 4: 23 GB
of f32 weights in this container; the widths, head counts, expert count and vocabulary are the real ones), the weights are the
device generator's (oracle/synth.py == omx_qwen3_synth_weights).

Writes tests/golden/mixtral_fullwidth_pin.npz: the prompt, and from ONE batched oracle pass over it the greedy token, top-8
(index, logit) pairs, a fixed 256-entry sample of the logits and the top-1 margin at every position, plus the smallest routing gap
seen.  tests/test_gpu_fullsize_pin.py replays it through the engine's batched route (grouped matrix-core GEMMs) and its decode step
(expert-selected GEMVs).  Build container only:   python tools/mixtral_pin.py [n_layers] [n_prompt]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ref_core as rc, ref_moe, ref_qwen3 as rq  # noqa: E402

NL = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N_PROMPT = int(sys.argv[2]) if len(sys.argv) > 2 else 12
d = dict(bench.MIXTRAL_8X7B)
cfg = rq.Qwen3Config(hidden_size=d["hidden_size"], num_hidden_layers=NL, intermediate_size=d["intermediate_size"],
                     num_attention_heads=d["num_attention_heads"], num_key_value_heads=d["num_key_value_heads"], head_dim=d["head_dim"],
                     vocab_size=d["vocab_size"], rms_norm_eps=d["rms_norm_eps"], rope_theta=d["rope_theta"], tie_word_embeddings=False,
                     num_experts=d["num_experts"], num_experts_per_tok=d["num_experts_per_tok"],
                     moe_intermediate_size=d["moe_intermediate_size"], moe_mode="mixtral", qk_norm=False)
t0 = time.time()
weights = rq.synth_weights(cfg)
print(f"weights of {NL} layers generated in {time.time() - t0:.0f} s "
      f"({sum(w.nbytes for w in weights.values()) / 1e9:.1f} GB as float32)", flush=True)
oracle = rq.Qwen3Oracle(cfg, weights)

gaps = []          # (gap between the 2nd and 3rd router logit, max |logit|) of every routing decision since the last clear
_route = ref_moe.route_logits_mixtral


def recording_route(gates, k, dt="bf16"):
    g = np.sort(np.asarray(gates, np.float64), axis=-1)[..., ::-1]
    for row in g.reshape(-1, g.shape[-1]):
        gaps.append((float(row[k - 1] - row[k]), float(np.abs(row).max())))
    return _route(gates, k, dt)


ref_moe.route_logits_mixtral = recording_route
SAFETY = 2.0 * 2.0 ** -6


def clear(decisions):
    return all(gap > SAFETY * mx for gap, mx in decisions)


caches, prompt, tries = [], [], 0
cand = iter(int(t) for t in bench.prompt_ids(4096, cfg.vocab_size))
t0 = time.time()
while len(prompt) < N_PROMPT:
    tok = next(cand)
    tries += 1
    gaps.clear()
    oracle.forward(np.array([[tok]]), caches)
    if clear(gaps):
        prompt.append(tok)
        print(f"  position {len(prompt) - 1}: token {tok} kept (smallest gap / max logit {min(g / m for g, m in gaps):.4f}, {time.time() - t0:.0f} s)", flush=True)
    else:
        for c in caches:
            c.trim(1)
print(f"{N_PROMPT} tokens kept out of {tries} candidates", flush=True)

# the fixture comes from ONE batched pass (the form Model.verify computes); its routing gaps are checked again
gaps.clear()
logits = oracle.forward(np.array([prompt]), [])[0].astype(np.float32)
assert clear(gaps), "the batched pass routes closer to a tie than the token-by-token pass did"
route_rel = min(g / m for g, m in gaps)
order = np.argsort(-logits, axis=-1, kind="stable")
top_idx = order[:, :8].astype(np.int64)
top_val = np.take_along_axis(logits, top_idx, axis=-1)
sub_idx = np.sort(np.random.default_rng(7).choice(cfg.vocab_size, 256, replace=False)).astype(np.int64)
out = os.path.join(ROOT, "tests", "golden", "mixtral_fullwidth_pin.npz" if NL == 4 else f"mixtral_fullwidth_pin_{NL}l.npz")
np.savez_compressed(out, n_layers=NL, prompt=np.asarray(prompt, np.int64), greedy=order[:, 0].astype(np.int64), top_idx=top_idx,
                    top_val=top_val, margin=(top_val[:, 0] - top_val[:, 1]).astype(np.float32), sub_idx=sub_idx,
                    sub_val=logits[:, sub_idx], max_abs=np.abs(logits).max(axis=-1).astype(np.float32),
                    route_min_rel_gap=np.float32(route_rel), route_safety=np.float32(SAFETY),
                    route_decisions=np.int64(len(gaps)))
print(f"{len(gaps)} routing decisions, smallest (2nd - 3rd logit) / max|logit| = {route_rel:.4f} (kept above {SAFETY:.4f}); "
      f"logit max {np.abs(logits).max():.3f}, top-1 margins {np.round(top_val[:, 0] - top_val[:, 1], 3).tolist()} -> {out}", flush=True)
