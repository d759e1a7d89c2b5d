"""Fixtures of BASELINE.json's full decode protocols at the REAL shapes with token ids as a hard assert (VERDICT r3 "Next" 7a):
the C port of the oracle (oracle/c/omx_oracle.c, OpenMP; cross-checked against the numpy oracle by tests/test_oracle_c.py) runs the
PEAKED synthetic checkpoint (oracle/ref_qwen3.py synth_weights(peaked=True) == omx_qwen3_synth_weights_peaked on the device) token by
token through

    c1: Qwen3-0.6B shapes (28 layers, hidden 1024, 16 / 8 heads of 128, FFN 3072, vocabulary 151 936), 128-token prompt + 32 tokens
    c2: Qwen3-8B shapes   (36 layers, hidden 4096, 32 / 8 heads of 128, FFN 12288, vocabulary 151 936), 2 048-token prompt + 256 tokens

and writes tests/golden/qwen3_<c1|c2>_protocol_pin.npz: the greedy tokens of the whole protocol, and for a handful of steps the
top-8 (index, logit) pairs + the top-1 / top-2 margin.  tests/test_gpu_fullsize_pin.py replays both on the GPU and asserts the token
ids EQUAL.  Build container only (c2: ~17 GB of memory, 0.5-1 h on 8 cores):

    python tools/protocol_pin.py c1|c2 [new_tokens]

Round 5 (VERDICT r4 "Next" 4a): the PEAKED fixture is a plumbing check -- its margins (74-80 against a bound of ~3.8) come from the
embedding and the head alone, an engine with its attention zeroed passes it.  `python tools/protocol_pin.py c2 256 iid` runs the same
protocol on the PLAIN i.i.d. checkpoint (embedding std 0.02, its own lm_head: what synth_weights() builds on the device), greedy on the
oracle's own tokens, and writes tests/golden/qwen3_c2_protocol_iid_pin.npz: all tokens, and at the pinned steps the top-8 (index,
logit) pairs, the top-1 / top-2 margin and the largest |logit|.  The GPU test replays it with the ORACLE's tokens forced and compares
LOGITS within 2^-7 * max|logit| * sqrt(layers) at 2 048 .. 2 304 tokens of context.
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import c_oracle, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c1"
CFG = {"c1": (dict(hidden_size=1024, num_hidden_layers=28, intermediate_size=3072, num_attention_heads=16, num_key_value_heads=8,
                   head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6), 128, 32),
       "c2": (dict(bench.QWEN3_8B), 2048, 256)}[which]
cfg, n_prompt, n_new = CFG
if len(sys.argv) > 2:
    n_new = int(sys.argv[2])
IID = len(sys.argv) > 3 and sys.argv[3] == "iid"
lib = c_oracle.load()
hd, I, H, Hkv, D, V, L = (cfg["hidden_size"], cfg["intermediate_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"],
                          cfg["head_dim"], cfg["vocab_size"], cfg["num_hidden_layers"])
cap = n_prompt + n_new + 8
AMP_W, AMP_N, AMP_E = np.float32(0.02 * np.sqrt(3.0)), np.float32(0.01 * np.sqrt(3.0)), np.float32(64.0 * np.sqrt(3.0))


def tensor(name, n, amp, off=0.0):
    a = np.empty(n, np.uint16)
    lib.oracle_fill_uniform_bf16(c_oracle.ptr(a), n, synth.name_seed(name), np.float32(amp), np.float32(off))
    return a


t0 = time.time()
layers = []
for i in range(L):
    p = f"model.layers.{i}."
    arrs = [tensor(p + "self_attn.q_proj.weight", H * D * hd, AMP_W), tensor(p + "self_attn.k_proj.weight", Hkv * D * hd, AMP_W),
            tensor(p + "self_attn.v_proj.weight", Hkv * D * hd, AMP_W), tensor(p + "self_attn.o_proj.weight", hd * H * D, AMP_W),
            tensor(p + "mlp.gate_proj.weight", I * hd, AMP_W), tensor(p + "mlp.up_proj.weight", I * hd, AMP_W),
            tensor(p + "mlp.down_proj.weight", hd * I, AMP_W), tensor(p + "self_attn.q_norm.weight", D, AMP_N, 1.0),
            tensor(p + "self_attn.k_norm.weight", D, AMP_N, 1.0), tensor(p + "input_layernorm.weight", hd, AMP_N, 1.0),
            tensor(p + "post_attention_layernorm.weight", hd, AMP_N, 1.0), np.zeros(Hkv * cap * D, np.uint16), np.zeros(Hkv * cap * D, np.uint16)]
    layers.append((arrs, c_oracle.Layer(*[c_oracle.ptr(a) for a in arrs])))
if IID:
    embed = tensor("model.embed_tokens.weight", V * hd, AMP_W).reshape(V, hd)
    head = tensor("lm_head.weight", V * hd, AMP_W)
else:
    embed = tensor("model.embed_tokens.weight", V * hd, AMP_E).reshape(V, hd)
    head = np.ascontiguousarray(np.roll(tensor("model.embed_tokens.weight", V * hd, AMP_W).reshape(V, hd), -1, axis=0)).reshape(-1)
norm_w = tensor("model.norm.weight", hd, AMP_N, 1.0)
lc = c_oracle.LayerCfg(hd, I, H, Hkv, D, cap, cfg["rms_norm_eps"], cfg["rope_theta"], 1.0)
scratch = np.zeros(lib.oracle_qwen3_scratch_elems(ctypes.byref(lc)) + hd, np.uint16)
logits = np.empty(V, np.uint16)
print(f"{which}: weights generated in {time.time() - t0:.0f} s", flush=True)

prompt = bench.prompt_ids(n_prompt, V)
# step 0 = the token sampled from the prompt.  Round 6 (VERDICT r5 "Next" 6a): the i.i.d. fixture pins every 16th step, and beside the
# top-8 a FIXED sample of 256 vocabulary entries (so that the bulk of the row is held, not only its maximum); the top-1 / top-2 margin is
# kept for EVERY step, which is what says how many greedy tokens two bf16 pipelines can be required to share
pin_steps = sorted({0, 1, n_new // 2, n_new - 1, n_new} | (set(range(0, n_new + 1, 16)) if IID else set()))
sub_idx = np.sort(np.random.default_rng(20260603).choice(V, 256, replace=False)).astype(np.int64)
tokens, top_idx, top_val, margins, absmax, sub_val, all_margins = [], [], [], [], [], [], []
tok = None
t0 = time.time()
for pos in range(n_prompt + n_new):
    cur = int(prompt[pos]) if pos < n_prompt else tok
    h = np.ascontiguousarray(embed[cur]).copy()
    for _, Ly in layers:
        lib.oracle_qwen3_layer_decode(ctypes.byref(lc), ctypes.byref(Ly), c_oracle.ptr(h), pos, c_oracle.ptr(scratch))
    if pos >= n_prompt - 1:      # the reference computes and discards the logits of the earlier prompt positions (model.rs:815)
        tok = int(lib.oracle_qwen3_head(c_oracle.ptr(h), c_oracle.ptr(norm_w), c_oracle.ptr(head), hd, V, cfg["rms_norm_eps"],
                                       c_oracle.ptr(logits), c_oracle.ptr(scratch)))
        tokens.append(tok)
        step = pos - (n_prompt - 1)
        lf = (logits.astype(np.uint32) << np.uint32(16)).view(np.float32)
        two = np.partition(lf, V - 2)[V - 2:]
        all_margins.append(float(two[1] - two[0]))
        if step in pin_steps:
            order = np.argsort(-lf, kind="stable")[:8]
            sub_val.append(lf[sub_idx].copy())
            top_idx.append(order.astype(np.int64)); top_val.append(lf[order]); margins.append(float(lf[order[0]] - lf[order[1]]))
            absmax.append(float(np.abs(lf).max()))
    if pos % 64 == 0:
        print(f"  position {pos} / {n_prompt + n_new}  ({time.time() - t0:.0f} s)", flush=True)
out = os.path.join(ROOT, "tests", "golden", f"qwen3_{which}_protocol_{'iid_' if IID else ''}pin.npz")
np.savez_compressed(out, prompt_len=n_prompt, tokens=np.asarray(tokens, np.int64), pin_steps=np.asarray(pin_steps, np.int64),
                    top_idx=np.stack(top_idx), top_val=np.stack(top_val), margins=np.asarray(margins, np.float32),
                    sub_idx=sub_idx, sub_val=np.stack(sub_val), all_margins=np.asarray(all_margins, np.float32),
                    logit_absmax=np.float32(max(absmax) if IID else np.abs(np.stack(top_val)).max()))
expect = [(int(prompt[-1]) - 1 - i) % V for i in range(len(tokens))]
print(f"{which}{' iid' if IID else ''}: {len(tokens)} tokens in {time.time() - t0:.0f} s, counting down from the last prompt token: {tokens == expect}; "
      f"margins {np.round(margins, 3).tolist()} -> {out}", flush=True)
