"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the Paraformer body pieces of SURVEY.md 8a row
a13.  Never imported by the product path.

Follows funasr-mlx/src/paraformer.rs (float32 model):
    SanmAttention::forward      :496-532   explicit softmax(q k^T * d^-1/2) v, FSMN depthwise conv (k=11, pad 5)
                                           over the v projection plus v, out_proj(attn) + fsmn
    FeedForward::forward        :560-570   Linear -> ReLU -> Linear
    SanmEncoderLayer::forward   :618-634   LN(1e-5) -> attention -> residual only if in_dim == dim -> LN -> FFN -> residual
    CIFPredictor::cif_fire      :779-879   integrate-and-fire, threshold 1.0, tail 0.45
Evaluated in float64 on the given inputs (the reference's f32 path); PARITY UNPINNED -- the reference tests only
constructor shapes for these (paraformer.rs:1592-1610) and "batch CIF == single CIF"
(examples/validate_correctness.rs part 2), the latter reproduced in tests/test_paraformer_oracle.py.
"""
from __future__ import annotations

import numpy as np


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def fsmn(v, w):
    """Depthwise Conv1d over time, NLC layout, weight [C, k], zero padding k//2, groups = C, no bias."""
    T, C = v.shape
    k = w.shape[1]
    pad = k // 2
    vp = np.pad(v, ((pad, pad), (0, 0)))
    out = np.zeros_like(v)
    for j in range(k):
        out += vp[j:j + T] * w[:, j][None, :]
    return out


def sanm_attention(x, p, heads):
    qkv = x @ p["qkv_w"].T + p["qkv_b"]
    dim = qkv.shape[1] // 3
    D = dim // heads
    q, k, v = qkv[:, :dim], qkv[:, dim:2 * dim], qkv[:, 2 * dim:]
    qh, kh, vh = (t.reshape(-1, heads, D).transpose(1, 0, 2) for t in (q, k, v))
    s = qh @ kh.transpose(0, 2, 1) * (np.float32(D) ** np.float32(-0.5))
    s = s - s.max(-1, keepdims=True)
    pr = np.exp(s)
    pr /= pr.sum(-1, keepdims=True)
    att = (pr @ vh).transpose(1, 0, 2).reshape(-1, dim)
    return att @ p["out_w"].T + p["out_b"] + (fsmn(v, p["fsmn_w"]) + v)


def sanm_encoder_layer(x, p, heads):
    x = np.asarray(x, np.float64)
    p = {k: np.asarray(v, np.float64) for k, v in p.items()}
    h = sanm_attention(layer_norm(x, p["norm1_w"], p["norm1_b"]), p, heads)
    dim = p["out_w"].shape[0]
    x = x + h if x.shape[1] == dim else h                             # :625-629
    h = layer_norm(x, p["norm2_w"], p["norm2_b"])
    h = np.maximum(h @ p["ffn_up_w"].T + p["ffn_up_b"], 0.0) @ p["ffn_down_w"].T + p["ffn_down_b"]
    return x + h


def cif_fire(hidden, alphas, threshold=1.0, tail_threshold=0.45):
    """hidden [B, T, H], alphas [B, T] (float32 arithmetic, as the reference) -> (frames [B, max, H], counts [B])."""
    hidden = np.asarray(hidden, np.float32)
    alphas = np.asarray(alphas, np.float32)
    B, T, H = hidden.shape
    all_frames, counts = [], []
    for b in range(B):
        integrate = np.float32(0.0)
        frame = np.zeros(H, np.float32)
        frames = []
        for t in range(T):
            alpha = alphas[b, t]
            completion = np.float32(1.0) - integrate
            integrate = np.float32(integrate + alpha)
            fire = integrate >= np.float32(threshold)
            if fire:
                integrate = np.float32(integrate - np.float32(1.0))
            cur = completion if fire else alpha
            remainds = np.float32(alpha - cur)
            frame = (frame + cur * hidden[b, t]).astype(np.float32)
            if fire:
                frames.append(frame.copy())
                frame = (remainds * hidden[b, t]).astype(np.float32)
        if integrate > np.float32(tail_threshold):
            frames.append(frame)
        all_frames.append(frames)
        counts.append(len(frames))
    mx = max(counts) if counts else 0
    out = np.zeros((B, mx, H), np.float32)
    for b, fr in enumerate(all_frames):
        for t, f in enumerate(fr):
            out[b, t] = f
    return out, np.array(counts, np.int32)


# ---- the rest of the model (paraformer.rs:418-439, 691-708, 761-768, 981-1053, 1144-1165, 1210-1230) ----

def position_encoding(T: int, dim: int) -> np.ndarray:
    """sinusoidal_position_encoding (:418-439): positions start at 1, [sin | cos] halves, float32 arithmetic."""
    half = dim // 2
    inc = np.float32(np.log(np.float32(10000.0))) / (np.float32(half) - np.float32(1.0))
    inv = np.exp(-(np.arange(half, dtype=np.float32)) * inc).astype(np.float32)
    st = (np.arange(1, T + 1, dtype=np.float32)[:, None] * inv[None, :]).astype(np.float32)
    return np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float64)


def encoder_embed(mel):
    """SanmEncoder::forward prologue (:697-703): x * sqrt(512) + PE."""
    mel = np.asarray(mel, np.float64)
    return mel * np.sqrt(np.float32(512.0)).astype(np.float64) + position_encoding(*mel.shape)


def predictor_alphas(enc, conv_w, conv_b, proj_w, proj_b):
    """CIFPredictor::compute_alphas (:761-768); conv_w in MLX layout [out, k, in], zero padding k//2."""
    enc = np.asarray(enc, np.float64)
    conv_w, conv_b, proj_w, proj_b = (np.asarray(a, np.float64) for a in (conv_w, conv_b, proj_w, proj_b))
    T, C = enc.shape
    k = conv_w.shape[1]
    pad = k // 2
    xp = np.pad(enc, ((pad, pad), (0, 0)))
    h = np.zeros((T, conv_w.shape[0]))
    for j in range(k):
        h += xp[j:j + T] @ conv_w[:, j, :].T
    h = np.maximum(h + conv_b, 0.0)
    z = h @ proj_w.reshape(-1) + proj_b.reshape(-1)[0]
    return 1.0 / (1.0 + np.exp(-z))


def decoder_layer(x, enc, p, heads):
    """ParaformerDecoderLayer::forward (:1030-1053) with cross_attention (:981-1017); FFN down has no bias (:1421)."""
    x, enc = np.asarray(x, np.float64), np.asarray(enc, np.float64)
    p = {k: np.asarray(v, np.float64) for k, v in p.items()}
    h = layer_norm(x, p["norm1_w"], p["norm1_b"])
    h = np.maximum(h @ p["ffn_up_w"].T + p["ffn_up_b"], 0.0)
    tgt = layer_norm(h, p["ffn_norm_w"], p["ffn_norm_b"]) @ p["ffn_down_w"].T
    h = layer_norm(tgt, p["norm2_w"], p["norm2_b"])
    x1 = x + (fsmn(h, p["fsmn_w"]) + h)
    h = layer_norm(x1, p["norm3_w"], p["norm3_b"])
    q = h @ p["q_w"].T + p["q_b"]
    kv = enc @ p["kv_w"].T + p["kv_b"]
    dim = q.shape[1]
    D = dim // heads
    qh, kh, vh = (t.reshape(-1, heads, D).transpose(1, 0, 2) for t in (q, kv[:, :dim], kv[:, dim:]))
    s = qh @ kh.transpose(0, 2, 1) * (np.float32(D) ** np.float32(-0.5))
    s = s - s.max(-1, keepdims=True)
    pr = np.exp(s)
    pr /= pr.sum(-1, keepdims=True)
    att = (pr @ vh).transpose(1, 0, 2).reshape(-1, dim)
    return x1 + (att @ p["out_w"].T + p["out_b"])


def decoder_tail(x, p):
    """ParaformerDecoder::forward tail (:1157-1165)."""
    x = np.asarray(x, np.float64)
    p = {k: np.asarray(v, np.float64) for k, v in p.items()}
    h = layer_norm(x, p["norm1_w"], p["norm1_b"])
    h = np.maximum(h @ p["up_w"].T + p["up_b"], 0.0)
    h = layer_norm(h, p["ffn_norm_w"], p["ffn_norm_b"]) @ p["down_w"].T
    return layer_norm(h, p["after_norm_w"], p["after_norm_b"]) @ p["out_w"].T + p["out_b"]


# checkpoint keys exactly as funasr-mlx's loader reads them (load_paraformer_weights, :1300-1477); conv weights in the
# PyTorch layout [out, in/groups, k] that `get_conv_weight` transposes (:1293-1298)
def checkpoint_shapes(cfg: dict) -> dict:
    E, F, D, G, V, k = cfg["encoder_dim"], cfg["encoder_ffn_dim"], cfg["decoder_dim"], cfg["decoder_ffn_dim"], cfg["vocab_size"], cfg["sanm_kernel_size"]
    in0 = cfg["n_mels"] * cfg["lfr_m"]
    ck = cfg["cif_l_order"] + cfg["cif_r_order"] + 1
    s = {}

    def enc_layer(prefix, in_dim):
        s.update({f"{prefix}.self_attn.linear_q_k_v.weight": (3 * E, in_dim), f"{prefix}.self_attn.linear_q_k_v.bias": (3 * E,),
                  f"{prefix}.self_attn.out_proj.weight": (E, E), f"{prefix}.self_attn.out_proj.bias": (E,),
                  f"{prefix}.self_attn.fsmn_block.weight": (E, 1, k),
                  f"{prefix}.ffn.up_proj.weight": (F, E), f"{prefix}.ffn.up_proj.bias": (F,),
                  f"{prefix}.ffn.down_proj.weight": (E, F), f"{prefix}.ffn.down_proj.bias": (E,),
                  f"{prefix}.norm1.weight": (in_dim,), f"{prefix}.norm1.bias": (in_dim,),
                  f"{prefix}.norm2.weight": (E,), f"{prefix}.norm2.bias": (E,)})

    enc_layer("encoder.encoders0.0", in0)
    for i in range(cfg["encoder_layers"] - 1):
        enc_layer(f"encoder.layers.{i}", E)
    s.update({"encoder.after_norm.weight": (E,), "encoder.after_norm.bias": (E,),
              "predictor.conv.weight": (E, E, ck), "predictor.conv.bias": (E,),
              "predictor.output_proj.weight": (1, E), "predictor.output_proj.bias": (1,)})
    for i in range(cfg["decoder_layers"]):
        p = f"decoder.layers.{i}"
        s.update({f"{p}.self_attn.fsmn_block.weight": (D, 1, k),
                  f"{p}.src_attn.q_proj.weight": (D, D), f"{p}.src_attn.q_proj.bias": (D,),
                  f"{p}.src_attn.linear_k_v.weight": (2 * D, E), f"{p}.src_attn.linear_k_v.bias": (2 * D,),
                  f"{p}.src_attn.out_proj.weight": (D, D), f"{p}.src_attn.out_proj.bias": (D,),
                  f"{p}.ffn.up_proj.weight": (G, D), f"{p}.ffn.up_proj.bias": (G,), f"{p}.ffn.down_proj.weight": (D, G),
                  f"{p}.feed_forward.norm.weight": (G,), f"{p}.feed_forward.norm.bias": (G,)})
        for n in ("norm1", "norm2", "norm3"):
            s.update({f"{p}.{n}.weight": (D,), f"{p}.{n}.bias": (D,)})
    t = "decoder.decoders3.0"
    s.update({f"{t}.norm1.weight": (D,), f"{t}.norm1.bias": (D,), f"{t}.ffn.up_proj.weight": (G, D), f"{t}.ffn.up_proj.bias": (G,),
              f"{t}.feed_forward.norm.weight": (G,), f"{t}.feed_forward.norm.bias": (G,), f"{t}.ffn.down_proj.weight": (D, G),
              "decoder.after_norm.weight": (D,), "decoder.after_norm.bias": (D,),
              "decoder.output_proj.weight": (V, D), "decoder.output_proj.bias": (V,)})
    return s


def synth_checkpoint(cfg: dict, seed: int = 5) -> dict:
    """Random checkpoint with the reference's keys (bf16-representable values): N(0, fan_in^-1/2) matrices, norm weights
    around 1, a positive predictor bias so that CIF fires a few tokens on short inputs."""
    from . import ref_core as rc
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in checkpoint_shapes(cfg).items():
        if name.endswith("norm.weight") or ".norm1.weight" in name or ".norm2.weight" in name or ".norm3.weight" in name or "after_norm.weight" in name:
            a = 1.0 + 0.05 * g.standard_normal(shape)
        elif name.endswith(".bias"):
            a = 0.05 * g.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = g.standard_normal(shape) / np.sqrt(fan_in)
        out[name] = rc.bf16_round(a.astype(np.float32))
    out["predictor.output_proj.bias"] = rc.bf16_round(np.array([0.3], np.float32))
    return out


def _enc_params(w, prefix):
    return {"norm1_w": w[f"{prefix}.norm1.weight"], "norm1_b": w[f"{prefix}.norm1.bias"],
            "qkv_w": w[f"{prefix}.self_attn.linear_q_k_v.weight"], "qkv_b": w[f"{prefix}.self_attn.linear_q_k_v.bias"],
            "out_w": w[f"{prefix}.self_attn.out_proj.weight"], "out_b": w[f"{prefix}.self_attn.out_proj.bias"],
            "fsmn_w": w[f"{prefix}.self_attn.fsmn_block.weight"][:, 0, :],
            "norm2_w": w[f"{prefix}.norm2.weight"], "norm2_b": w[f"{prefix}.norm2.bias"],
            "ffn_up_w": w[f"{prefix}.ffn.up_proj.weight"], "ffn_up_b": w[f"{prefix}.ffn.up_proj.bias"],
            "ffn_down_w": w[f"{prefix}.ffn.down_proj.weight"], "ffn_down_b": w[f"{prefix}.ffn.down_proj.bias"]}


def _dec_params(w, p):
    return {"norm1_w": w[f"{p}.norm1.weight"], "norm1_b": w[f"{p}.norm1.bias"],
            "ffn_up_w": w[f"{p}.ffn.up_proj.weight"], "ffn_up_b": w[f"{p}.ffn.up_proj.bias"],
            "ffn_norm_w": w[f"{p}.feed_forward.norm.weight"], "ffn_norm_b": w[f"{p}.feed_forward.norm.bias"],
            "ffn_down_w": w[f"{p}.ffn.down_proj.weight"],
            "norm2_w": w[f"{p}.norm2.weight"], "norm2_b": w[f"{p}.norm2.bias"], "fsmn_w": w[f"{p}.self_attn.fsmn_block.weight"][:, 0, :],
            "norm3_w": w[f"{p}.norm3.weight"], "norm3_b": w[f"{p}.norm3.bias"],
            "q_w": w[f"{p}.src_attn.q_proj.weight"], "q_b": w[f"{p}.src_attn.q_proj.bias"],
            "kv_w": w[f"{p}.src_attn.linear_k_v.weight"], "kv_b": w[f"{p}.src_attn.linear_k_v.bias"],
            "out_w": w[f"{p}.src_attn.out_proj.weight"], "out_b": w[f"{p}.src_attn.out_proj.bias"]}


def _tail_params(w):
    t = "decoder.decoders3.0"
    return {"norm1_w": w[f"{t}.norm1.weight"], "norm1_b": w[f"{t}.norm1.bias"], "up_w": w[f"{t}.ffn.up_proj.weight"],
            "up_b": w[f"{t}.ffn.up_proj.bias"], "ffn_norm_w": w[f"{t}.feed_forward.norm.weight"],
            "ffn_norm_b": w[f"{t}.feed_forward.norm.bias"], "down_w": w[f"{t}.ffn.down_proj.weight"],
            "after_norm_w": w["decoder.after_norm.weight"], "after_norm_b": w["decoder.after_norm.bias"],
            "out_w": w["decoder.output_proj.weight"], "out_b": w["decoder.output_proj.bias"]}


def transcribe_from_mel(mel, w: dict, cfg: dict):
    """Paraformer::transcribe_from_mel (:1236-1256) for one utterance: mel [T, n_mels*lfr_m] ->
    (token ids [N], logits [N, V], encoder_out [T, E], alphas [T], acoustic_embeds [N, E])."""
    h = encoder_embed(mel)
    h = sanm_encoder_layer(h, _enc_params(w, "encoder.encoders0.0"), cfg["encoder_heads"])
    for i in range(cfg["encoder_layers"] - 1):
        h = sanm_encoder_layer(h, _enc_params(w, f"encoder.layers.{i}"), cfg["encoder_heads"])
    enc = layer_norm(h, np.asarray(w["encoder.after_norm.weight"], np.float64), np.asarray(w["encoder.after_norm.bias"], np.float64))
    conv_w = np.asarray(w["predictor.conv.weight"]).transpose(0, 2, 1)                     # get_conv_weight: [out, in, k] -> [out, k, in]
    alphas = predictor_alphas(enc, conv_w, w["predictor.conv.bias"], w["predictor.output_proj.weight"], w["predictor.output_proj.bias"])
    frames, counts = cif_fire(enc[None].astype(np.float32), alphas[None].astype(np.float32), cfg["cif_threshold"], cfg["cif_tail_threshold"])
    n = int(counts[0])
    if n == 0:
        return np.zeros(0, np.int32), np.zeros((0, cfg["vocab_size"])), enc, alphas, np.zeros((0, enc.shape[1]))
    x = frames[0, :n].astype(np.float64)
    for i in range(cfg["decoder_layers"]):
        x = decoder_layer(x, enc, _dec_params(w, f"decoder.layers.{i}"), cfg["decoder_heads"])
    logits = decoder_tail(x, _tail_params(w))
    return np.argmax(logits, -1).astype(np.int32), logits, enc, alphas, frames[0, :n]
