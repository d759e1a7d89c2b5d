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
