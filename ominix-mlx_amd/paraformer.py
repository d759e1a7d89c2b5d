"""Host mirror of funasr-mlx's Paraformer (paraformer.rs): `SanmEncoderLayer` (:573-640), `SanmEncoder` (:646-717),
`CIFPredictor` (:723-889), `ParaformerDecoderLayer` (:900-1068), `ParaformerDecoder` (:1071-1180) and
`Paraformer::transcribe_from_mel` (:1236-1256) over the fused entry points of csrc/paraformer.hip."""
from __future__ import annotations

import ctypes

import numpy as np

from . import FLOAT32, INT32, check, lib
from .ops import Tensor

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p
_FIELDS = ("norm1_w", "norm1_b", "qkv_w", "qkv_b", "out_w", "out_b", "fsmn_w", "norm2_w", "norm2_b", "ffn_up_w", "ffn_up_b",
           "ffn_down_w", "ffn_down_b")


class SanmLayerWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in _FIELDS]


_DEC_FIELDS = ("norm1_w", "norm1_b", "ffn_up_w", "ffn_up_b", "ffn_norm_w", "ffn_norm_b", "ffn_down_w", "norm2_w", "norm2_b", "fsmn_w",
               "norm3_w", "norm3_b", "q_w", "q_b", "kv_w", "kv_b", "out_w", "out_b")
_TAIL_FIELDS = ("norm1_w", "norm1_b", "up_w", "up_b", "ffn_norm_w", "ffn_norm_b", "down_w", "after_norm_w", "after_norm_b", "out_w", "out_b")


class DecoderLayerWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in _DEC_FIELDS]


class TailWeights(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in _TAIL_FIELDS]


PARAFORMER_SIGNATURES = {
    "omx_paraformer_embed": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "omx_cif_alphas": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_paraformer_decoder_layer": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(DecoderLayerWeights), c_int, c_int, c_int,
                                             c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_paraformer_attention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_int, c_int,
                                             c_int, c_void_p]),
    "omx_paraformer_decoder_stack": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(DecoderLayerWeights), c_int, c_int, c_int, c_int, c_int,
                                             c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "omx_paraformer_decoder_tail": (c_int, [c_void_p, c_void_p, ctypes.POINTER(TailWeights), c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_cast": (c_int, [c_void_p, c_int, c_void_p, c_int, ctypes.c_int64, c_void_p]),
    "omx_sanm_encoder_layer": (c_int, [c_void_p, c_void_p, ctypes.POINTER(SanmLayerWeights), c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_sanm_encoder_stack": (c_int, [c_void_p, c_void_p, ctypes.POINTER(SanmLayerWeights), c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "omx_cif_fire": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_void_p]),
}
for _n, (_r, _a) in PARAFORMER_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


class SanmEncoderLayer:
    """weights: dict with the keys of SanmLayerWeights (Linear [out,in] + bias, fsmn_w [dim, k]), numpy arrays."""

    def __init__(self, weights: dict, heads: int = 4, kernel_size: int = 11, dtype: str = "f32"):
        """dtype "f32": the reference's arithmetic (f32 weights and activations); "bf16": narrower and faster."""
        self.dtype = dtype
        self._t = {k: Tensor.from_numpy(np.asarray(weights[k]), dtype) for k in _FIELDS}
        self.w = SanmLayerWeights(*[self._t[k].ptr for k in _FIELDS])
        self.heads, self.kernel_size = heads, kernel_size
        self.in_dim = self._t["qkv_w"].shape[1]
        self.dim = self._t["out_w"].shape[0]
        self.ffn_dim = self._t["ffn_up_w"].shape[0]

    def forward(self, x: Tensor) -> Tensor:
        T = x.shape[-2]
        out = Tensor(tuple(x.shape[:-1]) + (self.dim,), x.dtype)
        check(lib.omx_sanm_encoder_layer(out.ptr, x.ptr, ctypes.byref(self.w), T, self.in_dim, self.dim, self.heads, self.ffn_dim,
                                         self.kernel_size, out.dtype, None))
        return out


def cif_fire(hidden: Tensor, alphas: Tensor, threshold: float = 1.0, tail_threshold: float = 0.45):
    """-> (frames Tensor [B, max_tokens, H] f32, counts np.ndarray [B]); max_tokens is trimmed on the host like the
    reference pads to the longest item (paraformer.rs:841-872)."""
    B, T, H = hidden.shape
    frames = Tensor((B, T + 1, H), FLOAT32)
    counts = Tensor((B,), "u32")
    check(lib.omx_cif_fire(frames.ptr, counts.ptr, hidden.ptr, alphas.ptr, B, T, H, threshold, tail_threshold, T + 1, None))
    cnt = counts.numpy().astype(np.int32)
    mx = int(cnt.max()) if B else 0
    return frames.numpy()[:, :mx], cnt


DEFAULT_CONFIG = dict(n_mels=80, lfr_m=7, lfr_n=6, encoder_dim=512, encoder_layers=50, encoder_heads=4, encoder_ffn_dim=2048,
                      decoder_dim=512, decoder_layers=16, decoder_heads=4, decoder_ffn_dim=2048, vocab_size=8404,
                      sanm_kernel_size=11, cif_l_order=1, cif_r_order=1, cif_threshold=1.0, cif_tail_threshold=0.45)   # :110-145


def _cast(x: Tensor, dtype) -> Tensor:
    out = Tensor(x.shape, dtype)
    check(lib.omx_cast(out.ptr, out.dtype, x.ptr, x.dtype, x.size, None))
    return out


class Paraformer:
    """Encoder -> CIF predictor -> decoder of one utterance on one GPU (the frontend is audio.MelFrontend).
    `weights`: the checkpoint dict the reference's loader reads (load_paraformer_weights, :1300-1477), conv
    weights in the PyTorch layout it transposes (:1293-1298)."""

    def __init__(self, weights: dict, config: dict = None, dtype: str = "f32"):
        """dtype "f32" (default): the reference's arithmetic -- f32 weights as its loader converts them (paraformer.rs:1300-1477),
        f32 activations, exact-f32 matrix-core GEMMs; "bf16": bf16 weights / activations with fp32 accumulation."""
        self.cfg = dict(DEFAULT_CONFIG, **(config or {}))
        self.dtype = dtype
        c, self._keep = self.cfg, []
        self._scratch_bufs = {}
        if c["cif_l_order"] != c["cif_r_order"]:
            raise ValueError("CIF asymmetric padding (l_order != r_order) not yet supported")   # :736-740

        def dev(key, transform=None):
            if key not in weights:
                raise KeyError(f"Missing weight: {key}")                                       # get_weight, :1287-1291
            a = np.asarray(weights[key])
            t = Tensor.from_numpy(transform(a) if transform else a, dtype)
            self._keep.append(t)
            return t.ptr

        depthwise = lambda a: a[:, 0, :]                                  # [C, 1, k] -> [C, k]
        dense_conv = lambda a: np.ascontiguousarray(a.transpose(0, 2, 1))  # [out, in, k] -> [out, k, in]

        def enc_layer(p):
            return SanmLayerWeights(dev(f"{p}.norm1.weight"), dev(f"{p}.norm1.bias"), dev(f"{p}.self_attn.linear_q_k_v.weight"),
                                    dev(f"{p}.self_attn.linear_q_k_v.bias"), dev(f"{p}.self_attn.out_proj.weight"),
                                    dev(f"{p}.self_attn.out_proj.bias"), dev(f"{p}.self_attn.fsmn_block.weight", depthwise),
                                    dev(f"{p}.norm2.weight"), dev(f"{p}.norm2.bias"), dev(f"{p}.ffn.up_proj.weight"),
                                    dev(f"{p}.ffn.up_proj.bias"), dev(f"{p}.ffn.down_proj.weight"), dev(f"{p}.ffn.down_proj.bias"))

        self.enc_layers = [enc_layer("encoder.encoders0.0")] + [enc_layer(f"encoder.layers.{i}") for i in range(c["encoder_layers"] - 1)]
        self.after_norm = (dev("encoder.after_norm.weight"), dev("encoder.after_norm.bias"))
        self.pred = (dev("predictor.conv.weight", dense_conv), dev("predictor.conv.bias"), dev("predictor.output_proj.weight"),
                     dev("predictor.output_proj.bias"))
        # the decoder's linear_k_v weights / biases of all layers back to back in ONE buffer each: omx_paraformer_decoder_stack then projects the
        # encoder output for every layer with a single GEMM (the projection does not depend on the decoder state)
        L, esz = c["decoder_layers"], (4 if dtype == "f32" else 2)
        for i in range(L):
            for leaf in ("weight", "bias"):
                if f"decoder.layers.{i}.src_attn.linear_k_v.{leaf}" not in weights:
                    raise KeyError(f"Missing weight: decoder.layers.{i}.src_attn.linear_k_v.{leaf}")
        kv_w = Tensor.from_numpy(np.concatenate([np.asarray(weights[f"decoder.layers.{i}.src_attn.linear_k_v.weight"]) for i in range(L)], axis=0), dtype)
        kv_b = Tensor.from_numpy(np.concatenate([np.asarray(weights[f"decoder.layers.{i}.src_attn.linear_k_v.bias"]) for i in range(L)], axis=0), dtype)
        self._keep += [kv_w, kv_b]
        kv_rows = kv_w.shape[0] // L
        self.dec_layers = []
        for i in range(L):
            p = f"decoder.layers.{i}"
            self.dec_layers.append(DecoderLayerWeights(
                dev(f"{p}.norm1.weight"), dev(f"{p}.norm1.bias"), dev(f"{p}.ffn.up_proj.weight"), dev(f"{p}.ffn.up_proj.bias"),
                dev(f"{p}.feed_forward.norm.weight"), dev(f"{p}.feed_forward.norm.bias"), dev(f"{p}.ffn.down_proj.weight"),
                dev(f"{p}.norm2.weight"), dev(f"{p}.norm2.bias"), dev(f"{p}.self_attn.fsmn_block.weight", depthwise),
                dev(f"{p}.norm3.weight"), dev(f"{p}.norm3.bias"), dev(f"{p}.src_attn.q_proj.weight"), dev(f"{p}.src_attn.q_proj.bias"),
                kv_w.ptr + i * kv_rows * kv_w.shape[1] * esz, kv_b.ptr + i * kv_rows * esz, dev(f"{p}.src_attn.out_proj.weight"),
                dev(f"{p}.src_attn.out_proj.bias")))
        # the layer tables the two stack entry points take, built once
        self._enc_arr = (SanmLayerWeights * len(self.enc_layers))(*self.enc_layers)
        self._dec_arr = (DecoderLayerWeights * len(self.dec_layers))(*self.dec_layers)
        t = "decoder.decoders3.0"
        self.tail = TailWeights(dev(f"{t}.norm1.weight"), dev(f"{t}.norm1.bias"), dev(f"{t}.ffn.up_proj.weight"), dev(f"{t}.ffn.up_proj.bias"),
                                dev(f"{t}.feed_forward.norm.weight"), dev(f"{t}.feed_forward.norm.bias"), dev(f"{t}.ffn.down_proj.weight"),
                                dev("decoder.after_norm.weight"), dev("decoder.after_norm.bias"), dev("decoder.output_proj.weight"),
                                dev("decoder.output_proj.bias"))

    def _scratch(self, name: str, shape, dt) -> Tensor:
        """A scratch buffer the model keeps between calls (hipMalloc / hipFree per call cost more than the launches they served, and hipFree
        waits for the device); calls are ordered on one stream, so the next call's launches queue behind the last reader."""
        need = int(np.prod(shape, dtype=np.int64)) * (4 if dt in ("f32", "u32") else 2)
        buf = self._scratch_bufs.get(name)
        if buf is None or buf.nbytes < need:                      # grows to the largest request seen, one buffer per role
            buf = self._scratch_bufs[name] = Tensor((need,), "u8")
        return Tensor(shape, dt, ptr=buf.ptr, owner=buf)

    def encode(self, mel: Tensor, out: Tensor = None) -> Tensor:
        """SanmEncoder::forward (:691-708): mel f32 [T, n_mels*lfr_m] (device) -> encoder_out [T, encoder_dim] (into `out` when given)."""
        from .ops import layer_norm
        c = self.cfg
        T, in0 = mel.shape[-2], c["n_mels"] * c["lfr_m"]
        dt = self.dtype
        h = self._scratch("enc_in", (T, in0), dt)
        check(lib.omx_paraformer_embed(h.ptr, mel.ptr, T, in0, h.dtype, None))
        # the layer loop + after_norm in one C call (round 6: no 50 ctypes round trips, and in float32 each layer's last launch computes
        # the next layer's norm1); scratch: two activations and two normalised inputs, alternating
        D = c["encoder_dim"]
        act = [self._scratch("enc_act0", (T, D), dt), self._scratch("enc_act1", (T, D), dt)]
        nrm = [self._scratch("enc_nrm0", (T, max(in0, D)), dt), self._scratch("enc_nrm1", (T, max(in0, D)), dt)]
        out = Tensor((T, D), dt) if out is None else out
        n, arr = len(self.enc_layers), self._enc_arr
        check(lib.omx_sanm_encoder_stack(out.ptr, h.ptr, arr, n, T, in0, D, c["encoder_heads"], c["encoder_ffn_dim"], c["sanm_kernel_size"],
                                         self.after_norm[0], self.after_norm[1], act[0].ptr, act[1].ptr, nrm[0].ptr, nrm[1].ptr, out.dtype, None))
        return out

    def predict(self, enc: Tensor, scratch: bool = False):
        """CIFPredictor::forward (:883-889): -> (acoustic_embeds f32 [N, E] device Tensor or None, N, alphas Tensor).  scratch: the returned
        tensors live in buffers the model keeps (valid until the next call) instead of fresh allocations."""
        c = self.cfg
        T, E = enc.shape
        new = (lambda name, shape, dt: self._scratch("cif_" + name, shape, dt)) if scratch else (lambda name, shape, dt: Tensor(shape, dt))
        alphas, hidden = new("alphas", (1, T), "f32"), new("hidden", (1, T, E), "f32")
        check(lib.omx_cif_alphas(alphas.ptr, hidden.ptr, enc.ptr, *self.pred, T, E, c["cif_l_order"] + c["cif_r_order"] + 1, enc.dtype, None))
        frames = new("frames", (1, T + 1, E), "f32")
        counts = new("counts", (1,), "u32")
        check(lib.omx_cif_fire(frames.ptr, counts.ptr, hidden.ptr, alphas.ptr, 1, T, E, c["cif_threshold"], c["cif_tail_threshold"], T + 1, None))
        n = int(counts.numpy()[0])
        return (frames.slice_rows(0, (n, E)) if n else None), n, alphas

    def decode(self, embeds: Tensor, enc: Tensor, out: Tensor = None) -> Tensor:
        """ParaformerDecoder::forward (:1144-1166): acoustic_embeds f32 [N, D] -> logits [N, vocab] (into `out` when given)."""
        c = self.cfg
        N, Ts = embeds.shape[0], enc.shape[0]
        dt = self.dtype
        x = embeds if dt == "f32" else _cast(embeds, dt)
        # the layer loop in one C call (round 6): the encoder output's k | v for all layers from one GEMM, each layer's norm1 from the previous
        # layer's last launch; scratch: two activations, two normalised inputs, the [Ts, layers * 2 * dim] projection
        D, n = c["decoder_dim"], len(self.dec_layers)
        act = [self._scratch("dec_act0", (N, D), dt), self._scratch("dec_act1", (N, D), dt)]
        nrm = [self._scratch("dec_nrm0", (N, D), dt), self._scratch("dec_nrm1", (N, D), dt)]
        kv_all = self._scratch("dec_kv", (Ts, n * 2 * D), dt)
        x_out = self._scratch("dec_out", (N, D), dt)
        arr = self._dec_arr
        check(lib.omx_paraformer_decoder_stack(x_out.ptr, x.ptr, enc.ptr, arr, n, N, Ts, D, c["encoder_dim"], c["decoder_heads"],
                                               c["decoder_ffn_dim"], c["sanm_kernel_size"], act[0].ptr, act[1].ptr, nrm[0].ptr, nrm[1].ptr,
                                               kv_all.ptr, x_out.dtype, None))
        x = x_out
        logits = Tensor((N, c["vocab_size"]), dt) if out is None else out
        check(lib.omx_paraformer_decoder_tail(logits.ptr, x.ptr, ctypes.byref(self.tail), N, c["decoder_dim"], c["decoder_ffn_dim"],
                                              c["vocab_size"], logits.dtype, None))
        return logits

    def transcribe_from_mel(self, mel: Tensor):
        """Paraformer::transcribe_from_mel (:1236-1256): -> (token ids np.int32 [N], N)."""
        # every intermediate in buffers the model keeps: a hipMalloc per tensor costs more than the launches that fill it, and a hipFree
        # waits for the device
        c = self.cfg
        enc = self.encode(mel, out=self._scratch("t_enc", (mel.shape[-2], c["encoder_dim"]), self.dtype))
        embeds, n, _ = self.predict(enc, scratch=True)
        if n == 0:
            return np.zeros(0, np.int32), 0
        logits = self.decode(embeds, enc, out=self._scratch("t_logits", (n, c["vocab_size"]), self.dtype))
        tokens = self._scratch("t_tokens", (n,), "u32")
        check(lib.omx_argmax(tokens.ptr, logits.ptr, n, c["vocab_size"], logits.dtype, None))
        return tokens.numpy().astype(np.int32), n


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


def random_checkpoint(cfg: dict = None, seed: int = 3) -> dict:
    """Synthetic checkpoint with the reference's keys and layouts (benchmarks; there is no network for real weights)."""
    cfg = dict(DEFAULT_CONFIG, **(cfg or {}))
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in checkpoint_shapes(cfg).items():
        if name.endswith("norm.weight") or any(f".{n}.weight" in name for n in ("norm1", "norm2", "norm3")):
            a = 1.0 + 0.05 * g.standard_normal(shape)
        elif name.endswith(".bias"):
            a = 0.05 * g.standard_normal(shape)
        else:
            a = g.standard_normal(shape) / np.sqrt(int(np.prod(shape[1:])))
        out[name] = a.astype(np.float32)
    out["predictor.output_proj.bias"] = np.array([0.3], np.float32)
    return out
