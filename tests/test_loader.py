"""CPU: the checkpoint formats next to the path: safetensors container, shard index,
config.json -> model arguments, Mixtral expert stacking (mixtral-mlx/src/model.rs:480-510)."""
import json
import os

import numpy as np
import pytest


@pytest.fixture()
def loader():
    import omx_import
    omx_import.load_package()
    from ominix_mlx_amd import loader
    return loader


def test_safetensors_roundtrip_matches_the_reference_container(loader, tmp_path):
    """Our parser against the `safetensors` package (the format MLX's load_safetensors reads, ops/io.rs:51-58)."""
    from safetensors.numpy import load_file, save_file
    g = np.random.default_rng(0)
    tensors = {"a.weight": g.standard_normal((7, 5)).astype(np.float32), "b": g.integers(0, 2 ** 32, (3, 4), dtype=np.uint32),
               "c.h": g.standard_normal((2, 3, 4)).astype(np.float16)}
    p = str(tmp_path / "x.safetensors")
    save_file(tensors, p)
    got = loader.read_safetensors(p)
    assert set(got) == set(tensors)
    for k in tensors:
        np.testing.assert_array_equal(got[k], tensors[k])
    # and the other way round: our writer, their reader (+ a BF16 tensor as raw bits)
    bits = (g.standard_normal((4, 8)).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    p2 = str(tmp_path / "y.safetensors")
    loader.write_safetensors(p2, {"w": bits, "v": tensors["a.weight"]}, bf16_names=("w",))
    back = loader.read_safetensors(p2)
    assert type(back["w"]).__name__ == "Bf16Bits" and back["w"].dtype == np.uint16
    np.testing.assert_array_equal(np.asarray(back["w"]), bits)
    from safetensors import safe_open
    with safe_open(p2, framework="np") as f:        # their reader on our file (numpy has no bf16: read the f32 tensor only)
        assert set(f.keys()) == {"w", "v"}
        np.testing.assert_array_equal(f.get_tensor("v"), tensors["a.weight"])
    with open(str(tmp_path / "bad.safetensors"), "wb") as f:
        f.write(b"\x01\x02")
    with pytest.raises(ValueError, match="short header"):
        loader.read_safetensors(str(tmp_path / "bad.safetensors"))


def test_sharded_directory_and_config(loader, tmp_path):
    d = str(tmp_path)
    json.dump({"hidden_size": 1024, "num_hidden_layers": 3, "intermediate_size": 3072, "num_attention_heads": 8,
               "num_key_value_heads": 2, "vocab_size": 4096, "rms_norm_eps": 1e-6, "rope_theta": 1e6, "tie_word_embeddings": True,
               "quantization": {"bits": 4, "group_size": 64}}, open(os.path.join(d, "config.json"), "w"))
    a = {"model.norm.weight": np.ones(8, np.float32)}
    b = {"model.layers.0.mlp.up_proj.weight": np.arange(12, dtype=np.uint32).reshape(3, 4)}
    loader.write_safetensors(os.path.join(d, "model-00001-of-00002.safetensors"), a)
    loader.write_safetensors(os.path.join(d, "model-00002-of-00002.safetensors"), b)
    json.dump({"metadata": {}, "weight_map": {"model.norm.weight": "model-00001-of-00002.safetensors",
                                             "model.layers.0.mlp.up_proj.weight": "model-00002-of-00002.safetensors"}},
              open(os.path.join(d, "model.safetensors.index.json"), "w"))
    w = loader.load_all_weights(d)
    assert set(w) == {"model.norm.weight", "model.layers.0.mlp.up_proj.weight"}
    args = loader.model_args(d)
    assert args["head_dim"] == 128 and args["quantization"] == {"bits": 4, "group_size": 64} and args["tie_word_embeddings"] is True
    with pytest.raises(FileNotFoundError):
        loader.load_all_weights(str(tmp_path / "nowhere"))


def test_mixtral_expert_stacking(loader):
    """experts.{e}.w1/w2/w3 -> switch_mlp.gate_proj/down_proj/up_proj stacked on a new leading axis, for every
    component of a quantized triplet; a missing expert is WeightNotFound; stacked checkpoints pass through."""
    E, L = 4, 2
    w = {}
    for l in range(L):
        for e in range(E):
            for old, shape in (("w1", (6, 8)), ("w2", (8, 6)), ("w3", (6, 8))):
                for comp in ("weight", "scales", "biases"):
                    w[f"model.layers.{l}.block_sparse_moe.experts.{e}.{old}.{comp}"] = np.full(shape, 100 * l + 10 * e + len(comp), np.float32)
        w[f"model.layers.{l}.block_sparse_moe.gate.weight"] = np.zeros((E, 8), np.float32)
    out = loader.sanitize_weights(w, L, E)
    g = out["model.layers.1.block_sparse_moe.switch_mlp.gate_proj.weight"]
    assert g.shape == (E, 6, 8) and g[2, 0, 0] == 100 + 20 + 6
    assert out["model.layers.0.block_sparse_moe.switch_mlp.down_proj.scales"].shape == (E, 8, 6)
    assert out["model.layers.0.block_sparse_moe.switch_mlp.up_proj.biases"].shape == (E, 6, 8)
    assert not any(".experts." in k for k in out) and "model.layers.0.block_sparse_moe.gate.weight" in out
    assert loader.sanitize_weights(out, L, E) is out
    del w["model.layers.1.block_sparse_moe.experts.3.w2.scales"]
    with pytest.raises(KeyError, match="WeightNotFound"):
        loader.sanitize_weights(w, L, E)


def test_sharders_keep_the_bf16_bit_pattern_marking(loader):
    """np.ascontiguousarray / np.concatenate / np.asarray drop the Bf16Bits subclass, after which
    the upload would value-convert bit patterns (0x3F80 -> 16256.0).  Every sharder must hand the marking on."""
    from ominix_mlx_amd import ep, klein, tp
    g = np.random.default_rng(1)
    bits = lambda *shape: (g.standard_normal(shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16).view(loader.Bf16Bits)
    w = bits(8, 6)
    for name in ("model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.o_proj.weight", "model.norm.weight"):
        s = tp.shard(name, w, 1, 2)
        assert isinstance(s, loader.Bf16Bits), name
    np.testing.assert_array_equal(np.asarray(tp.shard("x.mlp.down_proj.weight", w, 1, 2)), np.asarray(w)[:, 3:])
    tied = tp.shard_state_dict({"model.embed_tokens.weight": w}, 0, 2, tie_word_embeddings=True)
    assert isinstance(tied["lm_head.weight"], loader.Bf16Bits) and tied["lm_head.weight"].shape == (4, 6)
    e = ep.shard_experts(bits(4, 6, 8), 1, 2)
    assert isinstance(e, loader.Bf16Bits) and e.shape == (2, 6, 8)
    kw = {"double_blocks.0.img_attn.to_q.weight": bits(8, 8), "double_blocks.0.img_mlp.mlp_out.weight": bits(8, 12),
          "single_blocks.0.to_qkv_mlp.weight": bits(3 * 8 + 2 * 12, 8), "single_blocks.0.to_out.weight": bits(8, 8 + 12)}
    for k, v in klein.shard_state_dict(kw, 8, 12, 1, 2).items():
        assert isinstance(v, loader.Bf16Bits), k
    stacked = loader.sanitize_weights({f"model.layers.0.block_sparse_moe.experts.{e_}.{o}.weight": bits(6, 8)
                                       for e_ in range(2) for o in ("w1", "w2", "w3")}, 1, 2)
    assert all(isinstance(v, loader.Bf16Bits) for v in stacked.values())


def test_upload_rejects_unmarked_16_bit_integers(loader):
    """Tensor.from_numpy(dtype='bf16') must not value-convert a plain uint16 array (raises before touching the device)."""
    from ominix_mlx_amd import OmxError
    from ominix_mlx_amd.ops import Tensor
    with pytest.raises(OmxError, match="ambiguous"):
        Tensor.from_numpy(np.full((2, 2), 0x3F80, np.uint16), "bf16")


def test_expected_shape_table_follows_the_config(loader):
    """load_weights checks every known tensor against the shape the kernels will read."""
    from ominix_mlx_amd import engine
    c = engine.Qwen3Config(hidden_size=512, num_hidden_layers=2, intermediate_size=1536, num_attention_heads=8,
                           num_key_value_heads=4, head_dim=64, vocab_size=2048, tp_rank=1, tp_size=2, ep_size=1)
    es = lambda n: engine.expected_shape(c, n)
    assert es("model.layers.1.self_attn.q_proj.weight") == (256, 512) and es("model.layers.1.self_attn.k_proj.weight") == (128, 512)
    assert es("model.layers.0.self_attn.o_proj.weight") == (512, 256) and es("model.layers.0.mlp.down_proj.weight") == (512, 768)
    assert es("lm_head.weight") == (1024, 512) and es("model.embed_tokens.weight") == (2048, 512)
    assert es("model.layers.0.self_attn.q_norm.weight") == (64,) and es("model.layers.0.self_attn.q_proj.bias") == (256,)
    assert es("some.other.tensor") is None and es("model.layers.0.rotary_emb.inv_freq") is None
    c.tp_size, c.quant_bits, c.quant_group = 1, 4, 64
    assert es("model.layers.0.mlp.gate_proj.weight") == (1536, 64) and es("model.layers.0.mlp.gate_proj.scales") == (1536, 8)
    assert es("model.embed_tokens.biases") == (2048, 8)
    c.quant_bits, c.num_experts, c.moe_intermediate_size, c.ep_size = 0, 8, 256, 2
    assert es("model.layers.0.block_sparse_moe.switch_mlp.down_proj.weight") == (4, 512, 256)
    assert es("model.layers.0.mlp.switch_mlp.gate_proj.weight") == (4, 256, 512) and es("model.layers.0.mlp.gate.weight") == (8, 512)
