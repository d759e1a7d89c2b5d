"""Generation front-end: tokenizer loading, chat template rendering
(mlx-rs/mlx-lm-utils/src/tokenizer.rs:242-258, 430-530) and the streaming text loop of
qwen3-mlx/examples/generate_qwen3.rs:31-101.  The reference's own tests for this need a downloaded Qwen3-4B
(tokenizer.rs:543-545, "how to test this in CI?"); here a small word-level tokenizer and a ChatML template are
built on the spot with the same libraries' file formats."""
import json

import numpy as np
import pytest


CHATML = ("{% for message in messages %}{{ '<|im_start|>' + message['role'] + '\\n' + message['content'] + '<|im_end|>' + '\\n' }}"
          "{% endfor %}{% if add_generation_prompt %}{{ '<|im_start|>assistant\\n' }}{% endif %}")


def _gen():
    import omx_import
    omx_import.load_package()
    from ominix_mlx_amd import generate
    return generate


def _write_tokenizer(path, vocab_size):
    from tokenizers import Tokenizer, models, pre_tokenizers
    words = ["<unk>", "<|im_start|>", "<|im_end|>"] + [f"w{i}" for i in range(vocab_size - 3)]
    tok = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.add_special_tokens(["<unk>", "<|im_start|>", "<|im_end|>"])
    tok.save(str(path / "tokenizer.json"))
    (path / "tokenizer_config.json").write_text(json.dumps({"chat_template": CHATML, "eos_token": "<|im_end|>"}))
    return tok


def test_chat_template_loading_and_rendering(tmp_path):
    g = _gen()
    _write_tokenizer(tmp_path, 64)
    tpl = g.load_model_chat_template_from_file(tmp_path / "tokenizer_config.json")
    assert tpl == CHATML
    assert g.load_model_chat_template_from_str('{"other": 1}') is None
    chat = [{"role": "user", "content": "hello"}]
    assert g.apply_chat_template(tpl, [chat]) == ["<|im_start|>user\nhello<|im_end|>\n"]
    assert g.apply_chat_template(tpl, [chat], add_generation_prompt=True) == ["<|im_start|>user\nhello<|im_end|>\n<|im_start|>assistant\n"]
    two = g.apply_chat_template(tpl, [chat, chat + [{"role": "assistant", "content": "hi  "}]], continue_final_message=True)
    assert two[0] == "<|im_start|>user\nhello" and two[1].endswith("<|im_start|>assistant\nhi  ")
    with pytest.raises(ValueError):
        g.apply_chat_template(tpl, [chat], add_generation_prompt=True, continue_final_message=True)
    tok = g.load_tokenizer(tmp_path)
    enc = g.apply_chat_template_and_encode(tok, "{% for m in messages %}{{ m['content'] }} {% endfor %}", [[{"role": "user", "content": "w1 w2 w5"}]])
    assert list(enc[0].ids) == [4, 5, 8]
    with pytest.raises(FileNotFoundError):
        g.load_tokenizer(tmp_path / "missing")


@pytest.mark.gpu
@pytest.mark.parametrize("temperature", [0.0, 0.7])
def test_generate_text_streams_what_the_engine_generates(omx, tmp_path, temperature):
    from ominix_mlx_amd import generate
    from test_gpu_qwen3 import CONFIGS, _engine
    cfg = CONFIGS["gqa4_d128"]
    tok = _write_tokenizer(tmp_path, cfg.vocab_size)
    prompt = "w10 w11 w12 w500 w7"
    ids = tok.encode(prompt, add_special_tokens=True).ids
    chunks = []
    out = generate.generate_text(_engine(omx, cfg), generate.load_tokenizer(tmp_path), prompt, temperature=temperature,
                                 max_tokens=25, seed=3, emit=chunks.append)
    ref = _engine(omx, cfg)
    ref.set_sampler(temperature, 3)
    want = [ref.prefill(np.array(ids, np.uint32))] + [int(t) for t in ref.decode(24)]
    assert out["tokens"] == want and out["prompt_tokens"] == len(ids)
    assert len(chunks) == 3                                   # 10 + 10 + the flushed 5 (generate_qwen3.rs:62-82)
    assert out["text"] == "".join(chunks) == "".join(tok.decode(want[i:i + 10], skip_special_tokens=True) for i in (0, 10, 20))
    # a stop token ends the stream right after it is produced
    stop_at = want[7]
    short = generate.generate_text(_engine(omx, cfg), tok, prompt, temperature=temperature, max_tokens=25, seed=3,
                                   stop_token_ids=[stop_at])
    first = want.index(stop_at)
    assert short["tokens"] == want[:first + 1]
