"""Generation front-end either side of the decode engine, host side only:

    load_tokenizer                       mlx-rs-core/src/lib.rs:73-76 (HF `tokenizer.json` through the `tokenizers` library)
    load_model_chat_template_from_*      mlx-rs/mlx-lm-utils/src/tokenizer.rs:242-258 (`chat_template` of tokenizer_config.json)
    apply_chat_template                  tokenizer.rs:430-530 (Jinja render of `messages`, add_generation_prompt,
                                         continue_final_message; the reference uses minijinja + pycompat, this uses jinja2)
    generate_text                        qwen3-mlx/examples/generate_qwen3.rs:31-101 (encode with special tokens,
                                         Generate at `temperature`, decode + emit every 10 tokens, flush the rest)

The model side is `engine.Model` / `engine.Generate` (the HIP decode engine); nothing here touches the GPU."""
from __future__ import annotations

import json
import os
import time
from typing import Callable, Iterable, List, Optional, Sequence


def load_tokenizer(model_dir):
    from tokenizers import Tokenizer
    path = os.path.join(os.fspath(model_dir), "tokenizer.json")
    if not os.path.exists(path):
        raise FileNotFoundError(f"Tokenizer: {path} not found")
    return Tokenizer.from_file(path)


def load_model_chat_template_from_str(content: str) -> Optional[str]:
    value = json.loads(content)
    tpl = value.get("chat_template") if isinstance(value, dict) else None
    return tpl if isinstance(tpl, str) else None


def load_model_chat_template_from_file(path) -> Optional[str]:
    with open(path, "r", encoding="utf-8") as fh:
        return load_model_chat_template_from_str(fh.read())


_ENV = None


def _jinja_env():
    global _ENV
    if _ENV is None:
        import jinja2
        from jinja2.sandbox import ImmutableSandboxedEnvironment

        def raise_exception(message):
            raise jinja2.exceptions.TemplateError(message)

        _ENV = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
        _ENV.globals["raise_exception"] = raise_exception
        _ENV.filters["tojson"] = lambda x, **kw: json.dumps(x, ensure_ascii=False, **{k: v for k, v in kw.items() if k == "indent"})
    return _ENV


def apply_chat_template(model_template: str, conversations: Sequence[Sequence[dict]], documents=None,
                        add_generation_prompt: Optional[bool] = None, continue_final_message: Optional[bool] = None) -> List[str]:
    """One rendered string per conversation (a conversation = list of {"role", "content"} messages)."""
    add_generation_prompt = bool(add_generation_prompt)
    continue_final_message = bool(continue_final_message)
    if add_generation_prompt and continue_final_message:
        raise ValueError("continue_final_message and add_generation_prompt are not compatible")
    template = _jinja_env().from_string(model_template)
    out = []
    for chat in conversations:
        chat = [dict(m) for m in chat]
        rendered = template.render(messages=chat, documents=documents, add_generation_prompt=add_generation_prompt)
        if continue_final_message:
            final = str(chat[-1]["content"])
            loc = rendered.rfind(final.strip())
            if loc < 0:
                raise ValueError("continue_final_message is set but the final message does not appear in the chat after "
                                 "applying the chat template")
            keep = len(final.lstrip())
            # the template kept the message's trailing spacing (or it has none): cut after it; otherwise after the trimmed text
            rendered = rendered[:loc + keep] if rendered[loc:loc + keep] == final else rendered[:loc + len(final.strip())]
        out.append(rendered)
    return out


def apply_chat_template_and_encode(tokenizer, model_template: str, conversations, **kw):
    """tokenizer.rs:128-160: render, then encode each string WITHOUT adding special tokens (the template wrote them)."""
    return [tokenizer.encode(text, add_special_tokens=False) for text in apply_chat_template(model_template, conversations, **kw)]


def generate_text(model, tokenizer, prompt: str, temperature: float = 0.7, max_tokens: int = 100, seed: int = 0,
                  emit: Optional[Callable[[str], None]] = None, flush_every: int = 10,
                  stop_token_ids: Optional[Iterable[int]] = None, prompt_ids: Optional[Sequence[int]] = None) -> dict:
    """generate_qwen3.rs:31-101.  Returns {"text", "tokens", "prompt_tokens", "seconds", "tokens_per_sec"}; `emit` receives
    each decoded chunk as the example prints it.  `stop_token_ids` (not in the example, which always runs max_tokens)
    ends the stream after such a token."""
    from .engine import Generate
    ids = list(prompt_ids) if prompt_ids is not None else list(tokenizer.encode(prompt, add_special_tokens=True).ids)
    if not ids:
        raise ValueError("generate_text: the prompt encodes to no tokens")
    stop = set(int(t) for t in stop_token_ids) if stop_token_ids is not None else set()
    start = time.perf_counter()
    pending: List[int] = []
    all_tokens: List[int] = []
    pieces: List[str] = []

    def flush():
        if pending:
            text = tokenizer.decode(pending, skip_special_tokens=True)
            pieces.append(text)
            if emit is not None:
                emit(text)
            pending.clear()

    for i, token in enumerate(Generate(model, temperature, ids, chunk=flush_every, seed=seed)):
        token = int(token)
        pending.append(token)
        all_tokens.append(token)
        if len(pending) % flush_every == 0:
            flush()
        if token in stop or i >= max_tokens - 1:
            break
    flush()
    seconds = time.perf_counter() - start
    return {"text": "".join(pieces), "tokens": all_tokens, "prompt_tokens": len(ids), "seconds": seconds,
            "tokens_per_sec": len(all_tokens) / seconds if seconds > 0 else 0.0}
