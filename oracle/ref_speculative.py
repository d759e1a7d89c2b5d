"""TEST INFRASTRUCTURE ONLY -- CPU restatement of mlx-rs-core/src/speculative.rs (SpeculativeGenerate) over two oracle models.

What the reference module does (file:line) and where it stops short of what its own comments describe:
  * :202-246  Prefill: the prompt through BOTH models, first token sampled from the target's last position.        (restated)
  * :111-127  generate_draft_tokens: `num_draft_tokens` sequential draft-model steps starting from the last token.  (restated)
  * :132-161  verify_draft_tokens: ONE target forward over [last token, draft 1..k]; a token sampled at every position. (restated)
  * :276-292  acceptance: the longest prefix of draft tokens equal to the target's tokens is accepted.              (restated)
  * :165-169  trim_cache is a stub ("full implementation would need proper cache trimming support in KeyValueCache"): the rejected
              tokens stay in both caches, so every later position is wrong.  Here the caches ARE trimmed (KVCache offset moved back)
              -- the target keeps [last, accepted drafts], the draft the same (ingesting the last draft token when all k were
              accepted, which the reference also omits).
  * :294-314  the target's own token at the first rejected position (or the bonus token when all were accepted) becomes `last_token`
              but is never yielded when at least one draft token was accepted, so the emitted sequence skips tokens.  Here it is
              yielded after the accepted draft tokens, as the module header describes ("verified by the larger target model").
With those two completions the greedy (temperature 0) output is EXACTLY the target model's own greedy sequence, whatever the draft
model proposes -- the size-independent property tests/ hold the GPU implementation to.  The reference has no test for this module
(:305-308 is an empty test module): parity is pinned on that invariant, not on vectors.
  * :104-109  sample(): temperature != 0 -> categorical(logits * (1 / temperature)) with NO key: each call takes the next key of MLX's
              global sequence (mlx-rs/src/random.rs:21-41).  Program order of the draws: the first token (:222), then per round the k
              draft draws (:118-121) and the k + 1 target draws (:145-148).  `speculative_generate(..., temp, seed)` restates exactly
              that with one `mlx_rng.RandomState(seed)`; acceptance stays token equality (:277-281), so every emitted token is the
              target's own draw at its position.  (The all-accepted case's extra draft forward draws nothing here: only its cache matters.)
"""
from __future__ import annotations

from typing import Iterator, List, Tuple

import numpy as np

from . import ref_core as rc


def _trim(caches: List, n: int) -> None:
    for c in caches:
        c.trim(n)


def speculative_generate(target, draft, prompt: np.ndarray, num_draft_tokens: int, max_tokens: int, temp: float = 0.0,
                         seed: int = 0) -> Iterator[Tuple[int, bool, np.ndarray]]:
    """Yields (token, from_draft, target logprobs [V]) -- SpeculativeToken, speculative.rs:18-25 -- for `max_tokens` tokens; temp == 0:
    greedy, otherwise every draw is categorical(logits / temp) with the next key of ONE sequence seeded with `seed`."""
    from . import mlx_rng
    state = mlx_rng.RandomState(seed)

    def draw(lg):      # speculative.rs:104-109 over rows [n, V]: one key per row, in row order
        if temp == 0.0:
            return rc.sample_greedy(lg)
        return np.array([int(rc.sample(row[None, :], temp, state.next())[0]) for row in np.asarray(lg)], np.uint32)

    t_cache, d_cache = [], []
    prompt = np.asarray(prompt)[None, :].astype(np.int64)
    logits = target.forward(prompt, t_cache)[:, -1, :]                     # :208-214, 222
    draft.forward(prompt, d_cache)                                        # :217-219
    last = int(draw(logits)[0])
    yield last, False, _logprobs(logits[0])
    emitted = 1
    while emitted < max_tokens:
        drafts, cur = [], last
        for _ in range(num_draft_tokens):                                  # :111-127
            lg = draft.forward(np.array([[cur]], np.int64), d_cache)[:, -1, :]
            cur = int(draw(lg)[0])
            drafts.append(cur)
        seq = np.array([[last] + drafts], np.int64)                        # :257-270
        lg = target.forward(seq, t_cache)[0]                               # :137  [k + 1, V]
        t_tokens = [int(t) for t in draw(lg)]
        accepted = 0
        while accepted < num_draft_tokens and drafts[accepted] == t_tokens[accepted]:   # :277-292
            accepted += 1
        # caches keep [last, drafts[:accepted]]
        _trim(t_cache, num_draft_tokens - accepted)
        if accepted == num_draft_tokens:
            draft.forward(np.array([[drafts[-1]]], np.int64), d_cache)    # the draft never saw its own last proposal
        else:
            _trim(d_cache, num_draft_tokens - accepted - 1)
        for i in range(accepted):
            if emitted < max_tokens:
                yield drafts[i], True, _logprobs(lg[i])
                emitted += 1
        last = t_tokens[accepted]                                          # :295-299
        if emitted < max_tokens:
            yield last, False, _logprobs(lg[accepted])
            emitted += 1


def _logprobs(logits: np.ndarray) -> np.ndarray:
    """:150-152  logits - logsumexp(logits) in float32."""
    x = np.asarray(logits, np.float64)
    m = x.max()
    return (x - (m + np.log(np.exp(x - m).sum()))).astype(np.float32)
