"""Speculative decoding over two engine models -- host-side mirror of mlx-rs-core/src/speculative.rs (`SpeculativeGenerate`,
`SpeculativeToken`, :18-104, :184-316).

A small draft model proposes `num_draft_tokens` tokens with ordinary decode steps; the target model checks all of them in ONE batched
matrix-core pass (`Model.verify`, speculative.rs:132-161) and the longest agreeing prefix is accepted.  Two things the reference
module leaves open are implemented here (oracle/ref_speculative.py lists the lines): the caches are trimmed after a rejection
(`Model.trim`, the `KeyValueCache::trim` that speculative.rs:165-169 says is missing), and the target's own token at the first
disagreement is emitted after the accepted draft tokens.  Greedy only (temperature 0): the emitted sequence is then the target model's
own greedy sequence, whatever the draft model proposes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterator, List, Optional

import numpy as np

from .engine import Model


def _log_softmax(logits: np.ndarray) -> np.ndarray:
    x = logits.astype(np.float64)
    m = x.max()
    return (x - (m + np.log(np.exp(x - m).sum()))).astype(np.float32)


@dataclass
class SpeculativeToken:
    """speculative.rs:18-25.  `logprobs`: logits - logsumexp(logits) of the TARGET model at this position (:150-152), float32 [V];
    None unless the generator was created with with_logprobs=True (one [V] row per token crosses PCIe then)."""
    token: int
    from_draft: bool
    logprobs: Optional[np.ndarray] = None


class SpeculativeGenerate:
    """Iterator of SpeculativeToken (speculative.rs:184-316).  `target` / `draft`: engine.Model instances sharing a vocabulary, both
    reset; `prompt`: token ids."""

    def __init__(self, target: Model, draft: Model, num_draft_tokens: int, temperature: float, prompt, with_logprobs: bool = False):
        if temperature != 0.0:
            raise NotImplementedError("speculative decoding is greedy here (temperature 0): comparing two independently sampled "
                                      "tokens, as speculative.rs:106-109 + :277-281 would, is not a valid acceptance rule")
        if num_draft_tokens < 1:
            raise ValueError("num_draft_tokens must be >= 1")
        self.target, self.draft, self.k = target, draft, int(num_draft_tokens)
        self.prompt = np.ascontiguousarray(np.asarray(prompt, dtype=np.uint32).ravel())
        self.with_logprobs = with_logprobs
        self.pending: List[SpeculativeToken] = []
        self.last = None
        self.token_count = 0
        self.accepted_total = 0
        self.rounds = 0

    def __iter__(self) -> Iterator[SpeculativeToken]:
        return self

    def __next__(self) -> SpeculativeToken:
        if self.pending:
            self.token_count += 1
            return self.pending.pop(0)
        if self.last is None:                                   # SpeculativeState::Prefill, :202-246
            first = int(self.target.prefill(self.prompt))
            lp = _log_softmax(self.target.last_logits()) if self.with_logprobs else None
            self.draft.prefill(self.prompt)
            self.draft.trim(0, first)                           # the draft continues from the TARGET's token
            self.last = first
            self.token_count = 1
            return SpeculativeToken(first, False, lp)
        k, last = self.k, self.last
        drafts = [int(t) for t in self.draft.decode(k)]         # generate_draft_tokens, :111-127
        t_tokens = [int(t) for t in self.target.verify([last] + drafts)]      # verify_draft_tokens, :132-161
        accepted = 0
        while accepted < k and drafts[accepted] == t_tokens[accepted]:       # :277-292
            accepted += 1
        final = t_tokens[accepted]                              # the target's correction, or its bonus token when all were accepted
        lps = [_log_softmax(self.target.verify_logits(i)) for i in range(accepted + 1)] if self.with_logprobs else [None] * (accepted + 1)
        # both caches keep [last, drafts[:accepted]]; `final` is the next input of both models
        self.target.trim(k - accepted, final)
        if accepted == k:
            self.draft.decode(1)                                # ingest the last proposal (its output is not used)
            self.draft.trim(0, final)
        else:
            self.draft.trim(k - accepted - 1, final)
        toks = [SpeculativeToken(drafts[i], True, lps[i]) for i in range(accepted)] + [SpeculativeToken(final, False, lps[accepted])]
        self.last = final
        self.accepted_total += accepted
        self.rounds += 1
        self.pending = toks[1:]
        self.token_count += 1
        return toks[0]
