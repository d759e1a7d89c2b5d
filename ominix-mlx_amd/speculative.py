"""Speculative decoding over two engine models -- host-side mirror of mlx-rs-core/src/speculative.rs (`SpeculativeGenerate`,
`SpeculativeToken`, :18-104, :184-316).

A small draft model proposes `num_draft_tokens` tokens with ordinary decode steps; the target model checks all of them in ONE batched
matrix-core pass (`Model.verify`, speculative.rs:132-161) and the longest agreeing prefix is accepted.  Two things the reference
module leaves open are implemented here (oracle/ref_speculative.py lists the lines): the caches are trimmed after a rejection
(`Model.trim`, the `KeyValueCache::trim` that speculative.rs:165-169 says is missing), and the target's own token at the first
disagreement is emitted after the accepted draft tokens.  At temperature 0 the emitted sequence is then the target model's own greedy
sequence, whatever the draft model proposes.

temperature != 0 (speculative.rs:104-109, :145-148, :277-281): every token -- the draft's proposals and the target's token at each
verified position -- is drawn categorical(logits / temperature), and a proposal is accepted when the two draws are EQUAL.  Every emitted
token is therefore the target's own draw at its position given the emitted prefix, so the output is a valid sample of the target model
(at a lower acceptance rate than rejection sampling would give; that is the reference's rule).  The reference draws all of them from
MLX's one global key sequence in program order: first token, then per round k draft draws and k + 1 target draws.  The two engine
models each keep a sequence on the device, so its two-word state is handed over at every switch (`Model.sampler_state`).
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

    def __init__(self, target: Model, draft: Model, num_draft_tokens: int, temperature: float, prompt, with_logprobs: bool = False,
                 seed: int = 0, record: bool = False):
        if temperature < 0.0 or temperature != temperature:
            raise ValueError("temperature must be >= 0")
        if num_draft_tokens < 1:
            raise ValueError("num_draft_tokens must be >= 1")
        self.target, self.draft, self.k = target, draft, int(num_draft_tokens)
        self.temperature = float(temperature)
        # both samplers start from mlx_rs::random::seed(seed); the draft's copy of the state is overwritten before its first draw
        target.set_sampler(self.temperature, seed)
        draft.set_sampler(self.temperature, seed)
        # record=True (tests): the draft proposes with k single steps and every round keeps the logits each draw was made from --
        # rounds[i] = {"draft_logits": [k, V], "drafts", "target_logits": [k + 1, V], "target_tokens", "accepted"}
        self.record = [] if record else None
        self.first_logits = None
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
            first = int(self.target.prefill(self.prompt))       # (sampled from the target's last position when temperature != 0, :222)
            lp = _log_softmax(self.target.last_logits()) if self.with_logprobs else None
            if self.record is not None:
                self.first_logits = self.target.last_logits()
            self.draft.prefill(self.prompt)                     # (its own draw, if any, is discarded: the reference's draft only forwards)
            self.draft.trim(0, first)                           # the draft continues from the TARGET's token
            self.last = first
            self.token_count = 1
            return SpeculativeToken(first, False, lp)
        k, last = self.k, self.last
        sampling = self.temperature != 0.0
        if sampling:
            self.draft.set_sampler_state(self.target.sampler_state())         # one key sequence: the draft draws next
        if self.record is None:
            drafts = [int(t) for t in self.draft.decode(k)]     # generate_draft_tokens, :111-127
        else:
            drafts, d_logits = [], []
            for _ in range(k):
                drafts.append(int(self.draft.decode(1)[0]))
                d_logits.append(self.draft.last_logits())
        if sampling:
            self.target.set_sampler_state(self.draft.sampler_state())         # ... then the target, k + 1 draws
        t_tokens = [int(t) for t in self.target.verify([last] + drafts)]      # verify_draft_tokens, :132-161
        accepted = 0
        while accepted < k and drafts[accepted] == t_tokens[accepted]:       # :277-292
            accepted += 1
        final = t_tokens[accepted]                              # the target's correction, or its bonus token when all were accepted
        lps = [_log_softmax(self.target.verify_logits(i)) for i in range(accepted + 1)] if self.with_logprobs else [None] * (accepted + 1)
        # both caches keep [last, drafts[:accepted]]; `final` is the next input of both models
        self.target.trim(k - accepted, final)
        if self.record is not None:
            self.record.append({"draft_logits": np.stack(d_logits), "drafts": list(drafts), "target_tokens": list(t_tokens), "accepted": accepted,
                                "target_logits": np.stack([self.target.verify_logits(i) for i in range(k + 1)])})
        if accepted == k:
            self.draft.decode(1)                                # ingest the last proposal (its output -- and its draw -- are not used)
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
