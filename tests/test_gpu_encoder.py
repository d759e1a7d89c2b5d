"""GPU parity of the FLUX.2-klein text encoder ("Qwen3-4B text encoder, hidden-state taps ->
7680-dim"): `omx_qwen3_encode` against oracle `Qwen3Oracle.encode` (flux-klein-mlx/src/qwen3_encoder.rs:141-224,
403-455).  Same stack, weights and tolerance rule as the decoder tests: bf16 hidden states after L layers,
|d| <= 2^-7 * max|h| * sqrt(L) (fp32 summation order differs; every op output is rounded to bf16 once).
Padding rows (attention_mask == 0) are compared too: the mask is the reference's additive -1e9, not a skip."""
import numpy as np
import pytest

from oracle import ref_qwen3 as rq
from oracle import synth
from test_gpu_qwen3 import _engine

pytestmark = pytest.mark.gpu

# Qwen3-4B proportions scaled down: hidden 1024 (2560), 6 layers (36), taps at 1/4, 1/2, 3/4 of the depth
CFG = rq.Qwen3Config(1024, 6, 3072, 8, 2, 128, 4096, 1e-6, 1e6, True)
TAPS = (1, 3, 4)


def _check(got, want, n_layers):
    bound = 2.0 ** -7 * np.abs(want).max() * np.sqrt(n_layers)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= bound


@pytest.mark.parametrize("n_tokens,n_pad", [(77, 0), (128, 37), (512, 200)])
def test_encode_matches_oracle(omx, n_tokens, n_pad):
    oracle = rq.Qwen3Oracle(CFG, rq.synth_weights(CFG))
    ids = synth.prompt_ids(n_tokens, CFG.vocab_size)
    am = None
    if n_pad:
        am = np.ones(n_tokens, np.uint8)
        am[n_tokens - n_pad:] = 0
        ids = ids.copy()
        ids[n_tokens - n_pad:] = 7            # the pad token id repeated, as a tokenizer pads
    want = oracle.encode(ids, am, TAPS)
    m = _engine(omx, CFG, max_context=512)
    got = m.encode(ids, am, TAPS).numpy()
    assert got.shape == (n_tokens, len(TAPS) * CFG.hidden_size)
    _check(got, want, max(TAPS) + 1)
    # real-token rows do not depend on what follows them (causal + right padding)
    if n_pad:
        # (the causal-only launch takes the mask-free softmax path, the padded one the additive-mask path: same
        # values to rounding, not the same bits)
        short = m.encode(ids[:n_tokens - n_pad], None, TAPS).numpy()
        _check(short, got[:n_tokens - n_pad], max(TAPS) + 1)


def test_encode_without_mask_is_causal_and_single_tap_is_a_prefix(omx):
    oracle = rq.Qwen3Oracle(CFG, rq.synth_weights(CFG))
    ids = synth.prompt_ids(96, CFG.vocab_size)
    m = _engine(omx, CFG, max_context=256)
    all_taps = m.encode(ids, None, TAPS).numpy()
    one = m.encode(ids, None, (TAPS[0],)).numpy()
    np.testing.assert_array_equal(one, all_taps[:, :CFG.hidden_size])
    _check(all_taps, oracle.encode(ids, None, TAPS), max(TAPS) + 1)
    with pytest.raises(omx.OmxError):
        m.encode(ids, None, (3, 1))            # taps must ascend
    with pytest.raises(omx.OmxError):
        m.encode(ids, None, (CFG.num_hidden_layers,))
