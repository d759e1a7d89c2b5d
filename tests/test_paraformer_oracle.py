"""CPU: properties of oracle/ref_paraformer.py that the reference itself checks
(examples/validate_correctness.rs part 2 / compare_cif_batch.rs: batched CIF == per-item CIF)."""
import numpy as np

from oracle import ref_paraformer as rp


def test_cif_batch_equals_single_and_conserves_mass():
    g = np.random.default_rng(0)
    B, T, H = 3, 120, 16
    hidden = g.standard_normal((B, T, H)).astype(np.float32)
    alphas = (g.random((B, T)) * 0.4).astype(np.float32)
    frames, counts = rp.cif_fire(hidden, alphas)
    for b in range(B):
        f1, c1 = rp.cif_fire(hidden[b:b + 1], alphas[b:b + 1])
        assert c1[0] == counts[b]
        np.testing.assert_array_equal(f1[0], frames[b, :counts[b]])
    # every fired frame integrates exactly one unit of alpha: count == floor(sum alpha) (+1 if the tail > 0.45)
    for b in range(B):
        s = float(alphas[b].astype(np.float64).sum())
        assert counts[b] in (int(np.floor(s)), int(np.floor(s)) + 1)


def test_cif_known_small_case():
    hidden = np.array([[[1.0], [2.0], [4.0]]], np.float32)
    alphas = np.array([[0.6, 0.6, 0.5]], np.float32)
    frames, counts = rp.cif_fire(hidden, alphas)
    # t0: 0.6 ; t1: fires with completion 0.4 -> 0.6*1 + 0.4*2 = 1.4, remainder 0.2*2 ; t2: 0.2+0.5=0.7 < 1 ; tail 0.7 > 0.45
    assert counts.tolist() == [2]
    np.testing.assert_allclose(frames[0, :, 0], [1.4, 0.2 * 2 + 0.5 * 4], rtol=1e-6)


def test_fsmn_is_centered_depthwise_conv():
    v = np.zeros((9, 2)); v[4, 0] = 1.0; v[4, 1] = 2.0
    w = np.zeros((2, 3)); w[0] = [1, 2, 3]; w[1] = [0, 1, 0]
    out = rp.fsmn(v, w)
    np.testing.assert_array_equal(out[:, 0], [0, 0, 0, 3, 2, 1, 0, 0, 0])      # correlation, not convolution (as Conv1d)
    np.testing.assert_array_equal(out[:, 1], v[:, 1])


def test_position_encoding_and_model_wiring():
    """paraformer.rs:418-439 (positions start at 1, [sin | cos]) and the reference's own shape tests (:1592-1610):
    PE [100, 512]; a tiny model with the reference's checkpoint keys runs end to end."""
    from oracle import ref_paraformer as rp
    pe = rp.position_encoding(100, 512)
    assert pe.shape == (100, 512)
    assert abs(pe[0, 0] - np.sin(1.0)) < 1e-6 and abs(pe[0, 256] - np.cos(1.0)) < 1e-6
    assert abs(pe[4, 255] - np.sin(5.0 * 1e-4)) < 1e-6                      # last timescale = 1/10000
    cfg = dict(n_mels=80, lfr_m=7, encoder_dim=128, encoder_layers=2, encoder_heads=1, encoder_ffn_dim=256, decoder_dim=128,
               decoder_layers=1, decoder_heads=1, decoder_ffn_dim=256, vocab_size=50, sanm_kernel_size=11, cif_l_order=1,
               cif_r_order=1, cif_threshold=1.0, cif_tail_threshold=0.45)
    w = rp.synth_checkpoint(cfg, 1)
    assert w["predictor.conv.weight"].shape == (128, 128, 3) and w["encoder.encoders0.0.self_attn.fsmn_block.weight"].shape == (128, 1, 11)
    tok, logits, enc, alphas, emb = rp.transcribe_from_mel(np.random.default_rng(2).standard_normal((30, 560)) * 0.5, w, cfg)
    assert enc.shape == (30, 128) and logits.shape == (len(tok), 50) and emb.shape == (len(tok), 128)
    assert abs(len(tok) - alphas.sum()) < 1.0                               # CIF emits ~ sum(alphas) tokens
