import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
from oracle import ref_core as rc, ref_qwen3 as rq, synth

def run(name, cfg, n_prompt, max_context=512):
    w = rq.synth_weights(cfg)
    o = rq.Qwen3Oracle(cfg, w)
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    rt, rl = o.generate(prompt, 4, return_logits=True)
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                     intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, vocab_size=cfg.vocab_size,
                     rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, tie_word_embeddings=cfg.tie_word_embeddings,
                     rope_scaling=cfg.rope_scaling, max_context=max_context)
    m.synth_weights()
    first = m.prefill(prompt)
    l0 = m.last_logits()
    rest = m.decode(3)
    print(f"{name:28s} n_prompt={n_prompt:4d} err0={np.abs(l0-rl[0]).max():.4f} max|l|={np.abs(rl).max():.3f} "
          f"tok={[int(first)]+[int(t) for t in rest]} ref={[int(t) for t in rt]}", flush=True)

C = rq.Qwen3Config
run("base d64 g2", C(512, 2, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False), 32, 256)
run("base d64 g2", C(512, 2, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False), 128)
run("1 layer d64", C(512, 1, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False), 16)
run("d128 g2 h512", C(512, 2, 1536, 4, 2, 128, 2048, 1e-6, 1e6, False), 32)
run("d128 g4 h1024 untied", C(1024, 3, 3072, 8, 2, 128, 4096, 1e-6, 1e6, False), 32)
run("d128 g4 h1024 tied", C(1024, 3, 3072, 8, 2, 128, 4096, 1e-6, 1e6, True), 32)
run("d128 g4 h1024 i1536", C(1024, 1, 1536, 8, 2, 128, 4096, 1e-6, 1e6, False), 8)
run("d128 g1 h1024", C(1024, 1, 1024, 8, 8, 128, 4096, 1e-6, 1e6, False), 8)
run("d64 g4 h512", C(512, 1, 1024, 8, 2, 64, 2048, 1e-6, 1e6, False), 8)
run("d64 g1 h512 1tok", C(512, 1, 1024, 8, 8, 64, 2048, 1e-6, 1e6, False), 1)
