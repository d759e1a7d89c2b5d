import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import omx_import
omx = omx_import.load_package()
from ominix_mlx_amd import engine
from oracle import ref_core as rc, ref_qwen3 as rq, synth
C = rq.Qwen3Config
def run(name, cfg, n_prompt, max_context):
    w = rq.synth_weights(cfg); o = rq.Qwen3Oracle(cfg, w)
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    # oracle logits after each prefix length
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                     intermediate_size=cfg.intermediate_size, num_attention_heads=cfg.num_attention_heads,
                     num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, vocab_size=cfg.vocab_size,
                     tie_word_embeddings=cfg.tie_word_embeddings, max_context=max_context)
    m.synth_weights()
    errs = []
    for n in sorted(set([1, 2, 31, 32, 33, 63, 64, 65, 66, 96, 127, 128, 129, n_prompt])):
        if n > n_prompt: continue
        m.reset()
        m.prefill(prompt[:n])
        l0 = m.last_logits()
        caches = []
        rl = o.forward(prompt[:n][None, :].astype(np.int64), caches)[0, -1]
        errs.append((n, float(np.abs(l0 - rl).max())))
    print(name, "cap", max_context, "graph" if not os.environ.get("OMX_NO_GRAPH") else "eager", errs, flush=True)
run("d64g2", C(512, 2, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False), 130, 256)
run("d64g2", C(512, 2, 1536, 8, 4, 64, 2048, 1e-6, 1e6, False), 130, 512)
run("d128g4", C(1024, 3, 3072, 8, 2, 128, 4096, 1e-6, 1e6, True), 130, 512)
run("d128g4-1layer", C(1024, 1, 3072, 8, 2, 128, 4096, 1e-6, 1e6, True), 130, 512)
