"""qwen3-mlx/examples/generate_qwen3.rs on the MI355X engine:  python tools/generate_qwen3.py <model_dir> [prompt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import
omx_import.load_package()
from ominix_mlx_amd import generate, loader

if len(sys.argv) < 2:
    sys.exit(f"Usage: {sys.argv[0]} <model_dir> [prompt]")
model_dir = sys.argv[1]
prompt = sys.argv[2] if len(sys.argv) > 2 else "Hello, I am a language model,"
tokenizer = generate.load_tokenizer(model_dir)
model = loader.load_model(model_dir)
print(f"Prompt: {prompt}\n---")
out = generate.generate_text(model, tokenizer, prompt, temperature=0.7, max_tokens=100, emit=lambda t: print(t, end="", flush=True))
print(f"\n---\nGenerated {len(out['tokens'])} tokens in {out['seconds']:.2f}s ({out['tokens_per_sec']:.1f} tok/s)")
