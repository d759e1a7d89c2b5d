"""bench.py's Paraformer leg on its own (mel frontend + transcribe_from_mel, 30 s of audio): python tools/paraformer_transcribe_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
print(json.dumps(bench.paraformer_secondary(omx, reps=8)))
