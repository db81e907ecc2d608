"""Build-container only: runs the independent Python model (pymodel.py) over the inputs of stream_cases.py and
writes model_streams.json -- SHA-256 of every input and of the stream the model produces, plus what the case
exercises (tables, rescaling, block cuts).  libbz2 must decode every stream back to its input.
    python tests/golden/gen_streams.py
"""
import bz2
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden import pymodel, stream_cases  # noqa: E402

out = {"note": "model-derived (tests/golden/pymodel.py, an independent Python restatement of reference lib/*.rs); "
               "not produced by the Rust binary (no toolchain in the build image)", "cases": {}}
for name, make in stream_cases.CASES.items():
    level, data = make()
    info = {}
    stream = pymodel.encode(data, level, info)
    assert bz2.decompress(stream) == data, name
    out["cases"][name] = {
        "level": level, "input_len": len(data), "input_sha256": hashlib.sha256(data).hexdigest(),
        "stream_len": len(stream), "stream_sha256": hashlib.sha256(stream).hexdigest(),
        "blocks_consumed_rle": info["blocks"], "tables": info.get("tables"), "max_code_len": info.get("maxlen"),
        "rescaled_to": info.get("rescaled", 1),
    }
    print(name, out["cases"][name])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "model_streams.json"), "w"), indent=1)
