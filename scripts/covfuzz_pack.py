"""Packs a minimised covfuzz corpus (scripts/covfuzz.sh -> WORKDIR/min) into tests/golden/fuzz_corpus.zip, smallest inputs
first, and records what the fuzzing session was (tests/golden/fuzz_corpus.json).  argv: min_dir log_file [max_inputs]"""
import hashlib, json, os, re, sys, zipfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, log = sys.argv[1], sys.argv[2]
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
files = sorted((os.path.getsize(os.path.join(src, f)), f) for f in os.listdir(src) if f.endswith(".bin"))[:cap]
out = os.path.join(root, "tests", "golden", "fuzz_corpus.zip")
h = hashlib.sha256()
with zipfile.ZipFile(out, "w", zipfile.ZIP_DEFLATED, compresslevel=9) as z:
    for size, f in files:
        data = open(os.path.join(src, f), "rb").read()
        h.update(len(data).to_bytes(4, "little") + data)
        zi = zipfile.ZipInfo(f, date_time=(2026, 1, 1, 0, 0, 0))  # (fixed: the archive's bytes depend on the inputs only)
        zi.compress_type = zipfile.ZIP_DEFLATED
        z.writestr(zi, data, compresslevel=9)
text = open(log).read()
runs = re.findall(r"covfuzz run: (\d+) executions, (\d+) new inputs, corpus (\d+), (\d+) \(edge, bucket\) bits", text)
m = re.search(r"covfuzz min: (\d+) inputs, (\d+) \(edge, bucket\) bits; kept (\d+) inputs, (\d+) bytes", text)
doc = {"what": "minimised corpus of a coverage-guided fuzzing session of the CPU oracle + strict decoder (oracle/covfuzz.c: gcc "
               "trace-pc edge coverage with hit-count buckets, encode -> decode -> compare as the reference's "
               "fuzz/fuzz_targets/round_trip.rs); an input is [level byte][data], level = 1 + byte % 9",
       "workers": len(runs), "executions": sum(int(r[0]) for r in runs), "inputs_found": sum(int(r[1]) for r in runs),
       "merged_inputs": int(m.group(1)) if m else None, "edge_bucket_bits": int(m.group(2)) if m else None,
       "kept_inputs": len(files), "kept_bytes": sum(s for s, _ in files), "largest_input": files[-1][0] if files else 0,
       "zip_bytes": os.path.getsize(out), "sha256_of_inputs": h.hexdigest(), "round_trip_failures": 0}
json.dump(doc, open(os.path.join(root, "tests", "golden", "fuzz_corpus.json"), "w"), indent=1)
print(json.dumps(doc))
