"""Wall clock of the bnz-compatible CLI (banzai_amd/bnzhip): process start to exit, file read and .bz2 written on /tmp;
best of 3 after one warm-up run.  argv: out.json"""
import bz2, json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banzai_amd import corpus
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(root, "banzai_amd", "bnzhip")
out = {"note": "banzai_amd/bnzhip -k <file>: process start to exit, file read and .bz2 written on /tmp; best of 3 after one "
               "warm-up run; most of both times is process start + HIP runtime initialisation", "runs": {}}
for name, n in (("1MB", 1_000_000), ("100MB", 100_000_000)):
    data = corpus.workload(n)[0].tobytes()
    path = f"/tmp/bnz_wall_{name}.bin"
    open(path, "wb").write(data)
    best, rc = None, 0
    for it in range(4):
        if os.path.exists(path + ".bz2"):
            os.remove(path + ".bz2")
        t = time.perf_counter()
        rc = subprocess.call([exe, "-k", path])
        dt = time.perf_counter() - t
        if it:
            best = dt if best is None or dt < best else best
    ok = rc == 0 and bz2.decompress(open(path + ".bz2", "rb").read()) == data
    out["runs"][name] = {"bytes": n, "seconds": round(best, 4), "MB/s": round(n / best / 1e6, 1), "rc": rc, "libbz2_roundtrip": bool(ok)}
    os.remove(path); os.remove(path + ".bz2")
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cli_wallclock.json", "w"), indent=1)
print(json.dumps(out["runs"]))
