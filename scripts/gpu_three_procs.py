"""Three unserialized processes on ONE GPU (not the supported deployment; what a shared box does to the look-backs of small
batches): `bnzhip` compresses a 1 MB file N times alone, then three of them do so at once.  A look-back that loses its
predecessor to another process gives up after 20 ms of wall clock and the sort runs again pinned to one XCD -- no call may
take more than twice the slowest solo call + the give-up budget.  argv: [N] [out.json]"""
import json, os, subprocess, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from banzai_amd import corpus

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/three_procs.json"
exe = os.path.join(root, "banzai_amd", "bnzhip")
tmp = tempfile.mkdtemp()
src = os.path.join(tmp, "in.bin")
open(src, "wb").write(corpus.workload(1_000_000)[0].tobytes())
want = subprocess.run([exe, "-c", src], capture_output=True, check=True).stdout


def loop(tag, times, bad):
    for _ in range(N):
        t = time.perf_counter()
        p = subprocess.run([exe, "-c", src], capture_output=True)
        times.append(time.perf_counter() - t)
        if p.returncode != 0 or p.stdout != want:
            bad.append((tag, p.returncode, p.stderr[-200:].decode(errors="replace")))


solo, bad = [], []
loop("solo", solo, bad)
trio = [[], [], []]
th = [threading.Thread(target=loop, args=(f"p{k}", trio[k], bad)) for k in range(3)]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join()
wall3 = time.perf_counter() - t0
allt = [x for l in trio for x in l]
doc = {"calls_each": N, "solo_s": {"mean": sum(solo) / N, "max": max(solo)}, "three_at_once_s": {"mean": sum(allt) / len(allt), "max": max(allt), "wall": wall3},
       "failures": bad[:5], "ok": not bad and max(allt) <= 2 * max(solo) + 0.1}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc))
sys.exit(0 if doc["ok"] else 1)
