"""Per-kernel FETCH_SIZE / WRITE_SIZE sums of two rocprofv3 --pmc runs (scripts/collect_profiles.sh) -> JSON.
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
streaming reads (MI355X_MICROARCH.md, HBM section); narrower accesses are uncalibrated, ratios between kernels hold."""
import csv, glob, json, os, re, sys

out, tag = sys.argv[1], sys.argv[2]


def load(counter):
    acc = {}
    for f in glob.glob(os.path.join(out, f"pmc_{counter}", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            a = acc.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fe, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
kern = {}
for k in sorted(set(fe) | set(wr)):
    n = max(fe.get(k, [0])[0], wr.get(k, [0])[0])
    f, w = fe.get(k, [0, 0.0])[1], wr.get(k, [0, 0.0])[1]
    kern[k] = {"launches": n, "fetch_kb": f, "write_kb": w, "hbm_bytes_per_launch": (2 * f + w) * 1024 / max(1, n)}
rs = [v for k, v in kern.items() if k.startswith("radix_scatter")]
nrs = sum(v["launches"] for v in rs)
# passes of the hot path in the run = launches of a kernel every pass launches exactly once
steps_in_run = max((v["launches"] for k, v in kern.items() if k.startswith("plan_split")), default=0)
doc = {"steps_in_run": steps_in_run,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 1 --no-cpu "
               "--no-extra` (`steps_in_run` passes of the hot path: first touch, profiled, warm-up, timed); values in KB as reported; hbm_bytes = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reads half of wide streaming reads, MI355X_MICROARCH.md "
               "HBM section; 8-byte-per-lane and narrower accesses are uncalibrated)",
       "tag": tag,
       "radix_scatter_all": {"launches": nrs, "hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in rs) / max(1, nrs)},
       "kernels": kern}
print(json.dumps(doc, indent=1))
