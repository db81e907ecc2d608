"""profiles/<tag>_msd_ablation.json: the bucket-first initial sort (the default for text) against the 8 radix passes it
replaces (BZH_INIT=lsd) on the bench workload -- both bench lines of ONE box, the rocprofv3 kernel tables of both runs (the
bucket-first one is <tag>_kernel_stats_bench_n1.csv), the plan's unit / level counts and chunk_finish's cycles per phase
(BZH_MSD_DBG=16).  argv: directory of scripts/collect_profiles.sh, tag"""
import csv, json, os, re, sys
d, tag = sys.argv[1], sys.argv[2]


def line(name):
    return json.loads(open(os.path.join(d, name)).read().strip().splitlines()[-1])


lsd, msd = line("msd_lsd_line.json"), line("msd_msd_line.json")
doc = {"workload": lsd["config"]["workload"],
       "eight_radix_passes": {k: lsd[k] for k in ("value", "ms_per_step", "stage_ms_per_step", "bwt_rounds", "A_over_n")},
       "bucket_first": {k: msd[k] for k in ("value", "ms_per_step", "stage_ms_per_step", "bwt_rounds", "A_over_n")},
       "eight_radix_passes_kernels_us_per_step": {k["kernel"]: k["us_per_step"] for k in lsd["roofline"]["kernels"]},
       "bucket_first_kernels_us_per_step": {k["kernel"]: k["us_per_step"] for k in msd["roofline"]["kernels"]}}
rows = list(csv.DictReader(open(os.path.join(d, f"{tag}_kernel_stats_bench_n1.csv"))))
want = ("bigram_hist", "bigram_plan", "bigram_scatter", "seg_count", "seg_plan", "seg_scatter", "chunk_finish", "rank_apply")
doc["rocprofv3_avg_us"] = {r["Name"].split("(")[0]: {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1)}
                           for r in rows if r["Name"].startswith(want)}
rows = list(csv.DictReader(open(os.path.join(d, f"{tag}_kernel_stats_lsd.csv"))))
want = ("void radix_scatter<8, 0>", "void radix_scatter<8, 2>", "void radix_scatter<8, 3>", "void refine_one<true>", "rank_apply", "byte_count")
doc["rocprofv3_avg_us_eight_radix_passes"] = {r["Name"].split("(")[0].replace("void ", ""): {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 1)}
                                              for r in rows if r["Name"].startswith(want)}
tr = open(os.path.join(d, "msd_trace.txt")).read()
m = re.findall(r"initial sort: (\d+) blocks bucket-first, (\d+) blocks 8-pass; (\d+) units; ([0-9.]+) % of the suffixes in oversized 2-byte buckets; oversized buckets per level (\d+) (\d+) (\d+) (\d+) (\d+) \(tiles (\d+) (\d+) (\d+) (\d+) (\d+)\)", tr)
if m:
    v = [int(x) for x in m[-1][:3]] + [int(x) for x in m[-1][4:]]
    doc["plan"] = {"blocks_bucket_first": v[0], "blocks_8_pass": v[1], "units": v[2],
                   "pct_of_suffixes_in_oversized_2_byte_buckets": float(m[-1][3]), "oversized_buckets_per_level": v[3:8],
                   "oversized_tiles_per_level": v[8:13]}
m = re.findall(r"chunk_finish, 16-cycle ticks over all workgroups: (.*)", tr)
if m:
    parts = re.findall(r"([a-z+ ]+?) (\d+)(?:,|$)", m[-1])
    tot = sum(int(b) for _, b in parts)
    doc["chunk_finish_phase_share"] = {a.strip(): round(int(b) / tot, 3) for a, b in parts}
    doc["chunk_finish_ticks_total_x16_cycles"] = tot
print(json.dumps(doc, indent=1))
