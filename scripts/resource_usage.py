#!/usr/bin/env python3
"""Compiler's own resource report for the kernels of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage):
python scripts/resource_usage.py banzai_amd/csrc/bwt.hip [name-filter].  tests/test_abi.py fails on any ScratchSize != 0."""
import os, re, subprocess, sys, tempfile

FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-DBZH_BUILD", "-Rpass-analysis=kernel-resource-usage"]


def report(src):
    with tempfile.TemporaryDirectory() as td:
        p = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", src, "-o", os.path.join(td, "x.o")], capture_output=True, text=True)
    if p.returncode:
        raise RuntimeError(p.stderr[-2000:])
    out, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?): (.*?) \[-Rpass", line) or re.search(r"remark: +(.*?): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
            out.append(cur)
        elif cur is not None:
            cur[k] = v
    return out


if __name__ == "__main__":
    rows = report(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for r in rows:
        if flt in r["name"]:
            print(f"{r['name'][:60]:60s} VGPR {r.get('VGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>3s} SGPR spill {r.get('SGPRs Spill','?'):>3s} "
                  f"scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} LDS {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}")
