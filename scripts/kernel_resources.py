#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
    python scripts/kernel_resources.py spleeterrt_amd/csrc/srt_nn2.hip [-DSRT_TUNING]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + root + "/include", "-I" + root + "/spleeterrt_amd/csrc",
       "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+([^:]+): (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"kernel": re.sub(r"\(.*", "", name).replace("void ", "")}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    print("%-62s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s lds %6s spill v%s s%s" % (
        r["kernel"][:62], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
        r.get("LDS Size [bytes/block]"), r.get("VGPRs Spill"), r.get("SGPRs Spill")))
