#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
    python tools/kernel_resources.py st-llm_amd/csrc/gemm_w4_bf16.hip [filter]
A hand-ordered K loop must never touch scratch: check after every edit of a hot loop (NOTES.md)."""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-x", "hip", "-c", src,
                    "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark:\s+(Function Name|SGPRs Spill|VGPRs Spill|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs):\s*(\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = v
        rows[cur] = {}
    elif cur:
        rows[cur][k.replace(' [bytes/lane]', '').replace(' [waves/SIMD]', '')] = v
dem = subprocess.run(["c++filt"] + list(rows), capture_output=True, text=True).stdout.splitlines()
for name, d in zip(dem, rows.values()):
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
    if flt and flt not in name:
        continue
    print(f"{name:70s} vgpr {d.get('VGPRs', '?'):>4} agpr {d.get('AGPRs', '?'):>4} sgpr {d.get('SGPRs', '?'):>4} scratch {d.get('ScratchSize', '?'):>5} "
          f"vgpr-spill {d.get('VGPRs Spill', '?'):>4} sgpr-spill {d.get('SGPRs Spill', '?'):>4}")
if r.returncode:
    print(r.stderr[-3000:])
    sys.exit(r.returncode)
