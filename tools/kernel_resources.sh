#!/bin/bash
# Register / scratch / LDS usage of every kernel in csrc/kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage):
#   bash tools/kernel_resources.sh [filter-regex]
# A non-zero ScratchSize on an MFMA kernel means spills: far more expensive than the instructions suggest.
cd /tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c /root/repo/handwritten-chinese-ocr-samples_amd/csrc/kernels.hip \
  -o /tmp/kernel_resources.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":",1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
import subprocess
for r in rows:
    name = subprocess.run(["/usr/bin/c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("hctr::", "").replace("(ConvArgs)", "")
    if flt and not flt.search(name): continue
    print("%-78s VGPR %-4s AGPR %-4s SGPR %-4s scratch %-5s occ %-2s LDS %s" % (name[:78], r.get("VGPRs"), r.get("AGPRs"),
          r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
' "$@"
