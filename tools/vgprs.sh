#!/bin/bash
# Register use of every raster kernel for a set of -D flags: tools/vgprs.sh [-DSWR_...]   (run from anywhere; compiles swr_kernels.hip only)
cd "$(dirname "$0")/../software-renderer_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -w "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/vgprs_$$.o csrc/swr_kernels.hip 2>&1 | python3 -c '
import re, sys
name = None
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m: name = m.group(1); row = {}
    for k, pat in (("VGPRs", r"\sVGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                   ("waves", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, l)
        if m and name: row[k] = int(m.group(1))
    if name and "LDS Size" in l and ("k_raster" in name or "k_bin" in name):
        import subprocess
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("swr::", "").replace("(swr::RasterArgs)", "")
        print("%-50s VGPRs %3d  spill %2d  scratch %3d  waves/SIMD %d  static LDS %6d" % (d, row.get("VGPRs", -1), row.get("spill", 0), row.get("scratch", 0), row.get("waves", 0), row.get("lds", 0)))
'
rm -f /tmp/vgprs_$$.o
