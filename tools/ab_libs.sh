#!/bin/bash
# A/B of several builds of the library (ablation builds: lib/ab_*.so) on cfg4: k_raster isolated, variants 0 / 3 / 4
for lib in "$@"; do
  echo "== $lib"
  python - <<PY
import os, subprocess, sys
ROOT = os.getcwd()
code = open('tools/ablate.py').read().split('CODE = r"""')[1].split('""" % ROOT')[0] % ROOT
for v in os.environ.get("VARIANTS", "0 3 4").split():
    r = subprocess.run([sys.executable, "-c", code], env={**os.environ, "SWR_LIBRARY": os.path.join(ROOT, "$lib"), "SWR_DEBUG_VARIANT": v}, capture_output=True, text=True)
    print(f"  variant {v}: k_raster us (depth-only, colour+depth) = {r.stdout.strip()} {r.stderr.strip()[-200:]}", flush=True)
PY
done
