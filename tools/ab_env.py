#!/usr/bin/env python3
"""Alternating A/B of ENVIRONMENT settings and swr_debug_set hooks on one box (same library):
    python tools/ab_env.py [--scene cfg4] [--reps 3] "k32=0" "insort=0" "SWR_PIPELINE=0 binmode=1" ""
(lower-case key=value: swr_debug_set — order, cull, binmode, oneshot, k32, insort; upper-case: environment)
Per setting: pipelined ms/frame (300 untimed frames), k_raster alone (pipelining off, events around the kernel)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
src = open(os.path.join(ROOT, "tools", "ab.py")).read()
CHILD = src[src.index("CHILD = r'''") + len("CHILD = r'''"):src.index("''' % ROOT")] % ROOT
CHILD = CHILD.replace("    ctx.scene_upload(", "    import os\n    for kv in os.environ.get('SWR_AB_HOOKS', '').split():\n        k, v = kv.split('='); ctx.debug_set({'order': 1, 'cull': 2, 'binmode': 3, 'oneshot': 4, 'k32': 5, 'insort': 6}[k], int(v))\n    ctx.scene_upload(", 1)
args = sys.argv[1:]
scene, reps = "cfg4", 3
while args and args[0].startswith("--"):
    if args[0] == "--scene": scene = args[1]
    if args[0] == "--reps": reps = int(args[1])
    args = args[2:]
res = {a: [] for a in args}
for r in range(reps):
    for a in args:
        env = dict(os.environ)
        hooks = []
        for kv in a.split():
            k, v = kv.split("=", 1)
            if k.islower(): hooks.append(kv)
            else: env[k] = v
        env["SWR_AB_HOOKS"] = " ".join(hooks)
        p = subprocess.run([sys.executable, "-c", CHILD, scene], env=env, capture_output=True, text=True)
        try:
            res[a].append(json.loads(p.stdout.strip().splitlines()[-1]))
        except Exception:
            print(a, "FAILED", p.stderr[-300:]); continue
for a in args:
    v = res[a]
    if not v: continue
    print(f"{scene:6s} {a or '(default)':28s} pipelined ms/frame {' '.join('%.4f' % x['ms'] for x in v)}  (min {min(x['ms'] for x in v):.4f})   "
          f"k_raster alone us {' '.join('%.1f' % x['raster_us'] for x in v)}  (min {min(x['raster_us'] for x in v):.1f})", flush=True)
