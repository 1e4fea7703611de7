#!/usr/bin/env python3
"""Copy one tools/round_refresh.sh result set into profiles/ and rewrite profiles/traffic.json / profiles/valu.json from it.
    python tools/profiles_from_refresh.py <refresh prefix> <profiles prefix> [round dir]      e.g.  g e r03"""
import ast, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r04"
out = os.path.join(ROOT, "profiles", rnd)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", "refresh", src + "_*")):
    shutil.copy(f, os.path.join(out, dst + os.path.basename(f)[len(src):]))
# the stamp is the commit that last changed the kernels, and only a committed build may be stamped: a measurement of
# uncommitted kernel sources would carry the hash of different code
dirty = subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "software-renderer_amd/csrc", "software-renderer_amd/Makefile"], text=True).strip()
if dirty:
    sys.exit("profiles_from_refresh: uncommitted changes under software-renderer_amd/csrc — commit the kernels, measure again, then stamp:\n" + dirty)
commit = subprocess.check_output(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "software-renderer_amd/csrc", "software-renderer_amd/Makefile"], text=True).strip()
raw = json.load(open(os.path.join(out, f"{dst}_traffic_raw_KB.json")))
kr = next(k for k in raw if "k_raster_depth" in k or "k_raster<true, 0, false, false" in k)
t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
sys.path.insert(0, ROOT)
import bench
digest = bench.csrc_digest()
t.update({"commit": commit, "csrc_digest": digest, "all_kernels_raw_KB": raw, "source": t.get("source", "").rsplit("; ", 1)[0] + f"; profiles/{rnd}/{dst}_traffic_raw_KB.json",
          "k_raster_fetch_bytes_raw": raw[kr]["FETCH_SIZE"] * 1024, "k_raster_write_bytes": raw[kr]["WRITE_SIZE"] * 1024,
          "k_raster_bytes_per_launch": (raw[kr]["FETCH_SIZE"] + raw[kr]["WRITE_SIZE"]) * 1024})
json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
v = json.load(open(os.path.join(ROOT, "profiles", "valu.json")))
for line in open(os.path.join(out, f"{dst}_sq_counters.txt")):
    if ("k_raster_depth" in line or "k_raster<true, 0, false, false" in line) and "SQ_INSTS_VALU" in line:
        d = ast.literal_eval(line[line.index("{"):])
        v.update({"commit": commit, "csrc_digest": digest, "source": v.get("source", "").rsplit("; ", 1)[0] + f"; profiles/{rnd}/{dst}_sq_counters.txt", "k_raster_valu_wave_insts_per_launch": d["SQ_INSTS_VALU"],
                  "k_raster_salu_insts_per_launch": d["SQ_INSTS_SALU"], "k_raster_lds_insts_per_launch": d["SQ_INSTS_LDS"]})
json.dump(v, open(os.path.join(ROOT, "profiles", "valu.json"), "w"), indent=1)
print("profiles/%s/%s_* <- refresh %s_*, commit %s: k_raster %.2f M VALU, %.1f MB HBM" % (
    rnd, dst, src, commit, v["k_raster_valu_wave_insts_per_launch"] / 1e6, t["k_raster_bytes_per_launch"] / 1e6))
