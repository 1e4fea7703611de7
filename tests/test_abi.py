"""CPU-only checks of the boundary: the C-ABI library builds for gfx950, loads without a GPU,
exports every symbol include/swr.h declares, the host-only arithmetic works, and compute entry
points FAIL LOUDLY when there is no HIP device (no CPU fallback inside the product)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "swr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(swr_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_all_exported(swr):
    swr.build()
    lib = ctypes.CDLL(swr.library_path())
    syms = declared_symbols()
    assert len(syms) >= 33
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/swr.h but not exported"
    assert sorted(swr.binding.ABI_SYMBOLS) == syms


def test_library_is_a_gfx950_code_object(swr):
    swr.build()
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          f"--input={swr.library_path()}"], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        assert "gfx950" in out.stdout
    else:   # fall back to a byte search of the fat binary
        assert b"gfx950" in open(swr.library_path(), "rb").read()


def test_product_does_not_link_the_oracle(swr):
    swr.build()
    out = subprocess.run(["ldd", swr.library_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out
    blob = open(swr.library_path(), "rb").read()
    assert b"swro_render" not in blob and b"libswr_oracle" not in blob


def test_struct_layouts_match_the_swift_types(swr):
    """Vertex = 2 x SIMD3<Float> padded to 16 B (Renderer.swift:154-157); RenderPass fields."""
    B = swr.binding
    assert ctypes.sizeof(B.RenderPass) == 8 * 10 + 4 + 4 + 64 + 3 * 8 + 2 * 4 + 8 == 192
    assert B.RenderPass.transform.offset == 88
    assert B.RenderPass.attributes.offset == 152          # extended fragment stage (ABI 2)
    assert B.RenderPass.scene_id.offset == 184            # scene identity (ABI 4)
    assert ctypes.sizeof(B.Material) == 56
    assert ctypes.sizeof(B.Config) == 16 and B.Config.device_count.offset == 4 and B.Config.wait_budget_ms.offset == 8   # swr_config (ABI 4)
    assert ctypes.sizeof(B.RenderTimes) == 28
    assert ctypes.sizeof(B.Timings) == 5 * 4 + 4 + 3 * 8


def test_band_rows_partition(swr):
    tw, th = swr.tile_shape()
    assert th > 0 and tw > 0
    for H in (1, 31, 32, 33, 1080, 2160, 4320):
        for parts in (1, 2, 3, 4, 8):
            edges = [swr.band_rows(H, parts, k) for k in range(parts)]
            assert edges[0][0] == 0 and edges[-1][1] == H
            for (a0, a1), (b0, b1) in zip(edges, edges[1:]):
                assert a1 == b0 and a0 <= a1
            for a0, a1 in edges:
                assert a0 % th == 0 or a0 == H
    with pytest.raises(swr.SwrError):
        swr.band_rows(100, 0, 0)
    with pytest.raises(swr.SwrError):
        swr.band_rows(100, 2, 2)
    sizes = [b - a for a, b in (swr.band_rows(2160, 8, k) for k in range(8))]
    assert max(sizes) - min(sizes) <= th


def test_no_device_fails_loudly_or_device_works(swr):
    """Without a GPU the context cannot be created (SWR_ERR_HIP); with one it can."""
    try:
        ctx = swr.Context()
    except swr.SwrError as e:
        assert e.code == -4 and "no CPU fallback" in str(e)
    else:
        ctx.close()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "software-renderer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".swift")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "swr_oracle" not in text and "swro_" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_host_mirror_compiles_against_the_c_abi(swr):
    """software-renderer_amd/host/Renderer.hpp (C++ mirror of Renderer / GpuRenderer / RenderPass)
    builds with plain g++ against include/swr.h + the shared library — no HIP headers needed."""
    swr.build()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "software-renderer_amd"), "-s", "lib/host_mirror_test"])
    exe = os.path.join(ROOT, "software-renderer_amd", "lib", "host_mirror_test")
    assert os.access(exe, os.X_OK)
    hdr = open(os.path.join(ROOT, "software-renderer_amd", "host", "Renderer.hpp")).read()
    for name in ("class Renderer", "class GpuRenderer", "struct RenderPass", "struct Vertex", "struct Pixel",
                 "enum class PrimitiveType", "class Image", "void render(const RenderPass& renderPass)"):
        assert name in hdr


def test_abi_version_matches_the_header_and_the_entry_point(swr):
    """include/swr.h, the built library and __graft_entry__.build() must agree on SWR_ABI_VERSION."""
    import re
    hdr = open(os.path.join(ROOT, "include", "swr.h")).read()
    want = int(re.search(r"#define\s+SWR_ABI_VERSION\s+(\d+)", hdr).group(1))
    swr.build()
    assert swr.load_library().swr_abi_version() == want
    entry = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert f"swr_abi_version() == {want}" in entry
