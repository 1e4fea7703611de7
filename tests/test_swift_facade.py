"""The Swift facade (software-renderer_amd/host/swift/) is source only — this image has no Swift toolchain — so what CAN be
checked is checked as text: the two replacement files declare the reference's entry points and data model with the
reference's own signatures (strings below: /root/reference/renderer/Renderer.swift and GpuRenderer.swift at the cited lines),
forward to the C-ABI calls include/swr.h declares, and INTEGRATION.md describes a two-file replacement without hand edits."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWIFT = os.path.join(ROOT, "software-renderer_amd", "host", "swift")


def squeeze(text):
    text = re.sub(r"//[^\n]*", "", text)                     # comments
    return re.sub(r"\s+", " ", text)


def read(name):
    with open(os.path.join(SWIFT, name)) as f:
        return squeeze(f.read())


# (declaration as the reference spells it, reference file:line)
RENDERER_DECLS = [
    ("typealias ColorImage = Image<Pixel>", "Renderer.swift:5"),
    ("typealias DepthImage = Image<Float>", "Renderer.swift:6"),
    ("class Image<Pixel> {", "Renderer.swift:8"),
    ("init(pointer: UnsafeMutablePointer<Pixel>, width: Int, height: Int, bytesPerRow: Int)", "Renderer.swift:9"),
    ("private(set) var pointer: UnsafeMutablePointer<Pixel>", "Renderer.swift:16"),
    ("let width: Int", "Renderer.swift:19"), ("let height: Int", "Renderer.swift:20"), ("let bytesPerRow: Int", "Renderer.swift:21"),
    ("subscript(x: Int, y: Int) -> Pixel", "Renderer.swift:23"),
    ("func contains(x: Int, y: Int) -> Bool", "Renderer.swift:39"),
    ("struct Pixel { var b: UInt8 var g: UInt8 var r: UInt8 var a: UInt8 }", "Renderer.swift:44-49"),
    ("struct Vertex {", "Renderer.swift:154"),
    ("let xyz: vector_float3", "Renderer.swift:156"), ("let color: vector_float3", "Renderer.swift:157"),
    ("enum PrimitiveType { case triangle case line case vertices", "Renderer.swift:174-177"),
    ("var verticesCount: Int", "Renderer.swift:179"),
    ("struct RenderPass { var colorBuffer: ColorImage var depthBuffer: DepthImage var vertices: [Vertex] var indices: [Int] "
     "var primitiveType: PrimitiveType = .triangle var transform: matrix_float4x4 = .init(diagonal: .one) }", "Renderer.swift:191-200"),
    ("final class Renderer {", "Renderer.swift:202"),
    ("func render(renderPass: RenderPass) {", "Renderer.swift:204"),
]
GPU_DECLS = [
    ("final class GpuRenderer {", "GpuRenderer.swift:12"),
    ("func render(renderPass: RenderPass) {", "GpuRenderer.swift:35"),
]


def test_renderer_swift_keeps_the_reference_declarations():
    src = read("Renderer.swift")
    for decl, where in RENDERER_DECLS:
        assert squeeze(decl) in src, f"{where}: `{decl}` not declared in host/swift/Renderer.swift"
    # the as-written CPU path has its z-test commented out (Renderer.swift:257-261): off by default
    assert "var depthTest = false" in src
    assert "renderOnHIP" not in src


def test_gpu_renderer_swift_keeps_the_reference_declarations():
    src = read("GpuRenderer.swift")
    for decl, where in GPU_DECLS:
        assert squeeze(decl) in src, f"{where}: `{decl}` not declared in host/swift/GpuRenderer.swift"
    assert "var depthTest = true" in src                      # the Metal path z-tests (Shaders.metal:158-165)


def test_both_facades_forward_to_the_c_abi():
    with open(os.path.join(ROOT, "include", "swr.h")) as f:
        header = f.read()
    for name in ("Renderer.swift", "GpuRenderer.swift"):
        src = read(name)
        assert "import CSwr" in src
        called = set(re.findall(r"\b(swr_[a-z_]+)\(", src))
        assert {"swr_context_create", "swr_context_destroy", "swr_render"} <= called, (name, called)
        for fn in called:          # functions, or the C structs Swift initialises with `swr_config(...)` / `swr_render_pass()`
            assert re.search(r"\b%s\s*\(|struct\s+%s\b" % (fn, fn), header), f"{name} uses {fn}, which include/swr.h does not declare"
        for field in re.findall(r"\bpass\.([a-z_]+)\s*=", src):
            assert re.search(r"\b%s\b" % field, header), f"{name} sets swr_render_pass.{field}, unknown to include/swr.h"
    with open(os.path.join(SWIFT, "module.modulemap")) as f:
        mm = f.read()
    assert "module CSwr" in mm and "swr.h" in mm and 'link "swr_hip"' in mm
    assert sorted(os.listdir(SWIFT)) == ["GpuRenderer.swift", "Renderer.swift", "module.modulemap"]


def test_integration_doc_describes_a_two_file_replacement():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    assert "host/swift/Renderer.swift" in doc and "host/swift/GpuRenderer.swift" in doc
    assert "renderOnHIP" not in doc and "Renderer+HIP" not in doc
    assert "libswr_hip_earlyz" not in doc                     # deleted in round 3
