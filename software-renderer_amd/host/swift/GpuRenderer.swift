// GpuRenderer.swift — drop-in replacement body for renderer/GpuRenderer.swift:12-147.
// Same class name, same `render(renderPass:)` signature (GpuRenderer.swift:35), same RenderPass /
// Image / Vertex / Pixel types (declared in Renderer.swift of this directory, the replacement of the app's
// Renderer.swift; with the app's own file kept they are the ones at Renderer.swift:5-200).  Where the
// original builds MTLBuffers, encodes vertex_pass / roi_pass and one rasterizer_pass dispatch per
// triangle and blocks twice in scheduleAndWait, this forwards the pass to libswr_hip.so.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Swift toolchain; the app also imports Apple-only
// simd/Metal).  On a Linux host with Swift: swiftc -I <this dir> -L <lib dir> -lswr_hip ...
import CSwr

final class GpuRenderer {
    private var ctx: OpaquePointer?
    /// The Metal path z-tests (Shaders.metal:158-165); false reproduces Renderer.swift as written.
    var depthTest = true
    /// true: the Metal kernels' own rules (SWR_FLAG_METAL_RULES) instead of the CPU renderer's pixel rules.
    var metalRules = false
    /// Extended fragment stage (not in the original app): per-vertex normals / uvs parallel to
    /// renderPass.vertices, a material and an optional texture.  All nil = Shaders.metal:116-121.
    var attributes: [swr_vertex_attr]? = nil
    var material: swr_material? = nil
    var texture: Image<Pixel>? = nil
    /// The app draws the same mesh every display frame with a new transform (App.swift:153-185); the original keeps its
    /// MTLBuffers across calls (GpuRenderer.swift:32-33,41-67).  staticScene = true: the mesh of the first
    /// render(renderPass:) stays resident on the GPU until sceneVersion changes (or the counts differ) — the pass then
    /// costs one resident frame plus the gather instead of a 120 MB upload.  false: upload on every call.
    var staticScene = false
    var sceneVersion: UInt64 = 1

    /// deviceCount > 1: ONE renderer drives that many GPUs — the framebuffer is cut into tile-row bands, every band is
    /// copied straight into its rows of the caller's image (swr_config.device_count; include/swr.h).
    init(deviceCount: UInt32 = 0) {
        var cfg = swr_config(device: deviceCount > 1 ? 0 : -1, device_count: deviceCount, wait_budget_ms: 0, reserved: 0)
        let rc = swr_context_create(&cfg, &ctx)
        precondition(rc == SWR_OK, String(cString: swr_last_error(nil)))   // original: try! (GpuRenderer.swift:20-31)
    }

    deinit { swr_context_destroy(ctx) }

    func render(renderPass: RenderPass) {
        var pass = swr_render_pass()
        pass.color = UnsafeMutableRawPointer(renderPass.colorBuffer.pointer)
        pass.depth = renderPass.depthBuffer.pointer
        pass.width = Int64(renderPass.colorBuffer.width)
        pass.height = Int64(renderPass.colorBuffer.height)
        pass.color_bytes_per_row = Int64(renderPass.colorBuffer.bytesPerRow)
        pass.depth_bytes_per_row = Int64(renderPass.depthBuffer.bytesPerRow)
        pass.primitive_type = renderPass.primitiveType == .triangle ? 0 : (renderPass.primitiveType == .line ? 1 : 2)
        pass.flags = metalRules ? UInt32(SWR_FLAG_METAL_RULES) : (depthTest ? UInt32(SWR_FLAG_DEPTH_TEST) : 0)
        pass.scene_id = staticScene ? sceneVersion : 0
        withUnsafeBytes(of: renderPass.transform) { src in          // matrix_float4x4 = 4 float4 columns
            withUnsafeMutableBytes(of: &pass.transform) { $0.copyMemory(from: src) }
        }
        // Vertex is two SIMD3<Float> = 32 bytes, the layout of swr_vertex (Renderer.swift:154-157)
        renderPass.vertices.withUnsafeBytes { v in
            renderPass.indices.withUnsafeBufferPointer { i in
                pass.vertices = v.baseAddress?.assumingMemoryBound(to: swr_vertex.self)
                pass.vertex_count = Int64(renderPass.vertices.count)
                i.baseAddress!.withMemoryRebound(to: Int64.self, capacity: i.count) { pass.indices = $0 }
                pass.index_count = Int64(i.count)
                var mat = material ?? swr_material()
                let attrs = attributes ?? []
                let rc: Int32 = attrs.withUnsafeBufferPointer { a in
                    withUnsafePointer(to: &mat) { m in
                        if material != nil {
                            precondition(attrs.count == renderPass.vertices.count)
                            pass.attributes = a.baseAddress
                            pass.material = m
                            if let t = texture {
                                pass.texture = UnsafeRawPointer(t.pointer)
                                pass.tex_width = Int32(t.width); pass.tex_height = Int32(t.height)
                            }
                        }
                        return swr_render(ctx, &pass)      // pointers are only used during the call
                    }
                }
                precondition(rc == SWR_OK, String(cString: swr_last_error(ctx)))
            }
        }
    }
}
