// Renderer.swift — drop-in replacement for renderer/Renderer.swift of zhvrnkov/software-renderer.
//
// Keeps what the app (App.swift:80-101, :153-185) compiles against — `ColorImage` / `DepthImage`, `Image<T>`, `Pixel`, `Vertex`,
// `PrimitiveType`, `RenderPass` (Renderer.swift:5-49, :154-200) and
//
//     final class Renderer { func render(renderPass: RenderPass) }          // Renderer.swift:202-204
//
// — and replaces what is behind the call: instead of the scalar scanline loop (Renderer.swift:204-287, :467-494) the pass is
// handed to libswr_hip.so through the C-ABI (include/swr.h, `import CSwr`) and comes back with the pixels in the caller's
// images, synchronously, exactly as the CPU loop leaves them: painter's order (the z-test is commented out at
// Renderer.swift:257-261, so `depthTest` defaults to false and the depth image is +inf on return), truncating 8-bit
// quantiser, inclusive integer spans.  Together with GpuRenderer.swift (same directory) this is a two-file replacement: no
// hand edit in the app.  The 2-D toy primitives of the original file (rect, circle, 2-D line / triangle, :289-465) are not on
// the triangle path and are not reproduced.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Swift toolchain).  tests/test_swift_facade.py checks the declarations below
// against the reference's signatures as text.
import Foundation
import CSwr
#if canImport(simd)
import simd
#else
// Linux: the two simd names the data model needs, with the same memory layout (columns of four floats; a float3 is 16 bytes)
typealias vector_float3 = SIMD3<Float>
struct matrix_float4x4 {
    var columns: (SIMD4<Float>, SIMD4<Float>, SIMD4<Float>, SIMD4<Float>)
    init(diagonal d: SIMD4<Float>) {
        columns = (SIMD4(d.x, 0, 0, 0), SIMD4(0, d.y, 0, 0), SIMD4(0, 0, d.z, 0), SIMD4(0, 0, 0, d.w))
    }
}
#endif

typealias ColorImage = Image<Pixel>
typealias DepthImage = Image<Float>

/// A caller-owned 2-D image over a raw pointer; element (x, y) lives at pointer[y * width + x] (App.swift:351-360).
class Image<Pixel> {
    init(pointer: UnsafeMutablePointer<Pixel>, width: Int, height: Int, bytesPerRow: Int) {
        self.pointer = pointer
        self.width = width
        self.height = height
        self.bytesPerRow = bytesPerRow
    }

    private(set) var pointer: UnsafeMutablePointer<Pixel>
    let width: Int
    let height: Int
    let bytesPerRow: Int

    func contains(x: Int, y: Int) -> Bool { x >= 0 && x < width && y >= 0 && y < height }

    subscript(x: Int, y: Int) -> Pixel {
        get {
            precondition(contains(x: x, y: y))
            return pointer[y * width + x]
        }
        set { if contains(x: x, y: y) { pointer[y * width + x] = newValue } }     // out-of-bounds stores are dropped
    }
}

/// Memory order b, g, r, a (bgra8): 4 bytes.
struct Pixel {
    var b: UInt8
    var g: UInt8
    var r: UInt8
    var a: UInt8
}

/// 32 bytes: position (NDC: x, y in -1...1, z in 0...1) and colour, each a 16-byte float3 — the layout of `swr_vertex`.
struct Vertex {
    let xyz: vector_float3
    let color: vector_float3
}

enum PrimitiveType {
    case triangle
    case line
    case vertices

    var verticesCount: Int { self == .line ? 2 : 3 }
}

struct RenderPass {
    var colorBuffer: ColorImage
    var depthBuffer: DepthImage

    var vertices: [Vertex]
    var indices: [Int]
    var primitiveType: PrimitiveType = .triangle

    var transform: matrix_float4x4 = .init(diagonal: .one)
}

final class Renderer {
    /// false = Renderer.swift as written (its z-test is commented out, :257-261); true restores those five lines
    /// (strict `<`, first drawn wins ties).
    var depthTest = false
    /// The app draws one mesh with a new transform every frame (App.swift:153-185): true keeps the mesh of the first
    /// render(renderPass:) resident on the GPU until `sceneVersion` changes.
    var staticScene = false
    var sceneVersion: UInt64 = 1

    private var ctx: OpaquePointer?

    init() {
        var cfg = swr_config(device: -1, device_count: 0, wait_budget_ms: 0, reserved: 0)
        let rc = swr_context_create(&cfg, &ctx)
        precondition(rc == SWR_OK, String(cString: swr_last_error(nil)))
    }

    deinit { swr_context_destroy(ctx) }

    func render(renderPass: RenderPass) {
        precondition(renderPass.indices.count.isMultiple(of: renderPass.primitiveType.verticesCount))   // assert at :209
        var pass = swr_render_pass()
        pass.color = UnsafeMutableRawPointer(renderPass.colorBuffer.pointer)
        pass.depth = renderPass.depthBuffer.pointer
        pass.width = Int64(renderPass.colorBuffer.width)
        pass.height = Int64(renderPass.colorBuffer.height)
        pass.color_bytes_per_row = Int64(renderPass.colorBuffer.bytesPerRow)
        pass.depth_bytes_per_row = Int64(renderPass.depthBuffer.bytesPerRow)
        switch renderPass.primitiveType {
        case .triangle: pass.primitive_type = Int32(SWR_PRIMITIVE_TRIANGLE)
        case .line: pass.primitive_type = Int32(SWR_PRIMITIVE_LINE)
        case .vertices: pass.primitive_type = Int32(SWR_PRIMITIVE_VERTICES)
        }
        pass.flags = depthTest ? UInt32(SWR_FLAG_DEPTH_TEST) : 0
        pass.scene_id = staticScene ? sceneVersion : 0
        withUnsafeBytes(of: renderPass.transform) { src in          // 4 columns of 4 floats, column-major, 64 bytes
            withUnsafeMutableBytes(of: &pass.transform) { $0.copyMemory(from: src) }
        }
        let rc: Int32 = renderPass.vertices.withUnsafeBytes { v in
            renderPass.indices.withUnsafeBufferPointer { i in
                pass.vertices = v.baseAddress?.assumingMemoryBound(to: swr_vertex.self)
                pass.vertex_count = Int64(renderPass.vertices.count)
                pass.index_count = Int64(i.count)
                guard let base = i.baseAddress else { return swr_render(ctx, &pass) }
                return base.withMemoryRebound(to: Int64.self, capacity: i.count) { idx in      // Swift Int = int64
                    pass.indices = idx
                    return swr_render(ctx, &pass)            // synchronous: pixels are in the caller's images on return
                }
            }
        }
        precondition(rc == SWR_OK, String(cString: swr_last_error(ctx)))     // the original traps (fatalError / assert)
    }
}
