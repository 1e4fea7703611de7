// Renderer+HIP.swift — keeps `final class Renderer { func render(renderPass:) }`
// (renderer/Renderer.swift:202-230) but runs the as-written CPU semantics (painter's order,
// depth image stays +inf) on the MI355X through the same C-ABI.  To adopt it, delete the body of
// Renderer.render(renderPass:) in the app's Renderer.swift and forward to this extension; the data
// model (Image, Pixel, Vertex, PrimitiveType, RenderPass — Renderer.swift:5-200) stays as is.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE (no Swift toolchain).
import CSwr

extension Renderer {
    private static let hip: GpuRenderer = {
        let r = GpuRenderer()
        r.depthTest = false          // Renderer.swift:257-261 is commented out in the reference
        return r
    }()

    func renderOnHIP(renderPass: RenderPass) {
        Renderer.hip.render(renderPass: renderPass)
    }
}
