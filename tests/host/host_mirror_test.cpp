// Exercises the C++ mirror of the reference's host interface (software-renderer_amd/host/Renderer.hpp)
// the way App.swift:153-185 drives the reference: build a RenderPass, call renderer.render(pass).
// Checks the SURVEY.md §C.1 known answer and the z-test / painter's-order difference.
// Exit code 0 = pass.  Needs a HIP device (no CPU fallback).
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../software-renderer_amd/host/Renderer.hpp"

using namespace swr_host;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
    const long W = 256, H = 256;
    std::vector<Pixel> color(W * H, Pixel{9, 9, 9, 9});
    std::vector<float> depth(W * H, -1.0f);
    try {
        GpuRenderer gpuRenderer;            // App.swift:149
        Renderer renderer;                  // App.swift:148
        RenderPass pass{ColorImage(color.data(), W, H, W * 4), DepthImage(depth.data(), W, H, W * 4),
                        {Vertex(0.0f, 0.5f, 0.5f, 1.0f, 0.5f, 0.25f), Vertex(0.5f, -0.5f, 0.5f, 1.0f, 0.5f, 0.25f),
                         Vertex(-0.5f, -0.5f, 0.5f, 1.0f, 0.5f, 0.25f)},
                        {0, 1, 2}};
        pass.primitiveType = PrimitiveType::triangle;
        renderer.render(pass);              // as-written CPU semantics
        long covered = 0;
        for (long i = 0; i < W * H; i++) {
            if (color[i].a == 255) {
                covered++;
                CHECK(color[i].b == 63 && color[i].g == 127 && color[i].r == 255);
            } else {
                CHECK(color[i].b == 0 && color[i].g == 0 && color[i].r == 0 && color[i].a == 0);
            }
            CHECK(std::isinf(depth[i]) && depth[i] > 0);
        }
        CHECK(covered == 8193);
        CHECK(pass.colorBuffer.at(64, 192).a == 255 && pass.colorBuffer.at(65, 192).a == 0);   // flat-bottom quirk

        // two coplanar-in-xy triangles at different depth: painter's order vs z-test
        pass.vertices = {Vertex(0, .8f, .2f, 1, 0, 0), Vertex(.8f, -.8f, .2f, 1, 0, 0), Vertex(-.8f, -.8f, .2f, 1, 0, 0),
                         Vertex(0, .8f, .7f, 0, 0, 1), Vertex(.8f, -.8f, .7f, 0, 0, 1), Vertex(-.8f, -.8f, .7f, 0, 0, 1)};
        pass.indices = {0, 1, 2, 3, 4, 5};
        renderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).b == 255 && pass.colorBuffer.at(128, 128).r == 0);   // last drawn wins
        gpuRenderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).r == 255 && pass.colorBuffer.at(128, 128).b == 0);   // nearest wins
        CHECK(pass.depthBuffer.at(128, 128) == 0.2f);

        // transform path: matrix_float4x4(rows:) * identity, as App.swift:176-183
        const float rows[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 1, 1}};
        pass.transform = matrix_float4x4::fromRows(rows) * matrix_float4x4::identity();
        gpuRenderer.render(pass);           // w = z + 1: shrinks towards the centre
        CHECK(pass.colorBuffer.at(128, 128).a == 255);
        CHECK(pass.colorBuffer.at(128, 40).a == 0);

        // the Metal kernels' own rules: closed triangle, nearest-even UNORM (63.75 -> 64, 127.5 -> 128)
        pass.vertices = {Vertex(0.0f, 0.5f, 0.5f, 1.0f, 0.5f, 0.25f), Vertex(0.5f, -0.5f, 0.5f, 1.0f, 0.5f, 0.25f),
                         Vertex(-0.5f, -0.5f, 0.5f, 1.0f, 0.5f, 0.25f)};
        pass.indices = {0, 1, 2};
        pass.transform = matrix_float4x4::identity();
        gpuRenderer.metalRules = true;
        gpuRenderer.render(pass);
        gpuRenderer.metalRules = false;
        long mcov = 0;
        for (long i = 0; i < W * H; i++) mcov += color[i].a == 255;
        CHECK(mcov == 8192);
        CHECK(pass.colorBuffer.at(128, 128).b == 64 && pass.colorBuffer.at(128, 128).g == 128);
        CHECK(pass.depthBuffer.at(128, 128) == 0.5f);

        // extended fragment stage (not in the reference): per-pixel Phong, then a 1x1 texture on the base colour.
        // normal (0,0,-2), L = H = (0,0,-1): rgb = c * fl(0.1 + 0.5) + 0.25 -> (0.85, 0.55, 0.4) -> 216, 140, 102
        pass.attributes = {VertexAttributes(0, 0, -2, 0, 0), VertexAttributes(0, 0, -2, 1, 0), VertexAttributes(0, 0, -2, 0, 1)};
        pass.material.shader = Shader::phong;
        pass.material.shininessLog2 = 3;
        pass.material.ambient = 0.1f; pass.material.diffuse = 0.5f; pass.material.specular = 0.25f;
        gpuRenderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).r == 216 && pass.colorBuffer.at(128, 128).g == 140 &&
              pass.colorBuffer.at(128, 128).b == 102 && pass.colorBuffer.at(128, 128).a == 255);
        Pixel texel{0, 128, 255, 255};      // b,g,r,a: base = c * (1, 128/255, 0)
        pass.material.shader = Shader::texturedPhong;
        const Image<Pixel> texture(&texel, 1, 1, 4);
        pass.material.texture = &texture;
        gpuRenderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).r == 216 && pass.colorBuffer.at(128, 128).g == 102 &&
              pass.colorBuffer.at(128, 128).b == 63);
        pass.attributes.pop_back();         // one attribute per vertex, or the mirror throws
        bool threw = false;
        try { gpuRenderer.render(pass); } catch (const RenderError& e) { threw = e.code == SWR_ERR_BAD_ARG; }
        CHECK(threw);
        pass.attributes.clear();
        pass.material = Material{};         // back to Shaders.metal:116-121
        gpuRenderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).r == 255 && pass.colorBuffer.at(128, 128).g == 127);

        // the reference traps on a bad index (Renderer.swift:226); the mirror throws
        pass.indices = {0, 1, 99};
        threw = false;
        try { gpuRenderer.render(pass); } catch (const RenderError& e) { threw = e.code == SWR_ERR_INDEX_RANGE; }
        CHECK(threw);
        pass.primitiveType = PrimitiveType::line;   // stub in the reference (Renderer.swift:289-293): clears only
        pass.indices = {0, 1, 2};                   // verticesCount == 2 -> the assert at Renderer.swift:209
        threw = false;
        try { gpuRenderer.render(pass); } catch (const RenderError& e) { threw = e.code == SWR_ERR_INDEX_COUNT; }
        CHECK(threw);
        pass.indices = {0, 1};
        gpuRenderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).a == 0 && std::isinf(pass.depthBuffer.at(128, 128)));
        pass.primitiveType = PrimitiveType::vertices;   // Renderer.swift:295-302: one pixel per vertex reference
        pass.indices = {0, 1, 2};
        pass.transform = matrix_float4x4::identity();
        gpuRenderer.render(pass);
        long pts = 0;
        for (long i = 0; i < W * H; i++) pts += color[i].a == 255;
        CHECK(pts == 3);
        CHECK(pass.colorBuffer.at(128, 64).r == 255);   // vertex (0, .5) -> (128, 64)

        // The app's frame loop (App.swift:153-185): the same mesh every frame, a new transform.  staticScene keeps the mesh
        // resident (GpuRenderer.swift:32-33,41-67); editing the arrays WITHOUT telling the renderer keeps the old mesh,
        // staticScene = false (scene_id 0) or a new sceneVersion picks the edit up.
        pass.primitiveType = PrimitiveType::triangle;
        pass.vertices = {Vertex(0, .8f, .2f, 1, 0, 0), Vertex(.8f, -.8f, .2f, 1, 0, 0), Vertex(-.8f, -.8f, .2f, 1, 0, 0)};
        pass.indices = {0, 1, 2};
        gpuRenderer.staticScene = true;
        for (int frame = 0; frame < 3; frame++) {
            pass.transform = matrix_float4x4::identity();
            pass.transform.columns[3][0] = 0.1f * (float)frame;         // slide to the right
            gpuRenderer.render(pass);
            CHECK(pass.colorBuffer.at(128 + (int)(12.8f * (float)frame), 128).r == 255);
        }
        pass.transform = matrix_float4x4::identity();
        pass.vertices[0] = Vertex(0, .8f, .2f, 0, 1, 0);               // recolour a corner in place ...
        pass.vertices[1] = Vertex(.8f, -.8f, .2f, 0, 1, 0);
        pass.vertices[2] = Vertex(-.8f, -.8f, .2f, 0, 1, 0);
        gpuRenderer.render(pass);                                        // ... same sceneVersion: the resident (red) mesh is drawn
        CHECK(pass.colorBuffer.at(128, 128).r == 255 && pass.colorBuffer.at(128, 128).g == 0);
        gpuRenderer.sceneVersion++;                                      // tell the renderer: uploaded again
        gpuRenderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).g == 255 && pass.colorBuffer.at(128, 128).r == 0);
        pass.vertices[0] = Vertex(0, .8f, .2f, 0, 0, 1); pass.vertices[1] = Vertex(.8f, -.8f, .2f, 0, 0, 1);
        pass.vertices[2] = Vertex(-.8f, -.8f, .2f, 0, 0, 1);
        gpuRenderer.staticScene = false;                                 // scene_id 0: every call uploads (the default)
        gpuRenderer.render(pass);
        CHECK(pass.colorBuffer.at(128, 128).b == 255 && pass.colorBuffer.at(128, 128).g == 0);
    } catch (const RenderError& e) {
        std::printf("RenderError %d: %s\n", e.code, e.what());
        return 2;
    }
    std::printf(fails ? "host mirror: %d failures\n" : "host mirror: ok\n", fails);
    return fails ? 1 : 0;
}
