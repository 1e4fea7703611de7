/*
 * swr_oracle.c — CPU restatement of renderer/Renderer.swift's triangle path.
 * TEST INFRASTRUCTURE (checker + reported CPU baseline).  PARITY UNPINNED — see swr_oracle.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction, IEEE binary32 ops,
 * every intermediate rounded to float).  All "file:line" notes refer to
 * /root/reference/renderer/Renderer.swift unless another file is named.
 *
 * Deviation (documented in DESIGN.md §2.4): a triangle with a non-finite screen coordinate or
 * a screen coordinate whose magnitude is >= 2^30 is skipped.  The reference would trap there
 * (SIMD2<Int>(SIMD2<Float>) of a NaN / out-of-range value, :251,:271).
 *
 * det == 0 (vertices collinear after truncation) is NOT skipped: nothing traps in the reference.
 * T() (:95-100) divides by zero -> +-inf / NaN entries, the weights and the interpolated colour
 * become +-inf / NaN, simd_clamp (:119-122) = min(max(v,0),1) with fmax/fmin NaN handling maps
 * NaN -> 0, +inf -> 1, -inf -> 0, and UInt8() gets a finite value.  So in painter's mode the
 * degenerate span is written (mostly as (0,0,0,255)); with the z-test restored a NaN depth fails
 * '<' (:258) and only a -inf depth can pass.  What Apple's closed simd `inverse` returns for a
 * singular matrix is unknowable here (PARITY UNPINNED); this file uses adjugate / determinant in
 * IEEE arithmetic like everywhere else.
 */
#include "swr_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

#define COORD_LIMIT 1073741824.0f /* 2^30 */

typedef struct {
    /* Vertex after apply(transform:) and convertedToScreen (floats), in index order a,b,c */
    float sx[3], sy[3], sz[3];
    float col[3][3];
    float nrm[3][3], uv[3][2];   /* extended varyings (swro_shading), index order a,b,c */
    /* simd_long2(a.xyz.xy) etc. (:251): truncated integer vertices in index order */
    int64_t ix[3], iy[3];
    /* sorted-by-float-y list, truncated (:271) */
    int64_t sxs[3], sys_[3];
} tri_t;

typedef struct {
    uint8_t* color;
    float* depth;
    int64_t W, H;
    int64_t row_begin, row_end;
    uint32_t flags;
    swro_stats st;
} frame_t;

/* extended fragment stage of the current swro_render*_shaded call on this thread (NULL: the reference's) */
static __thread const swro_shading* tls_shading = NULL;

/* one texel channel triple as floats in [0,1]: Pixel is b,g,r,a (Renderer.swift:44-49) */
static void texel(const swro_shading* sh, int64_t x, int64_t y, float rgb[3]) {
    const int64_t tw = sh->tex_w, th = sh->tex_h;
    x = ((x % tw) + tw) % tw;                       /* repeat */
    y = ((y % th) + th) % th;
    const uint8_t* p = sh->texture + (size_t)(y * tw + x) * 4;
    rgb[0] = (float)p[2] / 255.0f;
    rgb[1] = (float)p[1] / 255.0f;
    rgb[2] = (float)p[0] / 255.0f;
}

/* see swr_oracle.h */
void swro_fragment(const swro_shading* sh, const float color[3], const float normal[3], const float uv[2],
                   float out[4]) {
    const swro_material* m = &sh->material;
    if (m->shader == 0) {                           /* Shaders.metal:116-121 */
        out[0] = color[0]; out[1] = color[1]; out[2] = color[2]; out[3] = 1.0f;
        return;
    }
    float len2 = normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2];
    float N[3] = {0.0f, 0.0f, 0.0f};
    if (len2 > 0.0f) {
        float len = sqrtf(len2);
        N[0] = normal[0] / len; N[1] = normal[1] / len; N[2] = normal[2] / len;
    }
    float ndl = N[0] * m->light_dir[0] + N[1] * m->light_dir[1] + N[2] * m->light_dir[2];
    ndl = fmaxf(ndl, 0.0f);
    float ndh = N[0] * m->half_dir[0] + N[1] * m->half_dir[1] + N[2] * m->half_dir[2];
    ndh = fmaxf(ndh, 0.0f);
    float s = ndh;
    for (int i = 0; i < m->shininess_log2; i++) s = s * s;
    float base[3] = {color[0], color[1], color[2]};
    if (m->shader == 2) {
        const float fu = uv[0] - floorf(uv[0]), fv = uv[1] - floorf(uv[1]);
        const float x = fu * (float)sh->tex_w - 0.5f, y = fv * (float)sh->tex_h - 0.5f;
        const float x0f = floorf(x), y0f = floorf(y);
        const float ax = x - x0f, ay = y - y0f;
        const int64_t x0 = (int64_t)x0f, y0 = (int64_t)y0f;
        float t00[3], t10[3], t01[3], t11[3];
        texel(sh, x0, y0, t00); texel(sh, x0 + 1, y0, t10);
        texel(sh, x0, y0 + 1, t01); texel(sh, x0 + 1, y0 + 1, t11);
        for (int k = 0; k < 3; k++) {
            float top = t00[k] + (t10[k] - t00[k]) * ax;
            float bot = t01[k] + (t11[k] - t01[k]) * ax;
            float t = top + (bot - top) * ay;
            base[k] = base[k] * t;
        }
    }
    const float lit = m->ambient + m->diffuse * ndl;
    const float spec = m->specular * s;
    for (int k = 0; k < 3; k++) out[k] = base[k] * lit + spec;
    out[3] = 1.0f;
}

/* Pixel.floats (:117-124): UInt8(simd_clamp(v, 0, 1) * 255), truncation toward zero.
 * simd_clamp = min(max(v,lo),hi) with fmax/fmin NaN handling (NaN -> lo). */
uint8_t swro_quantise(float v) {
    float c = fminf(fmaxf(v, 0.0f), 1.0f);
    float s = c * 255.0f;
    return (uint8_t)s;
}

/* Renderer.interpolate(values:t:) (:467-494). */
int64_t swro_interpolate(const int64_t* p, int n, int64_t t) {
    int base = 0;
    if (n == 3) {                       /* :469-475 */
        if (t >= p[2 * 2 + 1]) base = 2;
        else if (t >= p[1 * 2 + 1]) base = 1;
    }
    int next = base + 1;                /* :476 */
    int64_t sx = p[base * 2], sy = p[base * 2 + 1];
    if (next >= n) return sx;           /* :478-480 */
    int64_t ex = p[next * 2], ey = p[next * 2 + 1];
    int64_t diff = ex - sx;             /* :484 */
    int64_t dy = ey - sy;               /* :485 */
    if (dy == 0) return sx;             /* :486-488 */
    int64_t nt = t - sy;                /* :490 */
    return sx + diff * nt / dy;         /* :492  (C '/' truncates toward zero like Swift Int '/') */
}

/* setPixel (:245-269).  noinline + per-call T() keeps the reference's per-pixel cost
 * (the 2x2 inverse is recomputed for every pixel, :251-252 -> :88-100). */
__attribute__((noinline))
static void set_pixel(frame_t* f, const tri_t* t, int64_t x, int64_t y,
                      const float* tinv_hoisted) {
    /* :246-250 bounds check of both images (same size); band restriction for sharding */
    if (x < 0 || x >= f->W || y < 0 || y >= f->H) return;
    if (y < f->row_begin || y >= f->row_end) return;
    f->st.fragments++;

    float t00, t01, t10, t11;
    /* Triangle.ws(xy:) (:88-93) with T() (:95-100) */
    float cfx = (float)t->ix[2] + 0.5f, cfy = (float)t->iy[2] + 0.5f;   /* :89 */
    if (tinv_hoisted) {
        t00 = tinv_hoisted[0]; t01 = tinv_hoisted[1]; t10 = tinv_hoisted[2]; t11 = tinv_hoisted[3];
    } else {
        float afx = (float)t->ix[0] + 0.5f, afy = (float)t->iy[0] + 0.5f; /* :96 */
        float bfx = (float)t->ix[1] + 0.5f, bfy = (float)t->iy[1] + 0.5f; /* :97 */
        float m00 = afx - cfx, m10 = afy - cfy;   /* column 0 = af - cf (:99) */
        float m01 = bfx - cfx, m11 = bfy - cfy;   /* column 1 = bf - cf */
        float det = m00 * m11 - m01 * m10;
        if (f->flags & SWRO_INV_RCP) {
            float r = 1.0f / det;
            t00 = m11 * r; t01 = -m01 * r; t10 = -m10 * r; t11 = m00 * r;
        } else {                                  /* adjugate / determinant (Shaders.metal:15-31) */
            t00 = m11 / det; t01 = -m01 / det; t10 = -m10 / det; t11 = m00 / det;
        }
    }
    float px = (float)x + 0.5f, py = (float)y + 0.5f;   /* :252 */
    float dx = px - cfx, dy = py - cfy;                 /* :91 (xy - cf) */
    float w0 = t00 * dx + t01 * dy;                     /* :91 matrix*vector: col0*v.x + col1*v.y */
    float w1 = t10 * dx + t11 * dy;
    float w2 = 1.0f - w0 - w1;                          /* :92 */

    size_t at = (size_t)(y * f->W + x);                 /* App.swift:351-360 */
    if (f->flags & SWRO_DEPTH_TEST) {
        float d = t->sz[0] * w0 + t->sz[1] * w1 + t->sz[2] * w2;  /* :257 */
        if (!(d < f->depth[at])) return;                          /* :258-260 strict, NaN fails */
        f->depth[at] = d;                                         /* :261 */
    }
    f->st.fragments_written++;
    if (f->flags & SWRO_NO_COLOR) return;
    float c[3];
    for (int k = 0; k < 3; k++)                                   /* :266 */
        c[k] = t->col[0][k] * w0 + t->col[1][k] * w1 + t->col[2][k] * w2;
    float o[4] = {c[0], c[1], c[2], 1.0f};                        /* fragment_shader: float4(color, 1) */
    if (tls_shading) {
        float n[3], uv[2];
        for (int k = 0; k < 3; k++) n[k] = t->nrm[0][k] * w0 + t->nrm[1][k] * w1 + t->nrm[2][k] * w2;
        for (int k = 0; k < 2; k++) uv[k] = t->uv[0][k] * w0 + t->uv[1][k] * w1 + t->uv[2][k] * w2;
        swro_fragment(tls_shading, c, n, uv, o);
    }
    uint8_t* p = f->color + at * 4;                               /* Pixel(float3:) :126-128 */
    p[0] = swro_quantise(o[2]);  /* b */
    p[1] = swro_quantise(o[1]);  /* g */
    p[2] = swro_quantise(o[0]);  /* r */
    p[3] = swro_quantise(o[3]);  /* a */
}

/* draw(triangle:colorBuffer:depthBuffer:) (:238-287); input already transformed. */
static void draw_triangle(frame_t* f, tri_t* t) {
    /* :271 sorted { $0.xyz.y < $1.xyz.y } — stable 3-element insertion sort on FLOAT y */
    int ord[3] = {0, 1, 2};
    for (int i = 1; i < 3; i++)
        for (int j = i; j > 0 && t->sy[ord[j]] < t->sy[ord[j - 1]]; j--) {
            int tmp = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = tmp;
        }
    int64_t left[6], right[4];
    for (int k = 0; k < 3; k++) { left[2 * k] = t->ix[ord[k]]; left[2 * k + 1] = t->iy[ord[k]]; }
    right[0] = left[0]; right[1] = left[1]; right[2] = left[4]; right[3] = left[5];  /* :272-273 */

    float tinv[4];
    const float* hoisted = NULL;
    if (f->flags & SWRO_TINV_PER_TRIANGLE) {
        float cfx = (float)t->ix[2] + 0.5f, cfy = (float)t->iy[2] + 0.5f;
        float m00 = ((float)t->ix[0] + 0.5f) - cfx, m10 = ((float)t->iy[0] + 0.5f) - cfy;
        float m01 = ((float)t->ix[1] + 0.5f) - cfx, m11 = ((float)t->iy[1] + 0.5f) - cfy;
        float det = m00 * m11 - m01 * m10;
        if (f->flags & SWRO_INV_RCP) {
            float r = 1.0f / det;
            tinv[0] = m11 * r; tinv[1] = -m01 * r; tinv[2] = -m10 * r; tinv[3] = m00 * r;
        } else {
            tinv[0] = m11 / det; tinv[1] = -m01 / det; tinv[2] = -m10 / det; tinv[3] = m00 / det;
        }
        hoisted = tinv;
    }

    int64_t y0 = left[1], y1 = left[5];          /* :275 sorted.first!.y ... sorted.last!.y */
    int unclamped = (f->flags & SWRO_UNCLAMPED) != 0;
    if (!unclamped) {                             /* off-screen rows are no-ops (:246-250) */
        y0 = y0 < f->row_begin ? f->row_begin : y0;
        y1 = y1 > f->row_end - 1 ? f->row_end - 1 : y1;
    }
    for (int64_t y = y0; y <= y1; y++) {
        int64_t lx = swro_interpolate(left, 3, y);    /* :276 */
        int64_t rx = swro_interpolate(right, 2, y);   /* :277 */
        if (lx > rx) { int64_t tmp = lx; lx = rx; rx = tmp; }   /* :278-280 */
        if (!unclamped) { lx = lx < 0 ? 0 : lx; rx = rx > f->W - 1 ? f->W - 1 : rx; }
        for (int64_t x = lx; x <= rx; x++) set_pixel(f, t, x, y, hoisted);   /* :281-283 */
    }
}

/* Vertex.apply(transform:) (Renderer.swift:159-163): xyzw = M * (x,y,z,1), accumulated column by column.
 * Default: one rounding per multiply and per add (what -ffp-contract=off Swift / C gives).
 * SWRO_FMA_TRANSFORM: the OTHER plausible lowering of Apple's closed simd_mul(float4x4, float4) on arm64 — fmul for
 * the first column, then one fused multiply-add per further column (fmla) — used only to MEASURE how far that
 * undetermined choice can move the image (tests/test_oracle.py::test_fma_transform_sensitivity_bound); the GPU path
 * and every parity test use the default. */
static void apply_transform(const float M[16], float x, float y, float z, uint32_t flags, float r[4]) {
    for (int c = 0; c < 4; c++) {
        float a = M[0 + c] * x;
        if (flags & SWRO_FMA_TRANSFORM) {
            a = fmaf(M[4 + c], y, a);
            a = fmaf(M[8 + c], z, a);
            a = fmaf(M[12 + c], 1.0f, a);
        } else {
            a = a + M[4 + c] * y;
            a = a + M[8 + c] * z;
            a = a + M[12 + c] * 1.0f;
        }
        r[c] = a;
    }
}

/* The per-index vertex stage alone (Renderer.swift:159-171): screen x, y (fractional, before the truncation of :251)
 * and NDC z of every vertex — for tests that compare transform variants vertex by vertex. */
void swro_project(const swro_vertex* vertices, int64_t vertex_count, const float M[16], int64_t W, int64_t H,
                  uint32_t flags, float* sx, float* sy, float* sz) {
    const float fw = (float)W, fh = (float)H;
    for (int64_t i = 0; i < vertex_count; i++) {
        float r[4];
        apply_transform(M, vertices[i].xyz[0], vertices[i].xyz[1], vertices[i].xyz[2], flags, r);
        const float nx = r[0] / r[3], ny = r[1] / r[3], nz = r[2] / r[3];
        const float u = nx * 0.5f + 0.5f, w = ny * -0.5f + 0.5f;
        sx[i] = u * fw; sy[i] = w * fh; sz[i] = nz;
    }
}

int swro_render(uint8_t* color, float* depth, int64_t W, int64_t H,
                const swro_vertex* vertices, int64_t vertex_count,
                const int64_t* indices, int64_t index_count,
                const float M[16], uint32_t flags,
                int64_t row_begin, int64_t row_end, swro_stats* stats) {
    if (!depth || W <= 0 || H <= 0 || !M) return -1;
    if (!(flags & SWRO_NO_COLOR) && !color) return -1;
    if (index_count < 0 || vertex_count < 0) return -1;
    if (index_count > 0 && (!indices || !vertices)) return -1;
    if (index_count % 3 != 0) return -2;                          /* :209 */
    if (row_begin < 0 || row_end > H || row_begin > row_end) return -1;
    for (int64_t i = 0; i < index_count; i++)
        if (indices[i] < 0 || indices[i] >= vertex_count) return -3;

    frame_t f;
    memset(&f, 0, sizeof f);
    f.color = color; f.depth = depth; f.W = W; f.H = H;
    f.row_begin = row_begin; f.row_end = row_end; f.flags = flags;

    /* clear (:205-206, :232-236) — only this band's rows */
    if (!(flags & SWRO_NO_COLOR))
        memset(color + (size_t)row_begin * (size_t)W * 4, 0, (size_t)(row_end - row_begin) * (size_t)W * 4);
    for (int64_t i = row_begin * W; i < row_end * W; i++) depth[i] = INFINITY;

    float fw = (float)W, fh = (float)H;
    int64_t nprim = index_count / 3;                              /* :221 */
    for (int64_t p = 0; p < nprim; p++) {                         /* :222 in index order */
        tri_t t;
        int ok = 1;
        for (int k = 0; k < 3; k++) {
            const swro_vertex* v = &vertices[indices[3 * p + k]]; /* :223-227 */
            /* Vertex.apply(transform:) (:159-163): xyzw = M * (x,y,z,1) accumulated column by
             * column (col0*x, + col1*y, + col2*z, + col3*1); xyz / w */
            float x = v->xyz[0], y = v->xyz[1], z = v->xyz[2];
            float r[4];
            apply_transform(M, x, y, z, flags, r);
            float nx = r[0] / r[3], ny = r[1] / r[3], nz = r[2] / r[3];
            /* convertedToScreen (:165-171): uv = xy*(0.5,-0.5)+0.5 ; xy = uv*(W,H) (the
             * .rounded() at :168 applies to the (W,H) constant: a no-op) */
            float u = nx * 0.5f + 0.5f;
            float w = ny * -0.5f + 0.5f;
            t.sx[k] = u * fw;
            t.sy[k] = w * fh;
            t.sz[k] = nz;
            t.col[k][0] = v->color[0]; t.col[k][1] = v->color[1]; t.col[k][2] = v->color[2];
            if (tls_shading) {
                const swro_vertex_attr* av = &tls_shading->attrs[indices[3 * p + k]];
                t.nrm[k][0] = av->normal[0]; t.nrm[k][1] = av->normal[1]; t.nrm[k][2] = av->normal[2];
                t.uv[k][0] = av->uv[0]; t.uv[k][1] = av->uv[1];
            }
            if (!(fabsf(t.sx[k]) < COORD_LIMIT) || !(fabsf(t.sy[k]) < COORD_LIMIT)) { ok = 0; }
            else { t.ix[k] = (int64_t)t.sx[k]; t.iy[k] = (int64_t)t.sy[k]; }   /* :251 truncation */
        }
        /* det == 0 is drawn like any other triangle: see the note at the top of this file */
        if (!ok) { f.st.triangles_skipped++; continue; }
        f.st.triangles_drawn++;
        draw_triangle(&f, &t);                                    /* :228 */
    }
    if (stats) *stats = f.st;
    return 0;
}

/* Renderer.render with primitiveType .line / .vertices (Renderer.swift:210-229, :289-302). */
#define SWRO_LINE_MAX_STEPS (1 << 20)   /* longer lines are skipped (documented deviation, include/swr.h SWR_FLAG_REAL_LINES) */
int swro_render_primitives(uint8_t* color, float* depth, int64_t W, int64_t H,
                           const swro_vertex* vertices, int64_t vertex_count,
                           const int64_t* indices, int64_t index_count,
                           const float M[16], uint32_t flags, int32_t primitive_type,
                           int64_t row_begin, int64_t row_end, swro_stats* stats) {
    if (primitive_type == 0)
        return swro_render(color, depth, W, H, vertices, vertex_count, indices, index_count, M, flags,
                           row_begin, row_end, stats);
    if (primitive_type != 1 && primitive_type != 2) return -5;
    if (!depth || W <= 0 || H <= 0 || !M) return -1;
    if (!(flags & SWRO_NO_COLOR) && !color) return -1;
    if (index_count < 0 || vertex_count < 0) return -1;
    if (index_count > 0 && (!indices || !vertices)) return -1;
    const int per = primitive_type == 1 ? 2 : 3;                  /* verticesCount :179-188 */
    if (index_count % per != 0) return -2;                        /* :209 */
    if (row_begin < 0 || row_end > H || row_begin > row_end) return -1;
    for (int64_t i = 0; i < index_count; i++)
        if (indices[i] < 0 || indices[i] >= vertex_count) return -3;
    swro_stats st;
    memset(&st, 0, sizeof st);
    /* clear (:205-206) */
    if (!(flags & SWRO_NO_COLOR))
        memset(color + (size_t)row_begin * (size_t)W * 4, 0, (size_t)(row_end - row_begin) * (size_t)W * 4);
    for (int64_t i = row_begin * W; i < row_end * W; i++) depth[i] = INFINITY;
    if (primitive_type == 2 && !(flags & SWRO_NO_COLOR)) {
        float fw = (float)W, fh = (float)H;
        for (int64_t i = 0; i < index_count; i++) {               /* :222-229 then :296-301 */
            const swro_vertex* v = &vertices[indices[i]];
            float x = v->xyz[0], y = v->xyz[1], z = v->xyz[2];
            float r[4];
            apply_transform(M, x, y, z, flags, r);
            float nx = r[0] / r[3], ny = r[1] / r[3];
            float sx = (nx * 0.5f + 0.5f) * fw;                   /* convertedToScreen :165-171 */
            float sy = (ny * -0.5f + 0.5f) * fh;
            if (!(fabsf(sx) < COORD_LIMIT) || !(fabsf(sy) < COORD_LIMIT)) { st.triangles_skipped++; continue; }
            int64_t px = (int64_t)sx, py = (int64_t)sy;           /* Int(v.xyz.x) :298-299 */
            if (px < 0 || px >= W || py < row_begin || py >= row_end) continue;   /* setter drops OOB :30-36 */
            st.fragments++;
            st.fragments_written++;
            uint8_t* p = color + (size_t)(py * W + px) * 4;       /* Pixel(float3: vertex.color) :300 */
            p[0] = swro_quantise(v->color[2]);
            p[1] = swro_quantise(v->color[1]);
            p[2] = swro_quantise(v->color[0]);
            p[3] = swro_quantise(1.0f);
        }
    }
    if (primitive_type == 1 && (flags & SWRO_REAL_LINES) && !(flags & SWRO_NO_COLOR)) {
        /* OPT-IN: the reference's own line DDA, draw(line:with:in:) (Renderer.swift:405-419), between the two transformed
         * endpoints truncated like draw(vertices:) does (:298-299); colour of the first vertex; in index order. */
        const float fw = (float)W, fh = (float)H;
        for (int64_t l = 0; l < index_count / 2; l++) {
            int64_t ex[2], ey[2];
            int ok = 1;
            for (int k = 0; k < 2; k++) {
                const swro_vertex* v = &vertices[indices[2 * l + k]];
                float r[4];
                apply_transform(M, v->xyz[0], v->xyz[1], v->xyz[2], flags, r);
                const float nx = r[0] / r[3], ny = r[1] / r[3];
                const float sx = (nx * 0.5f + 0.5f) * fw, sy = (ny * -0.5f + 0.5f) * fh;
                if (!(fabsf(sx) < COORD_LIMIT) || !(fabsf(sy) < COORD_LIMIT)) { ok = 0; break; }
                ex[k] = (int64_t)sx; ey[k] = (int64_t)sy;
            }
            if (!ok) { st.triangles_skipped++; continue; }
            const int64_t dx = ex[1] - ex[0], dy = ey[1] - ey[0];                       /* :406-407 */
            const int64_t adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
            const int64_t steps = adx > ady ? adx : ady;                                 /* :408 */
            if (steps > SWRO_LINE_MAX_STEPS) { st.triangles_skipped++; continue; }
            st.triangles_drawn++;
            const float xs = (float)dx / (float)steps, ys = (float)dy / (float)steps;   /* :409-410 */
            float x = (float)ex[0], y = (float)ey[0];                                   /* :412-413 */
            const swro_vertex* va = &vertices[indices[2 * l]];
            for (int64_t k = 0; k < steps; k++) {                                        /* :414 */
                const int64_t px = (int64_t)roundf(x), py = (int64_t)roundf(y);         /* :415 rounded(): half away from zero */
                if (px >= 0 && px < W && py >= row_begin && py < row_end) {              /* the setter drops OOB (:30-36) */
                    st.fragments++;
                    st.fragments_written++;
                    uint8_t* p = color + (size_t)(py * W + px) * 4;
                    p[0] = swro_quantise(va->color[2]);
                    p[1] = swro_quantise(va->color[1]);
                    p[2] = swro_quantise(va->color[0]);
                    p[3] = swro_quantise(1.0f);
                }
                x += xs;                                                                 /* :416-417 */
                y += ys;
            }
        }
    }
    if (stats) *stats = st;
    return 0;
}

/* ---- the Metal path's rules (Shaders.metal / GpuRenderer.swift), see swr_oracle.h ------------- */
static uint8_t unorm8(float v) {            /* bgra8Unorm store: clamp, *255, round to nearest even */
    float c = fminf(fmaxf(v, 0.0f), 1.0f);
    return (uint8_t)rintf(c * 255.0f);
}

int swro_render_metal(uint8_t* color, float* depth, int64_t W, int64_t H,
                      const swro_vertex* vertices, int64_t vertex_count,
                      const int64_t* indices, int64_t index_count,
                      const float M[16], uint32_t flags,
                      int64_t row_begin, int64_t row_end, swro_stats* stats) {
    if (!depth || W <= 0 || H <= 0 || !M) return -1;
    if (!(flags & SWRO_NO_COLOR) && !color) return -1;
    if (index_count < 0 || vertex_count < 0) return -1;
    if (index_count > 0 && (!indices || !vertices)) return -1;
    if (index_count % 3 != 0) return -2;
    if (row_begin < 0 || row_end > H || row_begin > row_end) return -1;
    for (int64_t i = 0; i < index_count; i++)
        if (indices[i] < 0 || indices[i] >= vertex_count) return -3;
    swro_stats st;
    memset(&st, 0, sizeof st);
    /* clear_depth_buffer (Shaders.metal:33-37) + colour clear (GpuRenderer.swift:78) */
    if (!(flags & SWRO_NO_COLOR))
        memset(color + (size_t)row_begin * (size_t)W * 4, 0, (size_t)(row_end - row_begin) * (size_t)W * 4);
    for (int64_t i = row_begin * W; i < row_end * W; i++) depth[i] = INFINITY;

    const float fw = (float)W, fh = (float)H;
    for (int64_t p = 0; p < index_count / 3; p++) {               /* serial dispatches, GpuRenderer.swift:117 */
        float px[3], py[3], pz[3], col[3][3], nrm[3][3] = {{0}}, tuv[3][2] = {{0}};
        int ok = 1;
        for (int k = 0; k < 3; k++) {
            const swro_vertex* v = &vertices[indices[3 * p + k]];
            float x = v->xyz[0], y = v->xyz[1], z = v->xyz[2];
            float r[4];
            for (int c = 0; c < 4; c++) {                         /* vertex_shader :50 */
                float a = M[0 + c] * x;
                a = a + M[4 + c] * y;
                a = a + M[8 + c] * z;
                a = a + M[12 + c] * 1.0f;
                r[c] = a;
            }
            float nx = r[0] / r[3], ny = r[1] / r[3], nz = r[2] / r[3];   /* :68 */
            float u = nx * 0.5f + 0.5f, w = ny * -0.5f + 0.5f;             /* :70 */
            px[k] = roundf(u * fw);                                         /* :71 round = half away from zero */
            py[k] = roundf(w * fh);
            pz[k] = nz;
            for (int c = 0; c < 3; c++) col[k][c] = v->color[c];
            if (tls_shading) {
                const swro_vertex_attr* av = &tls_shading->attrs[indices[3 * p + k]];
                for (int c = 0; c < 3; c++) nrm[k][c] = av->normal[c];
                tuv[k][0] = av->uv[0]; tuv[k][1] = av->uv[1];
            }
            /* uint2(pos.xy) (:102-104): undefined for negative / non-finite -> skip (documented) */
            if (!(px[k] >= 0.0f && px[k] < COORD_LIMIT) || !(py[k] >= 0.0f && py[k] < COORD_LIMIT)) ok = 0;
        }
        if (!ok) { st.triangles_skipped++; continue; }
        int64_t ax = (int64_t)px[0], ay = (int64_t)py[0], bx = (int64_t)px[1], by = (int64_t)py[1];
        int64_t cx = (int64_t)px[2], cy = (int64_t)py[2];
        int64_t minx = ax < bx ? (ax < cx ? ax : cx) : (bx < cx ? bx : cx);
        int64_t maxx = ax > bx ? (ax > cx ? ax : cx) : (bx > cx ? bx : cx);
        int64_t miny = ay < by ? (ay < cy ? ay : cy) : (by < cy ? by : cy);
        int64_t maxy = ay > by ? (ay > cy ? ay : cy) : (by > cy ? by : cy);
        if (minx == 0 || miny == 0) { st.triangles_skipped++; continue; }   /* GpuRenderer.swift:122-124 */
        st.triangles_drawn++;
        /* rasterizer_pass :133-166, one "thread" per ROI pixel; texture bounds + band scissor */
        const float p1x = px[0], p1y = py[0], p2x = px[1], p2y = py[1], p3x = px[2], p3y = py[2];
        const float divider = (p1x - p3x) * (p2y - p3y) - (p2x - p3x) * (p1y - p3y);        /* :143 */
        int64_t y0 = miny < row_begin ? row_begin : miny, y1 = maxy > row_end - 1 ? row_end - 1 : maxy;
        int64_t x0 = minx, x1 = maxx > W - 1 ? W - 1 : maxx;
        for (int64_t y = y0; y <= y1; y++)
            for (int64_t x = x0; x <= x1; x++) {
                const float sxp = (float)x + 0.5f, syp = (float)y + 0.5f;                    /* :133 */
                float w0 = (p2y - p3y) * (sxp - p3x) + (p3x - p2x) * (syp - p3y);            /* :144 */
                w0 = w0 / divider;                                                           /* :145 */
                float w1 = (p3y - p1y) * (sxp - p3x) + (p1x - p3x) * (syp - p3y);            /* :147 */
                w1 = w1 / divider;                                                           /* :148 */
                const float w2 = 1.0f - w0 - w1;                                             /* :149 */
                st.fragments++;
                if (!(0.0f <= w0 && w0 <= 1.0f && 0.0f <= w1 && w1 <= 1.0f && 0.0f <= w2 && w2 <= 1.0f))
                    continue;                                                                /* :153 */
                const float z = w0 * pz[0] + w1 * pz[1] + w2 * pz[2];                        /* :157,159 */
                const size_t at = (size_t)(y * W + x);
                if (!(z < depth[at])) continue;                                              /* :161 */
                st.fragments_written++;
                if (!(flags & SWRO_NO_COLOR)) {
                    float c[3];
                    for (int ch = 0; ch < 3; ch++)                                           /* :162 */
                        c[ch] = w0 * col[0][ch] + w1 * col[1][ch] + w2 * col[2][ch];
                    float o[4] = {c[0], c[1], c[2], 1.0f};                                   /* fragment_shader :116-121 */
                    if (tls_shading) {
                        float n[3], uv[2];
                        for (int ch = 0; ch < 3; ch++) n[ch] = w0 * nrm[0][ch] + w1 * nrm[1][ch] + w2 * nrm[2][ch];
                        for (int ch = 0; ch < 2; ch++) uv[ch] = w0 * tuv[0][ch] + w1 * tuv[1][ch] + w2 * tuv[2][ch];
                        swro_fragment(tls_shading, c, n, uv, o);
                    }
                    uint8_t* q = color + at * 4;                                             /* :163 bgra8Unorm */
                    q[0] = unorm8(o[2]); q[1] = unorm8(o[1]); q[2] = unorm8(o[0]); q[3] = unorm8(o[3]);
                }
                depth[at] = z;                                                               /* :164 */
            }
    }
    if (stats) *stats = st;
    return 0;
}

/* ---- extended fragment stage: the same two renderers with swro_fragment at the colour store ---- */
static int shading_ok(const swro_shading* sh, int64_t vertex_count) {
    if (!sh || sh->material.shader == 0) return 1;
    if (sh->material.shader != 1 && sh->material.shader != 2) return 0;
    if (vertex_count > 0 && !sh->attrs) return 0;
    if (sh->material.shininess_log2 < 0 || sh->material.shininess_log2 > 16) return 0;
    if (sh->material.shader == 2 && (!sh->texture || sh->tex_w <= 0 || sh->tex_h <= 0)) return 0;
    return 1;
}

int swro_render_shaded(uint8_t* color, float* depth, int64_t W, int64_t H,
                       const swro_vertex* vertices, int64_t vertex_count,
                       const int64_t* indices, int64_t index_count,
                       const float M[16], uint32_t flags,
                       int64_t row_begin, int64_t row_end, swro_stats* stats, const swro_shading* sh) {
    if (!shading_ok(sh, vertex_count)) return -1;
    tls_shading = (sh && sh->material.shader != 0) ? sh : NULL;
    int rc = swro_render(color, depth, W, H, vertices, vertex_count, indices, index_count, M, flags,
                         row_begin, row_end, stats);
    tls_shading = NULL;
    return rc;
}

int swro_render_metal_shaded(uint8_t* color, float* depth, int64_t W, int64_t H,
                             const swro_vertex* vertices, int64_t vertex_count,
                             const int64_t* indices, int64_t index_count,
                             const float M[16], uint32_t flags,
                             int64_t row_begin, int64_t row_end, swro_stats* stats, const swro_shading* sh) {
    if (!shading_ok(sh, vertex_count)) return -1;
    tls_shading = (sh && sh->material.shader != 0) ? sh : NULL;
    int rc = swro_render_metal(color, depth, W, H, vertices, vertex_count, indices, index_count, M, flags,
                               row_begin, row_end, stats);
    tls_shading = NULL;
    return rc;
}
