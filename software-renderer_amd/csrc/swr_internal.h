// swr_internal.h — structures shared between the HIP kernels (swr_kernels.hip) and the C-ABI
// host code (swr_api.hip).  Nothing here is part of the public boundary (include/swr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/swr.h"

namespace swr {

// Screen-space tile owned by one workgroup of the raster kernel.  The per-pixel visibility
// key tile (8 B/pixel) lives in LDS: 64x32 -> 16 KiB, i.e. up to 8 workgroups per CU by LDS.
constexpr int TILE_W = 64;
constexpr int TILE_H = 32;
constexpr int RASTER_THREADS = 256;

// Per-triangle record written by the setup kernel and gathered by the raster kernel:
// 32 bytes = 2 x 16-byte loads.  B and C are stored relative to A as int16, which is exact for
// every GEOM_SMALL triangle (bbox extents < 2^15); the rare others keep their absolute
// coordinates in GeomFull.  T() is recomputed from the integer vertices where it is needed (same
// expressions -> same bits), so it does not travel through HBM.
struct GeomRec {
    int32_t ax, ay;                // truncated screen vertex A (Renderer.swift:251)
    int16_t dbx, dby;              // B - A
    int16_t dcx, dcy;              // C - A
    float za, zb, zc;              // NDC z of a,b,c (Renderer.swift:254-256)
    uint32_t flags;                // GEOM_* below
};
static_assert(sizeof(GeomRec) == 32, "GeomRec must be 32 bytes");

// Absolute integer vertices, written and read only for triangles without GEOM_SMALL.
struct GeomFull {
    int32_t ax, ay, bx, by, cx, cy, pad0, pad1;
};
static_assert(sizeof(GeomFull) == 32, "GeomFull must be 32 bytes");

enum : uint32_t {
    GEOM_VALID = 1u << 0,
    GEOM_SMALL = 1u << 1,          // bbox extents < 2^15: int16 deltas and 32-bit span arithmetic are exact
    GEOM_ORD_SHIFT = 2,            // 3 x 2 bits: which of a,b,c is S0,S1,S2 of the y-sorted list (:271)
    GEOM_ORIG_SHIFT = 8            // bits 8..31: the primitive's ORIGINAL index (scenes below 2^24 primitives;
                                   // larger scenes are not reordered, so slot == original index)
};
constexpr int64_t SORT_MAX_TRIS = 1ll << 24;

struct Target {
    int32_t width, height;         // full framebuffer
    int32_t row_begin, row_end;    // band owned by this context
    int32_t tiles_x, tiles_y;      // tiles in the band
};

// device-side frame counters (one 32-bit word each)
enum { CNT_PAIRS = 0, CNT_OVERFLOW = 1, CNT_BAD_INDEX = 2, CNT_WORDS = 8 };

// Geometry of the LDS binning path (see plan_binning in swr_kernels.hip).
struct BinPlan {
    bool use_lds;
    int threads;        // workgroup size of the two binning walks
    int G;              // workgroups = rows of the count matrix
    int chunk;          // primitives per workgroup
    size_t lds_bytes;   // tiles * 4
};
BinPlan plan_binning(int64_t ntri, int ntiles, bool force_atomic);
int live_groups_per_workgroup(int64_t ntri, int G);

// Everything one frame needs, all device pointers.  colour/depth are band-local: element
// (x, y) of the full image lives at [(y - row_begin) * width + x].
struct DeviceFrame {
    const swr_vertex* vertices;    // as uploaded (AoS)
    const int64_t* indices;
    // the triangle stream (swr_upload.hip): primitives in Morton order of their centroid, de-indexed
    const float4* tri_xyz;         // [ni] corner positions; w of corner 0 = original primitive index (bits)
    const uint32_t* inv;           // [ntri] sorted slot of original primitive o
    const float4* box64;           // [2 * ceil(ntri/64)] object-space min / max of each 64-slot group
    int32_t reordered;             // 1: slots are a permutation (original index in GeomRec.flags); 0: slot == index
    const float4* tri_rgb;         // [ni] colours per slot corner (48 B / triangle); lane 3 = v
    const float4* tri_nrm;         // [ni] (nx,ny,nz,u) per primitive corner (extended fragment stage)
    swr_material material;         // shader == SWR_SHADER_PASSTHROUGH: the reference's stage
    const float4* texels;          // texture of swr_texture_upload, converted to (r,g,b,a) floats
    int32_t tex_w, tex_h;
    int64_t vertex_count;
    int64_t index_count;           // (.vertices / .line passes: ntri = index_count / 3 is the triangle count only)
    int64_t ntri;
    GeomRec* geo;
    GeomFull* geo_full;
    uint32_t* tile_count;          // [tiles] (triangle,tile) pairs per tile
    uint32_t* tile_start;          // [tiles+1] exclusive scan of tile_count
    uint32_t* counters;            // [CNT_WORDS]
    uint32_t* host_counters;       // device-visible address of the pinned host copy
    uint32_t* host_max;            // pinned host word: entries of the frame's fullest bin (written by k_fill_lds; may be NULL)
    int32_t insort;                // 1: k_raster_depth sorts every bin itself (no k_sort_bins launch for this frame)
    int32_t k32;                   // 1: depth-only z-tested frames take the 32-bit depth keys (k_raster_depth)
    uint32_t* redo_dev;            // device word: tiles k_raster_depth had to raster again (sampled)
    uint32_t* host_redo;           // pinned host word that receives it one launch later
    int32_t defer_big;             // 1: the previous frame had triangles for the deferred list (k_bin<DEFER>)
    int32_t skip_sort;             // 1: no k_sort_bins for this frame (every bin of the previous frames fitted two chunks)
    uint32_t* tile_cursor;         // [tiles] running fill position (starts as tile_start)
    uint2* ranges;                 // [ntri] band-clipped pixel bbox (x0|x1<<16, y0|y1<<16, y band-relative)
    uint32_t* bins;                // [capacity] primitive ids grouped by tile (exact bins) / [tiles * cap_tile] (fixed-stride bins)
    // fixed-stride bins (k_bin, one launch): tile t owns bins[t * cap_tile, +fill[CNT_WORDS + t])
    int32_t fixed_bins;            // 1: this frame is binned by k_bin
    uint32_t cap_tile;             // entries per tile region
    uint32_t* fill;                // [CNT_WORDS counters][tiles] of this frame, zero when k_bin starts
    uint32_t* fill_next;           // the next frame's block (k_bin zeroes it)
    uint32_t* host_fill;           // pinned host word: the frame's largest fill (overflow test)
    uint4* biglist;                // [1024] triangles of the frame that cover too many tiles to be scattered by k_bin: k_sort_bins appends them per tile
    uint32_t* bin_matrix;          // [G][tiles] per-workgroup tile counts -> prefixes (LDS path)
    uint32_t* live;                // [G][1 + ceil(groups/G)] per binning workgroup: count + the stream groups that survived its cull
    int32_t live_parity;           // >= 0: cull the groups against the band; -1: every group is live
    BinPlan plan;
    uint32_t capacity;
    uint8_t* color;
    float* depth;
    Target tg;
    float m[16];                   // column-major transform
    uint32_t flags;                // SWR_FLAG_*
};

void launch_validate_indices(const int64_t* indices, int64_t count, int64_t vertex_count,
                             uint32_t* counters, hipStream_t s);
hipError_t prepare_device();      // per-device kernel attributes (after hipSetDevice)

// swr_upload.hip: the once-per-scene triangle stream
struct StreamBuild {
    const swr_vertex* vertices; int64_t nv;
    const int64_t* indices; int64_t ntri;
    bool sort;                     // false: identity order (scenes of 2^24 primitives or more, or SWR_SORT=0)
    uint32_t* scratch;             // 4 * ntri + 8 words
    void* sort_temp; size_t sort_temp_bytes;
    float4* tri_xyz; float4* tri_rgb; uint32_t* inv; float4* box64;
};
size_t stream_sort_temp_bytes(int64_t ntri);
hipError_t launch_build_stream(const StreamBuild& b, hipStream_t s);
hipError_t launch_build_stream_range(const StreamBuild& b, int64_t t0, int64_t t1, hipStream_t s);   // index order, primitives [t0, t1)
void launch_gather_attrs(const swr_vertex_attr* attrs, int64_t nv, const int64_t* indices, int64_t ntri,
                         const float4* tri_xyz, float4* tri_nrm, float4* tri_rgb, hipStream_t s);
void launch_texture_to_float(const uint32_t* bgra, int64_t n, float4* out, hipStream_t s);
uint32_t fixed_cap_max(int64_t ntri, int ntiles);   // largest tile region k_bin can fill (0: the frame needs the exact-size path)
bool launch_bin(const DeviceFrame& f, hipStream_t s, hipEvent_t stop = nullptr);
void launch_setup_bin(const DeviceFrame& f, hipStream_t s);
void launch_scan(const DeviceFrame& f, hipStream_t s);
// stop != NULL: the event is bound to the (last) kernel launched, as its completion; returns whether a kernel carries it
bool launch_fill(const DeviceFrame& f, hipStream_t s, hipEvent_t stop = nullptr);
bool launch_sort_bins(const DeviceFrame& f, hipStream_t s, hipEvent_t stop = nullptr);
bool launch_raster(const DeviceFrame& f, hipStream_t s, hipEvent_t stop = nullptr);
bool frame_uses_k32(const DeviceFrame& f);   // the frame's raster is k_raster_depth (32-bit depth keys, sorts its bins itself)
void launch_points_or_lines(const DeviceFrame& f, int primitive_type, hipStream_t s);

}  // namespace swr
