// mesh_class.h -- host side: is an object's mesh a closed orientable surface, which triangles are wound inwards, and the order
// in which the rasteriser walks the triangles (mesh_class.hip).
#pragma once

#include <cstdint>
#include <vector>

namespace roft {

// The render contract's classification (oracle/ro_meshclass.c states the rules; this is the engine's own implementation of
// them): true and flip[t] = 1 for every triangle wound clockwise seen from outside when the mesh is a closed orientable
// surface; false (flip zeroed) for anything else -- such a mesh is drawn whole, exactly as the reference draws every mesh
// (depth test LESS, no culling: src/roft-lib/src/SICAD.cpp:271-272).
bool classify_mesh(const float* verts, int n_verts, const int32_t* tris, int n_tris, std::vector<uint8_t>& flip);

// The device copy of a CLOSED mesh's triangles: the same triangles (vertex order inside a triangle untouched -- the per-pixel
// arithmetic depends on it), reordered so that triangles whose outward normals point alike are neighbours (24 direction
// buckets: cube face x 2 x 2).  The render's result does not depend on the order (nearest depth per pixel); the order decides
// which triangles share a wave, and a wave whose 64 triangles all face away is skipped whole.  order[k] = index of the
// triangle walked k-th.
void facing_coherent_order(const float* verts, const int32_t* tris, int n_tris, const std::vector<uint8_t>& flip, std::vector<int32_t>& order);

// What roft_object_add uploads: closed -> the reordered triangles + their flip bits, else the caller's triangles as they are.
struct PreparedMesh {
    bool closed = false;
    std::vector<int32_t> reordered;   // 3 x n_tris, closed meshes only
    std::vector<uint8_t> flip;        // n_tris, in the reordered order
    const int32_t* tris(const int32_t* callers) const { return closed ? reordered.data() : callers; }
};
void prepare_mesh(const float* verts, int n_verts, const int32_t* tris, int n_tris, PreparedMesh& out);

}  // namespace roft
