/*
 * rt_from_reference.h -- the reference-side binding as code: flattens the reference's OWN objects (its Scene with
 * RayTraceMesh / Sphere / Square / Light / Material / BVH_Node, its Camera) into the rt_scene_desc / rt_camera of
 * include/rtgpu.h, so that main.cpp keeps its types, its setup_* functions and its trees, and only the body of
 * ray_trace_from_camera() changes (integration/main_gpu.patch).
 *
 * Header-only and written as templates over the member NAMES of the reference (cited below), so that
 *   - inside the reference tree it is instantiated with the reference's classes (include it after Scene.h / Camera.h);
 *   - in this repository, which may not hold the reference's sources, tests/test_integration_binding.py instantiates it with
 *     stand-in types that carry the same member names (integration/test/ref_like.h), fills them from an rth::Scene and
 *     compares the result byte for byte with rth::Scene::flatten().
 * Scene::meshes / spheres / squares / lights / sceneBvh (Scene.h:43-51), RayTraceMesh::triangleTree (Mesh.h:211) and the
 * fields of Camera (Camera.h:52-64) are private in the reference: the patch adds `friend struct rtref::Access;` to those
 * three classes -- three lines, nothing else of them changes.
 *
 * Names used:
 *   Scene        meshes, spheres, squares, lights, sceneBvh                                   Scene.h:43-49
 *   Mesh         vertices[i].position / .normal, triangles[t].v[0..2], material               Mesh.h:26-63,103-111
 *   RayTraceMesh triangleTree (std::unique_ptr<BVH_Node>)                                     Mesh.h:210-213
 *   Sphere       m_center, m_radius                                                           Sphere.h:39-40
 *   Square       vertices[0], [1], [3] (.position) after the transformations                  Square.h:47-50,67-69
 *   Light        pos, radius, material                                                        Scene.h:20-34
 *   Material     ambient_material, diffuse_material, specular_material, shininess, type, index_medium, transparency,
 *                reflectivity, hasTexture, textureData, textureWidth/Height/Channels, hasNormalMap, normalMapData,
 *                normalMapWidth/Height/Channels, materialScale                                Material.h:18-38
 *   BVH_Node     leaf, l_child, r_child, trianglesIdx, meshes (BuildingMesh: ref, type), boundingBox (AABB: min, max)
 *                                                                                             AccelerationStruct.h:11-18,88-106
 *   Camera       x, y, z, _zoom, curquat, fovAngle, aspectRatio, nearPlane, farPlane          Camera.h:52-64
 *
 * The trees are written out in the order the reference's constructors built them (node, left subtree, right subtree:
 * AccelerationStruct.cpp:151-243): inner node k of a tree becomes rt_bvh_node k with the boxes of BOTH children
 * (l_child->boundingBox, r_child->boundingBox -- what BVH_Node::intersect tests, AccelerationStruct.cpp:262-263,321-322),
 * leaf j becomes rt_bvh_leaf j, link = node index or -1 - leaf index, items in leaf order. Object ordinals count squares,
 * then spheres, then meshes (AccelerationStruct.cpp:123-134).
 */
#ifndef RT_FROM_REFERENCE_H
#define RT_FROM_REFERENCE_H
#include "rtgpu.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <vector>

namespace rtref {

/* what rt_flatten fills; `desc` points into the vectors (keep the object alive until rt_upload_scene has returned) */
struct Flattened {
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_square> squares;
    std::vector<rt_sphere> spheres;
    std::vector<rt_mesh> meshes;
    std::vector<float> positions, normals;
    std::vector<uint32_t> tris, items;
    std::vector<rt_bvh_node> nodes, top_nodes;
    std::vector<rt_bvh_leaf> leaves, top_leaves;
    std::vector<uint32_t> top_items;
    std::vector<rt_texture> textures;
    int32_t top_root = -1, top_depth = 0;
    rt_scene_desc desc;
};

/* The one place that touches private members: befriended by Scene, RayTraceMesh and Camera (integration/main_gpu.patch). */
struct Access {
    template <class SceneT> static auto& meshes(const SceneT& s) { return s.meshes; }
    template <class SceneT> static auto& spheres(const SceneT& s) { return s.spheres; }
    template <class SceneT> static auto& squares(const SceneT& s) { return s.squares; }
    template <class SceneT> static auto& lights(const SceneT& s) { return s.lights; }
    template <class SceneT> static auto& sceneBvh(const SceneT& s) { return s.sceneBvh; }
    template <class MeshT> static auto* triangleTree(const MeshT& m) { return m.triangleTree.get(); }
    template <class CameraT> static rt_camera camera(const CameraT& c) {
        rt_camera r;
        r.x = c.x; r.y = c.y; r.z = c.z; r.zoom = c._zoom;
        for (int i = 0; i < 4; i++) r.quat[i] = c.curquat[i];
        r.fovy_deg = c.fovAngle; r.aspect = c.aspectRatio; r.znear = c.nearPlane; r.zfar = c.farPlane;
        return r;
    }
};

namespace detail {

template <class BoxT>
inline void store_boxes(rt_bvh_node& n, const BoxT& l, const BoxT& r) {
    for (int k = 0; k < 3; k++) { n.lmin[k] = l.min[k]; n.lmax[k] = l.max[k]; n.rmin[k] = r.min[k]; n.rmax[k] = r.max[k]; }
}

/* One BVH_Node tree -> flat arrays, numbered as the reference's constructors recursed. item_of(node, i) is the i-th item of a
 * leaf (a triangle index, or an object ordinal); count_of(node) their number. Returns the link of the subtree. */
template <class NodeT, class CountOf, class ItemOf>
int32_t flatten_tree(const NodeT& n, CountOf count_of, ItemOf item_of, std::vector<rt_bvh_node>& nodes, std::vector<rt_bvh_leaf>& leaves,
                     std::vector<uint32_t>& items, int level, int32_t& depth) {
    if (n.leaf || !n.l_child || !n.r_child) {
        rt_bvh_leaf lf;
        lf.first = (uint32_t)items.size();
        lf.count = (uint32_t)count_of(n);
        for (uint32_t i = 0; i < lf.count; i++) items.push_back((uint32_t)item_of(n, i));
        leaves.push_back(lf);
        depth = std::max(depth, (int32_t)level);
        return -(int32_t)leaves.size();                  /* link = -1 - leaf index */
    }
    const int32_t me = (int32_t)nodes.size();
    nodes.emplace_back();
    std::memset(&nodes[(size_t)me], 0, sizeof(rt_bvh_node));
    const int32_t l = flatten_tree(*n.l_child, count_of, item_of, nodes, leaves, items, level + 1, depth);
    const int32_t r = flatten_tree(*n.r_child, count_of, item_of, nodes, leaves, items, level + 1, depth);
    store_boxes(nodes[(size_t)me], n.l_child->boundingBox, n.r_child->boundingBox);
    nodes[(size_t)me].left = l;
    nodes[(size_t)me].right = r;
    return me;
}

/* Material (Material.h:18-38) -> rt_material; images are collected once per distinct texel pointer */
template <class MaterialT>
int32_t add_material(Flattened& out, const MaterialT& m) {
    rt_material r;
    std::memset(&r, 0, sizeof(r));
    for (int k = 0; k < 3; k++) { r.ka[k] = m.ambient_material[k]; r.kd[k] = m.diffuse_material[k]; r.ks[k] = m.specular_material[k]; }
    r.shininess = m.shininess;                  /* uninitialised in the reference unless the scene sets it (Material.h:22,40-47): set it */
    r.type = (int32_t)m.type;                   /* Material_Diffuse_Blinn_Phong, _Glass, _Mirror, _Checker = RT_MAT_* 0..3 */
    r.ior = m.index_medium; r.transparency = m.transparency; r.reflectivity = m.reflectivity;
    auto image = [&](bool has, const unsigned char* data, int w, int h, int ch) -> int32_t {
        if (!has || !data || w <= 0 || h <= 0) return 0;                       /* a failed stbi_load leaves the material plain (Material.h:128-138) */
        for (size_t i = 0; i < out.textures.size(); i++) if (out.textures[i].texels == data) return (int32_t)i + 1;
        rt_texture t;
        std::memset(&t, 0, sizeof(t));
        t.width = w; t.height = h; t.channels = ch; t.texels = data;
        out.textures.push_back(t);
        return (int32_t)out.textures.size();
    };
    r.texture = image(m.hasTexture, m.textureData, m.textureWidth, m.textureHeight, m.textureChannels);
    r.normal_map = image(m.hasNormalMap, m.normalMapData, m.normalMapWidth, m.normalMapHeight, m.normalMapChannels);
    r.material_scale = m.materialScale;
    out.materials.push_back(r);
    return (int32_t)out.materials.size() - 1;
}

}  // namespace detail

/* Scene (after its setup_* function and applySettings, main.cpp:387-395) -> out.desc. Returns RT_OK, or RT_ERR_ARG if a mesh has
 * no tree yet (applySettings / build_arrays not run) or a top-level leaf refers to an object the scene does not hold. */
template <class SceneT>
int rt_flatten(const SceneT& scene, Flattened& out) {
    out = Flattened();
    const auto& meshes = Access::meshes(scene);
    const auto& spheres = Access::spheres(scene);
    const auto& squares = Access::squares(scene);
    for (const auto& l : Access::lights(scene)) {
        rt_light r;
        std::memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) { r.pos[k] = l.pos[k]; r.color[k] = l.material[k]; }
        r.radius = l.radius;
        out.lights.push_back(r);
    }
    for (const auto& s : squares) {                      /* Square::intersect reads vertices[0], [1], [3] (Square.h:67-69) */
        rt_square r;
        for (int k = 0; k < 3; k++) { r.v0[k] = s.vertices[0].position[k]; r.v1[k] = s.vertices[1].position[k]; r.v3[k] = s.vertices[3].position[k]; }
        r.material = detail::add_material(out, s.material);
        out.squares.push_back(r);
    }
    for (const auto& s : spheres) {
        rt_sphere r;
        std::memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) r.center[k] = s.m_center[k];
        r.radius = s.m_radius;
        r.material = detail::add_material(out, s.material);
        out.spheres.push_back(r);
    }
    for (const auto& m : meshes) {
        const auto* tree = Access::triangleTree(m);
        if (!tree) return RT_ERR_ARG;
        rt_mesh r;
        std::memset(&r, 0, sizeof(r));
        r.vert_first = (int64_t)out.positions.size() / 3; r.vert_count = (int64_t)m.vertices.size();
        r.tri_first = (int64_t)out.tris.size() / 3; r.tri_count = (int64_t)m.triangles.size();
        r.node_first = (int64_t)out.nodes.size(); r.leaf_first = (int64_t)out.leaves.size();
        r.material = detail::add_material(out, m.material);
        for (const auto& v : m.vertices)
            for (int k = 0; k < 3; k++) { out.positions.push_back(v.position[k]); out.normals.push_back(v.normal[k]); }
        for (const auto& t : m.triangles) for (int k = 0; k < 3; k++) out.tris.push_back((uint32_t)t.v[k]);
        /* the mesh's tree with links, leaf ranges and items local to the mesh, as rt_mesh wants them */
        std::vector<rt_bvh_node> nodes;
        std::vector<rt_bvh_leaf> leaves;
        std::vector<uint32_t> items;
        int32_t depth = 0;
        r.root = detail::flatten_tree(*tree, [](const auto& n) { return n.trianglesIdx.size(); },
                                      [](const auto& n, uint32_t i) { return n.trianglesIdx[i]; }, nodes, leaves, items, 0, depth);
        r.depth = depth;
        if ((int64_t)items.size() != r.tri_count) return RT_ERR_ARG;      /* every triangle sits in exactly one leaf */
        r.node_count = (int64_t)nodes.size(); r.leaf_count = (int64_t)leaves.size();
        out.nodes.insert(out.nodes.end(), nodes.begin(), nodes.end());
        out.leaves.insert(out.leaves.end(), leaves.begin(), leaves.end());
        out.items.insert(out.items.end(), items.begin(), items.end());
        out.meshes.push_back(r);
    }
    /* top level: BuildingMesh::ref points at the object inside the scene's vectors, ::type says which vector (0 Square, 1 Sphere,
     * 2 RayTraceMesh: AccelerationStruct.cpp:123-134) */
    bool bad = false;
    const size_t n_sq = squares.size(), n_sp = spheres.size(), n_me = meshes.size();
    auto ordinal = [&](const auto& n, uint32_t i) -> uint32_t {
        const auto& bm = n.meshes[i];
        size_t k = 0, limit = 0, base = 0;
        if (bm.type == 0) { k = (size_t)(static_cast<const typename std::decay<decltype(squares[0])>::type*>(bm.ref) - squares.data()); limit = n_sq; base = 0; }
        else if (bm.type == 1) { k = (size_t)(static_cast<const typename std::decay<decltype(spheres[0])>::type*>(bm.ref) - spheres.data()); limit = n_sp; base = n_sq; }
        else { k = (size_t)(static_cast<const typename std::decay<decltype(meshes[0])>::type*>(bm.ref) - meshes.data()); limit = n_me; base = n_sq + n_sp; }
        if (k >= limit) { bad = true; return 0u; }
        return (uint32_t)(base + k);
    };
    if (n_sq + n_sp + n_me > 0)
        out.top_root = detail::flatten_tree(Access::sceneBvh(scene), [](const auto& n) { return n.meshes.size(); }, ordinal, out.top_nodes, out.top_leaves,
                                            out.top_items, 0, out.top_depth);
    if (bad) return RT_ERR_ARG;

    rt_scene_desc& d = out.desc;
    std::memset(&d, 0, sizeof(d));
    d.n_textures = (int32_t)out.textures.size(); d.textures = out.textures.empty() ? nullptr : out.textures.data();
    d.n_materials = (int32_t)out.materials.size(); d.materials = out.materials.data();
    d.n_lights = (int32_t)out.lights.size(); d.lights = out.lights.data();
    d.n_squares = (int32_t)out.squares.size(); d.squares = out.squares.data();
    d.n_spheres = (int32_t)out.spheres.size(); d.spheres = out.spheres.data();
    d.n_meshes = (int32_t)out.meshes.size(); d.meshes = out.meshes.data();
    d.n_verts = (int64_t)out.positions.size() / 3; d.positions = out.positions.data(); d.normals = out.normals.data();
    d.n_tris = (int64_t)out.tris.size() / 3; d.tri_indices = out.tris.data(); d.mesh_items = out.items.data();
    d.n_mesh_nodes = (int64_t)out.nodes.size(); d.mesh_nodes = out.nodes.data();
    d.n_mesh_leaves = (int64_t)out.leaves.size(); d.mesh_leaves = out.leaves.data();
    d.n_top_nodes = (int32_t)out.top_nodes.size(); d.top_nodes = out.top_nodes.data();
    d.n_top_leaves = (int32_t)out.top_leaves.size(); d.top_leaves = out.top_leaves.data();
    d.n_top_items = (int32_t)out.top_items.size(); d.top_items = out.top_items.data();
    d.top_root = out.top_root; d.top_depth = out.top_depth;
    return RT_OK;
}

/* Camera (Camera.h) -> rt_camera: the trackball quaternion and the translation Camera::apply() would load into GL
 * (Camera.cpp:125-132), the projection of Camera::resize (Camera.cpp:41-50). */
template <class CameraT>
rt_camera rt_pack_camera(const CameraT& c) { return Access::camera(c); }

}  // namespace rtref
#endif
