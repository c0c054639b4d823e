/*
 * The reference-side binding (integration/rt_from_reference.h) against the host mirror: every named scene of rth::Scene is
 * rebuilt as objects with the reference's member names (ref_like.h: pointer trees of BVH_Node, BuildingMesh::ref into the
 * scene's own vectors, raw texel pointers in the materials), flattened by rtref::rt_flatten, and compared with
 * rth::Scene::flatten() array by array, byte for byte. Exit status 0 = all equal. No GPU is touched.
 */
#include "ref_like.h"
#include "../rt_from_reference.h"
#include "rthost.h"

#include <cstdio>
#include <cstring>
#include <string>

namespace reflike {
struct Filler {
    static std::unique_ptr<BVH_Node> tree(const rth::FlatTree& t, int32_t link, const AABB& box, bool top, Scene* sc) {
        auto n = std::make_unique<BVH_Node>();
        n->boundingBox = box;
        if (link < 0) {
            const rt_bvh_leaf& lf = t.leaves[(size_t)(-1 - link)];
            for (uint32_t i = 0; i < lf.count; i++) {
                const uint32_t it = t.items[lf.first + i];
                if (!top) { n->trianglesIdx.push_back(it); continue; }
                BuildingMesh bm;
                const size_t nsq = sc->squares.size(), nsp = sc->spheres.size();
                if (it < nsq) { bm.type = 0; bm.ref = &sc->squares[it]; }
                else if (it < nsq + nsp) { bm.type = 1; bm.ref = &sc->spheres[it - nsq]; }
                else { bm.type = 2; bm.ref = &sc->meshes[it - nsq - nsp]; }
                n->meshes.push_back(bm);
            }
            return n;
        }
        const rt_bvh_node& f = t.nodes[(size_t)link];
        AABB l, r;
        for (unsigned k = 0; k < 3; k++) { l.min[k] = f.lmin[k]; l.max[k] = f.lmax[k]; r.min[k] = f.rmin[k]; r.max[k] = f.rmax[k]; }
        n->leaf = false;
        n->l_child = tree(t, f.left, l, top, sc);
        n->r_child = tree(t, f.right, r, top, sc);
        return n;
    }
    static Material material(const rth::Material& m, rth::Scene& src) {
        Material r;
        for (unsigned k = 0; k < 3; k++) { r.ambient_material[k] = m.ambient_material[k]; r.diffuse_material[k] = m.diffuse_material[k]; r.specular_material[k] = m.specular_material[k]; }
        r.shininess = m.shininess; r.type = (MaterialType)m.type; r.index_medium = m.index_medium; r.transparency = m.transparency; r.reflectivity = m.reflectivity;
        r.materialScale = m.materialScale;
        if (m.texture >= 0) { rth::Image& im = src.textures[(size_t)m.texture]; r.hasTexture = true; r.textureData = im.data.data(); r.textureWidth = im.width; r.textureHeight = im.height; r.textureChannels = im.channels; }
        if (m.normalMap >= 0) { rth::Image& im = src.textures[(size_t)m.normalMap]; r.hasNormalMap = true; r.normalMapData = im.data.data(); r.normalMapWidth = im.width; r.normalMapHeight = im.height; r.normalMapChannels = im.channels; }
        return r;
    }
    static void fill(Scene& out, rth::Scene& src) {
        for (const rth::Light& l : src.lights) { Light r; for (unsigned k = 0; k < 3; k++) { r.pos[k] = l.pos[k]; r.material[k] = l.material[k]; } r.radius = l.radius; out.lights.push_back(r); }
        for (const rth::Square& s : src.squares) {
            Square r;
            r.vertices.resize(4);
            for (int j = 0; j < 4; j++) for (unsigned k = 0; k < 3; k++) r.vertices[(size_t)j].position[k] = s.vertices[j][k];
            r.material = material(s.material, src);
            out.squares.push_back(std::move(r));
        }
        for (const rth::Sphere& s : src.spheres) {
            Sphere r;
            for (unsigned k = 0; k < 3; k++) r.m_center[k] = s.m_center[k];
            r.m_radius = s.m_radius; r.material = material(s.material, src);
            out.spheres.push_back(std::move(r));
        }
        for (const rth::Mesh& m : src.meshes) {
            RayTraceMesh r;
            r.vertices.resize(m.positions.size());
            for (size_t i = 0; i < m.positions.size(); i++) for (unsigned k = 0; k < 3; k++) { r.vertices[i].position[k] = m.positions[i][k]; r.vertices[i].normal[k] = m.normals[i][k]; }
            r.triangles.resize(m.triangleCount());
            for (size_t t = 0; t < m.triangleCount(); t++) for (int k = 0; k < 3; k++) r.triangles[t].v[k] = m.triangles[3 * t + (size_t)k];
            r.material = material(m.material, src);
            AABB root;
            for (unsigned k = 0; k < 3; k++) { root.min[k] = m.triangleTree.rootBox.min[k]; root.max[k] = m.triangleTree.rootBox.max[k]; }
            r.set_tree(tree(m.triangleTree, m.triangleTree.root, root, false, nullptr));
            out.meshes.push_back(std::move(r));
        }
        AABB root;
        for (unsigned k = 0; k < 3; k++) { root.min[k] = src.sceneBvh.rootBox.min[k]; root.max[k] = src.sceneBvh.rootBox.max[k]; }
        if (!src.sceneBvh.leaves.empty() || !src.sceneBvh.nodes.empty()) out.sceneBvh = std::move(*tree(src.sceneBvh, src.sceneBvh.root, root, true, &out));
    }
    static void camera(Camera& c, const rth::Camera& s) {
        c.fovAngle = s.fovAngle; c.aspectRatio = s.aspectRatio; c.nearPlane = s.nearPlane; c.farPlane = s.farPlane;
        for (int i = 0; i < 4; i++) c.curquat[i] = s.curquat[i];
        c.x = s.x; c.y = s.y; c.z = s.z; c._zoom = s._zoom;
    }
};
}  // namespace reflike

template <class T>
static bool same(const char* what, const T* a, const T* b, size_t n, const std::string& scene) {
    if (n == 0 || std::memcmp(a, b, n * sizeof(T)) == 0) return true;
    std::fprintf(stderr, "%s: %s differs\n", scene.c_str(), what);
    return false;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: test_flatten <asset dir> <image dir>\n"); return 2; }
    rth::set_asset_dir(argv[1]);
    rth::set_image_dir(argv[2]);
    const char* names[] = {"final", "refraction", "reflexion", "reflexion_chamber", "cornell_box", "mesh_gorilla", "mesh_cow", "stage_blob_flat", "soup:5000"};
    int failures = 0;
    for (const char* name : names) {
        rth::Scene src;
        if (!src.setup(name)) { std::fprintf(stderr, "%s: setup failed (asset directories?)\n", name); return 2; }
        rth::Settings set;
        set.tree_subdivide = 12;
        src.applySettings(set);
        src.buildSceneBvh();
        const rt_scene_desc& want = src.flatten();
        reflike::Scene ref;
        reflike::Filler::fill(ref, src);
        rtref::Flattened got;
        const int rc = rtref::rt_flatten(ref, got);
        if (rc != RT_OK) { std::fprintf(stderr, "%s: rt_flatten returned %d\n", name, rc); failures++; continue; }
        const rt_scene_desc& d = got.desc;
        bool ok = d.n_materials == want.n_materials && d.n_lights == want.n_lights && d.n_squares == want.n_squares && d.n_spheres == want.n_spheres &&
                  d.n_meshes == want.n_meshes && d.n_verts == want.n_verts && d.n_tris == want.n_tris && d.n_mesh_nodes == want.n_mesh_nodes &&
                  d.n_mesh_leaves == want.n_mesh_leaves && d.n_top_nodes == want.n_top_nodes && d.n_top_leaves == want.n_top_leaves &&
                  d.n_top_items == want.n_top_items && d.top_root == want.top_root && d.top_depth == want.top_depth && d.n_textures == want.n_textures;
        if (!ok) { std::fprintf(stderr, "%s: counts differ\n", name); failures++; continue; }
        const std::string s = name;
        ok = same("lights", d.lights, want.lights, (size_t)d.n_lights, s) & same("squares", d.squares, want.squares, (size_t)d.n_squares, s) &
             same("spheres", d.spheres, want.spheres, (size_t)d.n_spheres, s) & same("meshes", d.meshes, want.meshes, (size_t)d.n_meshes, s) &
             same("positions", d.positions, want.positions, 3 * (size_t)d.n_verts, s) & same("normals", d.normals, want.normals, 3 * (size_t)d.n_verts, s) &
             same("tri_indices", d.tri_indices, want.tri_indices, 3 * (size_t)d.n_tris, s) & same("mesh_items", d.mesh_items, want.mesh_items, (size_t)d.n_tris, s) &
             same("mesh_nodes", d.mesh_nodes, want.mesh_nodes, (size_t)d.n_mesh_nodes, s) & same("mesh_leaves", d.mesh_leaves, want.mesh_leaves, (size_t)d.n_mesh_leaves, s) &
             same("top_nodes", d.top_nodes, want.top_nodes, (size_t)d.n_top_nodes, s) & same("top_leaves", d.top_leaves, want.top_leaves, (size_t)d.n_top_leaves, s) &
             same("top_items", d.top_items, want.top_items, (size_t)d.n_top_items, s);
        /* materials: every field equal; the image a material names must be the same image (the binding numbers images in
         * the order it meets them, the mirror in the order its setup loaded them) */
        for (int i = 0; i < d.n_materials && ok; i++) {
            rt_material a = d.materials[i], b = want.materials[i];
            auto texels = [](const rt_scene_desc& sc, int32_t idx) { return idx ? sc.textures[idx - 1].texels : nullptr; };
            if (texels(d, a.texture) != texels(want, b.texture) || texels(d, a.normal_map) != texels(want, b.normal_map)) { std::fprintf(stderr, "%s: material %d names another image\n", name, i); ok = false; }
            a.texture = b.texture = 0; a.normal_map = b.normal_map = 0;
            if (std::memcmp(&a, &b, sizeof(a)) != 0) { std::fprintf(stderr, "%s: material %d differs\n", name, i); ok = false; }
        }
        std::printf("%-20s %s  (%d materials, %d squares, %d spheres, %d meshes, %lld triangles, %lld + %d nodes, %d images)\n", name, ok ? "same bytes" : "DIFFERS",
                    d.n_materials, d.n_squares, d.n_spheres, d.n_meshes, (long long)d.n_tris, (long long)d.n_mesh_nodes, d.n_top_nodes, d.n_textures);
        if (!ok) failures++;
    }
    {   /* camera */
        rth::Camera c;
        c.move(0.f, 0.f, -3.1f); c.resize(1920, 1080); c.curquat[0] = 0.06805f; c.curquat[1] = 0.20361f; c.curquat[2] = -0.07876f; c.curquat[3] = 0.9735f;
        reflike::Camera rc;
        reflike::Filler::camera(rc, c);
        const rt_camera a = rtref::rt_pack_camera(rc), b = c.pack();
        const bool ok = std::memcmp(&a, &b, sizeof(a)) == 0;
        std::printf("%-20s %s\n", "camera", ok ? "same bytes" : "DIFFERS");
        if (!ok) failures++;
    }
    return failures ? 1 : 0;
}
