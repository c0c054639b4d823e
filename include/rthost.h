/*
 * rthost.h -- host side of the MI355X render loop: a mirror of the reference's
 * Scene / Camera / Mesh / Material types and of its render entry point, above the
 * rtgpu.h C-ABI.  The reference's toolchain is C++, so the host side is C++ with the
 * reference's names (Mesh::loadOFF, Scene::setup_final_scene, Scene::applySettings,
 * BVH_Node::buildBVH, ray_trace_from_camera ...).  A C facade (rth_*) at the bottom
 * lets tests and bench.py drive it through ctypes.
 */
#ifndef RTHOST_H
#define RTHOST_H
#include "rtgpu.h"

#ifdef __cplusplus
#include <cstdint>
#include <string>
#include <vector>

namespace rth {

/* Vec3.h:12-119 */
struct Vec3 {
    float v[3];
    Vec3() : v{0.f, 0.f, 0.f} {}
    explicit Vec3(float s) : v{s, s, s} {}
    Vec3(float x, float y, float z) : v{x, y, z} {}
    float& operator[](unsigned i) { return v[i]; }
    float operator[](unsigned i) const { return v[i]; }
    float squareLength() const;
    float length() const;
    void normalize();
    static float dot(const Vec3& a, const Vec3& b);
    static Vec3 cross(const Vec3& a, const Vec3& b);
};
Vec3 operator+(const Vec3& a, const Vec3& b);
Vec3 operator-(const Vec3& a, const Vec3& b);
Vec3 operator*(float s, const Vec3& a);
Vec3 operator*(const Vec3& a, float s);
Vec3 operator/(const Vec3& a, float s);

/* Vec3.h:121-281, row-major 3x3 */
struct Mat3 {
    float m[9];
    Vec3 operator*(const Vec3& p) const;
};

/* Material.h:18-47 */
struct Material {
    Vec3 ambient_material, diffuse_material, specular_material;
    double shininess = 0.0;   /* unset in the reference's ctor; defined 0.0 here */
    int type = RT_MAT_DIFFUSE;
    float index_medium = 1.0f, transparency = 0.0f, reflectivity = 0.0f;
    /* Material.h:24-31: the images live in Scene::textures here (the reference keeps raw stbi pointers per material) */
    int texture = -1, normalMap = -1;                  /* hasTexture / hasNormalMap = index >= 0 */
    double materialScale = 1.;
};
struct Image { int width = 0, height = 0, channels = 0; std::vector<uint8_t> data; };   /* stbi_load result */
/* Binary PPM (P6, maxval 255), the format of the reference's .ppm textures and normal maps. */
bool load_ppm(const std::string& filename, Image& out);
bool load_png(const std::string& filename, Image& out);     /* 8-bit, non-interlaced grey / grey+alpha / RGB / RGBA */
bool load_image(const std::string& filename, Image& out);   /* by extension: .ppm, .png; x.jpg reads the decoded x.ppm beside it */

struct AABB {                 /* AccelerationStruct.h:11-70 */
    Vec3 min, max;
    AABB();
    AABB(const Vec3& mn, const Vec3& mx) : min(mn), max(mx) {}
    Vec3 getCenter() const;
};

/* One flattened BVH (inner nodes with both child boxes, leaves, leaf items). */
/* thrown by the tree builders that can fail (the GPU builder: no device, tree too deep); the C facade turns it into its code */
struct BuildError { int code; std::string message; };

struct FlatTree {
    std::vector<rt_bvh_node> nodes;
    std::vector<rt_bvh_leaf> leaves;
    std::vector<uint32_t> items;
    int32_t root = -1;        /* link */
    int32_t depth = 0;
    AABB rootBox;
};

/* Mesh.h:68-207 + RayTraceMesh Mesh.h:210-227 */
class Mesh {
public:
    std::vector<Vec3> positions, normals;
    std::vector<uint32_t> triangles;   /* 3 per triangle */
    Material material;
    FlatTree triangleTree;

    bool loadOFF(const std::string& filename);          /* Mesh.cpp:5-23 */
    bool loadOFFColoured(const std::string& filename);  /* OFF and COFF (per-vertex colour columns are skipped), faces of any size fanned into triangles; scene files only */
    void recomputeNormals();                            /* Mesh.cpp:25-38 */
    void centerAndScaleToUnit();                        /* Mesh.cpp:40-53 */
    void translate(const Vec3& t);                      /* Mesh.h:128-132 */
    void apply_transformation_matrix(const Mat3& t);    /* Mesh.h:134-137 */
    void scale(const Vec3& s);                          /* Mesh.h:144-149 */
    void rotate_x(float angle);                         /* Mesh.h:151-157 */
    void rotate_y(float angle);                         /* Mesh.h:159-165 */
    void rotate_z(float angle);                         /* Mesh.h:167-173 */
    void unweld();                                      /* every triangle gets vertices of its own (face normals after recomputeNormals); no counterpart in the reference */
    void build_arrays() { recomputeNormals(); }         /* Mesh.h:119-125 (GL arrays not needed) */
    void buildTree(unsigned nb_of_subdivide_tree, int tree_mode = 0);      /* Mesh.cpp:144-179; tree_mode: Settings::tree_mode */
    AABB getBoundingBox() const { return triangleTree.rootBox; }   /* Mesh.cpp:181-183 */
    size_t triangleCount() const { return triangles.size() / 3; }
};

struct Square {               /* Square.h:18-125: the four transformed vertices */
    Vec3 vertices[4];
    Material material;
    void setQuad(const Vec3& bottomLeft, const Vec3& rightVector, const Vec3& upVector, float width, float height);
    void translate(const Vec3& t);
    void scale(const Vec3& s);
    void rotate_x(float angle);
    void rotate_y(float angle);
    AABB getBoundingBox() const;                        /* Square.h:113-123 */
};
struct Sphere {               /* Sphere.h:37-43 */
    Vec3 m_center;
    float m_radius = 0.f;
    Material material;
    AABB getBoundingBox() const;                        /* Sphere.h:148-151 */
};
struct Light {                /* Scene.h:20-34 */
    Vec3 material{1.f, 1.f, 1.f};
    Vec3 pos;
    float radius = 0.f;
};
struct Settings {                                       /* Scene.h:37-39, main.cpp:116 */
    unsigned tree_subdivide = 12;
    /* not in the reference: 0 = the reference's spatial-median builder (AccelerationStruct.cpp:151-195), the default and the
     * only mode with parity; 1 = binned-SAH builder with free topology (tree_subdivide is still the depth limit): a few grazing
     * pixels may change their hit identity, see tests/test_sah_mode.py */
    int tree_mode = 0;
};

/* Camera.h / Camera.cpp:24-38 */
struct Camera {
    float fovAngle = 45.0f, aspectRatio = 1.0f, nearPlane = 4.1f, farPlane = 10000.0f;
    float x = 0.f, y = 0.f, z = 0.f, _zoom = 3.0f;
    float curquat[4] = {0.f, 0.f, 0.f, 1.f};
    void resize(int W, int H) { aspectRatio = static_cast<float>(W) / static_cast<float>(H); }   /* Camera.cpp:41-50 */
    void move(float dx, float dy, float dz) { x += dx; y += dy; z += dz; }                       /* Camera.cpp:85-89 */
    void zoom(float dz) { _zoom += dz; }                                                          /* Camera.cpp:120-122 */
    rt_camera pack() const;
};

/* Scene.h:42-971 */
class Scene {
public:
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Square> squares;
    std::vector<Light> lights;
    std::vector<Image> textures;                        /* Material::texture / normalMap index into this */
    FlatTree sceneBvh;
    int topDepthLimit = 3;                              /* AccelerationStruct.h:121 */

    void clear();
    void applySettings(Settings set);                   /* Scene.h:54-58 */
    void buildSceneBvh();                               /* sceneBvh = BVH_Node::buildBVH(squares, spheres, meshes) */
    bool setup_refraction_scene();                      /* Scene.h:468-552 */
    bool setup_reflexion_scene();                       /* Scene.h:554-599 */
    bool setup_reflexion_chamber();                     /* Scene.h:601-714 */
    bool setup_cornell_box();                           /* Scene.h:288-466 */
    bool setup_final_scene();                           /* Scene.h:800-970 */
    /* Material::loadTexture / loadNormalMap (Material.h:128-138) for files under the image directory; -1 on failure */
    int loadImage(const std::string& relative_path);
    int addImage(int width, int height, int channels, const uint8_t* texels);
    bool setup_mesh_scene(const std::string& off, bool unit);   /* Scene.h:716-781 (gorilla block) */
    bool setup_stage_scene(const std::string& off, bool flat, float scale, float rotx, const Vec3& move);   /* the scenes of the stage images with.ppm, without.ppm and gorilla_1.ppm (scene.cpp) */
    bool setup_soup_scene(int64_t ntris);               /* BASELINE config 5 */
    bool setup(const std::string& name);

    /* Scene description files (data-driven scenes instead of the hard-coded setup_* tables; the
     * reference has no such format). Text lines -- light / texture / square / sphere / mesh with the material inline --
     * where a mesh is either an OFF file plus the transformations of Mesh.h:128-173 (mesh_off, center_unit, scale,
     * rotate_x/y/z, translate), or a binary block of final positions and triangles in the file itself (the mesh
     * cache for large inputs; textures likewise). save() writes the second form and reproduces the scene exactly:
     * load(save(s)) flattens to the same bytes as s. load() returns false and sets *why on malformed input. */
    bool save(const std::string& filename) const;
    bool load(const std::string& filename, std::string* why = nullptr);

    /* Flatten into the rt_scene_desc layout; the returned desc points into this object. */
    const rt_scene_desc& flatten();

private:
    std::vector<rt_material> f_materials;
    std::vector<rt_light> f_lights;
    std::vector<rt_square> f_squares;
    std::vector<rt_sphere> f_spheres;
    std::vector<rt_mesh> f_meshes;
    std::vector<float> f_positions, f_normals;
    std::vector<uint32_t> f_tris, f_items;
    std::vector<rt_bvh_node> f_nodes;
    std::vector<rt_bvh_leaf> f_leaves;
    std::vector<rt_texture> f_textures;
    rt_scene_desc f_desc;
};

void set_asset_dir(const std::string& dir);
void set_image_dir(const std::string& dir);            /* stands for the reference's "assets/img/" */

/* BVH_Node::buildBVH for triangles (AccelerationStruct.cpp:91-113,151-195). */
void buildBVH(const std::vector<Vec3>& positions, const std::vector<uint32_t>& triangles, int deepLimit, FlatTree& out, int tree_mode = 0);   /* tree_mode 1: binned SAH, 2: built on the GPU (throws BuildError if that fails); neither is the reference's tree */
/* BVH_Node::buildBVH for objects (AccelerationStruct.cpp:115-148,197-243). */
void buildBVH(const std::vector<AABB>& objectBoxes, int deepLimit, FlatTree& out);

/* Drop-in for ray_trace_from_camera() (main.cpp:180-258): renders scene through ctx in
 * 32x32 tiles and writes the P3 file exactly as main.cpp:245-255. Returns rt_status. */
int ray_trace_from_camera(rt_ctx* ctx, Scene& scene, const Camera& camera, int w, int h,
                          const rt_render_params* params_or_null, const char* filename, double* seconds);
int write_ppm(const char* filename, const float* rgb, int w, int h);   /* main.cpp:245-255 */

}  // namespace rth
extern "C" {
#endif

/* ------------------------------------------------------------------ C facade */
typedef struct rth_scene rth_scene;
rth_scene* rth_scene_new(void);
void rth_scene_free(rth_scene*);
void rth_set_asset_dir(const char* dir);
void rth_set_image_dir(const char* dir);
/* Returns the value for rt_material::texture / normal_map (1 + index), or a negative rt_status. The data is copied. */
int rth_add_texture(rth_scene*, int width, int height, int channels, const uint8_t* texels);
/* "final", "refraction", "reflexion", "mesh_gorilla", "mesh_cow", "soup:<ntris>" */
int rth_scene_setup(rth_scene*, const char* name);
int rth_add_light(rth_scene*, const float pos[3], float radius, const float color[3]);
int rth_add_square(rth_scene*, const float verts[12], const rt_material*);
int rth_add_sphere(rth_scene*, const float center[3], float radius, const rt_material*);
int rth_add_mesh(rth_scene*, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt, const rt_material*);
/* Scene::applySettings (mesh trees) + BVH_Node::buildBVH (top level) + flatten. */
int rth_build(rth_scene*, int top_depth, int mesh_depth);
/* ... with Settings::tree_mode for the mesh trees (0 = the reference's builder; 1 = binned SAH on the host, 2 = linear BVH built on the GPU
 * (lbvh.hip; needs a GPU, RT_ERR_DEVICE without one, RT_ERR_LIMIT if the tree comes out deeper than RT_MAX_TREE_DEPTH): free topology,
 * no parity) */
int rth_build_mode(rth_scene*, int top_depth, int mesh_depth, int tree_mode);
const rt_scene_desc* rth_scene_desc(rth_scene*);
int rth_ray_trace_from_camera(rt_ctx*, rth_scene*, const rt_camera*, const rt_render_params*, const char* ppm_path, double* seconds);
int rth_write_ppm(const char* path, const float* rgb, int w, int h);
/* Scene::save / Scene::load (scene description files, see rthost.h); load replaces the scene's contents, rth_build comes after. */
int rth_scene_save(rth_scene*, const char* path);
int rth_scene_load(rth_scene*, const char* path);
int rth_load_off(const char* path, float* verts, int64_t max_v, uint32_t* tris, int64_t max_t, int64_t* nv, int64_t* nt);

#ifdef __cplusplus
}
#endif
#endif
