/*
 * rt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the
 * per-pixel render loop of Crhonopost/Ray-tracing (reference @ 2025-01-17).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product path (ray-tracing_amd/) never links or calls it.
 *
 * PARITY UNPINNED: every reference file on this path transitively includes
 * <GL/glut.h> (Material.h:7, Mesh.h:13, Scene.h:12), which this image lacks,
 * and the reference holds no tests / golden vectors.  The restatement is
 * pinned only by (a) the GL-free reference sources compiled unmodified into
 * oracle/_ref (Vec3.h, Line.h, Trackball.cpp) and (b) the known answers
 * recorded from the reference in SURVEY.md (scene-setup values, BVH
 * statistics, primary-hit ordinal, ray statistics).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Material record, Material.h:18-47 (only the fields the hot path reads). */
typedef struct {
    float ka[3], kd[3], ks[3];
    float pad0;
    double shininess;     /* Material.h:22 (double); 0.0 where the reference leaves it uninitialised */
    int32_t type;         /* Material.h:10-15: 0 diffuse, 1 glass, 2 mirror, 3 checker */
    float ior;            /* index_medium */
    float transparency;
    float reflectivity;
    int32_t texture;      /* 0 = none (hasTexture false), else 1 + index of an image added with orc_add_texture (Material::textureData) */
    int32_t normal_map;   /* same for Material::normalMapData / hasNormalMap */
    double material_scale;/* Material.h:31 materialScale; 0.0 is read as the reference's default 1.0 */
} orc_material;

/* Camera fields of Camera.h / Camera.cpp:24-38; the oracle turns them into the
 * matrices Camera::apply + gluPerspective would load (Camera.cpp:41-50,125-132)
 * and unprojects like matrixUtilities.h:188-222. */
typedef struct {
    float x, y, z;        /* Camera::move accumulations */
    float zoom;           /* _zoom */
    float quat[4];        /* curquat */
    float fovy_deg, aspect, znear, zfar;
} orc_camera;

typedef struct {
    int32_t img_w, img_h;
    int32_t spp;             /* main.cpp:189 (30) */
    int32_t max_depth;       /* Scene.h:284 (50) */
    int32_t light_samples;   /* Scene.h:214 (35) */
    int32_t shadow_mode;     /* 0 = soft (sampleSphereLight), 1 = hard (one isInShadow query per light, Scene.h:215) */
    int32_t jitter;          /* 0 = centre of pixel, 1 = counter-RNG jitter (replaces rand(), main.cpp:195-196) */
    uint32_t seed;
} orc_params;

typedef struct {
    uint64_t rays_primary, rays_secondary, rays_shadow;
    uint64_t box_tests, tri_tests, node_visits;          /* all rays */
    uint64_t sh_box_tests, sh_tri_tests, sh_node_visits; /* shadow rays only */
} orc_stats;

typedef struct {
    int32_t exists;
    int32_t obj;      /* ordinal: squares, then spheres, then meshes (AccelerationStruct.cpp:123-134); -1 = miss */
    int32_t tri;      /* index into Mesh::triangles; 0 for spheres/squares */
    float t;
    float pos[3];
    float nrm[3];
    float u, v;       /* w0,w1 for meshes (Mesh.cpp:133-134), u,v for squares, Sphere.h:141-142 for spheres (narrowed to float as Scene.h:184 does) */
} orc_hit;

orc_scene* orc_scene_new(void);
void orc_scene_free(orc_scene*);

/* Directory that stands for the reference's "assets/models/" (OFF files). */
void orc_set_asset_dir(const char* dir);
/* Mesh::loadOFF (Mesh.cpp:5-23) on a file; returns 0 and counts, fills caller arrays if non-null. */
int orc_load_off(const char* path, float* verts, int max_v, uint32_t* tris, int max_t, int* nv, int* nt);

/* Directory that stands for the reference's "assets/img/" (normalMaps/, sphereTextures/ below it). */
void orc_set_image_dir(const char* dir);
/* An image as stbi_load returns it (Material.h:128-138): rows top to bottom, `channels` (>= 3) bytes per texel.
 * Returns its index for orc_material::texture / normal_map, or -1. The data is copied. */
int orc_add_texture(orc_scene*, int width, int height, int channels, const uint8_t* texels);
/* Size and (if texels is non-null) bytes of image i. */
int orc_get_texture(const orc_scene*, int i, int* width, int* height, int* channels, uint8_t* texels);
/* Binary PPM (P6, maxval 255) as the reference's assets are; returns 0 and the size, fills texels (3 bytes each) if non-null. */
int orc_load_ppm(const char* path, int* width, int* height, uint8_t* texels, int64_t max_bytes);

/* Named scenes: "final", "refraction", "reflexion", "reflexion_chamber", "mesh_gorilla", "mesh_cow", "soup:<ntris>". */
int orc_scene_setup(orc_scene*, const char* name);

/* Free-form scenes (already-transformed geometry). */
int orc_add_light(orc_scene*, const float pos[3], float radius, const float color[3]);
int orc_add_square(orc_scene*, const float verts[12], const orc_material*);
int orc_add_sphere(orc_scene*, const float center[3], float radius, const orc_material*);
int orc_add_mesh(orc_scene*, const float* verts, int nv, const uint32_t* tris, int nt, const orc_material*);

/* BVH_Node::buildBVH (top level, deepLimit) + Scene::applySettings (mesh depth). */
int orc_build(orc_scene*, int top_depth, int mesh_depth);

/* Introspection (for tree / scene parity). */
int orc_counts(const orc_scene*, int* n_squares, int* n_spheres, int* n_meshes, int* n_lights);
int orc_get_square(const orc_scene*, int i, float verts[12], orc_material*);
int orc_get_sphere(const orc_scene*, int i, float center[3], float* radius, orc_material*);
int orc_get_light(const orc_scene*, int i, float pos[3], float* radius, float color[3]);
int orc_get_mesh_counts(const orc_scene*, int i, int64_t* nv, int64_t* nt);
int orc_get_mesh(const orc_scene*, int i, float* verts, float* normals, uint32_t* tris, orc_material*);
/* Tree dump, DFS pre-order (left first). Per node: leaf flag, member count (0 for inner), box (6 floats).
 * which = -1 top level, else mesh index. items receives leaf members in order
 * (object ordinals for the top level, triangle indices for a mesh). Returns node count
 * (call with NULLs to size). */
int64_t orc_dump_tree(const orc_scene*, int which, int32_t* leaf, int32_t* count, float* boxes,
                      int64_t max_nodes, int64_t* items, int64_t max_items, int64_t* n_items);

/* Queries. */
void orc_intersect(const orc_scene*, const float* org, const float* dir, int64_t n, orc_hit* out);
void orc_shadow(const orc_scene*, const float* pts, const float* light_pos, int64_t n, int32_t* occluded);
/* rayTrace on explicit rays; pixel/sample key the light-sampling RNG. */
void orc_trace(const orc_scene*, const float* org, const float* dir, int64_t n, const orc_params*,
               const uint32_t* pixel, const uint32_t* sample, float* rgb);
void orc_camera_rays(const orc_camera*, const orc_params*, int x0, int y0, int w, int h,
                     float* org, float* dir); /* w*h*spp rays, sample-minor */
/* Full tile render on n_threads (ThreadPool-equivalent, one task per pixel). rgb = w*h*3 mean radiance;
 * prim = w*h*2 (obj, tri) of sample 0's primary hit (may be NULL). Returns seconds spent tracing. */
double orc_render(const orc_scene*, const orc_camera*, const orc_params*, int x0, int y0, int w, int h,
                  float* rgb, int32_t* prim, orc_stats* stats, int n_threads);
/* main.cpp:253 quantisation. */
void orc_quantize(const float* rgb, int64_t n, int32_t* out);
/* build_rotmatrix (Trackball.cpp:323-345), row-major float[16] as the C array m[4][4]. */
void orc_rotmatrix(const float q[4], float m[16]);

/* Counter RNG shared by definition with the GPU path (north_star: same RNG on both sides). */
uint32_t orc_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint64_t code, uint32_t light, uint32_t i, uint32_t which);

#ifdef __cplusplus
}
#endif
#endif
