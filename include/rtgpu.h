/*
 * rtgpu.h -- C-ABI of the MI355X render-loop library (librtgpu.so).
 *
 * Drop-in boundary for the ThreadPool CPU path of Crhonopost/Ray-tracing:
 * the body of ray_trace_from_camera() (main.cpp:180-258: ray generation,
 * per-pixel ThreadPool tasks, Scene::rayTrace, averaging) is replaced by
 *     rt_upload_scene -> rt_set_camera -> rt_begin_frame -> rt_render_tile* ->
 *     rt_download_framebuffer.
 * The reference has no FFI of its own; each entry point cites the
 * reference code it stands for. Plain pointers and sizes only; every function
 * returns 0 on success or a negative rt_status; no exception crosses the ABI.
 * A context is not thread-safe (one host thread drives it; with several GPUs it starts a worker thread per device itself).
 */
#ifndef RTGPU_H
#define RTGPU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum rt_status {
    RT_OK = 0,
    RT_ERR_ARG = -1,        /* null / out-of-range argument */
    RT_ERR_STATE = -2,      /* call order (no scene, no camera, no frame) */
    RT_ERR_DEVICE = -3,     /* HIP runtime error; see rt_last_error */
    RT_ERR_LIMIT = -4,      /* a documented limit is exceeded (tree depth, max_depth, counts) */
    RT_ERR_NOMEM = -5
};

/* Material.h:10-15 */
enum rt_material_type { RT_MAT_DIFFUSE = 0, RT_MAT_GLASS = 1, RT_MAT_MIRROR = 2, RT_MAT_CHECKER = 3 };

/* Material.h:18-47, the fields Scene::rayTraceRecursive reads (Scene.h:174-280). */
typedef struct {
    float ka[3], kd[3], ks[3];
    float pad0;
    double shininess;       /* Material.h:22 is a double; set 0.0 where the reference leaves it unset */
    int32_t type;           /* rt_material_type */
    float ior;              /* index_medium */
    float transparency;
    float reflectivity;
    /* Image texture and normal map (Material.h:24-31,86-138; applied at Scene.h:187-202). 0 = none (hasTexture /
     * hasNormalMap false), otherwise 1 + index into rt_scene_desc::textures, so a zeroed struct means "no images". */
    int32_t texture;
    int32_t normal_map;
    double material_scale;  /* materialScale (Material.h:31); 0.0 is read as the reference's default 1.0 */
} rt_material;

/* An image as stbi_load returns it (Material.h:128-138): rows top to bottom, `channels` (>= 3) bytes per texel,
 * the first three are read as r, g, b (Material.h:92-93,102-103). Decoding files is the caller's business
 * (the reference uses stb_image; include/rthost.h reads the binary PPMs the reference's scenes use). */
typedef struct { int32_t width, height, channels; int32_t pad0; const uint8_t* texels; } rt_texture;

/* Scene.h:20-34 (Light): spherical area light. */
typedef struct { float pos[3]; float radius; float color[3]; float pad0; } rt_light;

/* Square.h:62-71 reads vertices[0], [1], [3] of the transformed quad. */
typedef struct { float v0[3], v1[3], v3[3]; int32_t material; } rt_square;
/* Sphere.h:39-40 */
typedef struct { float center[3]; float radius; int32_t material; int32_t pad0[3]; } rt_sphere;

/* Flattened BVH_Node (AccelerationStruct.h:97-143): one 64-byte record per INNER node
 * holding both child boxes. Child link >= 0: inner node index (local to its tree);
 * link < 0: leaf number -1-link (local to its tree). */
typedef struct {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad0[2];
} rt_bvh_node;
typedef struct { uint32_t first, count; } rt_bvh_leaf;   /* range of the tree's item array */

/* RayTraceMesh (Mesh.h:210-227): ranges into the scene-wide arrays. */
typedef struct {
    int64_t vert_first, vert_count;     /* positions / normals (Mesh.h:26-42) */
    int64_t tri_first, tri_count;       /* vertex-index triples (Mesh.h:44-63), indices local to the mesh */
    int64_t node_first, node_count;     /* rt_bvh_node */
    int64_t leaf_first, leaf_count;     /* rt_bvh_leaf; leaf item ranges are local to [tri_first, +tri_count) of mesh_items */
    int32_t root;                       /* link of the root (a leaf link when the tree is a single leaf) */
    int32_t material;
    int32_t depth;                      /* tree depth (<= RT_MAX_TREE_DEPTH) */
    int32_t pad0;
} rt_mesh;

#define RT_MAX_TREE_DEPTH 40
#define RT_MAX_TOP_DEPTH 16
#define RT_MAX_LIGHTS 16
#define RT_MAX_PATH_DEPTH 60

/* Flat description of a Scene (Scene.h:42-51) after Scene::applySettings (Scene.h:54-58)
 * and BVH_Node::buildBVH (AccelerationStruct.cpp:91-148). The library copies everything;
 * the caller keeps ownership. Object ordinals count squares, then spheres, then meshes
 * (AccelerationStruct.cpp:123-134). */
typedef struct {
    int32_t n_materials;  const rt_material* materials;
    int32_t n_lights;     const rt_light* lights;
    int32_t n_squares;    const rt_square* squares;
    int32_t n_spheres;    const rt_sphere* spheres;
    int32_t n_meshes;     const rt_mesh* meshes;
    int64_t n_verts;      const float* positions;  const float* normals;     /* 3 floats each */
    int64_t n_tris;       const uint32_t* tri_indices;                       /* 3 per triangle */
    const uint32_t* mesh_items;      /* n_tris entries: per mesh, triangle indices (local) in leaf order */
    int64_t n_mesh_nodes; const rt_bvh_node* mesh_nodes;
    int64_t n_mesh_leaves; const rt_bvh_leaf* mesh_leaves;
    int32_t n_top_nodes;  const rt_bvh_node* top_nodes;
    int32_t n_top_leaves; const rt_bvh_leaf* top_leaves;
    int32_t n_top_items;  const uint32_t* top_items;                         /* object ordinals in leaf order */
    int32_t top_root;     int32_t top_depth;
    int32_t n_textures;   const rt_texture* textures;                        /* may be 0 / NULL */
} rt_scene_desc;

/* Camera fields (Camera.h, Camera.cpp:24-38). rt_set_camera derives what Camera::apply
 * (Camera.cpp:125-132), Camera::resize/gluPerspective (Camera.cpp:41-50) and the
 * unprojection of matrixUtilities.h:188-222 compute through the GL matrix stacks. */
typedef struct {
    float x, y, z;          /* Camera::move */
    float zoom;             /* _zoom */
    float quat[4];          /* curquat -> build_rotmatrix, Trackball.cpp:323-345 */
    float fovy_deg, aspect, znear, zfar;
} rt_camera;

/* The literals of the reference's render loop as parameters. */
typedef struct {
    int32_t img_w, img_h;       /* glutGet(GLUT_WINDOW_WIDTH/HEIGHT), main.cpp:184 */
    int32_t spp;                /* nsamples, main.cpp:189 (30) */
    int32_t max_depth;          /* Scene.h:284 (50) */
    int32_t light_samples;      /* Scene.h:214 (35) */
    int32_t shadow_mode;        /* 0 soft = sampleSphereLight (Scene.h:214); 1 hard = isInShadow per light (Scene.h:215) */
    int32_t jitter;             /* 0 pixel centre; 1 counter-RNG jitter in place of rand() (main.cpp:195-196) */
    uint32_t seed;
} rt_render_params;

typedef struct {
    uint64_t rays_primary, rays_secondary, rays_shadow;   /* one ray = one BVH_Node::intersectObjects query (rays_shadow: answered; see rays_shadow_grouped) */
    uint64_t nonfinite;         /* radiance contributions that were NaN/Inf (dropped from the sum) */
    uint64_t iterations;        /* wavefront iterations */
    uint64_t rays_shadow_culled; /* shadow rays NOT traced: their hit's whole Phong term, weighted by the path's throughput, lies below half a
                                  * unit of the frame's fixed-point accumulators whatever the shadow factor, so the frame's bits do not depend
                                  * on them (depth-50 glass recursions end in billions of such hits). Scene::isInShadow calls of the CPU path
                                  * = rays_shadow + rays_shadow_culled. RT_NO_CULL=1 traces them all. */
    double ms_total;            /* device time of the frame (HIP events) */
    double ms_closest;          /* sum over launches of the closest-hit+shade kernel */
    double ms_shadow;           /* sum over launches of the shadow kernel */
    uint64_t launches_closest, launches_shadow;
    uint64_t peak_states;       /* high-water mark of the path-state stack */
    uint64_t rays_shadow_grouped; /* of rays_shadow: shadow queries answered without a ray of their own -- all 35 rays of a (hit, light) pair
                                  * stay inside a swept cube from the hit point to the light's sphere that is clear of every object's box, so
                                  * Scene::isInShadow is false for each of them (k_shadow_group). The answers are the reference's; the rays were
                                  * not traced: throughput figures count rays_shadow - rays_shadow_grouped. RT_NO_GROUP_CULL=1 traces them all. */
} rt_stats;

/* One closest-hit result, RayIntersection (Ray.h:13-24) + identity. */
typedef struct {
    int32_t exists;
    int32_t obj;            /* object ordinal, -1 = miss */
    int32_t tri;            /* index into Mesh::triangles (0 for spheres/squares) */
    float t;
    float pos[3];
    float nrm[3];
    float u, v;
} rt_hit;

/* Creates a context on the n_devices GPUs of device_ids (1 <= n_devices <= 64; the ThreadPool of main.cpp:204-205, as GPUs).
 * One device: everything runs there. Several: the scene is replicated, rt_render_tile deals the tiles (round-robin along a
 * tile row with a hashed offset per row), rt_flush drives every device from a host thread of its own, and the frame is
 * gathered on device_ids[0] -- rt_resolve_device's buffer lives there: every device packs the pixels of its tiles into a slab
 * on its own stream, device_ids[0] pulls all slabs behind their events (hipMemcpyPeerAsync; rt_init enables peer access and
 * says so on stderr where it cannot, the copies then stage through host memory) and scatters them with one kernel. The
 * gather covers the tiles of the whole FRAME (all passes since rt_begin_frame), so a progressive pass over some tiles
 * only, or a resolve right after rt_continue_frame, still returns what the other devices rendered earlier.
 * The frame has the bits of a one-device render. rt_intersect / rt_occluded / rt_trace* run on device_ids[0].
 * (Scaling across processes -- one process per GPU, RCCL -- is ray-tracing_amd/dist.py + bench.py.) */
int rt_init(const int* device_ids, int n_devices, rt_ctx** out);
/* Diagnostic, needs no GPU: which of n_devices gets the tile whose corner is (x0, y0) (32-pixel tile grid) -- the deal of
 * rt_render_tile, the same rule as ray-tracing_amd/dist.py uses across processes. */
int rt_tile_owner(int x0, int y0, int n_devices);
void rt_shutdown(rt_ctx*);
const char* rt_last_error(const rt_ctx*);

/* Replaces the in-memory Scene the CPU path reads (Scene.h:42-51). */
int rt_upload_scene(rt_ctx*, const rt_scene_desc*);
/* camera.apply() + the GL read-back of matrixUtilities.h:188-217. */
int rt_set_camera(rt_ctx*, const rt_camera*);
/* Starts a frame: allocates/clears the img_w x img_h accumulator (std::vector<Vec3> image, main.cpp:203). */
int rt_begin_frame(rt_ctx*, const rt_render_params*);
/* Progressive accumulation, the part of SURVEY 8(f).4 that needs no GL (the reference redraws its GL preview while a frame
 * is being traced, main.cpp:154-177, and writes the PPM at the end; this lets a viewer show the frame after every pass).
 * Ends the current pass (runs what is queued) and opens the next one on the SAME accumulators: the tiles queued after it
 * add more_spp samples per pixel numbered on from the samples already taken, and every resolve divides by the total so
 * far. Passes of 4 + 4 + 8 samples leave the bits of one 16-sample frame (same sample numbers, same counter-RNG streams,
 * order-independent sums). Image size, camera and the other parameters stay those of rt_begin_frame. */
int rt_continue_frame(rt_ctx*, int32_t more_spp);
/* Sample sharding across contexts (one per GPU / rank; SURVEY 8(e): the reduce(sum) route for frames in which a few pixels dominate, where a
 * deal of tiles leaves the ranks unequal work). No counterpart in the reference, whose ThreadPool shares one image. rt_begin_frame_samples
 * opens a frame of p->spp samples per pixel of which THIS context renders samples first_sample .. first_sample + n_samples - 1 (of every tile
 * it is given: normally the whole image). Samples keep their numbers -- jitter and light-sampling streams are those of the undivided frame --
 * and the accumulators are order-independent fixed point, so after the contexts' accumulators have been added (rt_export_accumulators into
 * caller-owned DEVICE buffers: img_w*img_h*3 int64, img_w*img_h uint32 flag words; an all-reduce(sum) of the first and a bitwise OR of the
 * second; rt_import_accumulators) every resolve returns the bits of the frame one context renders alone. ray-tracing_amd/dist.py
 * (SampleReduce) does this over RCCL. Primary-hit ids are sample 0's: the context whose range starts at 0 has them. */
int rt_begin_frame_samples(rt_ctx*, const rt_render_params*, int32_t first_sample, int32_t n_samples);
int rt_export_accumulators(rt_ctx*, void* d_acc_i64, void* d_flags_u32);
int rt_import_accumulators(rt_ctx*, const void* d_acc_i64, const void* d_flags_u32);
/* Queues the pixels [x0,x0+w) x [y0,y0+h) (the per-pixel ThreadPool tasks of main.cpp:210-225).
 * Asynchronous: the work runs at the next rt_download_framebuffer / rt_resolve_device / rt_flush. */
int rt_render_tile(rt_ctx*, int x0, int y0, int w, int h);
/* The same for n tiles at once, xywh = n x (x0, y0, w, h): the enqueue loop of main.cpp:210-225 in one call (a 4K frame is
 * 8160 tiles of 32 x 32; through a foreign-function interface that is 8160 calls otherwise). Stops at the first tile that is
 * refused; the tiles before it stay queued. */
int rt_render_tiles(rt_ctx*, const int32_t* xywh, int64_t n);
/* Runs all queued tiles to completion on the device.
 * Queue capacities: a batch of B path states can append at most 2 B children (two per glass hit) and B diffuse records,
 * which is what the queues hold, and the path-state stack holds batch x (max_depth + 3): by construction they cannot
 * overflow. The kernels nevertheless bound every append (a state that does not fit is dropped, never written out of
 * range) and the host compares the counters with the capacities after every iteration: RT_ERR_LIMIT ("queue overflow")
 * would report a dropped state instead of returning a frame that misses it. The same holds for the per-lane stacks of the shadow walk:
 * their spill slices are sized at upload for the deepest stack the walk can build, a push beyond that is dropped, not written, and
 * the call returns RT_ERR_LIMIT ("the stack of the shadow walk overflowed"). */
int rt_flush(rt_ctx*);
/* pixel_color /= nsamples (main.cpp:219) for the whole image, rows top to bottom, into host memory.
 * prim_id (optional): 2 x uint32 per pixel = (object ordinal, triangle index) of sample 0's primary hit,
 * 0xFFFFFFFF = miss or pixel not rendered. stats optional. */
int rt_download_framebuffer(rt_ctx*, float* rgb, uint32_t* prim_id, rt_stats* stats);
/* Same resolve, into DEVICE memory of the caller (img_w*img_h*3 floats), for RCCL gathers. */
int rt_resolve_device(rt_ctx*, void* d_rgb);
/* The displayable frame, in DEVICE memory of the caller (img_w*img_h RGBA8 pixels, alpha 255): the bytes main.cpp:251-253
 * writes into rendu.ppm ((int)(255.f * std::min<float>(1.f, c)); values below 0, which the text file holds and a byte
 * cannot, become 0), rows top to bottom like the PPM or, bottom_up != 0, bottom to top as glDrawPixels and GL textures
 * want them. This is the device half of SURVEY 8(f).4 (a preview of the GPU frame in the GLUT window instead of
 * write-PPM-then-view): d_rgba may be a GL pixel-buffer object mapped with hipGraphicsGLRegisterBuffer /
 * hipGraphicsResourceGetMappedPointer (INTEGRATION.md shows the calls), or any other device pointer; with
 * rt_continue_frame it shows the frame after every pass. With several devices the pointer belongs to device_ids[0]. */
int rt_resolve_rgba8_device(rt_ctx*, void* d_rgba, int bottom_up);
int rt_get_stats(rt_ctx*, rt_stats* stats);
/* Diagnostic, no counterpart in the reference: 1 if the uploaded scene's shadow rays walk the 32-byte
 * quantised copies of the mesh nodes (all mesh boxes finite and nested, as bounding boxes of triangle
 * sets are), 0 if they walk the reference's nodes directly, negative on error. Results are identical. */
int rt_uses_quantised_shadow_nodes(const rt_ctx*);
/* Diagnostic, no counterpart in the reference, needs no GPU: the bound the distance pruning of rt_upload_scene relies
 * on. For the mesh triangle (v0, v1, v2) of a mesh whose largest |coordinate| is amax: how far from the triangle
 * record (the triangle scaled by 1.0001f, Mesh.cpp:107-109) a hit accepted by Triangle::getIntersection
 * (Triangle.h:101-107, barycentric sum in [0.9999, 1.00001]) can lie; infinity if the acceptance test gives no
 * bound (degenerate triangles). Tests hold it against the oracle's accepted hits. */
double rt_hit_excursion(const float v0[3], const float v1[3], const float v2[3], float amax);
/* No counterpart in the reference: the conservative plane filter the shadow walk runs in front of Triangle::getIntersection
 * (Triangle.h:86-121) on scenes with many objects, evaluated on the host for one triangle and one ray (1 = the triangle
 * goes on to the reference's test, 0 = rejected). Host only; for the tests that hold it against what the oracle accepts. */
int rt_filter_passes(const float v0[3], const float v1[3], const float v2[3], float amax, const float origin[3], const float direction[3]);

/* No counterpart in the reference: the group test that settles the shadow queries of a (hit point, light) pair without rays (k_shadow_group;
 * rt_stats::rays_shadow_grouped), for n pairs of the caller: out[i] = 1 if Scene::isInShadow (Scene.h:160-168) is proven false for EVERY ray from
 * pts[i] to a point within radii[i] of centres[i], 0 if not proven (or the scene has no object boxes). Diagnostic: the tests hold a 1 against
 * rt_occluded on sampled rays. Host arrays in and out. */
int rt_group_clear(rt_ctx*, const float* pts, const float* centres, const float* radii, int64_t n, int32_t* out);

/* Fine-grained seams, host arrays in and out:
 * Scene::computeIntersection (Scene.h:84-88) on n rays (origin, direction as given to Ray()). */
int rt_intersect(rt_ctx*, const float* org, const float* dir, int64_t n, rt_hit* out);
/* Scene::isInShadow (Scene.h:160-168) on n (point, light position) pairs. */
int rt_occluded(rt_ctx*, const float* pts, const float* light_pos, int64_t n, int32_t* out);
/* Scene::rayTrace (Scene.h:283-286) on n rays; pixel/sample key the light-sampling RNG. */
int rt_trace(rt_ctx*, const float* org, const float* dir, const uint32_t* pixel, const uint32_t* sample,
             int64_t n, const rt_render_params*, float* rgb);
/* rt_trace for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that must not disturb
 * a caching allocator). No counterpart in the reference. Same arguments, same radiance bits as rt_trace, but it calls no
 * hipMalloc / hipFree / hipEventCreate. It is NOT capturable into a HIP graph: like rt_trace it copies from and to the
 * caller's host arrays and synchronises the context's stream after every iteration of the wavefront loop (the host reads
 * the queue counters back to size the next launches). Staging, path states, queues, counters, shadow rays and the spill slices of the shadow walk
 * live in a file-scope __device__ array of RT_STATIC_BYTES = 8 MiB that every GPU the library runs on holds until the
 * process ends (address taken once per context, in rt_init). Limits (RT_ERR_LIMIT beyond them): n <= RT_STATIC_MAX_RAYS,
 * max_depth <= RT_STATIC_MAX_DEPTH, lights x light_samples <= 65536, and a scene whose 4-wide shadow walk spills at
 * most 64 stack entries per lane. All calls on one GPU share the array: calls that overlap on one GPU race. */
#define RT_STATIC_MAX_RAYS 4096
#define RT_STATIC_MAX_DEPTH 8
int rt_trace_static(rt_ctx*, const float* org, const float* dir, const uint32_t* pixel, const uint32_t* sample,
                    int64_t n, const rt_render_params*, float* rgb);
/* Primary rays of a tile exactly as the render kernels generate them (w*h*spp, sample-minor). */
int rt_camera_rays(rt_ctx*, const rt_render_params*, int x0, int y0, int w, int h, float* org, float* dir);

/* Counter RNG that replaces random()/rand() (Scene.h:115-116, main.cpp:195-196). */
uint32_t rt_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint64_t code, uint32_t light, uint32_t i, uint32_t which);

#ifdef __cplusplus
}
#endif
#endif
