/*
 * rtgpu.hip -- C-ABI of the MI355X render loop (see include/rtgpu.h) and the host
 * driver of the wavefront path-state loop that replaces Scene::rayTraceRecursive's
 * recursion and the per-pixel ThreadPool (main.cpp:203-226, Scene.h:174-286).
 *
 * Memory layout in HBM (uploaded once by rt_upload_scene):
 *   tri      4 x float4 per triangle in BVH leaf order (vertices pre-scaled by 1.0001f,
 *            unit normal, area, plane D precomputed exactly as Triangle.h:28-33,58)
 *   tri_ids  uint4 per triangle (3 global vertex ids, index into Mesh::triangles)
 *   normals  3 floats per vertex
 *   mnodes   64-byte inner nodes of all mesh trees; a child link is a global node index, or
 *            -2-first for a leaf whose triangle records start at `first` (the last record of a
 *            leaf is flagged; empty leaves share one never-hit record): no leaf table is read
 *   tnodes / tleaves / titems   top-level tree over object ordinals
 *   squares (64 B), spheres (32 B), materials (64 B), lights (32 B), sin/cos table
 * Per frame: int64 fixed-point accumulators (3 per pixel), prim ids, a LIFO stack of
 * 64-byte path states, child/diffuse queues for one batch.
 */
#include "rtgpu.h"
#include "rt_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <chrono>
#include <set>
#include <tuple>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include <utility>

#ifndef RT_FRONT_MAX_SHARE
#define RT_FRONT_MAX_SHARE 0.35    /* the front pass of the shadow rays stays on if it defers less than this share of them (measured: cow 13 % deferred,
                                    * frame -13 %; gorilla 27 %, -6 %; default scene 59 %, +22 %; chamber 61 %, +3 %; C5 100 %) */
#endif

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct EventPair { hipEvent_t a, b; int kind; };

}  // namespace

/* Scratch memory of rt_trace_static, the entry point that allocates nothing (include/rtgpu.h): every device the
 * library runs on holds its own copy of this array until the process ends. In host code the name is a stand-in,
 * not an address: rt_init asks hipGetSymbolAddress for this device's copy and only that pointer is ever used. */
#define RT_STATIC_BYTES (8u << 20)
static __device__ __align__(256) unsigned char rt_static_scratch[RT_STATIC_BYTES];

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool has_scene = false, has_camera = false, has_frame = false;

    DScene sc{};
    std::vector<DevBuf> scene_bufs;
    int n_lights = 0;
    CameraD cam{};
    FrameD fr{};
    std::vector<TileD> tiles;
    unsigned long long queued_samples = 0;
    int32_t resolve_spp = 0;        /* sample-sharded frames (rt_begin_frame_samples): the frame's samples per pixel over ALL contexts, what a resolve divides by; 0 = this context's own count */

    DevBuf acc, nf, prim, rgb, stack, child, drec, dcode, cnt, keys, ctr, tilebuf, srays, smask, dtex, ovf;
    uint32_t batch = 32u << 20;   /* path states per wavefront iteration at most (RT_BATCH). One batch of diffuse hits is one launch of the persistent shadow kernel, and
                                   * every launch ends in a tail of a few ms in which the longest rays finish alone: measured on C5 at 1 / 4 / 8 / 16 / 32 Mi states
                                   * 1112 / 1244 / 1334 / 1340 / 1358 Mrays/s (gorilla 75.6 / 65.8 / 61.8 ms per frame with 1 / 4 / 16 Mi). What a batch really
                                   * gets is bounded by the call's own primaries and by mem_budget (bufs_from_ctx). */
    unsigned long long mem_budget = 96ull << 30;    /* bytes the buffers of the wavefront loop may take together (RT_MEM_BUDGET_GB; HBM is 288 GB) */
    bool debug = false, count_steps = false;
    int persistent_blocks = 256 * 16;
    int group_state = -1;                          /* group culling of shadow rays on this scene: -1 undecided (runs, and counts), 0 off, 1 on */
    unsigned long long group_seen = 0, group_kept = 0, group_min_pairs = 8192;
    bool no_tail_swap = false;                     /* RT_NO_TAIL_SWAP (test knob): the children of every iteration are copied onto the stack */
    unsigned long long spread_lanes = 131072;      /* run_wavefront: lanes a small batch of bounced rays is spread over (RT_SPREAD_LANES, 0 = one state per lane always) */
    int wide_mode = 2;              /* closest hits over the 4-wide quantised tree: 0 never, 1 always, 2 for batches of bounced rays and rt_intersect (RT_WIDE_CLOSEST=0|1 overrides; see run_wavefront) */
    int cstack_wide = 0;            /* per-lane stack entries of that walk (three pending links per level); DScene::cstack_cap is what the binary walk needs */
    bool separate_raygen = false;   /* test knob RT_SEPARATE_RAYGEN=1: k_raygen writes the primaries out and k_closest loads them (the two ways must agree bit for bit) */
    int front_state = -1;       /* the front pass of the shadow rays (k_shadow_front): -1 = to be probed on this scene, 0 = off, 1 = on */
    Counters* h_ctr = nullptr;  /* pinned */
    Counters* d_hctr = nullptr; /* ... as the device sees it (k_publish writes the queue lengths of an iteration there) */
    rt_stats stats{};
    std::vector<EventPair> ev_pool;
    size_t ev_used = 0;
    hipEvent_t ev_frame_a = nullptr, ev_frame_b = nullptr;
    size_t lds_max = 65536;     /* sharedMemPerBlock of the device */
    /* a context over several GPUs (rt_init with n_devices > 1): one full context per device; this one only deals the tiles
     * and gathers the frame on subs[0]'s device */
    std::vector<rt_ctx*> subs;
    std::vector<std::vector<TileD>> sub_tiles;     /* per device: the tiles of this FRAME (the union over its progressive passes, each rectangle once), `first` counted in PIXELS */
    std::vector<std::set<std::tuple<int, int, int, int>>> sub_tile_keys;     /* ... the rectangles already in sub_tiles */
    std::vector<TileD> mg_all;                     /* the gathered tiles of all other devices, `first` counted along the concatenated slabs */
    DevBuf mg_slab, mg_tiles;                      /* on subs[0]'s device: the packed pixels of all other devices, one slab after the other / their tile list; on a rendering device: its own slab / tiles */
    hipEvent_t ev_pack = nullptr;                  /* (per device) this device's slab is packed */
    unsigned char* static_scratch = nullptr;   /* this device's copy of rt_static_scratch (rt_trace_static) */
};

namespace {

int fail(rt_ctx* c, int code, const char* fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}
#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(c, RT_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int ensure(rt_ctx* c, DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return RT_OK;
    if (b.p) { HIP_TRY(c, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
    if (bytes == 0) bytes = 256;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { b.p = nullptr; return fail(c, RT_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    b.bytes = bytes;
    return RT_OK;
}
void release(DevBuf& b) { if (b.p) (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }

template <class T>
int upload(rt_ctx* c, const std::vector<T>& v, const T** out) {
    DevBuf b;
    int rc = ensure(c, b, std::max<size_t>(v.size(), 1) * sizeof(T));
    if (rc) return rc;
    if (!v.empty()) HIP_TRY(c, hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    c->scene_bufs.push_back(b);
    *out = static_cast<const T*>(b.p);
    return RT_OK;
}

/* upload-time loops over millions of triangles: split over a few host threads (the per-item work is independent) */
template <typename F>
void parallel_for(int64_t n, F body) {
    unsigned hw = std::thread::hardware_concurrency();
    int nt = (int)std::min<int64_t>(std::min<unsigned>(hw ? hw : 1u, 16u), n / 65536 + 1);
    if (const char* e = getenv("OMP_NUM_THREADS")) nt = std::max(1, std::min(nt, atoi(e)));
    if (nt <= 1) { body((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([=] { body(n * t / nt, n * (t + 1) / nt); });
    for (auto& x : th) x.join();
}

/* host float helpers in the reference's operation order (this file is built with -ffp-contract=off) */
struct H3 { float x, y, z; };
inline H3 hsub(H3 a, H3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline H3 hcross(H3 a, H3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float hlen(H3 a) { return (float)std::sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z)); }

/* 4x4 double helpers, column-major as the GL matrices read back by glGetDoublev */
void m4_identity(double* m) { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
void m4_mul(const double* a, const double* b, double* o) {
    double t[16];
    for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += a[k * 4 + row] * b[col * 4 + k];
            t[col * 4 + row] = s;
        }
    memcpy(o, t, sizeof(t));
}
/* gluInvertMatrix (matrixUtilities.h:7-135): adjugate by cofactors. With M[r][c] = m[4r+c],
 * inv[4b+a] is the cofactor of element (a,b), expanded along the first remaining
 * column; products are taken left to right and the six terms summed left to right, which is
 * the evaluation order the reference's explicit expressions have. */
bool m4_inverse(const double* m, double* out) {
    double inv[16];
    for (int a = 0; a < 4; a++)          /* deleted row */
        for (int b = 0; b < 4; b++) {    /* deleted column */
            int R[3], C[3], nr = 0, nc = 0;
            for (int k = 0; k < 4; k++) { if (k != a) R[nr++] = k; if (k != b) C[nc++] = k; }
            double sgn = ((a + b) & 1) ? -1.0 : 1.0;
            double sum = 0.0;
            for (int k = 0; k < 3; k++) {
                int rA = R[k == 0 ? 1 : 0], rB = R[k == 2 ? 1 : 2];
                double first = m[4 * R[k] + C[0]];
                double s = (k == 0) ? sgn : -sgn * (k == 1 ? 1.0 : -1.0);
                double t1 = (s * first) * m[4 * rA + C[1]] * m[4 * rB + C[2]];
                double t2 = (-s * first) * m[4 * rA + C[2]] * m[4 * rB + C[1]];
                if (k == 0) sum = t1 + t2; else sum = sum + t1 + t2;
            }
            inv[4 * b + a] = sum;
        }
    double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) return false;
    det = 1.0 / det;
    for (int i = 0; i < 16; i++) out[i] = inv[i] * det;
    return true;
}

size_t lds_bytes(const rt_ctx* c, int block) { return (size_t)c->sc.stack_cap * block * sizeof(int); }
/* the closest-hit walks also keep the links of a fat leaf's sub-tree on the lane's stack */
size_t lds_bytes_closest(const rt_ctx* c, int block, bool wide = false) { return (size_t)(wide ? c->cstack_wide : c->sc.cstack_cap) * block * sizeof(int); }

EventPair* next_event(rt_ctx* c, int kind) {
    if (c->ev_used == c->ev_pool.size()) {
        EventPair p;
        if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
        c->ev_pool.push_back(p);
    }
    EventPair* e = &c->ev_pool[c->ev_used++];
    e->kind = kind;
    return e;
}

struct Source {                 /* where fresh path states come from */
    bool from_tiles;
    unsigned long long total;
    const float* d_org; const float* d_dir; const uint32_t* d_pixel; const uint32_t* d_sample;   /* explicit rays */
    int max_depth;
};

/* Device memory one run of the wavefront loop works in: the library's own growing buffers (bufs_from_ctx) or a
 * carve-up of the static scratch array (rt_trace_static, which must not allocate). */
struct WaveBufs {
    PathState* stack; size_t stack_cap;     /* LIFO of path states */
    PathState* child;                       /* 2 * batch */
    DiffuseRec* drec; unsigned long long* dcode; uint32_t* dtex;   /* batch each */
    uint32_t* cnt;                          /* batch * lights */
    uint32_t* keys;                         /* batch * lights: the (record, light) pairs k_shadow_group leaves; nullptr = no group culling */
    Counters* ctr;
    TileD* tilebuf;                         /* from_tiles only */
    float4* srays; unsigned long long srays_cap;    /* shadow-ray buffer, capacity in rays (2 float4 each) */
    uint32_t* smask;                        /* per shadow ray: the top-level leaves it reaches (k_shadow<5>); nullptr = that variant is not used */
    int* ovf; size_t ovf_bytes;             /* spill slices of the 4-wide shadow walk */
    uint32_t batch;
    int persistent_blocks;
    bool timing;                            /* record HIP events per launch (creates events on demand) */
};

int wide_ovf_cap(const rt_ctx* c, int* lds_cap_out) {
    /* stack entries of a lane kept in LDS (the first stack_cap = what the binary walks need; the 4-wide shadow walk
     * can need three per level and spills the rest -- rare -- to a global slice) */
    int lds_cap = c->sc.stack_cap;
    if (const char* e = getenv("RT_WIDE_LDS_CAP"))          /* test knob: force the spilling path */
        if (c->sc.quant_ok && atoi(e) > 0) lds_cap = std::max(c->sc.top_cap + 1, std::min(lds_cap, atoi(e)));
    if (lds_cap_out) *lds_cap_out = lds_cap;
    return c->sc.quant_ok ? std::max(0, c->sc.wstack_cap - lds_cap) : 0;
}

/* the library's own buffers, grown to what this run can need */
int bufs_from_ctx(rt_ctx* c, const Source& src, WaveBufs& w) {
    /* no more states per iteration than the call has primaries (a small frame must not size the buffers of a big one) */
    const int L = c->n_lights;
    const int S = c->fr.shadow_mode == 1 ? 1 : c->fr.light_samples;
    /* a batch costs, per state: its place on the stack at every depth, two children, a diffuse record with code, texel and
     * counters, and the shadow rays of its hit (32 bytes each, unless the scene has no meshes: no ray buffer then) */
    const bool no_ray_buffer = (c->sc.n_meshes == 0 && !getenv("RT_NO_DIRECT_SHADOW")) || L == 0;
    const unsigned long long per_state = (unsigned long long)sizeof(PathState) * (unsigned long long)(src.max_depth + 3 + 2) + sizeof(DiffuseRec) + 12ull +
                                         8ull * (unsigned long long)std::max(L, 1) + (no_ray_buffer ? 0ull : 32ull * (unsigned long long)std::max(L, 1) * (unsigned long long)S);
    const unsigned long long affordable = std::max<unsigned long long>(1ull << 20, (c->mem_budget / per_state) & ~1023ull);
    const uint32_t B = (uint32_t)std::min<unsigned long long>(std::min<unsigned long long>(c->batch, affordable), (std::max<unsigned long long>(src.total, 4096ull) + 1023ull) & ~1023ull);
    size_t need_stack = (size_t)B * (size_t)(src.max_depth + 3);
    int rc;
    if ((rc = ensure(c, c->stack, need_stack * sizeof(PathState)))) return rc;
    if ((rc = ensure(c, c->child, (size_t)2 * B * sizeof(PathState)))) return rc;
    if ((rc = ensure(c, c->drec, (size_t)B * sizeof(DiffuseRec)))) return rc;
    if ((rc = ensure(c, c->dcode, (size_t)B * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(c, c->dtex, (size_t)B * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->cnt, (size_t)B * std::max(L, 1) * sizeof(uint32_t)))) return rc;
    if (c->sc.n_obox > 0 && (rc = ensure(c, c->keys, (size_t)B * std::max(L, 1) * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->ctr, sizeof(Counters)))) return rc;
    if (src.from_tiles && (rc = ensure(c, c->tilebuf, std::max<size_t>(c->tiles.size(), 1) * sizeof(TileD)))) return rc;
    const unsigned long long per_rec = (unsigned long long)std::max(L, 1) * S;
    unsigned long long ray_cap = 1280ull << 20;                 /* rays per k_shadow launch at most (40 GiB of 32-byte rays): a batch of 32 Mi hits x 35 samples is one launch */
    if (const char* e = getenv("RT_SHADOW_RAYS")) ray_cap = std::max<unsigned long long>(1ull << 16, strtoull(e, nullptr, 10));
    const unsigned long long chunk = std::max<unsigned long long>(1ull, std::min<unsigned long long>(B, ray_cap / per_rec));
    const bool direct = c->sc.n_meshes == 0 && !getenv("RT_NO_DIRECT_SHADOW");
    if (!direct && L > 0 && (rc = ensure(c, c->srays, (size_t)(chunk * per_rec) * 2 * sizeof(float4)))) return rc;
    const bool want_mask = !direct && L > 0 && c->sc.n_flat > 0 && c->sc.has_topsub && c->sc.quant_ok;
    if (want_mask && (rc = ensure(c, c->smask, (size_t)(chunk * per_rec) * sizeof(uint32_t)))) return rc;
    const int ovf_cap = wide_ovf_cap(c, nullptr);
    if ((rc = ensure(c, c->ovf, std::max<size_t>(256, (size_t)c->persistent_blocks * ovf_cap * RT_BLOCK * sizeof(int))))) return rc;
    w.stack = static_cast<PathState*>(c->stack.p); w.stack_cap = c->stack.bytes / sizeof(PathState);
    w.child = static_cast<PathState*>(c->child.p);
    w.drec = static_cast<DiffuseRec*>(c->drec.p); w.dcode = static_cast<unsigned long long*>(c->dcode.p); w.dtex = static_cast<uint32_t*>(c->dtex.p);
    w.cnt = static_cast<uint32_t*>(c->cnt.p); w.ctr = static_cast<Counters*>(c->ctr.p);
    w.keys = c->sc.n_obox > 0 ? static_cast<uint32_t*>(c->keys.p) : nullptr;
    w.tilebuf = static_cast<TileD*>(c->tilebuf.p);
    w.srays = static_cast<float4*>(c->srays.p); w.srays_cap = c->srays.bytes / (2 * sizeof(float4));
    w.smask = want_mask ? static_cast<uint32_t*>(c->smask.p) : nullptr;
    w.ovf = static_cast<int*>(c->ovf.p); w.ovf_bytes = c->ovf.bytes;
    w.batch = B; w.persistent_blocks = c->persistent_blocks; w.timing = true;
    return RT_OK;
}

/* The wavefront loop. Path states sit on a LIFO stack; every iteration tops the batch up with
 * fresh primaries, pops up to `batch` states (deepest first, which bounds the live set by
 * batch*(max_depth+2)), runs closest-hit + material dispatch, the shadow pass and the Phong
 * pass for that batch, then pushes the children. Allocates nothing itself: all device memory comes in through `w`
 * (and no events are created when w.timing is off). */
int run_wavefront(rt_ctx* c, const Source& src, const WaveBufs& w, long long* acc, uint32_t* nonfinite, uint32_t* prim) {
    const uint32_t B = w.batch;
    const int L = c->n_lights;
    const int S = c->fr.shadow_mode == 1 ? 1 : c->fr.light_samples;
    if (w.stack_cap < (size_t)B * (size_t)(src.max_depth + 3)) return fail(c, RT_ERR_LIMIT, "path-state stack too small for depth %d", src.max_depth);
    if (src.from_tiles)
        HIP_TRY(c, hipMemcpyAsync(w.tilebuf, c->tiles.data(), c->tiles.size() * sizeof(TileD), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(w.ctr, 0, sizeof(Counters), c->stream));

    PathState* stack = w.stack;
    Counters* d_ctr = w.ctr;
    size_t top = 0;
    unsigned long long done = 0;
    unsigned long long front_rays = 0;      /* shadow rays that went through k_shadow_front in this call */
    size_t lds = lds_bytes(c, RT_BLOCK);
    c->ev_used = 0;
    /* threads per block of k_closest (one append atomic per block): big blocks where rays are cheap and the appends would
     * dominate, one wave per block under deep mesh trees, where a block would wait for its slowest wave
     * (64 / 128 / 256 threads: refraction scene 13.9 / 10.7 / 9.3 ms, gorilla 13.5 / 12.0 / 11.1, C5 reduced 25.6 / 28.0 / 28.6) */
    uint32_t cblock_light = (c->sc.stack_cap - c->sc.top_cap - 2 > 16) ? 64u : (uint32_t)RT_CBLOCK;
    if (lds_bytes_closest(c, RT_CBLOCK) + 64 > c->lds_max) cblock_light = 64u;      /* (deep sub-trees under a shallow tree: the big block's stacks would not fit) */
    uint32_t cblock = cblock_light;
    if (const char* e = getenv("RT_CBLOCK")) { int v = atoi(e); if ((v == 64 || v == 128 || v == 256) && lds_bytes_closest(c, v) + 64 <= c->lds_max) cblock = (uint32_t)v; }

    const auto t_loop0 = std::chrono::steady_clock::now();     /* (RT_DEBUG prints each iteration's time since here) */
    /* RT_SYNC_DEBUG=1: wait for every kernel of the loop and say which one has run (finding the kernel behind a device error) */
    const bool sync_debug = getenv("RT_SYNC_DEBUG") != nullptr;
    auto syncpoint = [&](const char* what) -> int {
        if (!sync_debug) return RT_OK;
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, "after %s: %s", what, hipGetErrorString(e));
        fprintf(stderr, "[rtgpu] done: %s\n", what);
        fflush(stderr);
        return RT_OK;
    };
    /* The tail of a frame -- no primaries left, the stack empty, the children of one batch fit the next -- does not go through
     * the stack: the children of one iteration are the batch of the next where they lie, k_closest writing to the child
     * buffer and to the (empty) stack buffer in turns. No copy per iteration then: a depth-50 path inside a glass mesh is fifty
     * iterations of a few hundred states, and their stream operations are what they cost. */
    PathState* alt_in = nullptr;
    uint32_t alt_n = 0;
    while (done < src.total || top > 0 || alt_n > 0) {
        /* 1. top up with fresh primaries */
        uint32_t room = top < B ? (uint32_t)(B - top) : 0u;
        uint32_t n_new = (uint32_t)std::min<unsigned long long>(src.total - done, room);
        const bool fuse_raygen = src.from_tiles && !c->separate_raygen;     /* the fresh primaries of queued tiles are made inside k_closest */
        FreshD fresh{};
        if (n_new) {
            dim3 g((n_new + 255) / 256), b(256);
            if (fuse_raygen) {
                fresh.cam = c->cam; fresh.fr = c->fr; fresh.tiles = static_cast<const TileD*>(w.tilebuf); fresh.n_tiles = (int)c->tiles.size();
                fresh.first = done;
            } else if (src.from_tiles)
                hipLaunchKernelGGL(k_raygen, g, b, 0, c->stream, c->cam, c->fr, static_cast<const TileD*>(w.tilebuf), (int)c->tiles.size(),
                                   done, n_new, stack + top);
            else
                hipLaunchKernelGGL(k_seed, g, b, 0, c->stream, src.d_org, src.d_dir, src.d_pixel, src.d_sample, (uint32_t)done, n_new,
                                   src.max_depth, stack + top);
            top += n_new;
            done += n_new;
        }
        /* 2. pop the batch */
        uint32_t nb = (uint32_t)std::min<size_t>(top, B);
        size_t base = top - nb;
        top = base;
        const PathState* in_ptr = stack + base;
        PathState* out_ptr = w.child;
        if (alt_n) {                       /* (top == 0 and n_new == 0 here) */
            nb = alt_n; base = 0;
            in_ptr = alt_in;
            out_ptr = alt_in == w.child ? stack : w.child;
            alt_in = nullptr; alt_n = 0;
        }
        /* (n_child and n_diffuse are zero here: the frame's fill above, then k_publish after every k_closest) */
        EventPair* e1 = nullptr;
        if (w.timing) {
            if (!(e1 = next_event(c, 0))) return fail(c, RT_ERR_DEVICE, "hipEventCreate failed");
            HIP_TRY(c, hipEventRecord(e1->a, c->stream));
        }
        /* (a batch of bounced rays is the heavy, divergent kind: default scene 147 ms with one wave per block, 169 with four) */
        if (!getenv("RT_CBLOCK")) cblock = (2ull * n_new >= nb) ? cblock_light : 64u;
        /* Which mesh walk: the 4-wide quantised tree halves the node fetches but spends more instructions per box, and needs a
         * deeper stack. Bounced rays are incoherent and wait for their loads: it wins there (k_closest -11 % on the default
         * scene, -15 % in the mirror chamber). The fresh primaries of a 16-spp frame come sixteen to a pixel, their lanes read
         * the same lines, and the binary walk over the reference's nodes is the faster one (C5, 32 Mi primaries: 37 ms against
         * 53). Both return the same hit, bit for bit. */
        DScene scl = c->sc;
        const bool fuse_hash = c->fr.shadow_mode == 0 && L > 0 && !getenv("RT_SEPARATE_HASH");      /* (test knob: the separate pass must leave the same bits) */
        scl.rng_hash = fuse_hash ? 1 : 0; scl.rng_seed = c->fr.seed;
        scl.wide_closest = (c->sc.wide_closest && (c->wide_mode == 1 || (c->wide_mode == 2 && 2ull * n_new < nb))) ? 1 : 0;
        const bool wide = scl.wide_closest != 0;
        /* a small batch of bounced rays (the depth-50 tail of a glass mesh: a few hundred states per iteration, fifty iterations)
         * is spread one state per 2^shift lanes over up to spread_lanes lanes */
        uint32_t spread = 0;
        if (n_new == 0 && cblock == 64u) while (spread < 6 && ((unsigned long long)nb << (spread + 1)) <= c->spread_lanes) spread++;
        scl.spread_shift = (int32_t)spread;
        if (wide) scl.cstack_cap = c->cstack_wide;         /* (what the kernel's stacks take in this launch: the words of its block-wide append sit behind them) */

        if (wide && cblock > 64u && lds_bytes_closest(c, cblock, true) + 64 > c->lds_max) cblock = 64u;
        auto launch_closest = [&](auto kern) {
            hipLaunchKernelGGL(kern, dim3((uint32_t)((((unsigned long long)nb << spread) + cblock - 1) / cblock)), dim3(cblock), lds_bytes_closest(c, cblock, wide) + (cblock > 64u ? (2 * (cblock / 64) + 2) * sizeof(uint32_t) : 0), c->stream, scl, in_ptr, nb,   /* (no word more than needed: LDS comes in large granules, and a few bytes can cost two blocks per CU) */
                               out_ptr, 2 * B, w.drec, w.dcode, w.dtex, B, d_ctr, acc, nonfinite, prim, fresh);
        };
        fresh.n_old = nb - n_new;
        const bool sub = c->sc.has_sub && c->sc.quant_ok;
        auto pick_closest = [&](auto fresh_c, auto wide_c) {
            constexpr bool F = decltype(fresh_c)::value, W = decltype(wide_c)::value;
            if (sub) { if (cblock == 64u) launch_closest(k_closest<true, 64, F, W>); else launch_closest(k_closest<true, RT_CBLOCK, F, W>); }
            else { if (cblock == 64u) launch_closest(k_closest<false, 64, F, W>); else launch_closest(k_closest<false, RT_CBLOCK, F, W>); }
        };
        if (fuse_raygen && n_new) { if (wide) pick_closest(std::true_type{}, std::true_type{}); else pick_closest(std::true_type{}, std::false_type{}); }
        else { if (wide) pick_closest(std::false_type{}, std::true_type{}); else pick_closest(std::false_type{}, std::false_type{}); }
        if (e1) HIP_TRY(c, hipEventRecord(e1->b, c->stream));
        if (int src_ = syncpoint("k_closest")) return src_;
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, c->stream, d_ctr, c->d_hctr, (c->front_state < 0 && front_rays) ? 1 : 0);
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipGetLastError());
        if (c->front_state < 0 && front_rays >= (16ull << 20)) {
            /* the share of deferred rays so far decides for the scene once enough of the frame has been seen (the first
             * tiles of an image are not the scene); until then the front pass runs */
            const unsigned long long deferred = c->h_ctr->deferred_sum;
            const bool enough = front_rays >= (1ull << 30) || done >= src.total;
            if ((double)deferred > RT_FRONT_MAX_SHARE * (double)front_rays) c->front_state = 0;
            else if (enough) c->front_state = 1;
            if (c->debug && c->front_state >= 0)
                fprintf(stderr, "[rtgpu] front pass: %llu of %llu shadow rays deferred to the persistent kernel -> %s for this scene\n", deferred, front_rays,
                        c->front_state ? "on" : "off");
        }
        uint32_t n_child = c->h_ctr->n_child, n_diff = c->h_ctr->n_diffuse;
        c->stats.rays_primary += n_new;                 /* the fresh primaries sit on top of the stack: all of them are in this batch */
        c->stats.rays_secondary += nb - n_new;
        if (c->debug) fprintf(stderr, "[rtgpu] iter %llu: new %u batch %u -> children %u diffuse %u (stack %zu) at %.1f us\n",
                              (unsigned long long)c->stats.iterations, n_new, nb, n_child, n_diff, top,
                              std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_loop0).count());
        if (n_child > 2 * B || n_diff > B) return fail(c, RT_ERR_LIMIT, "queue overflow (%u children, %u diffuse)", n_child, n_diff);
        /* 3. shadow + Phong for the diffuse records of this batch */
        /* Group culling (k_shadow_group, rt_kernels.h): the (record, light) pairs of a chunk whose 35 rays provably meet nothing are
         * dropped before any ray is made; the launches below then run over the pairs that are left. One more read-back per chunk, so
         * only for chunks worth it; the first million pairs of a scene decide whether the scene's objects leave enough pairs clear
         * to pay for the test (its cost grows with the number of objects). */
        auto group_pass = [&](uint32_t r0, uint32_t nr, DScene& scs, unsigned long long& threads) -> int {
            scs.group_keys = nullptr; scs.group_n = 0;
            const unsigned long long pairs = (unsigned long long)nr * (unsigned long long)L;
            if (!w.keys || c->sc.n_obox == 0 || c->group_state == 0 || pairs < c->group_min_pairs || pairs > 0xFFFFFF00ull) return RT_OK;
            HIP_TRY(c, hipMemsetAsync(&d_ctr->n_keys, 0, sizeof(unsigned long long), c->stream));
            DScene scq = c->sc;
            scq.group_debug = c->debug ? 1 : 0;
            hipLaunchKernelGGL(k_shadow_group, dim3((uint32_t)((pairs + 255) / 256)), dim3(256), 0, c->stream, scq, c->fr, w.drec + r0, nr, w.keys, d_ctr);
            HIP_TRY(c, hipMemcpyAsync(&c->h_ctr->n_keys, &d_ctr->n_keys, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const unsigned long long kept = c->h_ctr->n_keys;
            if (kept > pairs) return fail(c, RT_ERR_DEVICE, "group culling returned %llu of %llu pairs", kept, pairs);
            c->stats.rays_shadow_grouped += (pairs - kept) * (unsigned long long)S;
            if (c->group_state < 0) {
                c->group_seen += pairs; c->group_kept += kept;
                if (c->group_seen >= (1ull << 20)) {
                    const double clear = 1.0 - (double)c->group_kept / (double)c->group_seen;
                    c->group_state = clear > 0.02 + 0.004 * (double)c->sc.n_obox ? 1 : 0;
                    if (c->debug) fprintf(stderr, "[rtgpu] group culling: %.1f %% of %llu (record, light) pairs clear of all %d objects -> %s for this scene\n", 100.0 * clear,
                                          c->group_seen, c->sc.n_obox, c->group_state ? "on" : "off");
                }
            }
            scs.group_keys = w.keys; scs.group_n = (uint32_t)kept;
            threads = kept * (unsigned long long)S;
            return RT_OK;
        };
        if (n_diff && L > 0) {
            HIP_TRY(c, hipMemsetAsync(w.cnt, 0, (size_t)n_diff * L * sizeof(uint32_t), c->stream));
            const unsigned long long per_rec = (unsigned long long)L * S;
            if (c->fr.shadow_mode == 0 && !fuse_hash)
                hipLaunchKernelGGL(k_hash_records, dim3((n_diff + 255) / 256), dim3(256), 0, c->stream, c->fr.seed, w.drec, w.dcode, n_diff);
            /* no meshes: one thread per ray from record to verdict (k_shadow_direct), no ray buffer, no persistent kernel */
            const bool direct = c->sc.n_meshes == 0 && !getenv("RT_NO_DIRECT_SHADOW");
            if (direct) {
                const uint32_t dchunk = (uint32_t)std::max<unsigned long long>(1ull, std::min<unsigned long long>(n_diff, (1ull << 38) / per_rec));
                for (uint32_t r0 = 0; r0 < n_diff; r0 += dchunk) {
                    const uint32_t nr = std::min(dchunk, n_diff - r0);
                    unsigned long long threads = (unsigned long long)nr * per_rec;
                    c->stats.rays_shadow += threads;
                    DScene scs = c->sc;
                    if (int grc = group_pass(r0, nr, scs, threads)) return grc;
                    if (threads == 0) continue;
                    EventPair* e2 = nullptr;
                    if (w.timing) {
                        if (!(e2 = next_event(c, 1))) return fail(c, RT_ERR_DEVICE, "hipEventCreate failed");
                        HIP_TRY(c, hipEventRecord(e2->a, c->stream));
                    }
                    hipLaunchKernelGGL(k_shadow_direct, dim3((uint32_t)((threads + 255) / 256)), dim3(256), (size_t)c->sc.stack_cap * 256 * sizeof(int), c->stream,
                                       scs, c->fr, w.drec + r0, w.dcode + r0, nr, S, w.cnt + (size_t)r0 * L);
                    if (e2) HIP_TRY(c, hipEventRecord(e2->b, c->stream));
                }
            }
            /* the records are processed in chunks that fit the 32-byte ray buffer */
            if (!direct && w.srays_cap < per_rec) return fail(c, RT_ERR_LIMIT, "shadow-ray buffer too small for %d lights x %d samples", L, S);
            const uint32_t chunk = (uint32_t)std::max<unsigned long long>(1ull, std::min<unsigned long long>(n_diff, w.srays_cap / per_rec));
            for (uint32_t r0 = 0; r0 < n_diff && !direct; r0 += chunk) {
                const uint32_t nr = std::min(chunk, n_diff - r0);
                unsigned long long threads = (unsigned long long)nr * per_rec;
                c->stats.rays_shadow += threads;
                DScene scg = c->sc;
                if (int grc = group_pass(r0, nr, scg, threads)) return grc;
                if (threads == 0) continue;
                /* persistent waves: enough 64-lane blocks to fill the chip, never more than the work */
                const unsigned long long blocks = std::min<unsigned long long>((threads + RT_BLOCK - 1) / RT_BLOCK, (unsigned long long)w.persistent_blocks);
                unsigned long long blocks_now = blocks;
                HIP_TRY(c, hipMemsetAsync(&d_ctr->work, 0, sizeof(unsigned long long), c->stream));
                uint32_t* cnt0 = w.cnt + (size_t)r0 * L;
                /* Front pass: rays that stay clear of every mesh frame are settled one thread per ray, the rest go to the
                 * persistent kernel. Whether that pays is the scene's matter (a gorilla on a floor: most rays pass it by;
                 * C5's soup fills the scene: every ray enters it), so the first launch on a scene looks at the share of
                 * deferred rays and decides for the scene. */
                const size_t front_lds = (size_t)c->sc.top_cap * RT_FBLOCK * sizeof(int) + (RT_FBLOCK / 64 + 2) * sizeof(uint32_t);
                const bool front = c->sc.quant_ok && c->front_state != 0 && front_lds <= c->lds_max;
                EventPair* e2 = nullptr;
                if (front) {
                    HIP_TRY(c, hipMemsetAsync(&d_ctr->deferred, 0, sizeof(unsigned long long), c->stream));
                    if (w.timing) {
                        if (!(e2 = next_event(c, 1))) return fail(c, RT_ERR_DEVICE, "hipEventCreate failed");
                        HIP_TRY(c, hipEventRecord(e2->a, c->stream));
                    }
                    hipLaunchKernelGGL(k_shadow_front, dim3((uint32_t)((threads + RT_FBLOCK - 1) / RT_FBLOCK)), dim3(RT_FBLOCK),
                                       front_lds, c->stream, scg, c->fr,
                                       w.drec + r0, w.dcode + r0, nr, S, cnt0, w.srays, &d_ctr->deferred, &d_ctr->deferred_sum);
                    if (int src_ = syncpoint("k_shadow_front")) return src_;
                    blocks_now = (unsigned long long)w.persistent_blocks;      /* (how many rays come through is on the device) */
                    front_rays += threads;
                } else {
                    hipLaunchKernelGGL(k_shadow_setup, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, c->stream, scg, c->fr,
                                       w.drec + r0, w.dcode + r0, nr, S, w.srays, w.smask);
                    if (int src_ = syncpoint("k_shadow_setup")) return src_;
                    if (w.timing) {
                        if (!(e2 = next_event(c, 1))) return fail(c, RT_ERR_DEVICE, "hipEventCreate failed");
                        HIP_TRY(c, hipEventRecord(e2->a, c->stream));
                    }
                }
                /* small top-level tables (a few objects) are copied into LDS behind the stacks: a new ray walks them
                 * through four dependent loads while the other lanes of its wave wait */
                /* (room: 16 blocks per CU share 160 KB of LDS, so stacks + tables may take 10 KB per block) */
                const size_t top_room = std::max<size_t>(RT_TOP_LDS_MAX, lds < 10240 ? 10240 - lds : 0);
                const int top_lds = ((size_t)c->sc.top_blob_bytes <= top_room && !getenv("RT_TOP_GLOBAL")) ? 1 : 0;
                int lds_cap = 0;
                const int ovf_cap = wide_ovf_cap(c, &lds_cap);
                if (c->debug && r0 == 0 && c->stats.iterations == 0)
                    fprintf(stderr, "[rtgpu] k_shadow: %zu bytes of LDS stacks per wave, top-level tables %d bytes -> %s\n", lds, c->sc.top_blob_bytes, top_lds ? "LDS" : "global memory");
                if ((size_t)std::max<unsigned long long>(blocks, blocks_now) * ovf_cap * RT_BLOCK * sizeof(int) > w.ovf_bytes)
                    return fail(c, RT_ERR_LIMIT, "spill buffer of the shadow walk too small (%d entries per lane)", ovf_cap);
                int top_min = c->sc.n_top_items >= 24 ? 32 : 1;       /* see k_shadow; measured on the default scene: 1, 8, 16, 32, 48, 60 -> 2365, 2205, 2065, 1828, 1839, 2111 ms */
                /* a handful of objects around a mesh with fat leaves (k_shadow<2> anyway): rounds of eight lanes (1 / 4 / 8 / 16 / 32: mirror chamber 99.3 /
                 * 96.7 / 94.9 / 95.1 / 97.8 ms, cow 20.1 / 19.8 / 19.7 / 19.6 / 20.2, gorilla 31.8 / 32.0 / 31.9 / 32.2 / 33.8). Not for scenes without
                 * sub-trees: a top_min above 1 would take them off the plain kernel (C5) */
                if (top_min == 1 && c->sc.has_sub) top_min = 8;
                if (c->sc.has_topsub) top_min = 1;                     /* (with sub-trees over the objects a top-level round is short: 32 / 16 / 8 / 1 -> 94.7 / 93.1 / 92.6 / 91.9 ms on the default scene at 4 spp) */
                if (const char* e = getenv("RT_TOP_MIN")) top_min = std::max(1, atoi(e));
                const bool flat_top = c->sc.quant_ok && c->sc.has_topsub && c->sc.n_flat > 0 && w.smask && !front;     /* (its per-ray leaf masks come from k_shadow_setup) */      /* k_shadow<5>: the variant of mode 3 with the flat top-level pass (forced onto the scenes of mode 2 it loses: chamber 122 -> 137 ms, gorilla 42.3 -> 45.2, cow 29.1 -> 31.8 -- they pay for mode 3's plane filter and object sub-trees, and have two to four top-level leaves) */
                DScene scs = c->sc;
                if (!flat_top) scs.n_flat = 0;
                auto launch = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3((uint32_t)blocks_now), dim3(RT_BLOCK), lds + (top_lds ? (size_t)c->sc.top_blob_bytes : 0), c->stream, scs,
                                       static_cast<const float4*>(w.srays), threads, S, cnt0, d_ctr, &d_ctr->work, top_lds,
                                       w.ovf, lds_cap, ovf_cap, top_min,
                                       front ? static_cast<const unsigned long long*>(&d_ctr->deferred) : static_cast<const unsigned long long*>(nullptr),
                                       static_cast<const uint32_t*>(flat_top ? w.smask : nullptr));
                };
                if (flat_top) { if (c->count_steps) launch(k_shadow<5, true>); else launch(k_shadow<5, false>); }
                else if (c->sc.quant_ok && c->sc.has_topsub) { if (c->count_steps) launch(k_shadow<3, true>); else launch(k_shadow<3, false>); }
                else if (c->sc.quant_ok && (c->sc.has_sub || top_min > 1)) { if (c->count_steps) launch(k_shadow<2, true>); else launch(k_shadow<2, false>); }
                else if (c->sc.quant_ok) { if (c->count_steps) launch(k_shadow<1, true>); else launch(k_shadow<1, false>); }
                else { if (c->count_steps) launch(k_shadow<0, true>); else launch(k_shadow<0, false>); }
                if (e2) HIP_TRY(c, hipEventRecord(e2->b, c->stream));
                if (int src_ = syncpoint("k_shadow")) return src_;
            }
        }
        if (n_diff) {
            if (L == 0) HIP_TRY(c, hipMemsetAsync(w.cnt, 0, 256, c->stream));
            hipLaunchKernelGGL(k_shade, dim3((n_diff + 255) / 256), dim3(256), 0, c->stream, c->sc, w.drec,
                               w.dtex, n_diff, S, w.cnt, acc, nonfinite, d_ctr);
            if (int src_ = syncpoint("k_shade")) return src_;
        }
        /* 4. push the children */
        if (n_child) {
            if (top == 0 && done >= src.total && n_child <= B && !c->no_tail_swap) { alt_in = out_ptr; alt_n = n_child; }    /* the next batch, where it lies */
            else if (out_ptr == stack) top = n_child;                   /* (written to the empty stack buffer: in place already) */
            else {
                if (top + n_child > w.stack_cap) return fail(c, RT_ERR_LIMIT, "path-state stack overflow");
                HIP_TRY(c, hipMemcpyAsync(stack + top, w.child, (size_t)n_child * sizeof(PathState), hipMemcpyDeviceToDevice, c->stream));
                top += n_child;
            }
        }
        c->stats.iterations++;
        c->stats.peak_states = std::max<uint64_t>(c->stats.peak_states, top + alt_n);
    }
    HIP_TRY(c, hipMemcpyAsync(c->h_ctr, w.ctr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    if (c->h_ctr->walk_overflow) return fail(c, RT_ERR_LIMIT, "the stack of the shadow walk overflowed (%d entries per lane): the frame is void", c->sc.wstack_cap);
    c->stats.nonfinite += c->h_ctr->nonfinite;
    if (c->debug && c->sc.n_obox > 0 && c->group_state != 0)
        fprintf(stderr, "[rtgpu] group culling kept pairs because of: origin or length %llu, a square %llu, a sphere %llu, a mesh %llu\n",
                c->h_ctr->group_why[0], c->h_ctr->group_why[1], c->h_ctr->group_why[2], c->h_ctr->group_why[3]);
    c->stats.rays_shadow_culled += c->h_ctr->culled * (unsigned long long)L * (unsigned long long)S;
    if (c->count_steps)
        fprintf(stderr, "[rtgpu] shadow steps: node %llu tri %llu other %llu lane-iterations %llu rays %llu; wave iterations: node %llu tri %llu top %llu (top lanes %llu)\n",
                c->h_ctr->sh_node_steps, c->h_ctr->sh_tri_steps, c->h_ctr->sh_other_steps, c->h_ctr->sh_loop_iters, c->h_ctr->rays_shadow,
                c->h_ctr->sh_node_iters, c->h_ctr->sh_tri_iters, c->h_ctr->sh_top_iters, c->h_ctr->sh_top_lanes);
    if (c->count_steps)
        fprintf(stderr, "[rtgpu] shadow wave clocks: refill %llu top %llu node phase %llu triangle phase %llu; in mode 1: node steps %llu tri steps %llu\n", c->h_ctr->sh_clk_refill,
                c->h_ctr->sh_clk_top, c->h_ctr->sh_clk_node, c->h_ctr->sh_clk_tri, c->h_ctr->sh_m1_node, c->h_ctr->sh_m1_tri);
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev_pool[i].a, c->ev_pool[i].b) == hipSuccess) {
            if (c->ev_pool[i].kind == 0) { c->stats.ms_closest += ms; c->stats.launches_closest++; }
            else { c->stats.ms_shadow += ms; c->stats.launches_shadow++; }
        }
    }
    return RT_OK;
}


/* ---- where can an accepted hit of a triangle record lie? Pruning by distance (rt_kernels.h, "result-preserving
 * pruning") needs every accepted hit point to lie inside the box the prune test is made on. The reference accepts
 * a point P of the triangle's (float) plane when the float sum of the three sub-areas over the stored area is in
 * [0.9999, 1.00001] (Triangle.h:101-107) -- a NaN sum passes too -- so P can lie outside the triangle: by the
 * tolerance, by the rounding of the sum (absolute, so it weighs more the smaller the area), and without limit for
 * a triangle whose stored area is 0, NaN or lost in that rounding. With l the longest side, a the stored area,
 * u = 2^-24 and rho the distance of P from the triangle:
 *   sub-area sum S(P) >= A + g(P),  g >= h*rho/2 in the plane (h = least altitude),  S >= sqrt(A^2 + (z*l)^2) at
 *   height z over it;  computed sum within E(rho) = 32*u*l*(l + rho) + 16*u*l^2 of S;  |a - A| <= 16*u*l^2;
 *   accepted => g(P) <= G(rho) = 1.1e-5*a + 2*E(rho)
 *   => rho <= F(rho) = 2*G(rho)/h + sqrt(2*a*G(rho) + G(rho)^2)/l,   F concave, so the least fixed point bounds rho.
 * Returned: twice that fixed point (margin), or infinity if there is none. */
double tri_excursion(const H3& c0, const H3& c1, const H3& c2, float area_f) {
    const double u = 1.0 / 16777216.0;
    auto side = [](const H3& a, const H3& b) {
        double x = (double)a.x - b.x, y = (double)a.y - b.y, z = (double)a.z - b.z;
        return std::sqrt(x * x + y * y + z * z);
    };
    const double l = std::max(side(c0, c1), std::max(side(c1, c2), side(c2, c0)));
    const double a = (double)area_f;
    if (!(a > 0.0) || !(a < 1e30) || !(l > 0.0) || !(l < 1e15)) return INFINITY;
    const double A = a - 16.0 * u * l * l;
    if (!(A > 0.0)) return INFINITY;
    const double h = 2.0 * A / l;
    const double G0 = 1.1e-5 * a + 96.0 * u * l * l, Gs = 64.0 * u * l;     /* G(rho) = G0 + Gs*rho */
    if (!(2.0 * Gs / h + 2.0 * Gs / l < 0.25)) return INFINITY;             /* F must flatten well below the diagonal */
    double rho = 0.0;
    for (int it = 0; it < 200; it++) {
        const double G = G0 + Gs * rho;
        const double next = 2.0 * G / h + std::sqrt(2.0 * a * G + G * G) / l;
        if (!(next < 1e30)) return INFINITY;
        if (next <= rho * 1.001) { rho = next; break; }       /* (the slope is below 1/4: what is left is below 0.04 % of rho) */
        rho = next;
    }
    return 2.0 * rho;
}

/* The plane filter of the shadow walk (rt_kernels.h, tri_maybe): g1 = (e2 x n) / |n|^2, g2 = (n x e1) / |n|^2 with
 * e1 = c1 - c0, e2 = c2 - c0, n = e1 x e2 give u = g1.(P - c0), v = g2.(P - c0) with P = c0 + u e1 + v e2 + (a multiple
 * of n); on the triangle u, v >= 0 and u + v <= 1, and |grad u| = |g1|, |grad v| = |g2|, |grad (u+v)| <= |g1| + |g2|.
 * A point within e (the record's hit excursion) of the triangle therefore has u, v >= -e G and u + v <= 1 + e G,
 * G = |g1| + |g2|. The kernel evaluates with g1, g2 rounded to 16-bit floats (8 significant bits: each component off by
 * at most 2^-8 of itself, the dot product by at most 2^-8 sqrt(3) |P - c0| |g|, |P - c0| <= l + e with l the longest
 * side) and in float (P - c0 and the products and sums: a few units of 2^-24 times (|c|_max + l + e) |g|_1). The margin
 * m covers the three with room, is rounded up to a 16-bit float, and is infinite where e is (no bound) or where the
 * triangle is too degenerate for the planes to mean anything. Slot 7 = (g1.x | g1.y << 16, g1.z | g2.x << 16,
 * g2.y | g2.z << 16, m | flags << 16), each value the upper half of its float. */
uint32_t half_float_nearest(double v) {
    float f = (float)v;
    uint32_t b;
    memcpy(&b, &f, 4);
    if ((b & 0x7F800000u) == 0x7F800000u) return b >> 16;              /* inf / NaN as they are */
    b += 0x7FFFu + ((b >> 16) & 1u);                                  /* round to nearest even on bit 16 */
    return b >> 16;
}
uint32_t half_float_up(double v) {                                    /* v >= 0: the next 16-bit float at or above it */
    float f = (float)v;
    if ((double)f < v) f = std::nextafter(f, INFINITY);
    uint32_t b;
    memcpy(&b, &f, 4);
    if ((b & 0x7F800000u) == 0x7F800000u) return 0x7F80u;
    if (b & 0xFFFFu) b += 0x10000u;
    if ((b & 0x7F800000u) == 0x7F800000u) return 0x7F80u;
    return b >> 16;
}
uint4 filter_planes(const float4& q0, const float4& q1, const float4& q2, float exc, uint32_t flags) {
    const double c0[3] = {q0.x, q0.y, q0.z}, c1[3] = {q1.x, q1.y, q1.z}, c2[3] = {q2.x, q2.y, q2.z};
    double e1[3], e2[3], n[3], g1[3], g2[3];
    for (int k = 0; k < 3; k++) { e1[k] = c1[k] - c0[k]; e2[k] = c2[k] - c0[k]; }
    auto cross3 = [](const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
    cross3(e1, e2, n);
    const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
    const uint32_t pass_all = 0x7F80u | (flags << 16);                /* margin = inf: the filter rejects nothing */
    if (!(exc < 1e30f) || !(nn > 1e-280) || !(nn < 1e280)) return make_uint4(0u, 0u, 0u, pass_all);
    cross3(e2, n, g1);
    cross3(n, e1, g2);
    double G1 = 0, G2 = 0, G1s = 0, G2s = 0, cmax = 0, l = 0;
    for (int k = 0; k < 3; k++) {
        g1[k] /= nn; g2[k] /= nn;
        G1 += g1[k] * g1[k]; G2 += g2[k] * g2[k];
        G1s += std::fabs(g1[k]); G2s += std::fabs(g2[k]);
        cmax = std::max(cmax, std::max(std::fabs(c0[k]), std::max(std::fabs(c1[k]), std::fabs(c2[k]))));
    }
    auto side = [](const double* a, const double* b) { return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2])); };
    l = std::max(side(c0, c1), std::max(side(c1, c2), side(c2, c0)));
    const double G = std::sqrt(G1) + std::sqrt(G2), e = (double)exc;
    const double u24 = 1.0 / 16777216.0;
    double m = e * G                                                 /* where an accepted point can lie */
             + (l + e) * G * 1.7320508 / 256.0 * 2.0                 /* g1, g2 rounded to 8 bits (u + v: both) */
             + 16.0 * u24 * (cmax + l + e) * (G1s + G2s)             /* float evaluation */
             + 1e-6;
    m *= 1.05;
    if (!(m < 1e30) || !(G < 1e30)) return make_uint4(0u, 0u, 0u, pass_all);
    uint32_t h[6] = {half_float_nearest(g1[0]), half_float_nearest(g1[1]), half_float_nearest(g1[2]),
                     half_float_nearest(g2[0]), half_float_nearest(g2[1]), half_float_nearest(g2[2])};
    for (uint32_t v : h) if ((v & 0x7F80u) == 0x7F80u) return make_uint4(0u, 0u, 0u, pass_all);      /* a component beyond the 16-bit float range */
    return make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), half_float_up(m) | (flags << 16));
}

/* what the accepted hits below one child of a mesh node stay inside: the union, over its triangle records, of the
 * record's box grown by its excursion (+ a floor for the rounding of P = o + t*d itself); unbounded = some record
 * below has no bound */
struct HitBox {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool unbounded = false;
    void add(const HitBox& o) {
        for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], o.mn[k]); mx[k] = std::max(mx[k], o.mx[k]); }
        unbounded = unbounded || o.unbounded;
    }
};

/* ---- sub-trees inside fat leaves. The reference visits every triangle of a leaf whose box chain the ray passes
 * (AccelerationStruct.cpp:255-283, Mesh.cpp:97-123); with the spatial-median builder stopping at depth 12 a leaf
 * of the gorilla holds 299 of them. Which of them can be HIT is bounded by their HitBoxes, so inside such a leaf
 * the library may skip what the ray cannot reach: the records of a fat leaf are regrouped into short runs under a
 * binary SAH tree of HitBoxes (folded into the 4-wide shadow tree below the leaf, behind the reference's test of
 * the leaf's own box). The walks that need the reference's order inside a leaf (closest hit: the first of equal
 * t wins, Mesh.cpp:114) get each record's original position along. */
#ifndef RT_LEAF_SPLIT
#define RT_LEAF_SPLIT 3          /* leaves with more triangles than this get a sub-tree (6 / 3 / 2: gorilla frame 32.0 / 31.6 / 31.8 ms, cow 20.0 / 19.8 / 20.0, chamber 100.2 / 98.8 / 98.9, default scene 56.4 / 55.8 / 55.7) */
#endif
#ifndef RT_SPLIT_SHARE
#define RT_SPLIT_SHARE 0.02
#endif
#ifndef RT_SUB_RUN
#define RT_SUB_RUN 2             /* triangles per run (sub-tree leaf) */
#endif
#ifndef RT_TOP_SPLIT
#define RT_TOP_SPLIT 4           /* top-level leaves with more objects than this get a sub-tree over their objects */
#endif
struct SubNode { HitBox box[2]; int32_t link[2]; };     /* link >= 0: index into the sub-node array; < 0: a run, -2 - (first record | TRI_SUB) */
struct LeafSplitter {
    const std::vector<HitBox>& tb;          /* per record of the leaf (local index): its HitBox */
    const std::vector<H3>& cen;             /* per record: centre of the triangle's own box */
    std::vector<int>& order;                /* new position -> old local index */
    std::vector<SubNode>& subs;
    size_t first_record;                    /* global record index of the leaf's first triangle */
    std::vector<int>& run_last;             /* new positions that end a run */
    int depth = 0;
    static double half_area(const HitBox& b) {
        if (b.unbounded) return 1e30;
        if (!(b.mn[0] <= b.mx[0])) return 0.0;
        double x = (double)b.mx[0] - b.mn[0], y = (double)b.mx[1] - b.mn[1], z = (double)b.mx[2] - b.mn[2];
        return x * y + y * z + z * x;
    }
    int32_t build(int lo, int hi, HitBox& out, int level) {
        out = HitBox();
        for (int i = lo; i < hi; i++) out.add(tb[(size_t)order[(size_t)i]]);
        if (hi - lo <= RT_SUB_RUN) {
            run_last.push_back(hi - 1);
            return (int32_t)(-2 - (int64_t)((first_record + (size_t)lo) | (size_t)TRI_SUB));
        }
        depth = std::max(depth, level + 1);
        /* binned SAH over the three axes; a degenerate distribution falls back to halving the range */
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = lo; i < hi; i++) {
            const H3& c = cen[(size_t)order[(size_t)i]];
            const float v[3] = {c.x, c.y, c.z};
            for (int k = 0; k < 3; k++) { cmn[k] = std::min(cmn[k], v[k]); cmx[k] = std::max(cmx[k], v[k]); }
        }
        constexpr int NB = 16;
        double best = INFINITY;
        int best_axis = -1, best_bin = 0;
        for (int k = 0; k < 3; k++) {
            const float ext = cmx[k] - cmn[k];
            if (!(ext > 0.f) || !(ext < 1e30f)) continue;
            HitBox bins[NB];
            int cnt[NB] = {0};
            for (int i = lo; i < hi; i++) {
                const size_t o = (size_t)order[(size_t)i];
                const float v = k == 0 ? cen[o].x : k == 1 ? cen[o].y : cen[o].z;
                int b = (int)((v - cmn[k]) / ext * NB);
                b = std::min(std::max(b, 0), NB - 1);
                bins[b].add(tb[o]); cnt[b]++;
            }
            HitBox rb[NB];
            int rc[NB];
            HitBox acc; int n = 0;
            for (int b = NB - 1; b >= 0; b--) { acc.add(bins[b]); n += cnt[b]; rb[b] = acc; rc[b] = n; }
            acc = HitBox(); n = 0;
            for (int b = 0; b + 1 < NB; b++) {
                acc.add(bins[b]); n += cnt[b];
                if (n == 0 || rc[b + 1] == 0) continue;
                const double cost = half_area(acc) * n + half_area(rb[b + 1]) * rc[b + 1];
                if (cost < best) { best = cost; best_axis = k; best_bin = b; }
            }
        }
        int mid;
        if (best_axis < 0) mid = (lo + hi) / 2;
        else {
            const int k = best_axis;
            const float ext = cmx[k] - cmn[k];
            auto in_left = [&](int o) {
                const float v = k == 0 ? cen[(size_t)o].x : k == 1 ? cen[(size_t)o].y : cen[(size_t)o].z;
                int b = (int)((v - cmn[k]) / ext * NB);
                b = std::min(std::max(b, 0), NB - 1);
                return b <= best_bin;
            };
            mid = (int)(std::stable_partition(order.begin() + lo, order.begin() + hi, in_left) - order.begin());
            if (mid == lo || mid == hi) mid = (lo + hi) / 2;
        }
        const size_t me = subs.size();
        subs.emplace_back();
        HitBox bl, br;
        const int32_t l = build(lo, mid, bl, level + 1), r = build(mid, hi, br, level + 1);
        subs[me].box[0] = bl; subs[me].box[1] = br; subs[me].link[0] = l; subs[me].link[1] = r;
        return (int32_t)me;
    }
};

/* ---- quantised shadow nodes (see rt_kernels.h). Every child box of a mesh tree is rounded OUTWARD to
 * 16-bit planes in the frame (qmin, qstep) of the mesh, one extra quantum of padding on each side, and
 * stored with the two links in 32 bytes: x,y,z of the left box as lo | hi << 16, then the right box,
 * then the links. A child without triangles gets lo = 65535, hi = 0 (never entered). The first triangle
 * record of every leaf receives the leaf's exact box (slots 4-5 of its 128-byte line). Returns false if the boxes cannot be used this way: a non-finite or
 * inverted box, or a child box that is not inside its parent's box (then "leaf box hit => every box
 * above it hit" does not hold and the exact walk is used instead). */
bool quantise_mesh(const rt_scene_desc* d, const rt_mesh& m, const std::vector<rt_bvh_node>& nodes, int32_t empty_link, DMesh& dm,
                   std::vector<uint4>& qnodes, std::vector<float4>& lbox /* = the triangle records */,
                   const std::vector<HitBox>& hbox, std::vector<uint8_t>& qunp,
                   const std::vector<SubNode>& subs, size_t sub_lo, size_t sub_base /* index of subs[0] in qnodes */) {
    double gmin[3] = {INFINITY, INFINITY, INFINITY}, gmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t v = 0; v < m.vert_count; v++)
        for (int k = 0; k < 3; k++) {
            double x = d->positions[3 * (m.vert_first + v) + k];
            if (!(std::fabs(x) < 1e30)) return false;
            gmin[k] = std::min(gmin[k], x); gmax[k] = std::max(gmax[k], x);
        }
    /* A child box is widened to hold the hits accepted below it (HitBox), so that pruning by distance on the widened
     * box is exact; where no usable bound exists (further out than one mesh extent) the child keeps the reference
     * box and is marked: the walk then does not prune in the node that holds it. */
    const double reach = std::max(std::max(gmax[0] - gmin[0], gmax[1] - gmin[1]), std::max(gmax[2] - gmin[2], 1e-30));
    const double vmin[3] = {gmin[0], gmin[1], gmin[2]}, vmax[3] = {gmax[0], gmax[1], gmax[2]};
    auto usable = [&](const HitBox& hb) {
        if (hb.unbounded) return false;
        for (int k = 0; k < 3; k++)
            if (hb.mn[k] <= hb.mx[k] && (!((double)hb.mn[k] >= vmin[k] - reach) || !((double)hb.mx[k] <= vmax[k] + reach))) return false;
        return true;
    };
    auto cover = [&](const HitBox& hb) {
        if (!usable(hb)) return;
        for (int k = 0; k < 3; k++)
            if (hb.mn[k] <= hb.mx[k]) { gmin[k] = std::min(gmin[k], (double)hb.mn[k]); gmax[k] = std::max(gmax[k], (double)hb.mx[k]); }
    };
    for (int64_t i = 0; i < m.node_count; i++)
        for (int s = 0; s < 2; s++) cover(hbox[2 * (size_t)(m.node_first + i) + s]);
    for (size_t i = sub_lo; i < subs.size(); i++)
        for (int s = 0; s < 2; s++) cover(subs[i].box[s]);
    float qmin[3], qstep[3];
    for (int k = 0; k < 3; k++) {
        double step = std::max((gmax[k] - gmin[k]) / 65000.0, 1e-30);
        qstep[k] = (float)step;
        if (!(qstep[k] > 0.f)) qstep[k] = 1e-30f;
        qmin[k] = (float)(gmin[k] - 8.0 * (double)qstep[k]);
        dm.qmin[k] = qmin[k]; dm.qstep[k] = qstep[k];
    }
    bool ok = true;
    /* planes as the kernel decodes them: qmin + q*qstep with these very floats (its rounding is covered by the slack there) */
    auto lo_q = [&](float v, int k) -> uint32_t {
        double q = std::floor(((double)v - (double)qmin[k]) / (double)qstep[k]) - 1.0;
        if (!(q >= 0.0)) { if (!(q > -4.0)) ok = false; q = 0.0; }
        if (q > 65535.0) { ok = false; q = 65535.0; }
        return (uint32_t)q;
    };
    auto hi_q = [&](float v, int k) -> uint32_t {
        double q = std::ceil(((double)v - (double)qmin[k]) / (double)qstep[k]) + 1.0;
        if (!(q <= 65535.0)) { ok = false; q = 65535.0; }
        if (q < 0.0) { ok = false; q = 0.0; }
        return (uint32_t)q;
    };
    auto leaf_box = [&](int32_t link, const float* mn, const float* mx) {
        size_t first = (size_t)(-2 - (int64_t)link);
        lbox[RT_TRI_STRIDE * first + 4] = make_float4(mn[0], mn[1], mn[2], mx[0]);
        lbox[RT_TRI_STRIDE * first + 5] = make_float4(mx[1], mx[2], 0.f, 0.f);
    };
    const float ninf[3] = {-INFINITY, -INFINITY, -INFINITY}, pinf[3] = {INFINITY, INFINITY, INFINITY};
    const int32_t groot = nodes.empty() || m.root < 0 ? -1 : (int32_t)(m.root + m.node_first);
    /* the sub-trees of fat leaves: HitBoxes only (what lies below is reached through the leaf's own box) */
    for (size_t i = sub_lo; i < subs.size(); i++) {
        uint32_t w[6];
        int32_t link[2];
        for (int s = 0; s < 2; s++) {
            const HitBox& hb = subs[i].box[s];
            const bool widen = usable(hb) && hb.mn[0] <= hb.mx[0];
            qunp[2 * (sub_base + i) + s] = !widen;
            for (int k = 0; k < 3; k++) w[3 * s + k] = widen ? (lo_q(hb.mn[k], k) | (hi_q(hb.mx[k], k) << 16)) : 0xFFFF0000u;
            link[s] = subs[i].link[s] >= 0 ? (int32_t)(sub_base + (size_t)subs[i].link[s]) : subs[i].link[s];
        }
        qnodes[2 * (sub_base + i)] = make_uint4(w[0], w[1], w[2], w[3]);
        qnodes[2 * (sub_base + i) + 1] = make_uint4(w[4], w[5], (uint32_t)link[0], (uint32_t)link[1]);
    }
    if (m.root < 0) {      /* the whole mesh is one leaf: its triangles are always reached (the root has no box) */
        const rt_bvh_leaf& lf = d->mesh_leaves[m.leaf_first + (-1 - m.root)];
        if (lf.count) leaf_box((int32_t)(-2 - (int64_t)(lf.first + m.tri_first)), ninf, pinf);
        return ok;
    }
    (void)groot;
    for (int64_t i = 0; i < m.node_count; i++) {
        const rt_bvh_node& n = nodes[(size_t)(m.node_first + i)];
        const float* mn[2] = {n.lmin, n.rmin};
        const float* mx[2] = {n.lmax, n.rmax};
        const int32_t link[2] = {n.left, n.right};
        uint32_t w[6];
        for (int s = 0; s < 2; s++) {
            const bool empty = link[s] == empty_link;
            const HitBox& hb = hbox[2 * (size_t)(m.node_first + i) + s];
            const bool widen = usable(hb);
            qunp[2 * (size_t)(m.node_first + i) + s] = !empty && !widen;
            for (int k = 0; k < 3; k++) {
                if (empty) { w[3 * s + k] = 65535u; continue; }            /* lo = 65535, hi = 0 */
                if (!(mn[s][k] <= mx[s][k]) || !(std::fabs(mn[s][k]) < 1e30f) || !(std::fabs(mx[s][k]) < 1e30f)) { ok = false; w[3 * s + k] = 0xFFFF0000u; continue; }
                float lo = mn[s][k], hi = mx[s][k];
                if (widen && hb.mn[k] <= hb.mx[k]) { lo = std::min(lo, hb.mn[k]); hi = std::max(hi, hb.mx[k]); }
                w[3 * s + k] = lo_q(lo, k) | (hi_q(hi, k) << 16);
            }
            if (empty) continue;
            if (link[s] >= 0) {                                             /* nesting: the child's two boxes lie inside this one */
                const rt_bvh_node& ch = nodes[(size_t)link[s]];
                const float* cmn[2] = {ch.lmin, ch.rmin};
                const float* cmx[2] = {ch.lmax, ch.rmax};
                for (int t = 0; t < 2; t++)
                    for (int k = 0; k < 3; k++)
                        if (!(cmn[t][k] >= mn[s][k]) || !(cmx[t][k] <= mx[s][k])) ok = false;
            } else leaf_box(link[s], mn[s], mx[s]);
        }
        uint4 a = make_uint4(w[0], w[1], w[2], w[3]);
        uint4 b = make_uint4(w[4], w[5], (uint32_t)n.left, (uint32_t)n.right);
        qnodes[2 * (size_t)(m.node_first + i)] = a;
        qnodes[2 * (size_t)(m.node_first + i) + 1] = b;
    }
    return ok;
}


/* ---- 4-wide shadow tree: every other level of the (quantised) binary tree folded away. A wide node takes a
 * binary node's two children and keeps replacing the largest inner one by its own two children until four
 * slots are filled; children without triangles are dropped. 64 bytes: twelve packed plane pairs
 * (child-major: c0.x c0.y c0.z c1.x ...), four links; an unused slot has lo = 65535, hi = 0 and LINK_NONE.
 * The boxes are the binary tree's own quantised boxes, so everything said about them still holds. */
struct WideFold {
    const std::vector<uint4>& q;       /* 2 uint4 per binary node, links global */
    int32_t empty_link;
    const float* qstep;
    std::vector<uint4>& out;           /* 4 uint4 per wide node */
    const std::vector<uint8_t>& unp;   /* per binary child: hits below it have no usable bound (quantise_mesh) */
    int32_t link_flags = 0;            /* WIDE_SUB while folding the sub-tree of a fat leaf */
    int depth = 0;
    struct Cand { int32_t link; uint32_t w[3]; bool unp; };
    void children(int32_t node, Cand* c, int& n) const {
        const uint4 a = q[2 * (size_t)node], b = q[2 * (size_t)node + 1];
        Cand l = {(int32_t)b.z, {a.x, a.y, a.z}, unp[2 * (size_t)node] != 0}, r = {(int32_t)b.w, {a.w, b.x, b.y}, unp[2 * (size_t)node + 1] != 0};
        n = 0;
        if (l.link != empty_link) c[n++] = l;
        if (r.link != empty_link) c[n++] = r;
    }
    float area(const Cand& c) const {
        float e[3];
        for (int k = 0; k < 3; k++) e[k] = ((float)(c.w[k] >> 16) - (float)(c.w[k] & 0xFFFFu)) * qstep[k];
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
    int32_t build(int32_t node, int level) {
        Cand cand[4];
        int n = 0;
        children(node, cand, n);
        while (n > 0 && n < 4) {
            int pick = -1;
            float best = -1.f;
            for (int i = 0; i < n; i++)
                if (cand[i].link >= 0) { float a = area(cand[i]); if (a > best) { best = a; pick = i; } }
            if (pick < 0) break;
            Cand sub[2];
            int m = 0;
            children(cand[pick].link, sub, m);
            for (int i = pick; i + 1 < n; i++) cand[i] = cand[i + 1];
            n--;
            for (int i = 0; i < m; i++) cand[n++] = sub[i];
        }
        if (n == 0) return LINK_NONE;
        const size_t w = out.size() / 4;
        out.resize(out.size() + 4, make_uint4(0u, 0u, 0u, 0u));
        depth = std::max(depth, level + 1);
        uint32_t words[16];
        for (int i = 0; i < 12; i++) words[i] = 65535u;
        for (int i = 12; i < 16; i++) words[i] = (uint32_t)LINK_NONE;
        int used = 0;
        bool unpruned = false;             /* a slot of this node holds hits without a usable bound: its link says so */
        for (int i = 0; i < n; i++) {
            int32_t link = cand[i].link;
            if (link >= 0) { link = build(link, level + 1); if (link == LINK_NONE) continue; }
            for (int k = 0; k < 3; k++) words[3 * used + k] = cand[i].w[k];
            words[12 + used] = (uint32_t)link;
            unpruned = unpruned || cand[i].unp;
            used++;
        }
        if (used == 0) { out.resize(4 * w); return LINK_NONE; }
        for (int r = 0; r < 4; r++) out[4 * w + r] = make_uint4(words[4 * r], words[4 * r + 1], words[4 * r + 2], words[4 * r + 3]);
        return (int32_t)w | (unpruned ? (int32_t)WIDE_UNPRUNED : 0) | link_flags;
    }
};

/* The links must form a tree no deeper than declared: the device stacks are sized by the declared depth
 * and the kernels loop until the links run out, so a cycle or a shared child would never end.
 * (Links are already known to be in range.) Returns what is wrong, or nullptr. */
const char* tree_defect(const rt_bvh_node* nodes, int64_t n_nodes, int32_t root, int32_t depth) {
    if (depth < 0) return "negative tree depth";
    if (root < 0) return nullptr;                       /* a single leaf */
    std::vector<uint8_t> seen((size_t)n_nodes, 0);
    std::vector<std::pair<int32_t, int32_t>> todo(1, std::make_pair(root, 0));
    while (!todo.empty()) {
        const std::pair<int32_t, int32_t> cur = todo.back();
        todo.pop_back();
        if (seen[(size_t)cur.first]) return "a node has two parents or lies on a cycle";
        seen[(size_t)cur.first] = 1;
        if (cur.second >= depth) return "tree deeper than its declared depth";
        const rt_bvh_node& n = nodes[cur.first];
        if (n.left >= 0) todo.push_back(std::make_pair(n.left, cur.second + 1));
        if (n.right >= 0) todo.push_back(std::make_pair(n.right, cur.second + 1));
    }
    return nullptr;
}

int check_desc(rt_ctx* c, const rt_scene_desc* d) {
    /* counts and the arrays they size, before anything is read through them */
    if (d->n_materials < 0 || d->n_lights < 0 || d->n_squares < 0 || d->n_spheres < 0 || d->n_meshes < 0 || d->n_verts < 0 || d->n_tris < 0 ||
        d->n_mesh_nodes < 0 || d->n_mesh_leaves < 0 || d->n_top_nodes < 0 || d->n_top_leaves < 0 || d->n_top_items < 0 || d->n_textures < 0)
        return fail(c, RT_ERR_ARG, "negative count in the scene description");
    if ((d->n_materials && !d->materials) || (d->n_lights && !d->lights) || (d->n_squares && !d->squares) || (d->n_spheres && !d->spheres) ||
        (d->n_meshes && !d->meshes) || (d->n_verts && (!d->positions || !d->normals)) || (d->n_tris && (!d->tri_indices || !d->mesh_items)) ||
        (d->n_mesh_nodes && !d->mesh_nodes) || (d->n_mesh_leaves && !d->mesh_leaves) || (d->n_top_nodes && !d->top_nodes) ||
        (d->n_top_leaves && !d->top_leaves) || (d->n_top_items && !d->top_items))
        return fail(c, RT_ERR_ARG, "null array with a non-zero count in the scene description");
    if ((int64_t)d->n_squares + d->n_spheres + d->n_meshes > 0x3FFFFFFF) return fail(c, RT_ERR_LIMIT, "too many objects");
    if (d->n_lights > RT_MAX_LIGHTS) return fail(c, RT_ERR_LIMIT, "more than %d lights", RT_MAX_LIGHTS);
    if (d->n_materials <= 0 && (d->n_squares + d->n_spheres + d->n_meshes) > 0) return fail(c, RT_ERR_ARG, "no materials");
    if (d->n_textures < 0 || (d->n_textures > 0 && !d->textures)) return fail(c, RT_ERR_ARG, "textures");
    for (int i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        if (t.width <= 0 || t.height <= 0 || t.channels < 3 || !t.texels) return fail(c, RT_ERR_ARG, "texture %d: needs a size, >= 3 channels and texels", i);
    }
    for (int i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        if (m.texture < 0 || m.texture > d->n_textures || m.normal_map < 0 || m.normal_map > d->n_textures)
            return fail(c, RT_ERR_ARG, "material %d: texture / normal map index out of range", i);
    }
    if (d->top_depth > RT_MAX_TOP_DEPTH) return fail(c, RT_ERR_LIMIT, "top-level tree deeper than %d", RT_MAX_TOP_DEPTH);
    int n_obj = d->n_squares + d->n_spheres + d->n_meshes;
    auto link_ok = [](int32_t link, int64_t n_nodes, int64_t n_leaves) {
        return link >= 0 ? (int64_t)link < n_nodes : (int64_t)(-1 - link) < n_leaves;
    };
    if (n_obj > 0 && !link_ok(d->top_root, d->n_top_nodes, d->n_top_leaves)) return fail(c, RT_ERR_ARG, "top_root out of range");
    for (int i = 0; i < d->n_top_nodes; i++)
        if (!link_ok(d->top_nodes[i].left, d->n_top_nodes, d->n_top_leaves) || !link_ok(d->top_nodes[i].right, d->n_top_nodes, d->n_top_leaves))
            return fail(c, RT_ERR_ARG, "top node %d: child link out of range", i);
    if (n_obj > 0)
        if (const char* why = tree_defect(d->top_nodes, d->n_top_nodes, d->top_root, d->top_depth)) return fail(c, RT_ERR_ARG, "top-level tree: %s", why);
    for (int i = 0; i < d->n_top_leaves; i++)
        if ((int64_t)d->top_leaves[i].first + d->top_leaves[i].count > d->n_top_items) return fail(c, RT_ERR_ARG, "top leaf %d out of range", i);
    for (int i = 0; i < d->n_top_items; i++)
        if ((int)d->top_items[i] >= n_obj) return fail(c, RT_ERR_ARG, "top item %d: object ordinal out of range", i);
    for (int i = 0; i < d->n_squares; i++) if (d->squares[i].material < 0 || d->squares[i].material >= d->n_materials) return fail(c, RT_ERR_ARG, "square material");
    for (int i = 0; i < d->n_spheres; i++) if (d->spheres[i].material < 0 || d->spheres[i].material >= d->n_materials) return fail(c, RT_ERR_ARG, "sphere material");
    for (int mi = 0; mi < d->n_meshes; mi++) {
        const rt_mesh& m = d->meshes[mi];
        if (m.material < 0 || m.material >= d->n_materials) return fail(c, RT_ERR_ARG, "mesh material");
        if (m.depth > RT_MAX_TREE_DEPTH) return fail(c, RT_ERR_LIMIT, "mesh %d: tree deeper than %d", mi, RT_MAX_TREE_DEPTH);
        if (m.vert_first < 0 || m.vert_first + m.vert_count > d->n_verts || m.tri_first < 0 || m.tri_first + m.tri_count > d->n_tris ||
            m.node_first < 0 || m.node_first + m.node_count > d->n_mesh_nodes || m.leaf_first < 0 || m.leaf_first + m.leaf_count > d->n_mesh_leaves)
            return fail(c, RT_ERR_ARG, "mesh %d: range out of bounds", mi);
        if (m.tri_count <= 0 || !link_ok(m.root, m.node_count, m.leaf_count)) return fail(c, RT_ERR_ARG, "mesh %d: bad root", mi);
        for (int64_t i = 0; i < m.node_count; i++) {
            const rt_bvh_node& n = d->mesh_nodes[m.node_first + i];
            if (!link_ok(n.left, m.node_count, m.leaf_count) || !link_ok(n.right, m.node_count, m.leaf_count))
                return fail(c, RT_ERR_ARG, "mesh %d node %lld: child link out of range", mi, (long long)i);
        }
        for (int64_t i = 0; i < m.leaf_count; i++) {
            const rt_bvh_leaf& lf = d->mesh_leaves[m.leaf_first + i];
            if ((int64_t)lf.first + lf.count > m.tri_count) return fail(c, RT_ERR_ARG, "mesh %d leaf %lld out of range", mi, (long long)i);
        }
        if (const char* why = tree_defect(d->mesh_nodes + m.node_first, m.node_count, m.root, m.depth))
            return fail(c, RT_ERR_ARG, "mesh %d: %s", mi, why);
        for (int64_t i = 0; i < m.tri_count; i++) {
            if (d->mesh_items[m.tri_first + i] >= (uint64_t)m.tri_count) return fail(c, RT_ERR_ARG, "mesh %d: item out of range", mi);
            for (int k = 0; k < 3; k++)
                if (d->tri_indices[3 * (m.tri_first + i) + k] >= (uint64_t)m.vert_count) return fail(c, RT_ERR_ARG, "mesh %d: vertex index out of range", mi);
        }
    }
    if (d->top_depth > 60 || d->n_top_leaves > 4000) return fail(c, RT_ERR_LIMIT, "top-level tree too deep or too wide (k_shadow packs its top-level state)");
    if (d->n_tris > 0x0FFFFFF0ll || d->n_mesh_nodes > 0x7FFFFFF0ll || d->n_mesh_leaves > 0x7FFFFFF0ll) return fail(c, RT_ERR_LIMIT, "scene too large for 32-bit links");
    return RT_OK;
}

void free_scene(rt_ctx* c) {
    for (DevBuf& b : c->scene_bufs) release(b);
    c->scene_bufs.clear();
    c->has_scene = false;
}

}  // namespace

/* multi-GPU gather (rt_init with several devices): the pixels of a device's tiles, tile after tile, packed into a slab on the
 * device that rendered them and scattered into the frame on the gathering device. `first` of a tile counts pixels here. */
__global__ void __launch_bounds__(256) k_tiles_pack(const float* __restrict__ frame, int img_w, const TileD* tiles, int n_tiles, uint32_t n_pix, float* __restrict__ slab) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    FrameD one{}; one.spp = 1;
    int x, y, s_;
    locate_sample(tiles, n_tiles, one, i, x, y, s_);
    const size_t at = ((size_t)y * img_w + x) * 3;
    slab[3 * (size_t)i] = frame[at]; slab[3 * (size_t)i + 1] = frame[at + 1]; slab[3 * (size_t)i + 2] = frame[at + 2];
}
__global__ void __launch_bounds__(256) k_tiles_scatter(float* __restrict__ frame, int img_w, const TileD* tiles, int n_tiles, uint32_t n_pix, const float* __restrict__ slab) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix) return;
    FrameD one{}; one.spp = 1;
    int x, y, s_;
    locate_sample(tiles, n_tiles, one, i, x, y, s_);
    const size_t at = ((size_t)y * img_w + x) * 3;
    frame[at] = slab[3 * (size_t)i]; frame[at + 1] = slab[3 * (size_t)i + 1]; frame[at + 2] = slab[3 * (size_t)i + 2];
}

namespace {
/* the deal of tiles to devices: round-robin along a tile row with a hashed offset per row (the same rule as
 * ray-tracing_amd/dist.py: cost per pixel is heavy-tailed, and 120 tiles per row over 2/4/8 devices must not become column stripes) */
inline uint32_t tile_row_offset(uint32_t ty, uint32_t world) {
    uint32_t h = ty * 0x9E3779B1u + 0x7F4A7C15u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    return h % world;
}
inline int tile_owner(int x0, int y0, int world) { return (int)(((uint32_t)(x0 / 32) + tile_row_offset((uint32_t)(y0 / 32), (uint32_t)world)) % (uint32_t)world); }

template <typename F>
int on_all(rt_ctx* c, F body) {           /* body(sub) on every device, one host thread each; the first error wins */
    std::vector<int> rc(c->subs.size(), RT_OK);
    std::vector<std::thread> th;
    for (size_t k = 0; k < c->subs.size(); k++) th.emplace_back([&, k] { rc[k] = body(c->subs[k]); });
    for (auto& t : th) t.join();
    for (size_t k = 0; k < rc.size(); k++)
        if (rc[k]) return fail(c, rc[k], "device %d: %s", c->subs[k]->device, c->subs[k]->err.c_str());
    return RT_OK;
}
}  // namespace

extern "C" {

uint32_t rt_rng(uint32_t seed, uint32_t pixel, uint32_t sample, uint64_t code, uint32_t light, uint32_t i, uint32_t which) {
    return rt_rng_impl(seed, pixel, sample, code, light, i, which);
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->err.c_str() : "null context"; }

int rt_tile_owner(int x0, int y0, int n_devices) { return (n_devices < 1 || x0 < 0 || y0 < 0) ? RT_ERR_ARG : tile_owner(x0, y0, n_devices); }

double rt_hit_excursion(const float v0[3], const float v1[3], const float v2[3], float amax) {
    const float s = 1.0001f;                                   /* the record of rt_upload_scene */
    H3 c0 = {s * v0[0], s * v0[1], s * v0[2]}, c1 = {s * v1[0], s * v1[1], s * v1[2]}, c2 = {s * v2[0], s * v2[1], s * v2[2]};
    const float area = hlen(hcross(hsub(c1, c0), hsub(c2, c0))) / 2.f;
    return (double)(float)(tri_excursion(c0, c1, c2, area) + 1e-5 * (double)amax + 1e-6);
}

/* the plane filter of the shadow walk (rt_kernels.h, tri_maybe) evaluated on the host with the device's float operations
 * in the device's order, on the record rt_upload_scene would make of the triangle: 1 = the record goes on to the
 * reference's test, 0 = rejected. Host only; tests/test_hit_bounds.py holds it against what the oracle accepts. */
int rt_filter_passes(const float v0[3], const float v1[3], const float v2[3], float amax, const float o[3], const float d[3]) {
    const float s = 1.0001f;
    H3 c0 = {s * v0[0], s * v0[1], s * v0[2]}, c1 = {s * v1[0], s * v1[1], s * v1[2]}, c2 = {s * v2[0], s * v2[1], s * v2[2]};
    H3 nn = hcross(hsub(c1, c0), hsub(c2, c0));
    const float norm = hlen(nn);
    H3 n = {nn.x / norm, nn.y / norm, nn.z / norm};
    const float area = norm / 2.f, D = hdot(c0, n);
    const float exc = (float)(tri_excursion(c0, c1, c2, area) + 1e-5 * (double)amax + 1e-6);
    const uint4 q7 = filter_planes(make_float4(c0.x, c0.y, c0.z, D), make_float4(c1.x, c1.y, c1.z, area), make_float4(c2.x, c2.y, c2.z, 0.f), exc, 0u);
    auto asf = [](uint32_t b) { float f; memcpy(&f, &b, 4); return f; };
    const H3 O = {o[0], o[1], o[2]}, Dr = {d[0], d[1], d[2]};
    const float dn = hdot(Dr, n);
    if (dn == 0) return 0;
    const float t = (D - hdot(O, n)) / dn;
    const H3 P = {O.x + t * Dr.x, O.y + t * Dr.y, O.z + t * Dr.z};
    const H3 v = hsub(P, c0);
    const H3 g1 = {asf(q7.x << 16), asf(q7.x & 0xFFFF0000u), asf(q7.y << 16)}, g2 = {asf(q7.y & 0xFFFF0000u), asf(q7.z << 16), asf(q7.z & 0xFFFF0000u)};
    const float m = asf(q7.w << 16);
    const float a = hdot(v, g1), b = hdot(v, g2);
    return !(a < -m || b < -m || a + b > 1.0f + m) ? 1 : 0;
}

int rt_uses_quantised_shadow_nodes(const rt_ctx* c) {
    if (!c) return RT_ERR_ARG;
    if (!c->subs.empty()) return rt_uses_quantised_shadow_nodes(c->subs[0]);
    if (!c->has_scene) return RT_ERR_STATE;
    return c->sc.quant_ok ? 1 : 0;
}

int rt_init(const int* device_ids, int n_devices, rt_ctx** out) {
    if (!out) return RT_ERR_ARG;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return RT_ERR_ARG;
    if (n_devices > 1) {
        /* several GPUs behind one context: a full context per device (scene replicated), tiles dealt in rt_render_tile, every
         * device driven by its own host thread in rt_flush, the frame gathered on device_ids[0] */
        if (!device_ids) return RT_ERR_ARG;
        rt_ctx* p = nullptr;
        try { p = new rt_ctx(); } catch (...) { return RT_ERR_NOMEM; }
        p->device = device_ids[0];
        for (int k = 0; k < n_devices; k++) {
            rt_ctx* sub = nullptr;
            int rc = rt_init(device_ids + k, 1, &sub);
            if (rc) { for (rt_ctx* q : p->subs) rt_shutdown(q); delete p; return rc; }
            p->subs.push_back(sub);
        }
        p->sub_tiles.resize((size_t)n_devices);
        p->sub_tile_keys.resize((size_t)n_devices);
        /* the gather pulls every other device's slab over xGMI: ask for peer access from the gathering device (and back); without it
         * hipMemcpyPeerAsync still works but stages through host memory -- said once, not silently */
        for (int k = 1; k < n_devices; k++) {
            const int d0 = device_ids[0], dk = device_ids[k];
            if (dk == d0) continue;
            bool done = false;
            for (int j = 1; j < k; j++) if (device_ids[j] == dk) done = true;
            if (done) continue;
            int can = 0, back = 0;
            bool ok = hipDeviceCanAccessPeer(&can, d0, dk) == hipSuccess && can && hipDeviceCanAccessPeer(&back, dk, d0) == hipSuccess && back;
            if (ok) {
                hipError_t e1 = hipSetDevice(d0) == hipSuccess ? hipDeviceEnablePeerAccess(dk, 0) : hipErrorInvalidDevice;
                hipError_t e2 = hipSetDevice(dk) == hipSuccess ? hipDeviceEnablePeerAccess(d0, 0) : hipErrorInvalidDevice;
                ok = (e1 == hipSuccess || e1 == hipErrorPeerAccessAlreadyEnabled) && (e2 == hipSuccess || e2 == hipErrorPeerAccessAlreadyEnabled);
                (void)hipGetLastError();
            }
            if (!ok) fprintf(stderr, "[rtgpu] no peer access between devices %d and %d: the frame gather stages their slab through host memory\n", d0, dk);
            else if (getenv("RT_DEBUG")) fprintf(stderr, "[rtgpu] peer access between devices %d and %d enabled\n", d0, dk);
        }
        (void)hipSetDevice(device_ids[0]);
        *out = p;
        return RT_OK;
    }
    int dev = device_ids ? device_ids[0] : 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || dev < 0 || dev >= count) return RT_ERR_DEVICE;
    rt_ctx* c = new rt_ctx();
    c->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess || hipEventCreate(&c->ev_frame_a) != hipSuccess ||
        hipEventCreate(&c->ev_frame_b) != hipSuccess || hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return RT_ERR_DEVICE;
    }
    {   /* set-up that must not happen later: the pinned read-back block and this device's address of the static scratch */
        void* sp = nullptr;
        if (hipHostMalloc((void**)&c->h_ctr, sizeof(Counters), hipHostMallocDefault) != hipSuccess ||
            hipHostGetDevicePointer((void**)&c->d_hctr, c->h_ctr, 0) != hipSuccess || !c->d_hctr ||
            hipGetSymbolAddress(&sp, HIP_SYMBOL(rt_static_scratch)) != hipSuccess || !sp) {
            rt_shutdown(c);
            return RT_ERR_DEVICE;
        }
        c->static_scratch = static_cast<unsigned char*>(sp);
    }
    const char* b = getenv("RT_BATCH");
    if (b && atoi(b) >= 1024) c->batch = (uint32_t)atoi(b);
    if (const char* e = getenv("RT_MEM_BUDGET_GB")) { if (atof(e) >= 1.0) c->mem_budget = (unsigned long long)(atof(e) * 1073741824.0); }
    c->debug = getenv("RT_DEBUG") != nullptr;
    c->count_steps = getenv("RT_COUNT") != nullptr;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            c->persistent_blocks = prop.multiProcessorCount * 16;    /* measured on C5: 16 one-wave blocks per CU (4 per SIMD) beat 12/14/18/20; more rays in flight thrash L1/L2 */
        c->separate_raygen = getenv("RT_SEPARATE_RAYGEN") != nullptr;
        const char* pb = getenv("RT_PERSISTENT_BLOCKS");
        if (pb && atoi(pb) > 0) c->persistent_blocks = atoi(pb);
        if (const char* e = getenv("RT_SPREAD_LANES")) c->spread_lanes = strtoull(e, nullptr, 10);
        c->no_tail_swap = getenv("RT_NO_TAIL_SWAP") != nullptr;
    }
    *out = c;
    return RT_OK;
}

void rt_shutdown(rt_ctx* c) {
    if (!c) return;
    if (!c->subs.empty()) {
        (void)hipSetDevice(c->device);
        release(c->mg_slab); release(c->mg_tiles);
        for (rt_ctx* q : c->subs) rt_shutdown(q);
        delete c;
        return;
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    for (DevBuf* b : {&c->acc, &c->nf, &c->prim, &c->rgb, &c->stack, &c->child, &c->drec, &c->dcode, &c->cnt, &c->ctr, &c->tilebuf, &c->srays, &c->dtex, &c->ovf, &c->mg_slab, &c->mg_tiles}) release(*b);
    if (c->h_ctr) (void)hipHostFree(c->h_ctr);
    for (EventPair& e : c->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    (void)hipEventDestroy(c->ev_frame_a);
    (void)hipEventDestroy(c->ev_frame_b);
    if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

static int upload_scene_impl(rt_ctx* c, const rt_scene_desc* d);
int rt_upload_scene(rt_ctx* c, const rt_scene_desc* d) {
    if (!c || !d) return RT_ERR_ARG;
    if (!c->subs.empty()) {
        int rc = on_all(c, [&](rt_ctx* q) { return rt_upload_scene(q, d); });
        c->has_scene = rc == RT_OK;
        return rc;
    }
    /* the host-side preparation sizes vectors by the caller's counts: no exception may cross the ABI */
    try { return upload_scene_impl(c, d); }
    catch (const std::bad_alloc&) { free_scene(c); return fail(c, RT_ERR_NOMEM, "out of host memory while preparing the scene"); }
    catch (...) { free_scene(c); return fail(c, RT_ERR_ARG, "scene preparation failed"); }
}
static int upload_scene_impl(rt_ctx* c, const rt_scene_desc* d) {
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = check_desc(c, d);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_scene(c);
    DScene sc{};

    /* triangles in leaf order with the per-test quantities of Triangle.h precomputed */
    std::vector<float4> tri(((size_t)d->n_tris + 1) * RT_TRI_STRIDE, make_float4(0.f, 0.f, 0.f, 0.f));   /* +1: the record empty leaves point at */
    /* word 2.w of a record: bit 0 = last record of its run, bit 1 = last record of its (reference) leaf */
    auto flag_record = [&](size_t g, uint32_t bits) {
        uint32_t f;
        memcpy(&f, &tri[RT_TRI_STRIDE * g + 2].w, 4);
        f |= bits;
        memcpy(&tri[RT_TRI_STRIDE * g + 2].w, &f, 4);
    };
    flag_record((size_t)d->n_tris, 3u);
    std::vector<uint4> ids((size_t)d->n_tris);
    std::vector<rt_bvh_node> nodes(d->mesh_nodes, d->mesh_nodes + d->n_mesh_nodes);
    std::vector<DMesh> meshes((size_t)d->n_meshes);
    /* shadow walk (rt_kernels.h, "quantised shadow nodes"): 32-byte copies of the nodes + the exact box of every leaf */
    std::vector<uint4> qnodes((size_t)d->n_mesh_nodes * 2, make_uint4(0u, 0u, 0u, 0u));
    std::vector<uint4> wnodes;                                              /* 4-wide shadow tree, all meshes */
    std::vector<HitBox> hbox((size_t)d->n_mesh_nodes * 2);
    std::vector<uint8_t> qunp((size_t)d->n_mesh_nodes * 2, 0);
    std::vector<float> texc;
    std::vector<SubNode> subs;                                              /* sub-trees of fat leaves, all meshes */
    std::vector<HitBox> mesh_hit((size_t)d->n_meshes);                      /* per mesh: what every accepted hit on it stays inside */
    /* the sub-tree variants of the kernels cost a few per cent where no leaf is fat, so a handful of fat leaves
     * (C5 has five among millions) is not worth them: at least RT_SPLIT_SHARE of the triangles must sit in fat leaves */
    bool split_leaves = false;
    {
        int64_t fat = 0;
        for (int64_t i = 0; i < d->n_mesh_leaves; i++) if (d->mesh_leaves[i].count > (uint32_t)RT_LEAF_SPLIT) fat += d->mesh_leaves[i].count;
        split_leaves = !getenv("RT_NO_LEAF_SPLIT") && (double)fat >= RT_SPLIT_SHARE * (double)d->n_tris && fat > 0;
        if (getenv("RT_FORCE_LEAF_SPLIT")) split_leaves = fat > 0;       /* test knob */
    }
    bool quant_ok = true;
    int max_mesh_depth = 0, max_wide_depth = 0, max_sub_depth = 0, max_wide_sub_depth = 0, max_wide_top_depth = 0;
    bool wide_paths_ok = true;      /* every leaf's path from its root fits a 64-bit word (see the path words below) */
    for (int mi = 0; mi < d->n_meshes; mi++) {
        const rt_mesh& m = d->meshes[mi];
        max_mesh_depth = std::max(max_mesh_depth, m.depth);
        float amax = 0.f;
        for (int64_t v = 0; v < 3 * m.vert_count; v++) amax = std::max(amax, std::fabs(d->positions[3 * m.vert_first + v]));
        texc.assign((size_t)m.tri_count, 0.f);
        parallel_for(m.tri_count, [&](int64_t j_lo, int64_t j_hi) {
        for (int64_t j = j_lo; j < j_hi; j++) {
            uint32_t t = d->mesh_items[m.tri_first + j];
            const uint32_t* iv = d->tri_indices + 3 * (m.tri_first + t);
            H3 c0, c1, c2;
            {
                const float* p0 = d->positions + 3 * (m.vert_first + iv[0]);
                const float* p1 = d->positions + 3 * (m.vert_first + iv[1]);
                const float* p2 = d->positions + 3 * (m.vert_first + iv[2]);
                const float s = 1.0001f;                           /* Mesh.cpp:107-109 */
                c0 = {s * p0[0], s * p0[1], s * p0[2]};
                c1 = {s * p1[0], s * p1[1], s * p1[2]};
                c2 = {s * p2[0], s * p2[1], s * p2[2]};
            }
            H3 nn = hcross(hsub(c1, c0), hsub(c2, c0));            /* Triangle.h:28-33 */
            float norm = hlen(nn);
            H3 n = {nn.x / norm, nn.y / norm, nn.z / norm};
            float area = norm / 2.f;
            float D = hdot(c0, n);                                 /* Triangle.h:58 */
            size_t g = (size_t)(m.tri_first + j);
            tri[RT_TRI_STRIDE * g + 0] = make_float4(c0.x, c0.y, c0.z, D);
            tri[RT_TRI_STRIDE * g + 1] = make_float4(c1.x, c1.y, c1.z, area);
            tri[RT_TRI_STRIDE * g + 2] = make_float4(c2.x, c2.y, c2.z, 0.f);
            tri[RT_TRI_STRIDE * g + 3] = make_float4(n.x, n.y, n.z, 0.f);
            ids[g] = make_uint4((uint32_t)(m.vert_first + iv[0]), (uint32_t)(m.vert_first + iv[1]), (uint32_t)(m.vert_first + iv[2]), t);
            texc[(size_t)j] = (float)(tri_excursion(c0, c1, c2, area) + 1e-5 * (double)amax + 1e-6);
        }
        });
        auto glob = [&](int32_t link) {
            if (link >= 0) return (int32_t)(link + m.node_first);
            const rt_bvh_leaf& lf = d->mesh_leaves[m.leaf_first + (-1 - link)];
            return lf.count == 0 ? (int32_t)(-2 - d->n_tris) : (int32_t)(-2 - (int64_t)(lf.first + m.tri_first));
        };
        for (int64_t i = 0; i < m.node_count; i++) {
            rt_bvh_node& n = nodes[(size_t)(m.node_first + i)];
            n.left = glob(n.left);
            n.right = glob(n.right);
        }
        const size_t sub_lo = subs.size();
        std::vector<std::pair<size_t, int32_t>> gates;          /* fat leaves: first record, root of the sub-tree (index into subs) */
        for (int64_t i = 0; i < m.leaf_count; i++) {
            const rt_bvh_leaf& lf = d->mesh_leaves[m.leaf_first + i];
            if (!lf.count) continue;
            const size_t g0 = (size_t)(m.tri_first + lf.first), n = lf.count;
            flag_record(g0 + n - 1, 3u);
            auto set_position = [&](size_t g, uint32_t pos) { memcpy(&tri[RT_TRI_STRIDE * g + 3].w, &pos, 4); };
            if (!split_leaves || n <= (size_t)RT_LEAF_SPLIT) {
                for (size_t p = 0; p < n; p++) set_position(g0 + p, (uint32_t)p);
                continue;
            }
            /* a fat leaf: runs of records under a sub-tree; every record keeps its position in the reference's order */
            std::vector<HitBox> tb(n);
            std::vector<H3> cen(n);
            std::vector<int> order(n);
            for (size_t p = 0; p < n; p++) {
                const float4* q = &tri[RT_TRI_STRIDE * (g0 + p)];
                const float e = texc[g0 + p - (size_t)m.tri_first];
                const float cc[3][3] = {{q[0].x, q[0].y, q[0].z}, {q[1].x, q[1].y, q[1].z}, {q[2].x, q[2].y, q[2].z}};
                float lo[3], hi[3];
                for (int k = 0; k < 3; k++) {
                    lo[k] = std::min(cc[0][k], std::min(cc[1][k], cc[2][k]));
                    hi[k] = std::max(cc[0][k], std::max(cc[1][k], cc[2][k]));
                }
                cen[p] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
                if (!(e < 1e30f)) tb[p].unbounded = true;
                else for (int k = 0; k < 3; k++) { tb[p].mn[k] = lo[k] - e; tb[p].mx[k] = hi[k] + e; }
                order[p] = (int)p;
            }
            std::vector<int> run_last;
            LeafSplitter ls{tb, cen, order, subs, g0, run_last};
            HitBox all;
            const int32_t root = ls.build(0, (int)n, all, 0);
            max_sub_depth = std::max(max_sub_depth, ls.depth);
            gates.emplace_back(g0, root);
            std::vector<float4> rec(n * RT_TRI_STRIDE);
            std::vector<uint4> rid(n);
            std::vector<float> rex(n);
            for (size_t p = 0; p < n; p++) {
                const size_t o = g0 + (size_t)order[p];
                std::copy(tri.begin() + RT_TRI_STRIDE * o, tri.begin() + RT_TRI_STRIDE * (o + 1), rec.begin() + RT_TRI_STRIDE * p);
                rid[p] = ids[o];
                rex[p] = texc[o - (size_t)m.tri_first];
            }
            std::copy(rec.begin(), rec.end(), tri.begin() + RT_TRI_STRIDE * g0);
            std::copy(rid.begin(), rid.end(), ids.begin() + g0);
            std::copy(rex.begin(), rex.end(), texc.begin() + (g0 - (size_t)m.tri_first));
            for (size_t p = 0; p < n; p++) {
                set_position(g0 + p, (uint32_t)order[p]);
                uint32_t zero = 0u;
                memcpy(&tri[RT_TRI_STRIDE * (g0 + p) + 2].w, &zero, 4);
            }
            for (int p : run_last) flag_record(g0 + (size_t)p, 1u);
            flag_record(g0 + n - 1, 3u);
        }
        /* slot 7 of every record: the plane filter of the shadow walk (rt_kernels.h, tri_maybe) and a copy of the flags */
        parallel_for(m.tri_count, [&](int64_t j_lo, int64_t j_hi) {
            for (int64_t j = j_lo; j < j_hi; j++) {
                float4* q = &tri[RT_TRI_STRIDE * (size_t)(m.tri_first + j)];
                uint32_t flags;
                memcpy(&flags, &q[2].w, 4);
                const uint4 f = filter_planes(q[0], q[1], q[2], texc[(size_t)j], flags);
                memcpy(&q[7], &f, 16);
            }
        });
        meshes[mi].root = glob(m.root);
        meshes[mi].material = m.material;
        meshes[mi].amax = amax;
        meshes[mi].pad = 0; meshes[mi].wroot = LINK_NONE;
        /* per child of every node: the box the accepted hits below it stay inside (see tri_excursion) */
        const int32_t empty_link = (int32_t)(-2 - d->n_tris);
        std::function<HitBox(int32_t)> below = [&](int32_t link) -> HitBox {
            HitBox b;
            if (link == empty_link) return b;
            if (link < 0) {
                size_t g = (size_t)(-2 - (int64_t)link);
                for (bool last = false; !last; g++) {
                    const float4 *q = &tri[RT_TRI_STRIDE * g];
                    uint32_t flag;
                    memcpy(&flag, &q[2].w, 4);
                    last = (flag & 2u) != 0u;
                    const float e = texc[g - (size_t)m.tri_first];
                    if (!(e < 1e30f)) { b.unbounded = true; continue; }
                    const float c[3][3] = {{q[0].x, q[0].y, q[0].z}, {q[1].x, q[1].y, q[1].z}, {q[2].x, q[2].y, q[2].z}};
                    for (int k = 0; k < 3; k++) {
                        b.mn[k] = std::min(b.mn[k], std::min(c[0][k], std::min(c[1][k], c[2][k])) - e);
                        b.mx[k] = std::max(b.mx[k], std::max(c[0][k], std::max(c[1][k], c[2][k])) + e);
                    }
                }
                return b;
            }
            const rt_bvh_node& n = nodes[(size_t)link];
            HitBox l = below(n.left), r = below(n.right);
            hbox[2 * (size_t)link] = l; hbox[2 * (size_t)link + 1] = r;
            b.add(l); b.add(r);
            return b;
        };
        float worst = 0.f;
        mesh_hit[(size_t)mi] = below(meshes[mi].root);          /* (fills hbox on the way) */
        for (int64_t i = 0; i < m.node_count; i++) {       /* the exact walks prune in time: how far the hits leave the reference box */
            rt_bvh_node& n = nodes[(size_t)(m.node_first + i)];
            const float* mn[2] = {n.lmin, n.rmin};
            const float* mx[2] = {n.lmax, n.rmax};
            for (int s = 0; s < 2; s++) {
                const HitBox& hb = hbox[2 * (size_t)(m.node_first + i) + s];
                float e = 0.f;
                for (int k = 0; k < 3; k++)
                    if (hb.mn[k] <= hb.mx[k]) e = std::max(e, std::max(mn[s][k] - hb.mn[k], hb.mx[k] - mx[s][k]));
                if (hb.unbounded || !(e < 1e30f)) e = INFINITY;
                else worst = std::max(worst, e);
                memcpy(&n.pad0[s], &e, 4);
            }
        }
        meshes[mi].delta = worst;
        const size_t sub_base = (size_t)d->n_mesh_nodes;
        qnodes.resize(2 * (sub_base + subs.size()), make_uint4(0u, 0u, 0u, 0u));
        qunp.resize(2 * (sub_base + subs.size()), 0);
        quant_ok = quantise_mesh(d, m, nodes, empty_link, meshes[mi], qnodes, tri, hbox, qunp, subs, sub_lo, sub_base) && quant_ok;
        /* Words 22-23 of a leaf's first record: the leaf's path from the root (bit 63 = the root's decision, 0 left / 1 right,
         * one bit per level; the depth in the low six bits). The closest-hit walk over the 4-wide tree visits leaves in an
         * order of its own, and of two hits with the same t the reference keeps the one it reaches first (Mesh.cpp:114):
         * the order of two leaves is decided at their lowest common ancestor (the nearer child's leaves come first,
         * AccelerationStruct.cpp:270-279), which two path words locate (rt_kernels.h, leaf_comes_first). */
        if (meshes[mi].root >= 0) {
            std::vector<std::pair<int32_t, unsigned long long>> todo;      /* (inner node, path so far | depth) */
            todo.emplace_back(meshes[mi].root, 0ull);
            while (!todo.empty()) {
                const auto [node, word] = todo.back();
                todo.pop_back();
                const unsigned depth = (unsigned)(word & 63ull);
                const int32_t child[2] = {nodes[(size_t)node].left, nodes[(size_t)node].right};
                for (int side = 0; side < 2; side++) {
                    if (child[side] == empty_link) continue;
                    if (depth >= 57u) { wide_paths_ok = false; continue; }
                    const unsigned long long w = ((word & ~63ull) | ((unsigned long long)side << (63u - depth))) | (unsigned long long)(depth + 1u);
                    if (child[side] >= 0) todo.emplace_back(child[side], w);
                    else {
                        float4& q5 = tri[RT_TRI_STRIDE * (size_t)(-2 - (int64_t)child[side]) + 5];
                        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
                        memcpy(&q5.z, &lo, 4);
                        memcpy(&q5.w, &hi, 4);
                    }
                }
            }
        }
        {
            WideFold wf{qnodes, empty_link, meshes[mi].qstep, wnodes, qunp};
            if (meshes[mi].root >= 0) meshes[mi].wroot = wf.build(meshes[mi].root, 0);
            else meshes[mi].wroot = meshes[mi].root == wf.empty_link ? LINK_NONE : meshes[mi].root;
            max_wide_depth = std::max(max_wide_depth, wf.depth);
            /* the sub-trees hang behind their leaves: words 24-25 of the leaf's first record = (link, 1) */
            wf.link_flags = (int32_t)WIDE_SUB;
            for (const auto& gate : gates) {
                wf.depth = 0;
                const int32_t sl = wf.build((int32_t)(sub_base + (size_t)gate.second), 0);
                max_wide_sub_depth = std::max(max_wide_sub_depth, wf.depth);
                const uint32_t mark = 1u;
                float4 w6 = make_float4(0.f, 0.f, 0.f, 0.f);
                memcpy(&w6.x, &sl, 4); memcpy(&w6.y, &mark, 4);
                tri[RT_TRI_STRIDE * gate.first + 6] = w6;
            }
        }
    }
    if (d->n_verts > 0x7FFFFFF0ll) return fail(c, RT_ERR_LIMIT, "too many vertices");
    if (wnodes.size() / 4 + 4096 >= (size_t)WIDE_INDEX) return fail(c, RT_ERR_LIMIT, "too many mesh nodes");
    if (c->debug) {
        size_t unb = 0;
        float worst = 0.f;
        for (uint8_t f : qunp) unb += f;
        for (const DMesh& m : meshes) worst = std::max(worst, m.delta);
        fprintf(stderr, "[rtgpu] fat leaves (more than %d triangles): %zu sub-tree nodes, wide depth %d below the leaves\n", RT_LEAF_SPLIT, subs.size(), max_wide_sub_depth);
        fprintf(stderr, "[rtgpu] hit excursions: %zu of %zu mesh-node children without a usable bound, largest bounded one %g\n", unb, qunp.size(), (double)worst);
    }
    std::vector<float> normals(d->normals, d->normals + 3 * d->n_verts);

    std::vector<float4> squares((size_t)d->n_squares * 4);
    for (int i = 0; i < d->n_squares; i++) {                       /* Square.h:67-71,77,85-86 */
        const rt_square& s = d->squares[i];
        H3 bl = {s.v0[0], s.v0[1], s.v0[2]};
        H3 right = hsub({s.v1[0], s.v1[1], s.v1[2]}, bl), up = hsub({s.v3[0], s.v3[1], s.v3[2]}, bl);
        H3 nn = hcross(right, up);
        float L = hlen(nn);
        H3 n = {nn.x / L, nn.y / L, nn.z / L};
        float D = hdot(bl, n);
        float rs = right.x * right.x + right.y * right.y + right.z * right.z;
        float us = up.x * up.x + up.y * up.y + up.z * up.z;
        float mf;
        memcpy(&mf, &s.material, 4);
        squares[4 * i + 0] = make_float4(bl.x, bl.y, bl.z, D);
        squares[4 * i + 1] = make_float4(right.x, right.y, right.z, rs);
        squares[4 * i + 2] = make_float4(up.x, up.y, up.z, us);
        squares[4 * i + 3] = make_float4(n.x, n.y, n.z, mf);
    }
    std::vector<float4> spheres((size_t)d->n_spheres * 2);
    for (int i = 0; i < d->n_spheres; i++) {
        const rt_sphere& s = d->spheres[i];
        float mf;
        memcpy(&mf, &s.material, 4);
        spheres[2 * i] = make_float4(s.center[0], s.center[1], s.center[2], s.radius);
        spheres[2 * i + 1] = make_float4(mf, 0.f, 0.f, 0.f);
    }
    std::vector<DMat> mats((size_t)d->n_materials);
    for (int i = 0; i < d->n_materials; i++) {
        const rt_material& m = d->materials[i];
        DMat& o = mats[i];
        memset(&o, 0, sizeof(o));
        for (int k = 0; k < 3; k++) { o.ka[k] = m.ka[k]; o.kd[k] = m.kd[k]; o.ks[k] = m.ks[k]; }
        o.ior = m.ior; o.shininess = m.shininess; o.type = m.type; o.transparency = m.transparency; o.reflectivity = m.reflectivity;
        o.tex = m.texture - 1; o.nmap = m.normal_map - 1;              /* 0 = none in the C-ABI */
        o.scale = m.material_scale == 0.0 ? 1.0 : m.material_scale;
    }
    /* images: one texel buffer, a table of (pointer, size) records */
    std::vector<uint8_t> texels;
    std::vector<size_t> tex_off;
    for (int i = 0; i < d->n_textures; i++) {
        const rt_texture& t = d->textures[i];
        tex_off.push_back(texels.size());
        texels.insert(texels.end(), t.texels, t.texels + (size_t)t.width * t.height * t.channels);
    }
    const uint8_t* d_texels = nullptr;
    if ((rc = upload(c, texels, &d_texels))) { free_scene(c); return rc; }
    std::vector<DTex> textures((size_t)d->n_textures);
    for (int i = 0; i < d->n_textures; i++) {
        textures[i].data = d_texels + tex_off[i];
        textures[i].w = d->textures[i].width; textures[i].h = d->textures[i].height; textures[i].ch = d->textures[i].channels; textures[i].pad = 0;
    }
    std::vector<DLight> lights((size_t)d->n_lights);
    for (int i = 0; i < d->n_lights; i++) {
        memset(&lights[i], 0, sizeof(DLight));
        for (int k = 0; k < 3; k++) { lights[i].pos[k] = d->lights[i].pos[k]; lights[i].color[k] = d->lights[i].color[k]; }
        lights[i].radius = d->lights[i].radius;
    }
    /* sin/cos of the 360 integer-degree angles, evaluated by the host libm exactly where the
     * reference evaluates them (Scene.h:112-122: float angle, double sin/cos) */
    std::vector<double> sincos(720);
    {
        float degToRad = (float)(M_PI / 180.);
        for (int k = 0; k < 360; k++) {
            float ang = (float)(long)k * degToRad;
            sincos[k] = std::sin((double)ang);
            sincos[360 + k] = std::cos((double)ang);
        }
    }
    std::vector<rt_bvh_node> tnodes(d->top_nodes, d->top_nodes + d->n_top_nodes);
    std::vector<uint2> tleaves((size_t)d->n_top_leaves);
    for (int i = 0; i < d->n_top_leaves; i++) tleaves[i] = make_uint2(d->top_leaves[i].first, d->top_leaves[i].count);
    std::vector<uint32_t> titems(d->top_items, d->top_items + d->n_top_items);

    /* ---- sub-trees inside top-level leaves. A shadow query asks, per top-level leaf the reference reaches, whether SOME object
     * of the leaf is hit (at t <= lightDistance / at all, rt_kernels.h shadow_walk), so the order in which the objects of a
     * leaf are looked at is free, and an object the ray cannot hit need not be looked at. The reference's depth-3 top-level tree
     * leaves up to 22 objects in a leaf (default scene: 84 objects in 8 leaves), which the walk used to test one after the
     * other. A leaf of more than RT_TOP_SPLIT objects gets a small SAH tree over conservative boxes of its objects (what
     * every accepted hit on the object stays inside), in the same 4-wide quantised node format as the mesh shadow trees
     * (planes in the frame of the whole scene), hung behind the leaf: the walk opens the leaf as before (reachability and
     * the order of the leaves are the reference's), then walks this tree and tests only the objects whose box it enters. */
    std::vector<int32_t> tsub((size_t)std::max(d->n_top_leaves, 1), LINK_NONE);
    DMesh top_frame{};
    float top_omin[3] = {0.f, 0.f, 0.f}, top_omax[3] = {0.f, 0.f, 0.f};
    int max_top_sub_depth = 0;
    bool has_topsub = false;
    /* ---- conservative box per top-level object (what every accepted hit on it stays inside): for the sub-trees inside top-level
     * leaves below, and for the group culling of shadow rays (k_shadow_group) */
    const int n_obj = d->n_squares + d->n_spheres + d->n_meshes;
    /* the scene's extent (object boxes without padding) bounds where shadow rays start: origins further out than one
     * diagonal take the plain item scan (checked per ray in the kernel) */
    double emin[3] = {INFINITY, INFINITY, INFINITY}, emax[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto grow = [&](const double* lo, const double* hi) { for (int k = 0; k < 3; k++) { emin[k] = std::min(emin[k], lo[k]); emax[k] = std::max(emax[k], hi[k]); } };
    bool finite = true;
    for (int i = 0; i < d->n_squares && finite; i++) {
        const rt_square& q = d->squares[i];
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            const double v4 = (double)q.v1[k] + (double)q.v3[k] - (double)q.v0[k];
            lo[k] = std::min(std::min((double)q.v0[k], (double)q.v1[k]), std::min((double)q.v3[k], v4));
            hi[k] = std::max(std::max((double)q.v0[k], (double)q.v1[k]), std::max((double)q.v3[k], v4));
            if (!(std::fabs(lo[k]) < 1e15) || !(std::fabs(hi[k]) < 1e15)) finite = false;
        }
        if (finite) grow(lo, hi);
    }
    for (int i = 0; i < d->n_spheres && finite; i++) {
        const rt_sphere& sp = d->spheres[i];
        double lo[3], hi[3];
        if (!(sp.radius > 0.f) || !(sp.radius < 1e15f)) { finite = false; break; }
        for (int k = 0; k < 3; k++) { lo[k] = (double)sp.center[k] - sp.radius; hi[k] = (double)sp.center[k] + sp.radius; if (!(std::fabs(lo[k]) < 1e15) || !(std::fabs(hi[k]) < 1e15)) finite = false; }
        if (finite) grow(lo, hi);
    }
    for (int mi = 0; mi < d->n_meshes && finite; mi++) {
        const HitBox& hb = mesh_hit[(size_t)mi];
        if (hb.unbounded) { finite = false; break; }
        if (!(hb.mn[0] <= hb.mx[0])) continue;            /* a mesh that cannot be hit */
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) { lo[k] = hb.mn[k]; hi[k] = hb.mx[k]; if (!(std::fabs(lo[k]) < 1e15) || !(std::fabs(hi[k]) < 1e15)) finite = false; }
        if (finite) grow(lo, hi);
    }
    const double diag = finite && emin[0] <= emax[0] ? std::sqrt((emax[0] - emin[0]) * (emax[0] - emin[0]) + (emax[1] - emin[1]) * (emax[1] - emin[1]) + (emax[2] - emin[2]) * (emax[2] - emin[2])) : 0.0;
    const bool obox_ok = finite && diag > 0.0 && diag < 1e12;
    /* conservative box per object ordinal */
    std::vector<HitBox> obox((size_t)n_obj);
    double amax_scene = 0.0, Dmax = 0.0;
    const double u = 1.0 / 16777216.0;
    if (obox_ok) {
        for (int k = 0; k < 3; k++) amax_scene = std::max(amax_scene, std::max(std::fabs(emin[k] - diag), std::fabs(emax[k] + diag)));
        Dmax = 2.5 * diag;            /* largest distance between an admitted origin and a point of the scene */
        for (int i = 0; i < d->n_squares; i++) {
            /* Square::intersect accepts P = o + t*d (t > 0) whose u, v computed FROM P lie in [0, 1] (Square.h:81-89): P is
             * inside the quad up to the rounding of those dot products and off its plane by the rounding of t */
            const rt_square& q = d->squares[i];
            HitBox& b = obox[(size_t)i];
            double ext = 0.0;
            for (int k = 0; k < 3; k++) ext = std::max(ext, std::max(std::fabs((double)q.v1[k] - q.v0[k]), std::fabs((double)q.v3[k] - q.v0[k])));
            const float pad = (float)(1e-4 * ext + 64.0 * u * (amax_scene + Dmax) + 1e-6);
            for (int k = 0; k < 3; k++) {
                const float v4 = q.v1[k] + q.v3[k] - q.v0[k];
                b.mn[k] = std::min(std::min(q.v0[k], q.v1[k]), std::min(q.v3[k], v4)) - pad;
                b.mx[k] = std::max(std::max(q.v0[k], q.v1[k]), std::max(q.v3[k], v4)) + pad;
            }
        }
        for (int i = 0; i < d->n_spheres; i++) {
            /* Sphere::intersect (Sphere.h:86-146) accepts P = o + t*d for a float root t of a*t^2 + b*t + c: with the
             * coefficients and the root rounded, | |P - centre|^2 - r^2 | <= 96 u (|o - centre| + r)^2, hence P lies within
             * 96 u (D + r)^2 / r of the sphere (D: the largest admitted distance of an origin). The root may be NEGATIVE
             * (discriminant exactly 0, Sphere.h:113): nodes holding a sphere are tested against the ray's whole line (WIDE_LINE). */
            const rt_sphere& sp = d->spheres[i];
            HitBox& b = obox[(size_t)(d->n_squares + i)];
            const double r = sp.radius;
            const float pad = (float)(96.0 * u * (Dmax + r) * (Dmax + r) / r + 1e-5 * r + 64.0 * u * (amax_scene + Dmax) + 1e-6);
            for (int k = 0; k < 3; k++) { b.mn[k] = sp.center[k] - sp.radius - pad; b.mx[k] = sp.center[k] + sp.radius + pad; }
        }
        for (int mi = 0; mi < d->n_meshes; mi++) obox[(size_t)(d->n_squares + d->n_spheres + mi)] = mesh_hit[(size_t)mi];
    }
    if (quant_ok && !getenv("RT_NO_TOP_SPLIT") && d->n_top_items > 0) {
        if (obox_ok) {
            /* frame of the 16-bit planes: everything the boxes cover */
            double gmin[3] = {INFINITY, INFINITY, INFINITY}, gmax[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (const HitBox& b : obox)
                if (b.mn[0] <= b.mx[0]) for (int k = 0; k < 3; k++) { gmin[k] = std::min(gmin[k], (double)b.mn[k]); gmax[k] = std::max(gmax[k], (double)b.mx[k]); }
            float fqmin[3], fqstep[3];
            for (int k = 0; k < 3; k++) {
                double step = std::max((gmax[k] - gmin[k]) / 65000.0, 1e-30);
                fqstep[k] = (float)step;
                if (!(fqstep[k] > 0.f)) fqstep[k] = 1e-30f;
                fqmin[k] = (float)(gmin[k] - 8.0 * (double)fqstep[k]);
                top_frame.qmin[k] = fqmin[k]; top_frame.qstep[k] = fqstep[k];
                top_omin[k] = (float)(emin[k] - diag); top_omax[k] = (float)(emax[k] + diag);
            }
            top_frame.amax = (float)amax_scene;
            top_frame.root = LINK_NONE; top_frame.wroot = LINK_NONE;
            bool qok = true;
            auto lo_q = [&](float v, int k) -> uint32_t {
                double q = std::floor(((double)v - (double)fqmin[k]) / (double)fqstep[k]) - 1.0;
                if (!(q >= 0.0)) { if (!(q > -4.0)) qok = false; q = 0.0; }
                if (q > 65535.0) { qok = false; q = 65535.0; }
                return (uint32_t)q;
            };
            auto hi_q = [&](float v, int k) -> uint32_t {
                double q = std::ceil(((double)v - (double)fqmin[k]) / (double)fqstep[k]) + 1.0;
                if (!(q <= 65535.0)) { qok = false; q = 65535.0; }
                if (q < 0.0) { qok = false; q = 0.0; }
                return (uint32_t)q;
            };
            /* binary SAH trees over the items of the fat top-level leaves, as 32-byte quantised nodes behind the mesh nodes */
            struct TopNode { HitBox box[2]; int32_t link[2]; };
            std::vector<TopNode> tn;
            std::function<int32_t(std::vector<int>&, int, int, HitBox&, int, int&)> build = [&](std::vector<int>& ord, int lo, int hi, HitBox& out, int level, int& depth) -> int32_t {
                out = HitBox();
                for (int i = lo; i < hi; i++) out.add(obox[(size_t)ord[(size_t)i]]);
                if (hi - lo == 1) return (int32_t)(-2 - (int64_t)(TOP_ITEM | (uint32_t)ord[(size_t)lo]));
                depth = std::max(depth, level + 1);
                /* exhaustive sweep SAH: the lists are short (tens of objects) */
                double best = INFINITY;
                int best_axis = 0, best_mid = (lo + hi) / 2;
                std::vector<int> tmp(ord.begin() + lo, ord.begin() + hi);
                for (int k = 0; k < 3; k++) {
                    std::stable_sort(tmp.begin(), tmp.end(), [&](int a, int b) { return obox[(size_t)a].mn[k] + obox[(size_t)a].mx[k] < obox[(size_t)b].mn[k] + obox[(size_t)b].mx[k]; });
                    const int n = (int)tmp.size();
                    std::vector<double> right((size_t)n + 1, 0.0);
                    HitBox acc;
                    for (int i = n - 1; i >= 1; i--) { acc.add(obox[(size_t)tmp[(size_t)i]]); right[(size_t)i] = LeafSplitter::half_area(acc) * (n - i); }
                    acc = HitBox();
                    for (int i = 1; i < n; i++) {
                        acc.add(obox[(size_t)tmp[(size_t)i - 1]]);
                        const double cost = LeafSplitter::half_area(acc) * i + right[(size_t)i];
                        if (cost < best) { best = cost; best_axis = k; best_mid = lo + i; }
                    }
                }
                {
                    const int k = best_axis;
                    std::stable_sort(ord.begin() + lo, ord.begin() + hi, [&](int a, int b) { return obox[(size_t)a].mn[k] + obox[(size_t)a].mx[k] < obox[(size_t)b].mn[k] + obox[(size_t)b].mx[k]; });
                }
                const size_t me = tn.size();
                tn.emplace_back();
                HitBox bl, br;
                const int32_t l = build(ord, lo, best_mid, bl, level + 1, depth), r = build(ord, best_mid, hi, br, level + 1, depth);
                tn[me].box[0] = bl; tn[me].box[1] = br; tn[me].link[0] = l; tn[me].link[1] = r;
                return (int32_t)me;
            };
            struct Pending { int leaf; int32_t root; bool line; };
            std::vector<Pending> pend;
            for (int li = 0; li < d->n_top_leaves; li++) {
                const rt_bvh_leaf& lf = d->top_leaves[li];
                if (lf.count <= (uint32_t)RT_TOP_SPLIT) continue;
                std::vector<int> ord;
                bool line = false, usable = true;
                for (uint32_t i = 0; i < lf.count; i++) {
                    const int o = (int)d->top_items[lf.first + i];
                    const HitBox& b = obox[(size_t)o];
                    if (b.unbounded) usable = false;
                    if (!(b.mn[0] <= b.mx[0])) continue;          /* cannot be hit: not in the tree */
                    if (o >= d->n_squares && o < d->n_squares + d->n_spheres) line = true;
                    ord.push_back(o);
                }
                if (!usable || ord.size() < 2) continue;
                HitBox all;
                int depth = 0;
                const int32_t root = build(ord, 0, (int)ord.size(), all, 0, depth);
                max_top_sub_depth = std::max(max_top_sub_depth, depth);
                pend.push_back({li, root, line});
            }
            if (!pend.empty()) {
                const size_t base = qnodes.size() / 2;
                qnodes.resize(2 * (base + tn.size()), make_uint4(0u, 0u, 0u, 0u));
                qunp.resize(2 * (base + tn.size()), 0);
                for (size_t i = 0; i < tn.size(); i++) {
                    uint32_t w[6];
                    int32_t link[2];
                    for (int sd = 0; sd < 2; sd++) {
                        const HitBox& hb = tn[i].box[sd];
                        for (int k = 0; k < 3; k++) w[3 * sd + k] = lo_q(hb.mn[k], k) | (hi_q(hb.mx[k], k) << 16);
                        link[sd] = tn[i].link[sd] >= 0 ? (int32_t)(base + (size_t)tn[i].link[sd]) : tn[i].link[sd];
                    }
                    qnodes[2 * (base + i)] = make_uint4(w[0], w[1], w[2], w[3]);
                    qnodes[2 * (base + i) + 1] = make_uint4(w[4], w[5], (uint32_t)link[0], (uint32_t)link[1]);
                }
                if (qok) {
                    WideFold wf{qnodes, (int32_t)(-2 - d->n_tris), fqstep, wnodes, qunp};
                    for (const Pending& pe : pend) {
                        wf.depth = 0;
                        wf.link_flags = (int32_t)(WIDE_SUB | (pe.line ? WIDE_LINE : 0u));
                        tsub[(size_t)pe.leaf] = wf.build((int32_t)(base + (size_t)pe.root), 0);
                        max_wide_top_depth = std::max(max_wide_top_depth, wf.depth);
                    }
                    has_topsub = true;
                }
            }
        }
    }

    /* the same boxes for k_shadow_group (a few hundred objects at most: the test runs once per (record, light) pair) */
    std::vector<float4> gobox;
    if (obox_ok && n_obj > 0 && n_obj <= 256 && !getenv("RT_NO_GROUP_CULL")) {
        auto put = [&](const HitBox& b, int kind) {
            float kf;
            memcpy(&kf, &kind, 4);
            if (b.unbounded) { gobox.push_back(make_float4(-INFINITY, -INFINITY, -INFINITY, INFINITY)); gobox.push_back(make_float4(INFINITY, INFINITY, kf, 0.f)); }
            else if (!(b.mn[0] <= b.mx[0])) { gobox.push_back(make_float4(INFINITY, INFINITY, INFINITY, -INFINITY)); gobox.push_back(make_float4(-INFINITY, -INFINITY, kf, 0.f)); }   /* cannot be hit */
            else { gobox.push_back(make_float4(b.mn[0], b.mn[1], b.mn[2], b.mx[0])); gobox.push_back(make_float4(b.mx[1], b.mx[2], kf, 0.f)); }
        };
        /* squares and spheres one box each, in ordinal order (the kernel finds their records by it); a mesh as the hit boxes of up to
         * `per_mesh` nodes of its tree -- the box of a whole gorilla is mostly air: the node with the largest box is replaced by its
         * two children until the budget is spent (every accepted hit lies in the hit box of its leaf, hence in one of these) */
        for (int o = 0; o < d->n_squares + d->n_spheres; o++) put(obox[(size_t)o], o < d->n_squares ? 0 : 1);
        int per_mesh = d->n_meshes > 0 ? std::max(1, std::min(8, (256 - d->n_squares - d->n_spheres) / d->n_meshes)) : 1;      /* (1 / 8 / 32 / 64 boxes: gorilla frame 34.8 / 32.0 / 32.5 / 33.6 ms, cow 20.1 / 20.0 / 20.8 / 22.1) */
        if (const char* e = getenv("RT_GROUP_MESH_BOXES")) per_mesh = std::max(1, std::min(64, atoi(e)));
        for (int mi = 0; mi < d->n_meshes; mi++) {
            struct Part { HitBox box; int32_t link; };
            std::vector<Part> parts{{mesh_hit[(size_t)mi], meshes[(size_t)mi].root}};
            while ((int)parts.size() < per_mesh) {
                int pick = -1;
                double best = -1.0;
                for (int i = 0; i < (int)parts.size(); i++) {
                    const Part& pt = parts[(size_t)i];
                    if (pt.link < 0 || pt.box.unbounded || !(pt.box.mn[0] <= pt.box.mx[0])) continue;
                    const double a = LeafSplitter::half_area(pt.box);
                    if (a > best) { best = a; pick = i; }
                }
                if (pick < 0) break;
                const int32_t link = parts[(size_t)pick].link;
                const rt_bvh_node& n = nodes[(size_t)link];
                const Part l{hbox[2 * (size_t)link], n.left}, r{hbox[2 * (size_t)link + 1], n.right};
                parts[(size_t)pick] = l;
                parts.push_back(r);
            }
            for (const Part& pt : parts)
                if (pt.box.unbounded || pt.box.mn[0] <= pt.box.mx[0]) put(pt.box, 2);      /* (a part that cannot be hit needs no box) */
            if (parts.empty()) put(mesh_hit[(size_t)mi], 2);
        }
        for (int k = 0; k < 3; k++) { sc.gomin[k] = (float)(emin[k] - diag); sc.gomax[k] = (float)(emax[k] + diag); }
    }
    sc.n_obox = (int32_t)(gobox.size() / 2);
    if (gobox.empty()) gobox.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
    const rt_bvh_node* dn = nullptr;
    const rt_bvh_node* dtn = nullptr;
    if ((rc = upload(c, gobox, &sc.obox)) || (rc = upload(c, tri, &sc.tri)) || (rc = upload(c, ids, &sc.tri_ids)) || (rc = upload(c, normals, &sc.normals)) ||
        (rc = upload(c, nodes, &dn)) || (rc = upload(c, meshes, &sc.meshes)) ||
        (rc = upload(c, wnodes, &sc.wnodes)) ||
        (rc = upload(c, tnodes, &dtn)) || (rc = upload(c, tleaves, &sc.tleaves)) || (rc = upload(c, titems, &sc.titems)) ||
        (rc = upload(c, squares, &sc.squares)) || (rc = upload(c, spheres, &sc.spheres)) || (rc = upload(c, mats, &sc.mats)) ||
        (rc = upload(c, textures, &sc.textures)) ||
        (rc = upload(c, lights, &sc.lights)) || (rc = upload(c, sincos, &sc.sincos))) {
        free_scene(c);
        return rc;
    }
    sc.mnodes = reinterpret_cast<const float4*>(dn);
    sc.tnodes = reinterpret_cast<const float4*>(dtn);
    {   /* the same top-level tables as one blob (16-byte aligned sections) for k_shadow's LDS copy */
        std::vector<uint8_t> blob;
        auto section = [&](const void* p, size_t bytes, int k) {
            blob.resize((blob.size() + 15) & ~(size_t)15);
            sc.top_off[k] = (int32_t)blob.size();
            const uint8_t* b = static_cast<const uint8_t*>(p);
            blob.insert(blob.end(), b, b + bytes);
        };
        std::vector<uint2> tl = tleaves;
        if (tl.empty()) tl.push_back(make_uint2(0, 0));
        section(tnodes.data(), tnodes.size() * sizeof(rt_bvh_node), 0);
        section(tl.data(), tl.size() * sizeof(uint2), 1);
        section(titems.data(), titems.size() * sizeof(uint32_t), 2);
        section(squares.data(), squares.size() * sizeof(float4), 3);
        section(spheres.data(), spheres.size() * sizeof(float4), 4);
        section(meshes.data(), meshes.size() * sizeof(DMesh), 5);
        {
            static_assert(sizeof(TopFrame) % 16 == 0, "tsub follows the frame");
            TopFrame tf{};
            tf.frame = top_frame;
            for (int k = 0; k < 3; k++) { tf.omin[k] = top_omin[k]; tf.omax[k] = top_omax[k]; }
            std::vector<uint8_t> sec(sizeof(TopFrame) + tsub.size() * sizeof(int32_t));
            memcpy(sec.data(), &tf, sizeof(TopFrame));
            memcpy(sec.data() + sizeof(TopFrame), tsub.data(), tsub.size() * sizeof(int32_t));
            section(sec.data(), sec.size(), 6);
        }
        {   /* The flat top-level pass of k_shadow (rt_kernels.h, "flat top-level pass"): the top-level leaves with their own boxes in
             * the order the reference's left-first walk meets them (AccelerationStruct.cpp:315-336). Usable when the tree has an inner
             * root, at most 32 leaves, and finite boxes that are nested (a child's two boxes inside the box its parent has for it):
             * then "the leaf's own box passes AABB::intersect => every box above it passes" (monotone slab arithmetic, as for the
             * mesh trees), the reference reaches exactly the leaves whose own box passes, in this order. */
            std::vector<float4> flat;
            bool ok = !getenv("RT_NO_FLAT_TOP") && d->top_root >= 0 && d->n_top_leaves <= 32 && (d->n_squares + d->n_spheres + d->n_meshes) > 0;
            std::function<void(int32_t)> walk = [&](int32_t node) {
                const rt_bvh_node& n = tnodes[(size_t)node];
                const float* bmin[2] = {n.lmin, n.rmin};
                const float* bmax[2] = {n.lmax, n.rmax};
                const int32_t link[2] = {n.left, n.right};
                for (int sd = 0; sd < 2 && ok; sd++) {
                    const bool empty = !(bmin[sd][0] <= bmax[sd][0]) || !(bmin[sd][1] <= bmax[sd][1]) || !(bmin[sd][2] <= bmax[sd][2]);
                    if (!empty) for (int k = 0; k < 3; k++) if (!(std::fabs(bmin[sd][k]) < 1e30f) || !(std::fabs(bmax[sd][k]) < 1e30f)) ok = false;
                    if (link[sd] >= 0) {
                        if (empty) { ok = false; continue; }       /* (an inner node under an empty box: not a tree of bounding boxes) */
                        const rt_bvh_node& ch = tnodes[(size_t)link[sd]];
                        const float* cmn[2] = {ch.lmin, ch.rmin};
                        const float* cmx[2] = {ch.lmax, ch.rmax};
                        for (int t = 0; t < 2; t++) {
                            if (!(cmn[t][0] <= cmx[t][0])) continue;             /* an empty child box (inverted, +-inf) misses anyway */
                            for (int k = 0; k < 3; k++) if (!(cmn[t][k] >= bmin[sd][k]) || !(cmx[t][k] <= bmax[sd][k])) ok = false;
                        }
                        walk(link[sd]);
                    } else if (!empty) {
                        const int32_t leaf = -1 - link[sd];
                        float fl;
                        memcpy(&fl, &leaf, 4);
                        flat.push_back(make_float4(bmin[sd][0], bmin[sd][1], bmin[sd][2], bmax[sd][0]));
                        flat.push_back(make_float4(bmax[sd][1], bmax[sd][2], fl, 0.f));
                    }
                }
            };
            if (ok) walk(d->top_root);
            if (!ok || flat.size() > 64) flat.clear();
            sc.n_flat = (int32_t)(flat.size() / 2);
            if (flat.empty()) flat.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
            section(flat.data(), flat.size() * sizeof(float4), 7);
            if (c->debug) fprintf(stderr, "[rtgpu] flat top-level pass: %d leaves\n", sc.n_flat);
        }
        blob.resize((blob.size() + 15) & ~(size_t)15);
        const uint8_t* db = nullptr;
        if ((rc = upload(c, blob, &db))) { free_scene(c); return rc; }
        sc.top_blob = reinterpret_cast<const float4*>(db);
        sc.top_blob_bytes = (int32_t)blob.size();
    }
    sc.n_top_items = d->n_top_items;
    sc.n_lights = d->n_lights; sc.n_squares = d->n_squares; sc.n_spheres = d->n_spheres; sc.n_meshes = d->n_meshes;
    sc.top_root = (d->n_squares + d->n_spheres + d->n_meshes) > 0 ? d->top_root : -1;
    if ((d->n_squares + d->n_spheres + d->n_meshes) == 0) {
        /* empty scene: a single empty leaf */
        std::vector<uint2> one(1, make_uint2(0, 0));
        if ((rc = upload(c, one, &sc.tleaves))) { free_scene(c); return rc; }
    }
    sc.top_cap = d->top_depth + 2;
    sc.stack_cap = sc.top_cap + max_mesh_depth + 2;
    sc.cstack_cap = sc.stack_cap + (subs.empty() ? 0 : 3 * max_wide_sub_depth + 5);
    /* The deepest stack of the 4-wide shadow walk: three links pending per level of the mesh's tree, and TWO sub-trees of fat leaves on top of
     * each other -- a lane walks on with one leaf pending, so it can be inside the sub-tree of a fat leaf A (one entry for its root, three
     * per level) when the triangle round tests the next pending leaf B, which pushes B's sub-tree on top of what is left of A's. Until the
     * end of round 3 this counted one sub-tree: a walk that built both wrote into the next block's slice (found by tools/soak.sh). */
    sc.wstack_cap = sc.top_cap + 3 * (max_wide_depth + 2 * max_wide_sub_depth) + 4 + (has_topsub ? 3 * max_wide_top_depth + 2 : 0);
    sc.has_sub = subs.empty() ? 0 : 1;
    sc.has_topsub = has_topsub ? 1 : 0;
    {   /* contribution culling (k_closest): the bound needs finite, non-negative light colours */
        double sum = 0.0;
        /* without lights the reference's shadow factor is 1 - 0/0 = NaN on every diffuse hit (Scene.h:211-222), whatever its throughput:
         * nothing may be culled, or those NaNs would not reach the frame */
        bool ok = !getenv("RT_NO_CULL") && d->n_lights > 0;
        for (int i = 0; i < d->n_lights; i++) {
            double mx = 0.0;
            for (int k = 0; k < 3; k++) { const double v = std::fabs((double)d->lights[i].color[k]); if (!(v < 1e30)) ok = false; mx = std::max(mx, v); }
            sum += mx;
        }
        sc.light_sum = ok ? (float)(sum * 1.0001) : -1.f;
        /* culling of path states: G (growth of a throughput per bounce) and C (what a hit adds per unit of throughput) over all
         * materials; needs finite geometry (a NaN normal below a culled state would have reached the frame as a NaN) */
        bool states = ok;
        for (int64_t i = 0; i < 3 * d->n_verts && states; i++) if (!(std::fabs(d->normals[i]) < 1e30f) || !(std::fabs(d->positions[i]) < 1e30f)) states = false;
        /* ... and an interpolated normal that can vanish (Mesh.cpp:117-130 normalises n0 w0 + n1 w1 + n2 w2: 0/0 = NaN): a vertex normal of
         * length zero, or two normals of one triangle that point against each other */
        for (int64_t v = 0; v < d->n_verts && states; v++) {
            const float* n = d->normals + 3 * v;
            if (!((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2] > 1e-60)) states = false;
        }
        for (int mi = 0; mi < d->n_meshes && states; mi++) {
            const rt_mesh& m = d->meshes[mi];
            for (int64_t t = 0; t < m.tri_count && states; t++) {
                const float* nv[3];
                for (int k = 0; k < 3; k++) nv[k] = d->normals + 3 * ((int64_t)m.vert_first + (int64_t)d->tri_indices[3 * (m.tri_first + t) + k]);
                for (int k = 0; k < 3 && states; k++) {
                    const float* a = nv[k]; const float* b = nv[(k + 1) % 3];
                    const double ab = (double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2];
                    const double aa = (double)a[0] * a[0] + (double)a[1] * a[1] + (double)a[2] * a[2], bb = (double)b[0] * b[0] + (double)b[1] * b[1] + (double)b[2] * b[2];
                    if (!(ab > -0.9999 * std::sqrt(aa * bb))) states = false;
                }
            }
        }
        double G = 1.0, Cm = 1.0;                        /* the sky adds at most 1 per unit (Scene.h:274-278) */
        for (int i = 0; i < d->n_materials && states; i++) {
            const rt_material& m = d->materials[i];
            const double nmax = m.normal_map ? 3.0003 : 1.0002;     /* |N| after Material::applyTBN: at most |t| + |b| + |n| with texels in [0, 1] */
            double kmax = 0.0, amax = 0.0, kdmax = -INFINITY;
            for (int k = 0; k < 3; k++) {
                kmax = std::max(kmax, std::max(std::fabs((double)m.kd[k]), std::fabs((double)m.ks[k])));
                amax = std::max(amax, std::fabs((double)m.ka[k]));
                kdmax = std::max(kdmax, (double)m.kd[k]);
            }
            if (m.texture || m.type == RT_MAT_CHECKER) kmax = std::max(kmax, 1.0);
            if (!(kmax < 1e30) || !(amax < 1e30) || !(m.shininess >= 0.0) || !(m.shininess < 1e6)) { states = false; break; }
            const double spec = std::pow(2.0 * nmax * nmax + 1.0, m.shininess);
            Cm = std::max(Cm, sum * 1.0001 * (amax + kmax * (nmax + spec)) * 1.01);
            if (m.type == RT_MAT_MIRROR) {
                double ks = 0.0;
                for (int k = 0; k < 3; k++) ks = std::max(ks, std::fabs((double)m.ks[k]));
                if (m.texture) ks = std::max(ks, 1.0);          /* ks = the texel (Scene.h:189-191) */
                G = std::max(G, std::fabs((double)m.reflectivity) * ks * 1.001);
            } else if (m.type == RT_MAT_GLASS) {
                if (!(kdmax <= 1.0) || !(m.transparency >= 0.f) || !(m.ior > 0.f)) { states = false; break; }     /* absorption exp(-(1 - kd) t transparency) <= 1 needs these */
                const double ri = m.ior, r0 = ((1.0 - ri) / (1.0 + ri)) * ((1.0 - ri) / (1.0 + ri));
                const double rmax = r0 + (1.0 - r0) * std::pow(1.0 + nmax, 5.0);
                const double omtr = std::fabs(1.0 - (double)m.transparency);
                G = std::max(G, std::max(omtr * rmax, omtr * std::max(1.0, rmax - 1.0) + (double)m.transparency) * 1.001);
            }
        }
        if (!(Cm < 1e30) || !(G < 1e30)) states = false;
        std::vector<float> gpow((size_t)RT_MAX_PATH_DEPTH + 2, INFINITY);
        if (states) {
            double acc = 1.0;
            for (size_t r = 0; r < gpow.size(); r++) { gpow[r] = acc < 1e37 ? (float)(acc * 1.0001) : INFINITY; acc *= G; }
        }
        if ((rc = upload(c, gpow, &sc.cull_gpow))) { free_scene(c); return rc; }
        sc.cull_cmax = states ? (float)(Cm * 1.0001) : -1.f;
        if (c->debug) fprintf(stderr, "[rtgpu] contribution culling: hits %s, path states %s (growth per bounce <= %g, a hit adds <= %g per unit of throughput)\n",
                              ok ? "on" : "off", states ? "on" : "off", G, Cm);
    }

    sc.quant_ok = (quant_ok && !getenv("RT_SHADOW_EXACT_NODES")) ? 1 : 0;
    /* closest hits over the 4-wide quantised tree (rt_kernels.h, mesh_closest_wide): up to three links pending per level */
    c->wide_mode = 2;
    if (const char* e = getenv("RT_WIDE_CLOSEST")) c->wide_mode = atoi(e) ? 1 : 0;       /* test knob: 1 = every launch, 0 = never */
    sc.wide_closest = quant_ok && wide_paths_ok && !getenv("RT_SHADOW_EXACT_NODES") && c->wide_mode != 0 ? 1 : 0;
    c->cstack_wide = std::max(sc.cstack_cap, sc.top_cap + 3 * (max_wide_depth + max_wide_sub_depth) + 6);
    {   /* The per-lane stacks of the walks live in LDS (4 bytes x entries x threads of the block; k_shadow's wide stack spills
         * beyond stack_cap). What a launch may ask for is a documented limit, not a launch failure at rt_flush: the one-wave
         * build of k_closest / k_intersect must fit (the 256-thread build is only chosen when it fits, run_wavefront). */
        hipDeviceProp_t prop;
        size_t lds_max = 65536;
        if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.sharedMemPerBlock > 0) lds_max = prop.sharedMemPerBlock;
        if ((size_t)c->cstack_wide * RT_BLOCK * sizeof(int) + 4096 > lds_max) sc.wide_closest = 0;      /* (the binary walk needs less) */
        const size_t need = (size_t)std::max(sc.cstack_cap, sc.stack_cap) * RT_BLOCK * sizeof(int) + 4096;
        if (need > lds_max) {
            free_scene(c);
            return fail(c, RT_ERR_LIMIT, "traversal stacks need %zu bytes of LDS per block (%d entries per lane: tree depth %d, sub-tree depth %d), the device allows %zu",
                        need, std::max(sc.cstack_cap, sc.stack_cap), max_mesh_depth, max_wide_sub_depth, lds_max);
        }
        c->lds_max = lds_max;
    }
    if (c->debug) fprintf(stderr, "[rtgpu] scene: %lld mesh nodes, depth %d, quantised shadow nodes %s\n", (long long)d->n_mesh_nodes, max_mesh_depth,
                          sc.quant_ok ? "on" : "off");
    c->sc = sc;
    c->front_state = -1;
    if (const char* e = getenv("RT_FRONT")) c->front_state = atoi(e) ? 1 : 0;     /* test knob: 1 = always, 0 = never */
    c->group_state = -1; c->group_seen = 0; c->group_kept = 0;
    c->group_min_pairs = 8192;      /* (a chunk smaller than this is not worth the read-back) */
    if (const char* e = getenv("RT_GROUP_CULL")) { c->group_state = atoi(e) ? 1 : 0; c->group_min_pairs = 1; }  /* test knob: 1 = on every chunk, 0 = never */
    c->n_lights = d->n_lights;
    c->has_scene = true;
    if (const char* dir = getenv("RT_DUMP_SCENE")) {
        /* debugging aid (tools/emu): the device scene as files -- the DScene record and every buffer it points into, with the device
         * addresses, so that the walks can be replayed on the CPU under a memory checker */
        const std::string base(dir);
        if (FILE* f = fopen((base + "/dscene.bin").c_str(), "wb")) { fwrite(&c->sc, sizeof(DScene), 1, f); fclose(f); }
        if (FILE* m = fopen((base + "/manifest.txt").c_str(), "w")) {
            for (size_t k = 0; k < c->scene_bufs.size(); k++) {
                const DevBuf& b = c->scene_bufs[k];
                std::vector<unsigned char> host(b.bytes);
                if (hipMemcpy(host.data(), b.p, b.bytes, hipMemcpyDeviceToHost) != hipSuccess) continue;
                fprintf(m, "%zu %llx %zu\n", k, (unsigned long long)(uintptr_t)b.p, b.bytes);
                if (FILE* f = fopen((base + "/buf_" + std::to_string(k) + ".bin").c_str(), "wb")) { fwrite(host.data(), 1, b.bytes, f); fclose(f); }
            }
            fclose(m);
        }
    }
    return RT_OK;
}

int rt_set_camera(rt_ctx* c, const rt_camera* cam) {
    if (!c || !cam) return RT_ERR_ARG;
    if (!c->subs.empty()) {
        for (rt_ctx* q : c->subs) { int rc = rt_set_camera(q, cam); if (rc) return fail(c, rc, "%s", q->err.c_str()); }
        c->has_camera = true;
        return RT_OK;
    }
    /* Camera::apply (Camera.cpp:125-132): I * T(x,y,z) * T(0,0,-zoom) * R, all glMultMatrix calls */
    double mv[16], t[16], r[16];
    m4_identity(mv);
    m4_identity(t); t[12] = cam->x; t[13] = cam->y; t[14] = cam->z;
    m4_mul(mv, t, mv);
    m4_identity(t); t[14] = -(double)cam->zoom;
    m4_mul(mv, t, mv);
    {   /* build_rotmatrix, Trackball.cpp:323-345; the float[4][4] is read by glMultMatrixf as column-major */
        const float* q = cam->quat;
        float m[16];
        m[0] = (float)(1.0 - 2.0 * (q[1] * q[1] + q[2] * q[2]));
        m[1] = (float)(2.0 * (q[0] * q[1] - q[2] * q[3]));
        m[2] = (float)(2.0 * (q[2] * q[0] + q[1] * q[3]));
        m[3] = 0.0f;
        m[4] = (float)(2.0 * (q[0] * q[1] + q[2] * q[3]));
        m[5] = (float)(1.0 - 2.0 * (q[2] * q[2] + q[0] * q[0]));
        m[6] = (float)(2.0 * (q[1] * q[2] - q[0] * q[3]));
        m[7] = 0.0f;
        m[8] = (float)(2.0 * (q[2] * q[0] - q[1] * q[3]));
        m[9] = (float)(2.0 * (q[1] * q[2] + q[0] * q[3]));
        m[10] = (float)(1.0 - 2.0 * (q[1] * q[1] + q[0] * q[0]));
        m[11] = 0.0f;
        m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;
        for (int i = 0; i < 16; i++) r[i] = m[i];
    }
    m4_mul(mv, r, mv);
    /* gluPerspective(fovAngle, aspectRatio, nearPlane, farPlane), Camera.cpp:48 */
    double p[16];
    memset(p, 0, sizeof(p));
    double radians = (double)cam->fovy_deg / 2.0 * M_PI / 180.0;
    double sine = std::sin(radians);
    double deltaZ = (double)cam->zfar - (double)cam->znear;
    if (deltaZ == 0.0 || sine == 0.0 || cam->aspect == 0.0f) return fail(c, RT_ERR_ARG, "degenerate projection");
    double cot = std::cos(radians) / sine;
    p[0] = cot / (double)cam->aspect;
    p[5] = cot;
    p[10] = -((double)cam->zfar + (double)cam->znear) / deltaZ;
    p[11] = -1.0;
    p[14] = -2.0 * (double)cam->znear * (double)cam->zfar / deltaZ;
    if (!m4_inverse(p, c->cam.pinv) || !m4_inverse(mv, c->cam.mvinv)) return fail(c, RT_ERR_ARG, "singular camera matrix");
    /* cameraSpaceToWorldSpace(0,0,0), matrixUtilities.h:188-197 */
    const double* mi = c->cam.mvinv;
    double w = mi[15];
    c->cam.eye[0] = (float)(mi[12] / w);
    c->cam.eye[1] = (float)(mi[13] / w);
    c->cam.eye[2] = (float)(mi[14] / w);
    c->has_camera = true;
    return RT_OK;
}

static int check_params(rt_ctx* c, const rt_render_params* p) {
    if (p->img_w <= 0 || p->img_h <= 0 || (int64_t)p->img_w * p->img_h > 0x3FFFFFFFll) return fail(c, RT_ERR_ARG, "bad image size");
    if (p->spp <= 0 || p->spp > 65535) return fail(c, RT_ERR_ARG, "spp out of range");
    if (p->max_depth < 0 || p->max_depth > RT_MAX_PATH_DEPTH) return fail(c, RT_ERR_LIMIT, "max_depth above %d", RT_MAX_PATH_DEPTH);
    if (p->light_samples <= 0 || p->light_samples > 4096) return fail(c, RT_ERR_ARG, "light_samples out of range");
    if (p->shadow_mode < 0 || p->shadow_mode > 1) return fail(c, RT_ERR_ARG, "shadow_mode");
    return RT_OK;
}
static void set_frame(rt_ctx* c, const rt_render_params* p) {
    c->fr.img_w = p->img_w; c->fr.img_h = p->img_h; c->fr.spp = p->spp; c->fr.max_depth = p->max_depth;
    c->fr.light_samples = p->light_samples; c->fr.shadow_mode = p->shadow_mode; c->fr.jitter = p->jitter; c->fr.seed = p->seed;
    c->fr.sample_base = 0;
}

int rt_begin_frame(rt_ctx* c, const rt_render_params* p) {
    if (!c || !p) return RT_ERR_ARG;
    if (!c->subs.empty()) {
        int rc = on_all(c, [&](rt_ctx* q) { return rt_begin_frame(q, p); });
        if (rc) return rc;
        set_frame(c, p);
        for (auto& t : c->sub_tiles) t.clear();
        for (auto& t : c->sub_tile_keys) t.clear();
        memset(&c->stats, 0, sizeof(c->stats));
        c->has_frame = true;
        return RT_OK;
    }
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "rt_begin_frame before rt_upload_scene");
    int rc = check_params(c, p);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    set_frame(c, p);
    size_t npix = (size_t)p->img_w * p->img_h;
    if ((rc = ensure(c, c->acc, npix * 3 * sizeof(long long)))) return rc;
    if ((rc = ensure(c, c->prim, npix * 2 * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->nf, npix * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(c, c->rgb, npix * 3 * sizeof(float)))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->acc.p, 0, npix * 3 * sizeof(long long), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->prim.p, 0xFF, npix * 2 * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->nf.p, 0, npix * sizeof(uint32_t), c->stream));
    c->tiles.clear();
    c->queued_samples = 0;
    c->resolve_spp = 0;
    memset(&c->stats, 0, sizeof(c->stats));
    c->has_frame = true;
    return RT_OK;
}

/* Sample sharding ("when a few pixels dominate"): this context renders samples first .. first + count - 1 of every pixel
 * it is given, of a frame of p->spp samples per pixel that several contexts (ranks) share. The samples keep their numbers, hence their
 * jitter and their counter-RNG streams, and the sums are order-independent fixed point: once the accumulators of the contexts have been
 * added (rt_export_accumulators, all-reduce, rt_import_accumulators) a resolve gives the bits of the frame one context would have
 * rendered alone. Primary-hit ids belong to sample 0, i.e. to the context whose range starts at 0. */
int rt_begin_frame_samples(rt_ctx* c, const rt_render_params* p, int32_t first_sample, int32_t n_samples) {
    if (!c || !p) return RT_ERR_ARG;
    if (!c->subs.empty()) return fail(c, RT_ERR_ARG, "rt_begin_frame_samples: one device per context (the ranks of ray-tracing_amd/dist.py)");
    if (first_sample < 0 || n_samples < 1 || (long long)first_sample + n_samples > (long long)p->spp) return fail(c, RT_ERR_ARG, "rt_begin_frame_samples: sample range outside the frame's %d samples", p->spp);
    int rc = rt_begin_frame(c, p);
    if (rc) return rc;
    c->fr.sample_base = first_sample;
    c->fr.spp = n_samples;
    c->resolve_spp = p->spp;
    return RT_OK;
}

/* the frame's accumulators, out of the context and back in: 3 x int64 per pixel (2^-36 units) and one flag word per pixel (three bits per
 * channel: NaN, +inf, -inf seen), both in DEVICE memory of the caller. Export runs what is queued first. */
int rt_export_accumulators(rt_ctx* c, void* d_acc, void* d_flags) {
    if (!c || !d_acc || !d_flags) return RT_ERR_ARG;
    if (!c->subs.empty()) return fail(c, RT_ERR_ARG, "rt_export_accumulators: one device per context");
    if (!c->has_frame) return fail(c, RT_ERR_STATE, "rt_export_accumulators before rt_begin_frame");
    int rc = rt_flush(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npix = (size_t)c->fr.img_w * c->fr.img_h;
    HIP_TRY(c, hipMemcpyAsync(d_acc, c->acc.p, npix * 3 * sizeof(long long), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d_flags, c->nf.p, npix * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}
int rt_import_accumulators(rt_ctx* c, const void* d_acc, const void* d_flags) {
    if (!c || !d_acc || !d_flags) return RT_ERR_ARG;
    if (!c->subs.empty()) return fail(c, RT_ERR_ARG, "rt_import_accumulators: one device per context");
    if (!c->has_frame) return fail(c, RT_ERR_STATE, "rt_import_accumulators before rt_begin_frame");
    if (!c->tiles.empty()) return fail(c, RT_ERR_STATE, "rt_import_accumulators with tiles queued (they would add to the imported sums)");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npix = (size_t)c->fr.img_w * c->fr.img_h;
    HIP_TRY(c, hipMemcpyAsync(c->acc.p, d_acc, npix * 3 * sizeof(long long), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->nf.p, d_flags, npix * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

/* progressive accumulation: more samples into the accumulators of the frame that is open */
int rt_continue_frame(rt_ctx* c, int32_t more_spp) {
    if (!c) return RT_ERR_ARG;
    if (!c->has_frame) return fail(c, RT_ERR_STATE, "rt_continue_frame before rt_begin_frame");
    if (more_spp < 1 || (long long)c->fr.sample_base + c->fr.spp + more_spp > 0x00FFFFFFll) return fail(c, RT_ERR_ARG, "rt_continue_frame: samples per pixel out of range");
    if (c->resolve_spp > 0) return fail(c, RT_ERR_STATE, "rt_continue_frame on a sample-sharded frame (rt_begin_frame_samples)");
    int rc = rt_flush(c);                                   /* what is queued belongs to the pass that ends here */
    if (rc) return rc;
    if (!c->subs.empty()) {
        rc = on_all(c, [&](rt_ctx* q) { return rt_continue_frame(q, more_spp); });
        if (rc) return rc;
        /* (sub_tiles stays: it lists the tiles of the FRAME -- a later pass may cover only some of them, and the gather must still
         * bring home what the earlier passes rendered on the other devices; rt_begin_frame clears it) */
    } else {
        c->tiles.clear();
        c->queued_samples = 0;
    }
    c->fr.sample_base += c->fr.spp;
    c->fr.spp = more_spp;
    return RT_OK;
}

int rt_render_tile(rt_ctx* c, int x0, int y0, int w, int h) {
    if (!c) return RT_ERR_ARG;
    if (!c->subs.empty()) {
        if (!c->has_frame) return fail(c, RT_ERR_STATE, "rt_render_tile before rt_begin_frame");
        const int k = tile_owner(std::max(x0, 0), std::max(y0, 0), (int)c->subs.size());
        int rc = rt_render_tile(c->subs[(size_t)k], x0, y0, w, h);
        if (rc) return fail(c, rc, "%s", c->subs[(size_t)k]->err.c_str());
        if (w > 0 && h > 0 && c->sub_tile_keys[(size_t)k].insert(std::make_tuple(x0, y0, w, h)).second) {
            std::vector<TileD>& tl = c->sub_tiles[(size_t)k];
            TileD t;
            t.x0 = x0; t.y0 = y0; t.w = w; t.h = h;
            t.first = tl.empty() ? 0ull : tl.back().first + (unsigned long long)tl.back().w * tl.back().h;
            tl.push_back(t);
        }
        return RT_OK;
    }
    if (!c->has_frame) return fail(c, RT_ERR_STATE, "rt_render_tile before rt_begin_frame");
    if (!c->has_camera) return fail(c, RT_ERR_STATE, "rt_render_tile before rt_set_camera");
    if (w < 0 || h < 0 || x0 < 0 || y0 < 0 || x0 + w > c->fr.img_w || y0 + h > c->fr.img_h) return fail(c, RT_ERR_ARG, "tile outside the image");
    if (w == 0 || h == 0) return RT_OK;
    TileD t;
    t.x0 = x0; t.y0 = y0; t.w = w; t.h = h; t.first = c->queued_samples;
    c->tiles.push_back(t);
    c->queued_samples += (unsigned long long)w * h * c->fr.spp;
    return RT_OK;
}

int rt_render_tiles(rt_ctx* c, const int32_t* xywh, int64_t n) {
    if (!c || (n > 0 && !xywh) || n < 0) return RT_ERR_ARG;
    for (int64_t i = 0; i < n; i++) {
        const int rc = rt_render_tile(c, xywh[4 * i], xywh[4 * i + 1], xywh[4 * i + 2], xywh[4 * i + 3]);
        if (rc) return rc;
    }
    return RT_OK;
}

int rt_flush(rt_ctx* c) {
    if (!c) return RT_ERR_ARG;
    if (!c->subs.empty()) {
        if (!c->has_frame) return fail(c, RT_ERR_STATE, "rt_flush before rt_begin_frame");
        int rc = on_all(c, [&](rt_ctx* q) { return rt_flush(q); });
        if (rc) return rc;
        /* statistics of the frame: counts add up, the devices ran side by side */
        rt_stats t{};
        for (rt_ctx* q : c->subs) {
            const rt_stats& s = q->stats;
            t.rays_primary += s.rays_primary; t.rays_secondary += s.rays_secondary; t.rays_shadow += s.rays_shadow; t.rays_shadow_culled += s.rays_shadow_culled; t.rays_shadow_grouped += s.rays_shadow_grouped; t.nonfinite += s.nonfinite;
            t.iterations = std::max(t.iterations, s.iterations); t.peak_states = std::max(t.peak_states, s.peak_states);
            t.ms_total = std::max(t.ms_total, s.ms_total); t.ms_closest += s.ms_closest; t.ms_shadow += s.ms_shadow;
            t.launches_closest += s.launches_closest; t.launches_shadow += s.launches_shadow;
        }
        c->stats = t;
        return RT_OK;
    }
    if (!c->has_frame) return fail(c, RT_ERR_STATE, "rt_flush before rt_begin_frame");
    if (c->tiles.empty()) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    Source src{};
    src.from_tiles = true;
    src.total = c->queued_samples;
    src.max_depth = c->fr.max_depth;
    WaveBufs wb{};
    int rc = bufs_from_ctx(c, src, wb);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_frame_a, c->stream));
    rc = run_wavefront(c, src, wb, static_cast<long long*>(c->acc.p), static_cast<uint32_t*>(c->nf.p), static_cast<uint32_t*>(c->prim.p));
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev_frame_b, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev_frame_b));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_frame_a, c->ev_frame_b));
    c->stats.ms_total += ms;
    c->tiles.clear();
    c->queued_samples = 0;
    return RT_OK;
}

int rt_get_stats(rt_ctx* c, rt_stats* s) {
    if (!c || !s) return RT_ERR_ARG;
    *s = c->stats;
    return RT_OK;
}

int rt_resolve_device(rt_ctx* c, void* d_rgb) {
    if (!c || !d_rgb) return RT_ERR_ARG;
    if (!c->subs.empty()) {
        /* every device resolves its own frame (its tiles; the rest of it is never looked at); subs[0] resolves straight into
         * the caller's buffer, the tiles of the others arrive as packed slabs (peer copy) and are scattered in */
        int rc = rt_flush(c);
        if (rc) return rc;
        rt_ctx* d0 = c->subs[0];
        /* all devices at once, each on its own host thread and stream: resolve; the others also pack the pixels of their tiles
         * into a slab and record an event behind it */
        rc = on_all(c, [&](rt_ctx* q) -> int {
            if (q == d0) return rt_resolve_device(d0, d_rgb);
            int r2 = rt_resolve_device(q, q->rgb.p);
            if (r2) return r2;
            size_t k = 0;
            while (c->subs[k] != q) k++;
            const std::vector<TileD>& tl = c->sub_tiles[k];
            if (tl.empty()) return RT_OK;
            const uint32_t n_pix = (uint32_t)(tl.back().first + (unsigned long long)tl.back().w * tl.back().h);
            HIP_TRY(q, hipSetDevice(q->device));
            if ((r2 = ensure(q, q->mg_slab, (size_t)n_pix * 12)) || (r2 = ensure(q, q->mg_tiles, tl.size() * sizeof(TileD)))) return r2;
            HIP_TRY(q, hipMemcpyAsync(q->mg_tiles.p, tl.data(), tl.size() * sizeof(TileD), hipMemcpyHostToDevice, q->stream));
            hipLaunchKernelGGL(k_tiles_pack, dim3((n_pix + 255) / 256), dim3(256), 0, q->stream, static_cast<const float*>(q->rgb.p), c->fr.img_w,
                               static_cast<const TileD*>(q->mg_tiles.p), (int)tl.size(), n_pix, static_cast<float*>(q->mg_slab.p));
            HIP_TRY(q, hipEventRecord(q->ev_pack, q->stream));
            return RT_OK;
        });
        if (rc) return rc;
        /* the gathering device pulls every slab behind its event (peer copies over xGMI, all queued at once) into one
         * buffer, slab after slab, and scatters them with ONE kernel over the merged tile list */
        c->mg_all.clear();
        unsigned long long total = 0;
        for (size_t k = 1; k < c->subs.size(); k++) {
            const std::vector<TileD>& tl = c->sub_tiles[k];
            if (tl.empty()) continue;
            for (TileD t : tl) { t.first += total; c->mg_all.push_back(t); }
            total += tl.back().first + (unsigned long long)tl.back().w * tl.back().h;
        }
        if (total) {
            HIP_TRY(c, hipSetDevice(d0->device));
            if ((rc = ensure(c, c->mg_slab, (size_t)total * 12)) || (rc = ensure(c, c->mg_tiles, c->mg_all.size() * sizeof(TileD)))) return rc;
            unsigned long long off = 0;
            for (size_t k = 1; k < c->subs.size(); k++) {
                rt_ctx* q = c->subs[k];
                const std::vector<TileD>& tl = c->sub_tiles[k];
                if (tl.empty()) continue;
                const unsigned long long n_pix = tl.back().first + (unsigned long long)tl.back().w * tl.back().h;
                unsigned char* dst = static_cast<unsigned char*>(c->mg_slab.p) + (size_t)off * 12;
                HIP_TRY(c, hipStreamWaitEvent(d0->stream, q->ev_pack, 0));
                if (q->device == d0->device) HIP_TRY(c, hipMemcpyAsync(dst, q->mg_slab.p, (size_t)n_pix * 12, hipMemcpyDeviceToDevice, d0->stream));
                else HIP_TRY(c, hipMemcpyPeerAsync(dst, d0->device, q->mg_slab.p, q->device, (size_t)n_pix * 12, d0->stream));
                off += n_pix;
            }
            HIP_TRY(c, hipMemcpyAsync(c->mg_tiles.p, c->mg_all.data(), c->mg_all.size() * sizeof(TileD), hipMemcpyHostToDevice, d0->stream));
            hipLaunchKernelGGL(k_tiles_scatter, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, d0->stream, static_cast<float*>(d_rgb), c->fr.img_w,
                               static_cast<const TileD*>(c->mg_tiles.p), (int)c->mg_all.size(), (uint32_t)total, static_cast<const float*>(c->mg_slab.p));
            HIP_TRY(c, hipStreamSynchronize(d0->stream));
        }
        HIP_TRY(c, hipGetLastError());
        return RT_OK;
    }
    int rc = rt_flush(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));        /* (rt_flush returns before its own hipSetDevice when nothing is queued, and this may run on a worker thread of on_all) */
    uint32_t npix = (uint32_t)((size_t)c->fr.img_w * c->fr.img_h);
    hipLaunchKernelGGL(k_resolve, dim3((npix * 3 + 255) / 256), dim3(256), 0, c->stream, static_cast<const long long*>(c->acc.p),
                       static_cast<const uint32_t*>(c->nf.p), npix, c->resolve_spp > 0 ? c->resolve_spp : c->fr.sample_base + c->fr.spp, static_cast<float*>(d_rgb));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}

int rt_resolve_rgba8_device(rt_ctx* c, void* d_rgba, int bottom_up) {
    if (!c || !d_rgba) return RT_ERR_ARG;
    rt_ctx* d0 = c->subs.empty() ? c : c->subs[0];            /* the gathering device */
    if (!d0->rgb.p) return fail(c, RT_ERR_STATE, "rt_resolve_rgba8_device before rt_begin_frame");
    int rc = rt_resolve_device(c, d0->rgb.p);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(d0->device));
    const uint32_t npix = (uint32_t)((size_t)c->fr.img_w * c->fr.img_h);
    hipLaunchKernelGGL(k_display, dim3((npix + 255) / 256), dim3(256), 0, d0->stream, static_cast<const float*>(d0->rgb.p), c->fr.img_w, c->fr.img_h,
                       bottom_up ? 1 : 0, static_cast<uint32_t*>(d_rgba));
    HIP_TRY(c, hipStreamSynchronize(d0->stream));
    HIP_TRY(c, hipGetLastError());
    return RT_OK;
}

int rt_download_framebuffer(rt_ctx* c, float* rgb, uint32_t* prim_id, rt_stats* stats) {
    if (!c || !rgb) return RT_ERR_ARG;
    if (!c->subs.empty()) {
        rt_ctx* d0 = c->subs[0];
        if (!d0->rgb.p) return fail(c, RT_ERR_STATE, "rt_download_framebuffer before rt_begin_frame");
        int rc = rt_resolve_device(c, d0->rgb.p);
        if (rc) return rc;
        const size_t npix = (size_t)c->fr.img_w * c->fr.img_h;
        HIP_TRY(c, hipSetDevice(d0->device));
        HIP_TRY(c, hipMemcpy(rgb, d0->rgb.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
        if (prim_id) {
            /* ids: every device's buffer holds its own tiles (0xFFFFFFFF elsewhere); merged on the host tile by tile */
            std::vector<uint32_t> tmp(npix * 2);
            HIP_TRY(c, hipMemcpy(prim_id, d0->prim.p, npix * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (size_t k = 1; k < c->subs.size(); k++) {
                rt_ctx* q = c->subs[k];
                HIP_TRY(c, hipSetDevice(q->device));
                HIP_TRY(c, hipMemcpy(tmp.data(), q->prim.p, npix * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
                for (const TileD& t : c->sub_tiles[k])
                    for (int y = t.y0; y < t.y0 + t.h; y++)
                        memcpy(prim_id + 2 * ((size_t)y * c->fr.img_w + t.x0), tmp.data() + 2 * ((size_t)y * c->fr.img_w + t.x0), (size_t)t.w * 2 * sizeof(uint32_t));
            }
        }
        if (stats) *stats = c->stats;
        return RT_OK;
    }
    int rc = rt_resolve_device(c, c->rgb.p);
    if (rc) return rc;
    size_t npix = (size_t)c->fr.img_w * c->fr.img_h;
    HIP_TRY(c, hipMemcpy(rgb, c->rgb.p, npix * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (prim_id) HIP_TRY(c, hipMemcpy(prim_id, c->prim.p, npix * 2 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (stats) *stats = c->stats;
    return RT_OK;
}

/* ------------------------------------------------------------------ fine-grained seams */
static int stage_in(rt_ctx* c, DevBuf& b, const void* host, size_t bytes) {
    int rc = ensure(c, b, bytes);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice));
    return RT_OK;
}

int rt_intersect(rt_ctx* c, const float* org, const float* dir, int64_t n, rt_hit* out) {
    if (c && !c->subs.empty()) { int rc = rt_intersect(c->subs[0], org, dir, n, out); if (rc) c->err = c->subs[0]->err; return rc; }      /* explicit rays: the first device */
    if (!c || !org || !dir || !out || n < 0 || n > 0x7FFFFFFFll) return RT_ERR_ARG;
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "no scene");
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    static_assert(sizeof(HitOut) == sizeof(rt_hit), "hit layout");
    DevBuf a, b, o;
    int rc;
    if ((rc = stage_in(c, a, org, (size_t)n * 12)) || (rc = stage_in(c, b, dir, (size_t)n * 12)) || (rc = ensure(c, o, (size_t)n * sizeof(HitOut)))) {
        release(a); release(b); release(o);
        return rc;
    }
    auto launch_intersect = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(((uint32_t)n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), lds_bytes_closest(c, RT_BLOCK, c->sc.wide_closest != 0), c->stream, c->sc,
                           static_cast<const float*>(a.p), static_cast<const float*>(b.p), (uint32_t)n, static_cast<HitOut*>(o.p));
    };
    if (c->sc.wide_closest) { if (c->sc.has_sub && c->sc.quant_ok) launch_intersect(k_intersect<true, true>); else launch_intersect(k_intersect<false, true>); }
    else { if (c->sc.has_sub && c->sc.quant_ok) launch_intersect(k_intersect<true, false>); else launch_intersect(k_intersect<false, false>); }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, o.p, (size_t)n * sizeof(HitOut), hipMemcpyDeviceToHost);
    release(a); release(b); release(o);
    if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, "rt_intersect: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_occluded(rt_ctx* c, const float* pts, const float* lp, int64_t n, int32_t* out) {
    if (c && !c->subs.empty()) { int rc = rt_occluded(c->subs[0], pts, lp, n, out); if (rc) c->err = c->subs[0]->err; return rc; }      /* explicit rays: the first device */
    if (!c || !pts || !lp || !out || n < 0 || n > 0x7FFFFFFFll) return RT_ERR_ARG;
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "no scene");
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf a, b, o;
    int rc;
    if ((rc = stage_in(c, a, pts, (size_t)n * 12)) || (rc = stage_in(c, b, lp, (size_t)n * 12)) || (rc = ensure(c, o, (size_t)n * 4))) {
        release(a); release(b); release(o);
        return rc;
    }
    hipLaunchKernelGGL(k_occluded, dim3(((uint32_t)n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), lds_bytes(c, RT_BLOCK), c->stream, c->sc,
                       static_cast<const float*>(a.p), static_cast<const float*>(b.p), (uint32_t)n, static_cast<int32_t*>(o.p));
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, o.p, (size_t)n * 4, hipMemcpyDeviceToHost);
    release(a); release(b); release(o);
    if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, "rt_occluded: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_group_clear(rt_ctx* c, const float* pts, const float* centres, const float* radii, int64_t n, int32_t* out) {
    if (c && !c->subs.empty()) { int rc = rt_group_clear(c->subs[0], pts, centres, radii, n, out); if (rc) c->err = c->subs[0]->err; return rc; }
    if (!c || !pts || !centres || !radii || !out || n < 0 || n > 0x7FFFFFFFll) return RT_ERR_ARG;
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "no scene");
    if (n == 0) return RT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf a, b, r, o;
    int rc;
    if ((rc = stage_in(c, a, pts, (size_t)n * 12)) || (rc = stage_in(c, b, centres, (size_t)n * 12)) || (rc = stage_in(c, r, radii, (size_t)n * 4)) || (rc = ensure(c, o, (size_t)n * 4))) {
        release(a); release(b); release(r); release(o);
        return rc;
    }
    hipLaunchKernelGGL(k_group_probe, dim3(((uint32_t)n + 255) / 256), dim3(256), 0, c->stream, c->sc,
                       static_cast<const float*>(a.p), static_cast<const float*>(b.p), static_cast<const float*>(r.p), (uint32_t)n, static_cast<int32_t*>(o.p));
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, o.p, (size_t)n * 4, hipMemcpyDeviceToHost);
    release(a); release(b); release(r); release(o);
    if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, "rt_group_clear: %s", hipGetErrorString(e));
    return RT_OK;
}

int rt_trace(rt_ctx* c, const float* org, const float* dir, const uint32_t* pixel, const uint32_t* sample, int64_t n,
             const rt_render_params* p, float* rgb) {
    if (c && !c->subs.empty()) { int rc = rt_trace(c->subs[0], org, dir, pixel, sample, n, p, rgb); if (rc) c->err = c->subs[0]->err; return rc; }
    if (!c || !org || !dir || !pixel || !sample || !p || !rgb || n < 0 || n > 0x3FFFFFFFll) return RT_ERR_ARG;
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "no scene");
    if (n == 0) return RT_OK;
    rt_render_params q = *p;
    if (q.img_w <= 0) q.img_w = 1;
    if (q.img_h <= 0) q.img_h = 1;
    q.spp = 1;
    int rc = check_params(c, &q);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    FrameD saved = c->fr;
    rt_stats saved_stats = c->stats;
    set_frame(c, &q);
    DevBuf a, b, px, sm, acc, nf, out;
    auto cleanup = [&]() { release(a); release(b); release(px); release(sm); release(acc); release(nf); release(out); c->fr = saved; c->stats = saved_stats; };
    if ((rc = stage_in(c, a, org, (size_t)n * 12)) || (rc = stage_in(c, b, dir, (size_t)n * 12)) || (rc = stage_in(c, px, pixel, (size_t)n * 4)) ||
        (rc = stage_in(c, sm, sample, (size_t)n * 4)) || (rc = ensure(c, acc, (size_t)n * 3 * sizeof(long long))) ||
        (rc = ensure(c, nf, (size_t)n * sizeof(uint32_t))) || (rc = ensure(c, out, (size_t)n * 3 * sizeof(float)))) {
        cleanup();
        return rc;
    }
    hipError_t e = hipMemsetAsync(acc.p, 0, (size_t)n * 3 * sizeof(long long), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(nf.p, 0, (size_t)n * sizeof(uint32_t), c->stream);
    if (e != hipSuccess) { cleanup(); return fail(c, RT_ERR_DEVICE, "memset: %s", hipGetErrorString(e)); }
    Source src{};
    src.from_tiles = false;
    src.total = (unsigned long long)n;
    src.d_org = static_cast<const float*>(a.p); src.d_dir = static_cast<const float*>(b.p);
    src.d_pixel = static_cast<const uint32_t*>(px.p); src.d_sample = static_cast<const uint32_t*>(sm.p);
    src.max_depth = q.max_depth;
    WaveBufs wb{};
    rc = bufs_from_ctx(c, src, wb);
    if (rc == RT_OK) rc = run_wavefront(c, src, wb, static_cast<long long*>(acc.p), static_cast<uint32_t*>(nf.p), nullptr);
    if (rc == RT_OK) {
        hipLaunchKernelGGL(k_resolve, dim3(((uint32_t)n * 3 + 255) / 256), dim3(256), 0, c->stream, static_cast<const long long*>(acc.p),
                           static_cast<const uint32_t*>(nf.p), (uint32_t)n, 1, static_cast<float*>(out.p));
        e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess) e = hipMemcpy(rgb, out.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, RT_ERR_DEVICE, "rt_trace: %s", hipGetErrorString(e));
    }
    cleanup();
    return rc;
}

/* rt_trace without a single allocation: staging, path states, queues, counters, shadow rays and spill slices all live
 * in this device's copy of rt_static_scratch (address taken once, in rt_init). Same kernels and same host loop as
 * rt_trace (run_wavefront with a small batch), so the radiance has the same bits: the sums are order-independent fixed
 * point. Calls on one GPU share the array -- everything that must start at zero is cleared here (accumulators, flags,
 * counters; the per-batch queues are cleared by the loop) -- and calls that overlap on one GPU would race. */
int rt_trace_static(rt_ctx* c, const float* org, const float* dir, const uint32_t* pixel, const uint32_t* sample, int64_t n,
                    const rt_render_params* p, float* rgb) {
    if (c && !c->subs.empty()) { int rc = rt_trace_static(c->subs[0], org, dir, pixel, sample, n, p, rgb); if (rc) c->err = c->subs[0]->err; return rc; }
    if (!c || !org || !dir || !pixel || !sample || !p || !rgb || n < 0) return RT_ERR_ARG;
    if (!c->has_scene) return fail(c, RT_ERR_STATE, "no scene");
    if (n > RT_STATIC_MAX_RAYS) return fail(c, RT_ERR_LIMIT, "rt_trace_static: %lld rays, the limit is %d", (long long)n, RT_STATIC_MAX_RAYS);
    if (p->max_depth > RT_STATIC_MAX_DEPTH) return fail(c, RT_ERR_LIMIT, "rt_trace_static: max_depth %d, the limit is %d", p->max_depth, RT_STATIC_MAX_DEPTH);
    if (n == 0) return RT_OK;
    rt_render_params q = *p;
    if (q.img_w <= 0) q.img_w = 1;
    if (q.img_h <= 0) q.img_h = 1;
    q.spp = 1;
    int rc = check_params(c, &q);
    if (rc) return rc;
    if (!c->static_scratch) return fail(c, RT_ERR_STATE, "rt_trace_static: no scratch address (rt_init)");
    HIP_TRY(c, hipSetDevice(c->device));

    /* carve-up of the scratch array (256-byte aligned pieces) */
    constexpr uint32_t B = 1024;                    /* batch */
    constexpr unsigned long long SRAYS = 65536;     /* shadow rays per k_shadow launch */
    constexpr int BLOCKS = 256;                     /* persistent blocks */
    const int L = c->n_lights;
    const int S = q.shadow_mode == 1 ? 1 : q.light_samples;
    if ((unsigned long long)std::max(L, 1) * S > SRAYS)
        return fail(c, RT_ERR_LIMIT, "rt_trace_static: %d lights x %d samples, the limit is %llu shadow rays per hit", L, S, SRAYS);
    const int ovf_cap = wide_ovf_cap(c, nullptr);
    size_t off = 0;
    bool fits = true;
    auto take = [&](size_t bytes) -> unsigned char* {
        unsigned char* at = c->static_scratch + off;
        off += (bytes + 255) & ~(size_t)255;
        if (off > RT_STATIC_BYTES) fits = false;
        return at;
    };
    float* d_org = reinterpret_cast<float*>(take((size_t)n * 12));
    float* d_dir = reinterpret_cast<float*>(take((size_t)n * 12));
    uint32_t* d_px = reinterpret_cast<uint32_t*>(take((size_t)n * 4));
    uint32_t* d_sm = reinterpret_cast<uint32_t*>(take((size_t)n * 4));
    long long* d_acc = reinterpret_cast<long long*>(take((size_t)n * 3 * sizeof(long long)));
    uint32_t* d_nf = reinterpret_cast<uint32_t*>(take((size_t)n * 4));
    float* d_out = reinterpret_cast<float*>(take((size_t)n * 12));
    WaveBufs w{};
    w.batch = B; w.persistent_blocks = BLOCKS; w.timing = false;
    w.stack_cap = (size_t)B * (RT_STATIC_MAX_DEPTH + 3);
    w.stack = reinterpret_cast<PathState*>(take(w.stack_cap * sizeof(PathState)));
    w.child = reinterpret_cast<PathState*>(take((size_t)2 * B * sizeof(PathState)));
    w.drec = reinterpret_cast<DiffuseRec*>(take((size_t)B * sizeof(DiffuseRec)));
    w.dcode = reinterpret_cast<unsigned long long*>(take((size_t)B * 8));
    w.dtex = reinterpret_cast<uint32_t*>(take((size_t)B * 4));
    w.cnt = reinterpret_cast<uint32_t*>(take((size_t)B * RT_MAX_LIGHTS * 4));
    w.keys = nullptr;                       /* (no group culling on the static scratch: 4096 rays at most) */
    w.smask = nullptr;                      /* (... and the top-level tree walked node by node: k_shadow<3>) */
    w.ctr = reinterpret_cast<Counters*>(take(sizeof(Counters)));
    w.tilebuf = nullptr;
    w.srays = reinterpret_cast<float4*>(take((size_t)SRAYS * 2 * sizeof(float4))); w.srays_cap = SRAYS;
    w.ovf_bytes = std::max<size_t>(256, (size_t)BLOCKS * ovf_cap * RT_BLOCK * sizeof(int));
    w.ovf = reinterpret_cast<int*>(take(w.ovf_bytes));
    if (!fits) return fail(c, RT_ERR_LIMIT, "rt_trace_static: this scene's shadow walk needs %d spill entries per lane, more than the scratch array holds", ovf_cap);

    FrameD saved = c->fr;
    rt_stats saved_stats = c->stats;
    set_frame(c, &q);
    auto restore = [&]() { c->fr = saved; c->stats = saved_stats; };
    hipError_t e = hipMemcpyAsync(d_org, org, (size_t)n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_dir, dir, (size_t)n * 12, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_px, pixel, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sm, sample, (size_t)n * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_acc, 0, (size_t)n * 3 * sizeof(long long), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_nf, 0, (size_t)n * 4, c->stream);
    if (e != hipSuccess) { restore(); return fail(c, RT_ERR_DEVICE, "rt_trace_static staging: %s", hipGetErrorString(e)); }
    Source src{};
    src.from_tiles = false;
    src.total = (unsigned long long)n;
    src.d_org = d_org; src.d_dir = d_dir; src.d_pixel = d_px; src.d_sample = d_sm;
    src.max_depth = q.max_depth;
    rc = run_wavefront(c, src, w, d_acc, d_nf, nullptr);
    if (rc == RT_OK) {
        hipLaunchKernelGGL(k_resolve, dim3(((uint32_t)n * 3 + 255) / 256), dim3(256), 0, c->stream, d_acc, d_nf, (uint32_t)n, 1, d_out);
        e = hipMemcpyAsync(rgb, d_out, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, RT_ERR_DEVICE, "rt_trace_static: %s", hipGetErrorString(e));
    }
    restore();
    return rc;
}

int rt_camera_rays(rt_ctx* c, const rt_render_params* p, int x0, int y0, int w, int h, float* org, float* dir) {
    if (c && !c->subs.empty()) { int rc = rt_camera_rays(c->subs[0], p, x0, y0, w, h, org, dir); if (rc) c->err = c->subs[0]->err; return rc; }      /* explicit rays: the first device */
    if (!c || !p || !org || !dir || w <= 0 || h <= 0) return RT_ERR_ARG;
    if (!c->has_camera) return fail(c, RT_ERR_STATE, "no camera");
    int rc = check_params(c, p);
    if (rc) return rc;
    size_t n = (size_t)w * h * p->spp;
    if (n > 0x3FFFFFFFull) return fail(c, RT_ERR_LIMIT, "too many rays");
    HIP_TRY(c, hipSetDevice(c->device));
    FrameD fr{};
    fr.img_w = p->img_w; fr.img_h = p->img_h; fr.spp = p->spp; fr.max_depth = p->max_depth; fr.light_samples = p->light_samples;
    fr.shadow_mode = p->shadow_mode; fr.jitter = p->jitter; fr.seed = p->seed;
    DevBuf a, b;
    if ((rc = ensure(c, a, n * 12)) || (rc = ensure(c, b, n * 12))) { release(a); release(b); return rc; }
    hipLaunchKernelGGL(k_camera_rays, dim3(((uint32_t)n + 255) / 256), dim3(256), 0, c->stream, c->cam, fr, x0, y0, w, h, static_cast<float*>(a.p),
                       static_cast<float*>(b.p));
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(org, a.p, n * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(dir, b.p, n * 12, hipMemcpyDeviceToHost);
    release(a); release(b);
    if (e != hipSuccess) return fail(c, RT_ERR_DEVICE, "rt_camera_rays: %s", hipGetErrorString(e));
    return RT_OK;
}

}  // extern "C"
