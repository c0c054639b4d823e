/*
 * rt_kernels.h -- traversal, shading and wavefront kernels (gfx950, wave64).
 *
 * Data flow of one wavefront iteration (rtgpu.hip drives it):
 *   k_raygen      primary rays -> path states           main.cpp:191-201, matrixUtilities.h:198-222
 *   k_closest     closest hit + material dispatch        Scene.h:174-280, AccelerationStruct.cpp:286-337
 *                 -> child states (mirror/glass), diffuse records, sky radiance
 *   k_shadow      one lane per (diffuse hit, light, sample): area-light sample + ordered any-hit
 *                                                        Scene.h:110-134,160-168
 *   k_shade       Phong + shadow factor, accumulate      Scene.h:90-108,204-222
 *   k_resolve     pixel_color /= nsamples                main.cpp:219
 * Per-lane traversal stacks live in LDS (entry k of lane l at s_stack[k*blockDim+l]:
 * conflict-free, one bank per lane).
 */
#pragma once
#include "rt_device.h"

#define RT_BLOCK 64
#define RT_FIX_SCALE 68719476736.0   /* 2^36: radiance accumulators are int64 fixed point */

struct PathState {          /* 64 B */
    float o[3], d[3], thr[3];
    uint32_t slot;          /* accumulator index */
    uint32_t pixel;         /* RNG key */
    uint32_t sample;        /* RNG key */
    uint64_t code;          /* path code: 1 primary; child = 2*code (+1 for refraction) */
    int32_t remaining;      /* NRemainingBounces */
    uint32_t flags;         /* bit0: write prim id */
};
#ifndef RT_CBLOCK
#define RT_CBLOCK 256       /* threads per block of k_closest: one append atomic per block */
#endif
struct DiffuseRec {         /* 64 B + code */
    float P[3], N[3], V[3], thr[3];
    uint32_t mat;           /* material index | checker override: bit30 set, bit29 = colour (0 black, 1 white) */
    uint32_t slot, pixel, sample;
};
struct Counters {
    uint32_t n_child, n_diffuse;
    uint32_t walk_overflow;         /* a lane of k_shadow wanted more stack entries than wstack_cap allows (wpush): the launch's answers are void, rt_flush reports RT_ERR_LIMIT */
    uint32_t pad;
    unsigned long long rays_primary, rays_secondary, rays_shadow, nonfinite;
    unsigned long long work;        /* next unclaimed shadow-ray index of the running k_shadow launch */
    unsigned long long deferred;    /* rays k_shadow_front handed to the running k_shadow launch */
    unsigned long long deferred_sum; /* ... to all launches of this frame */
    unsigned long long culled;       /* diffuse hits k_closest dropped because they cannot change the frame (contribution culling) */
    unsigned long long n_keys;       /* (record, light) pairs k_shadow_group left for the running launch */
    unsigned long long group_why[4]; /* RT_DEBUG: pairs kept because of their origin or length / a square / a sphere / a mesh */
    unsigned long long sh_node_steps, sh_tri_steps, sh_other_steps, sh_loop_iters;   /* RT_COUNT=1 diagnostics */
    unsigned long long sh_node_iters, sh_tri_iters, sh_top_iters, sh_top_lanes;
    unsigned long long sh_m1_node, sh_m1_tri, sh_m1_rays;   /* steps taken and rays finished in mode 1 (the second walk of shadow_walk) */
    unsigned long long sh_clk_refill, sh_clk_top, sh_clk_node, sh_clk_tri;             /* wave clocks per section (s_memtime) */      /* wave iterations per phase; lanes entering the top-level step */
};
struct TileD { int32_t x0, y0, w, h; unsigned long long first; };
struct CameraD { double pinv[16], mvinv[16]; float eye[3]; float pad; };
struct FrameD {
    int32_t img_w, img_h, spp, max_depth, light_samples, shadow_mode, jitter;
    uint32_t seed;
    int32_t sample_base;    /* progressive accumulation (rt_continue_frame): samples per pixel of the earlier passes; this pass renders sample_base .. sample_base + spp - 1 */
};
struct HitD { int exists; float t; int obj; int tri; F3 n; F3 pos; float u, v; int mat; };

RT_DEV int lds_index(int k, int stride) { return k * stride; }

/* Child links of the mesh trees on the device: >= 0 inner node; <= -2 a leaf whose triangles start at
 * record -2-link (the last record of a leaf carries a flag in word 2.w), so that reaching a leaf costs
 * no extra fetch; a leaf without triangles points at one shared all-zero record (normal 0 => d.n == 0
 * => Triangle::isParallelTo rejects it); LINK_NONE = no link. */
#define LINK_NONE ((int)0x80000000)
#define WIDE_UNPRUNED 0x40000000    /* on a link to a 4-wide shadow node: do not prune by distance in that node */
#define WIDE_SUB 0x20000000         /* ... the node belongs to the sub-tree of a fat leaf: the reference's tMax does not apply */
#define WIDE_LINE 0x10000000        /* ... the node holds a sphere of a top-level leaf: its boxes are tested against the ray's whole line (a tangent hit may lie behind the origin, Sphere.h:113) */
#define WIDE_INDEX 0x0FFFFFFF
#define TOP_ITEM 0x20000000u        /* on a record index: not a triangle record but a top-level object ordinal (leaf of the sub-tree over the objects of a top-level leaf) */
#define LINK_SENTINEL (-1)          /* on a lane's stack: below this entry lies the walk of a top-level leaf's sub-tree (scene frame), above it a mesh walk */
#define TRI_SUB 0x40000000u         /* on a record index: a run inside a fat leaf (reached through the leaf's sub-tree) */
#define TRI_EXACT 0x10000000u       /* on a pending record index (quantised walk): the record passed the plane filter (tri_maybe), the reference's test is still to run */
#define TRI_LEAF_END 2u             /* record word 2.w: bit 0 = last of its run, bit 1 = last of its reference leaf */
RT_DEV uint32_t leaf_first(int link) { return (uint32_t)(-2 - link); }

/* Result-preserving pruning. A child whose every possible hit lies beyond the distance of interest cannot
 * change the reference's answer (the hit would lose the strict `<` of Mesh.cpp:114, or lie beyond the light),
 * so it is skipped. "Every possible hit" needs care: a triangle record is the mesh triangle scaled by 1.0001f
 * about the origin (Mesh.cpp:107-109) and accepted with a tolerance on the barycentric sum (Triangle.h:105), so
 * its hits can lie outside the (unscaled) box of its leaf -- a little for ordinary triangles, without limit for
 * degenerate ones. The upload bounds this per child (rtgpu.hip, tri_excursion / HitBox): words 14-15 of a
 * 64-byte mesh node hold, for the left and the right child, how far accepted hits below it can leave its box
 * (inf = no bound). Moving a box face by e moves a slab entry time by at most e*max|1/d|, hence every hit
 * below a child has  t >= (tEnter - e*max|1/d|) * (1 - 1e-5).  The reference's own slab test still decides
 * which children are entered and in which order. */
RT_DEV float max_abs_inv(const Ray& r) {
    return fmaxf(fabsf(r.inv.x), fmaxf(fabsf(r.inv.y), fabsf(r.inv.z)));    /* inf for axis-parallel rays: nothing is pruned */
}
/* NaN (0 * inf) compares false: not pruned */
RT_DEV bool beyond(float tEnter, float slack, float bound) { return (tEnter - slack) * 0.99999f > bound; }

/* ---- a ray in the 16-bit plane frame of a mesh (see "quantised shadow nodes" further down): plane time =
 * q*qa + qn (entry planes) / qf (exit planes).
 * A packed pair of planes (lo | hi << 16) is never converted to float: one v_perm_b32 drops the 16-bit plane into the
 * mantissa of 2^23 (bytes: 0x4B, 0x00, plane), which IS the float 8388608 + q, and which half it takes -- the plane the
 * ray enters through or the one it leaves through, AccelerationStruct.cpp:22 -- is the selector, a per-ray constant. That
 * replaces a rotate and two conversions per axis and child by two permutes. qn and qf absorb the 2^23: the constant
 * c - 2^23*qa is rounded to half a unit of |qa| at worst, i.e. half a quantum of the frame in time, so both are widened by a
 * whole one (on top of the slack for the rounding of the plane arithmetic itself). */
struct QuantRay {
    F3 qa, qn, qf;
    uint32_t seln[3];             /* v_perm_b32 selectors: the half of a plane pair the ray enters through (the other half: ^ QUANT_SEL_FLIP) */
};
#define QUANT_SEL_LO 0x070C0100u
#define QUANT_SEL_HI 0x070C0302u
#define QUANT_SEL_FLIP 0x00000202u
RT_DEV float quant_plane(uint32_t w, uint32_t sel) { return __uint_as_float(__builtin_amdgcn_perm(0x4B000000u, w, sel)); }
RT_DEV void quant_setup(QuantRay& q, const Ray& r, const DMesh& m) {
    float M = 1.01f * fmaxf(fmaxf(fabsf(r.o.x), fabsf(r.o.y)), fmaxf(fabsf(r.o.z), m.amax));
    float ix = fabsf(r.inv.x), iy = fabsf(r.inv.y), iz = fabsf(r.inv.z);
    /* an infinite 1/d gives NaN or +-inf plane times in both forms; its slack is irrelevant, 0 keeps the constants clean */
    q.qa = f3(m.qstep[0] * r.inv.x, m.qstep[1] * r.inv.y, m.qstep[2] * r.inv.z);
    float ex = 2e-6f * M * (ix < INFINITY ? ix : 0.f) + fabsf(q.qa.x), ey = 2e-6f * M * (iy < INFINITY ? iy : 0.f) + fabsf(q.qa.y),
          ez = 2e-6f * M * (iz < INFINITY ? iz : 0.f) + fabsf(q.qa.z);
    float bx = (m.qmin[0] - r.o.x) * r.inv.x, by = (m.qmin[1] - r.o.y) * r.inv.y, bz = (m.qmin[2] - r.o.z) * r.inv.z;
    q.qn = f3(__builtin_fmaf(-8388608.0f, q.qa.x, bx - ex), __builtin_fmaf(-8388608.0f, q.qa.y, by - ey), __builtin_fmaf(-8388608.0f, q.qa.z, bz - ez));
    q.qf = f3(__builtin_fmaf(-8388608.0f, q.qa.x, bx + ex), __builtin_fmaf(-8388608.0f, q.qa.y, by + ey), __builtin_fmaf(-8388608.0f, q.qa.z, bz + ez));
    const bool sx = r.inv.x < 0.0f, sy = r.inv.y < 0.0f, sz = r.inv.z < 0.0f;      /* a negative 1/d swaps the roles of the two planes (AccelerationStruct.cpp:22) */
    q.seln[0] = sx ? QUANT_SEL_HI : QUANT_SEL_LO;
    q.seln[1] = sy ? QUANT_SEL_HI : QUANT_SEL_LO;
    q.seln[2] = sz ? QUANT_SEL_HI : QUANT_SEL_LO;
}
/* conservative entry time of one quantised box (three packed plane pairs); hit = the slab is not empty within [lo, lim] */
RT_DEV float quant_entry(uint32_t wx, uint32_t wy, uint32_t wz, const QuantRay& s, float lim, bool& hit, float lo = 0.0f) {
    float a = __builtin_fmaf(quant_plane(wx, s.seln[0]), s.qa.x, s.qn.x), d = __builtin_fmaf(quant_plane(wx, s.seln[0] ^ QUANT_SEL_FLIP), s.qa.x, s.qf.x);
    float b = __builtin_fmaf(quant_plane(wy, s.seln[1]), s.qa.y, s.qn.y), e = __builtin_fmaf(quant_plane(wy, s.seln[1] ^ QUANT_SEL_FLIP), s.qa.y, s.qf.y);
    float c = __builtin_fmaf(quant_plane(wz, s.seln[2]), s.qa.z, s.qn.z), f = __builtin_fmaf(quant_plane(wz, s.seln[2] ^ QUANT_SEL_FLIP), s.qa.z, s.qf.z);
    float tMin = fmaxf(fmaxf(fmaxf(a, b), c), lo);
    float tMax = fminf(fminf(fminf(d, e), f), lim);
    hit = tMin <= tMax;
    return tMin;
}

/* BVH_Node::intersect + RayTraceMesh::intersect (AccelerationStruct.cpp:255-283, Mesh.cpp:97-123):
 * all leaves whose box chain is hit, near child first, first minimal t wins (strict <).
 * `bound` = the best t found so far in the leaf loop of intersectObjects: a mesh hit only counts
 * if it is strictly closer (AccelerationStruct.cpp:307). */
template <bool SUB>     /* SUB: fat leaves carry sub-trees (DScene::has_sub and quant_ok) */
RT_DEV void mesh_closest(const DScene& sc, const DMesh& m, const Ray& r, float bound, int* stk, int stride, int sp0,
                         bool& found, float& bestT, uint32_t& bestG, float& bw0, float& bw1, float& bw2) {
    found = false;
    bestT = FLT_MAX;
    bestG = 0; bw0 = bw1 = bw2 = 0.f;
    const float mi = max_abs_inv(r);
    QuantRay qr;
    bool have_q = false;
    int sp = sp0;
    int link = m.root;
    for (;;) {
        if (link >= 0) {
            const float4* nd = sc.mnodes + 4 * (size_t)link;
            float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
            float tL, tR;
            bool hL = slab(r, a.x, a.y, a.z, a.w, b.x, b.y, tL);
            bool hR = slab(r, b.z, b.w, c.x, c.y, c.z, c.w, tR);
            int left = __float_as_int(d.x), right = __float_as_int(d.y);
            float lim = fminf(bound, bestT);
            bool nearLeft = tL <= tR;                         /* AccelerationStruct.cpp:273 */
            hL = hL && !beyond(tL, d.z * mi, lim);
            hR = hR && !beyond(tR, d.w * mi, lim);
            if (hL && hR) {
                if (nearLeft) { stk[lds_index(sp++, stride)] = right; link = left; }
                else { stk[lds_index(sp++, stride)] = left; link = right; }
                continue;
            }
            if (hL) { link = left; continue; }
            if (hR) { link = right; continue; }
        } else {
            /* Mesh.cpp:111-119: strict <, so of equal t the first in the reference's order wins; the records of a fat
             * leaf are regrouped (rtgpu.hip, LeafSplitter), word 3.w is their position in that order */
            uint32_t g = leaf_first(link);
            bool mine = false;
            uint32_t bestP = 0;
            auto candidate = [&](uint32_t gi, const float4* q, float4 q2, float4 q3) {
                float t, w0, w1, w2;
                if (tri_test(r, q[0], q[1], q2, q3, t, w0, w1, w2)) {
                    const uint32_t p = __float_as_uint(q3.w);
                    if (t < bestT || (mine && t == bestT && p < bestP)) {
                        found = true; bestT = t; bestG = gi; bw0 = w0; bw1 = w1; bw2 = w2;
                        mine = true; bestP = p;
                    }
                }
            };
            const float4 gate = SUB ? sc.tri[RT_TRI_STRIDE * (size_t)g + 6] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (__float_as_uint(gate.y) != 0u) {
                /* a fat leaf: only the runs whose HitBox the ray reaches within the distance of interest can hold a
                 * hit that counts (same boxes and test as the shadow walk, rtgpu.hip quantise_mesh) */
                if (!have_q) { quant_setup(qr, r, m); have_q = true; }
                const int spl = sp;
                int sub = __float_as_int(gate.x);
                for (;;) {
                    if (sub >= 0) {
                        const uint4* nd = sc.wnodes + 4 * (size_t)(sub & WIDE_INDEX);
                        uint4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
                        const float lim = (sub & WIDE_UNPRUNED) ? INFINITY : fminf(bound, bestT) * 1.00002f;
                        bool h0, h1, h2, h3;
                        quant_entry(a.x, a.y, a.z, qr, lim, h0);
                        quant_entry(a.w, b.x, b.y, qr, lim, h1);
                        quant_entry(b.z, b.w, c.x, qr, lim, h2);
                        quant_entry(c.y, c.z, c.w, qr, lim, h3);
                        if (h0) stk[lds_index(sp++, stride)] = (int)d.x;
                        if (h1) stk[lds_index(sp++, stride)] = (int)d.y;
                        if (h2) stk[lds_index(sp++, stride)] = (int)d.z;
                        if (h3) stk[lds_index(sp++, stride)] = (int)d.w;
                    } else if (sub != LINK_NONE) {       /* (a ray with a NaN in it "hits" every slot of a wide node, the unused ones with their LINK_NONE included: quant_entry drops NaN plane times) */
                        uint32_t gi = leaf_first(sub) & ~TRI_SUB;
                        for (bool last = false; !last; gi++) {
                            const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)gi;
                            float4 q2 = q[2];
                            last = __float_as_uint(q2.w) != 0u;
                            candidate(gi, q, q2, q[3]);
                        }
                    }
                    if (sp == spl) break;
                    sub = stk[lds_index(--sp, stride)];
                }
            } else {
                for (bool last = false; !last; g++) {
                    const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)g;
                    float4 q2 = q[2];
                    last = (__float_as_uint(q2.w) & TRI_LEAF_END) != 0u;
                    candidate(g, q, q2, q[3]);
                }
            }
        }
        if (sp == sp0) break;
        link = stk[lds_index(--sp, stride)];
    }
}

/* Of two leaves of one mesh that the reference both reaches, does it reach A first? Their order is decided at their
 * lowest common ancestor: BVH_Node::intersect lists the leaves below the nearer child first (tL <= tR: left,
 * AccelerationStruct.cpp:270-279). The path words (rtgpu.hip: bit 63 = the root's decision, 1 = right; depth in the low
 * six bits) locate that node; both of its children are hit (each holds a reached leaf, and the boxes are nested), so both
 * entry times are the ones the reference compares. Only runs for two hits with the very same t in different leaves. */
RT_DEV bool leaf_comes_first(const DScene& sc, const DMesh& m, const Ray& r, unsigned long long wa, unsigned long long wb) {
    const unsigned long long x = (wa ^ wb) & ~63ull;
    if (x == 0ull) return false;
    const int k = __builtin_clzll(x);                   /* decisions the two paths share = depth of the common ancestor */
    int node = m.root;
    for (int i = 0; i < k && node >= 0; i++) {
        const float4 d = sc.mnodes[4 * (size_t)node + 3];
        node = ((wa >> (63 - i)) & 1ull) ? __float_as_int(d.y) : __float_as_int(d.x);
    }
    if (node < 0) return false;
    const float4* nd = sc.mnodes + 4 * (size_t)node;
    const float4 a = nd[0], b = nd[1], c = nd[2];
    float tL, tR;
    slab(r, a.x, a.y, a.z, a.w, b.x, b.y, tL);
    slab(r, b.z, b.w, c.x, c.y, c.z, c.w, tR);
    const bool nearLeft = tL <= tR;                     /* AccelerationStruct.cpp:273 */
    const bool aLeft = ((wa >> (63 - k)) & 1ull) == 0ull;
    return aLeft == nearLeft;
}

/* mesh_closest over the 4-wide quantised tree of the shadow walk (DScene::wide_closest): half the node visits, a quarter
 * of the node bytes. What the reference returns is the hit with the smallest t among the triangles of the leaves it
 * reaches, and of equal t the one it meets first. It reaches exactly the leaves whose own box passes its test (see
 * "quantised shadow nodes" below), so the route to the leaves is free: here the nearest hit child of a wide node goes
 * first and children beyond the best hit so far are dropped (their boxes hold every hit accepted below them). The
 * leaf's own box is tested with the reference's arithmetic, the triangles likewise, and a tie in t between two leaves is
 * settled by leaf_comes_first -- the same hit, bit for bit. */
template <bool SUB>
RT_DEV void mesh_closest_wide(const DScene& sc, const DMesh& m, const Ray& r, float bound, int* stk, int stride, int sp0,
                              bool& found, float& bestT, uint32_t& bestG, float& bw0, float& bw1, float& bw2) {
    found = false;
    bestT = FLT_MAX;
    bestG = 0; bw0 = bw1 = bw2 = 0.f;
    int link = m.wroot;
    if (link == LINK_NONE) return;
    QuantRay qr;
    quant_setup(qr, r, m);
    if (link >= 0) {            /* the mesh's whole frame, as in shadow_top_item */
        bool may;
        quant_entry(0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, qr, (link & WIDE_UNPRUNED) ? 100.0f : fminf(100.0f, bound * 1.00002f), may);
        if (!may) return;
    }
    uint32_t bestLeaf = 0xFFFFFFFFu, bestP = 0;
    unsigned long long bestPath = 0ull;
    int sp = sp0;
    for (;;) {
        if (link >= 0) {
            const float near = fminf(bound, bestT) * 1.00002f;
            const float lim = (link & WIDE_UNPRUNED) ? 100.0f : fminf(100.0f, near);      /* tMax of AABB::intersect */
            const uint4* nd = sc.wnodes + 4 * (size_t)(link & WIDE_INDEX);
            const uint4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
            bool h0, h1, h2, h3;
            float t0 = quant_entry(a.x, a.y, a.z, qr, lim, h0);
            float t1 = quant_entry(a.w, b.x, b.y, qr, lim, h1);
            float t2 = quant_entry(b.z, b.w, c.x, qr, lim, h2);
            float t3 = quant_entry(c.y, c.z, c.w, qr, lim, h3);
            const int l0 = (int)d.x, l1 = (int)d.y, l2 = (int)d.z, l3 = (int)d.w;
            t0 = h0 ? t0 : INFINITY; t1 = h1 ? t1 : INFINITY; t2 = h2 ? t2 : INFINITY; t3 = h3 ? t3 : INFINITY;
            float ta = t0; int la = l0, ia = 0;
            if (t1 < ta) { ta = t1; la = l1; ia = 1; }
            if (t2 < ta) { ta = t2; la = l2; ia = 2; }
            if (t3 < ta) { ta = t3; la = l3; ia = 3; }
            if (h0 || h1 || h2 || h3) {
                if (!(ta < INFINITY)) { la = h0 ? l0 : h1 ? l1 : h2 ? l2 : l3; ia = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3; }    /* (hit with a NaN or infinite entry time) */
                if (h0 && ia != 0) stk[lds_index(sp++, stride)] = l0;
                if (h1 && ia != 1) stk[lds_index(sp++, stride)] = l1;
                if (h2 && ia != 2) stk[lds_index(sp++, stride)] = l2;
                if (h3 && ia != 3) stk[lds_index(sp++, stride)] = l3;
                link = la;
                continue;
            }
        } else if (link != LINK_NONE) {          /* (LINK_NONE: the unused slot of a wide node, which only a ray with a NaN in it can "hit": quant_entry drops NaN plane times) */
            const uint32_t g = leaf_first(link);
            const float4* lq = sc.tri + RT_TRI_STRIDE * (size_t)g;
            const float4 b0 = lq[4], b1 = lq[5];
            float tEnter;
            if (slab(r, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tEnter)) {     /* the reference's test of the leaf's own box */
                const unsigned long long path = (unsigned long long)__float_as_uint(b1.z) | ((unsigned long long)__float_as_uint(b1.w) << 32);
                auto candidate = [&](uint32_t gi, const float4* q, float4 q2, float4 q3) {
                    float t, w0, w1, w2;
                    if (tri_test(r, q[0], q[1], q2, q3, t, w0, w1, w2)) {
                        const uint32_t p = __float_as_uint(q3.w);
                        bool better = t < bestT;
                        if (!better && found && t == bestT)      /* Mesh.cpp:114, strict <: of equal t the one the reference meets first */
                            better = bestLeaf == g ? p < bestP : leaf_comes_first(sc, m, r, path, bestPath);
                        if (better) {
                            found = true; bestT = t; bestG = gi; bw0 = w0; bw1 = w1; bw2 = w2;
                            bestLeaf = g; bestP = p; bestPath = path;
                        }
                    }
                };
                const float4 gate = SUB ? lq[6] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (__float_as_uint(gate.y) != 0u) {
                    /* a fat leaf: its sub-tree of runs, walked on the spot (the hits of the whole leaf count as one leaf's) */
                    const int spl = sp;
                    int sub = __float_as_int(gate.x);
                    for (;;) {
                        if (sub >= 0) {
                            const uint4* nd = sc.wnodes + 4 * (size_t)(sub & WIDE_INDEX);
                            const uint4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
                            const float lim = (sub & WIDE_UNPRUNED) ? INFINITY : fminf(bound, bestT) * 1.00002f;     /* (no tMax below the leaf's own box) */
                            bool h0, h1, h2, h3;
                            quant_entry(a.x, a.y, a.z, qr, lim, h0);
                            quant_entry(a.w, b.x, b.y, qr, lim, h1);
                            quant_entry(b.z, b.w, c.x, qr, lim, h2);
                            quant_entry(c.y, c.z, c.w, qr, lim, h3);
                            if (h0) stk[lds_index(sp++, stride)] = (int)d.x;
                            if (h1) stk[lds_index(sp++, stride)] = (int)d.y;
                            if (h2) stk[lds_index(sp++, stride)] = (int)d.z;
                            if (h3) stk[lds_index(sp++, stride)] = (int)d.w;
                        } else if (sub != LINK_NONE) {
                            uint32_t gi = leaf_first(sub) & ~TRI_SUB;
                            for (bool last = false; !last; gi++) {
                                const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)gi;
                                float4 q2 = q[2];
                                last = __float_as_uint(q2.w) != 0u;
                                candidate(gi, q, q2, q[3]);
                            }
                        }
                        if (sp == spl) break;
                        sub = stk[lds_index(--sp, stride)];
                    }
                } else {
                    uint32_t gi = g;
                    for (bool last = false; !last; gi++) {
                        const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)gi;
                        float4 q2 = q[2];
                        last = (__float_as_uint(q2.w) & TRI_LEAF_END) != 0u;
                        candidate(gi, q, q2, q[3]);
                    }
                }
            }
        }
        if (sp == sp0) break;
        link = stk[lds_index(--sp, stride)];
    }
}

/* Same candidate set, shadow use: is any triangle hit at t <= bound? (Order does not matter for
 * that answer, so the nearer child goes first to find an occluder early.) With prune=false and
 * bound just below FLT_MAX this is "does RayTraceMesh::intersect return a hit at all". */
RT_DEV bool mesh_any(const DScene& sc, const DMesh& m, const Ray& r, float bound, bool prune, int* stk, int stride, int sp0) {
    const float mi = prune ? max_abs_inv(r) : INFINITY;
    int sp = sp0;
    int link = m.root;
    for (;;) {
        if (link >= 0) {
            const float4* nd = sc.mnodes + 4 * (size_t)link;
            float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
            float tL, tR;
            bool hL = slab(r, a.x, a.y, a.z, a.w, b.x, b.y, tL);
            bool hR = slab(r, b.z, b.w, c.x, c.y, c.z, c.w, tR);
            int left = __float_as_int(d.x), right = __float_as_int(d.y);
            bool nearLeft = tL <= tR;
            hL = hL && !beyond(tL, d.z * mi, bound);
            hR = hR && !beyond(tR, d.w * mi, bound);
            if (hL && hR) {
                if (nearLeft) { stk[lds_index(sp++, stride)] = right; link = left; }
                else { stk[lds_index(sp++, stride)] = left; link = right; }
                continue;
            }
            if (hL) { link = left; continue; }
            if (hR) { link = right; continue; }
        } else {
            uint32_t g = leaf_first(link);
            for (bool last = false; !last; g++) {
                const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)g;
                float4 q2 = q[2];
                last = (__float_as_uint(q2.w) & TRI_LEAF_END) != 0u;
                float t, w0, w1, w2;
                if (tri_test(r, q[0], q[1], q2, q[3], t, w0, w1, w2) && t <= bound) return true;
            }
        }
        if (sp == sp0) break;
        link = stk[lds_index(--sp, stride)];
    }
    return false;
}

/* BVH_Node::intersectObjects(ray) with stopAtFirst=false (AccelerationStruct.cpp:286-337):
 * every hit child is visited, left first; ties keep the earlier result. */
template <bool SUB, bool WIDE>     /* WIDE: meshes are walked over the 4-wide quantised tree (mesh_closest_wide; a kernel of its own, the two walks in one cost the binary one a wave per SIMD) */
RT_DEV HitD closest_hit(const DScene& sc, const Ray& r, int* stk, int stride) {
    HitD best;
    best.exists = 0; best.t = FLT_MAX; best.obj = -1; best.tri = 0; best.mat = 0;
    best.n = f3s(0.f); best.pos = f3s(0.f); best.u = best.v = 0.f;
    int kind = -1;
    uint32_t bestG = 0;
    float bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
    int sp = 0;
    int link = sc.top_root;
    for (;;) {
        if (link >= 0) {
            const float4* nd = sc.tnodes + 4 * (size_t)link;
            float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
            float tL, tR;
            bool hL = slab(r, a.x, a.y, a.z, a.w, b.x, b.y, tL);
            bool hR = slab(r, b.z, b.w, c.x, c.y, c.z, c.w, tR);
            int left = __float_as_int(d.x), right = __float_as_int(d.y);
            if (hL && hR) { stk[lds_index(sp++, stride)] = right; link = left; continue; }
            if (hL) { link = left; continue; }
            if (hR) { link = right; continue; }
        } else {
            uint2 lf = sc.tleaves[-1 - link];
            for (uint32_t i = 0; i < lf.y; i++) {
                int ord = (int)sc.titems[lf.x + i];
                if (ord < sc.n_squares) {
                    const float4* q = sc.squares + 4 * ord;
                    float4 q3 = q[3];
                    float t, u, v;
                    if (square_test(r, q[0], q[1], q[2], q3, t, u, v) && t < best.t) {
                        best.exists = 1; best.t = t; best.obj = ord; best.tri = 0; kind = 0;
                        best.n = f3(q3.x, q3.y, q3.z); best.u = u; best.v = v; best.mat = __float_as_int(q3.w);
                    }
                } else if (ord < sc.n_squares + sc.n_spheres) {
                    const float4* q = sc.spheres + 2 * (ord - sc.n_squares);
                    float t; F3 n;
                    if (sphere_test(r, q[0], t, n) && t < best.t) {
                        best.exists = 1; best.t = t; best.obj = ord; best.tri = 0; kind = 1;
                        best.n = n; best.u = best.v = 0.f; best.mat = __float_as_int(q[1].x);
                    }
                } else {
                    DMesh m = sc.meshes[ord - sc.n_squares - sc.n_spheres];
                    bool found; float t; uint32_t g; float w0, w1, w2;
                    if (WIDE) mesh_closest_wide<SUB>(sc, m, r, best.t, stk, stride, sc.top_cap, found, t, g, w0, w1, w2);
                    else mesh_closest<SUB>(sc, m, r, best.t, stk, stride, sc.top_cap, found, t, g, w0, w1, w2);
                    if (found && t < best.t) {
                        best.exists = 1; best.t = t; best.obj = ord; kind = 2;
                        bestG = g; bw0 = w0; bw1 = w1; bw2 = w2; best.mat = m.material;
                    }
                }
            }
        }
        if (sp == 0) break;
        link = stk[lds_index(--sp, stride)];
    }
    if (best.exists) {
        if (kind == 2) {                                         /* Mesh.cpp:117-134 */
            uint4 id = sc.tri_ids[bestG];
            const float* nn = sc.normals;
            F3 n0 = f3(nn[3 * (size_t)id.x], nn[3 * (size_t)id.x + 1], nn[3 * (size_t)id.x + 2]);
            F3 n1 = f3(nn[3 * (size_t)id.y], nn[3 * (size_t)id.y + 1], nn[3 * (size_t)id.y + 2]);
            F3 n2 = f3(nn[3 * (size_t)id.z], nn[3 * (size_t)id.z + 1], nn[3 * (size_t)id.z + 2]);
            best.n = normalized(add(add(mul(bw0, n0), mul(bw1, n1)), mul(bw2, n2)));
            best.tri = (int)id.w;
            best.u = bw0; best.v = bw1;
        }
        best.pos = add(r.o, mul(best.t, r.d));                   /* Mesh.cpp:131, Sphere.h:136, Square.h:105 */
    }
    return best;
}

/* BVH_Node::intersectObjects(ray, lightDistance, stopAtFirst=true) as Scene::isInShadow uses it
 * (Scene.h:160-168): leaves in left-first order; the first leaf whose result exists decides --
 * occluded iff that result has t <= lightDistance, which inside a leaf is reached exactly when
 * some object of the leaf is hit at t <= lightDistance (AccelerationStruct.cpp:293-311,326-330).
 * mode 0: walk the leaves looking for an object hit at t <= lightDistance (meshes pruned by the
 *         light distance). Returns the number of leaves visited before the occluding one, or -1.
 * mode 1: for the first `limit` leaves, is any object hit at all (any t)? Such a leaf returns an
 *         existing result beyond the light and hides the occluder behind it
 *         (AccelerationStruct.cpp:327-330). Returns 1 / 0. */
/* FRONT (k_shadow_front): meshes are not walked. A mesh whose plane frame the ray misses cannot be hit (every box of
 * its shadow tree, hence every leaf the reference can reach, lies inside; needs DScene::quant_ok); one that it may hit
 * sets *defer and ends the walk with -2: that ray goes to the persistent kernel as a whole. */
template <bool FRONT>
RT_DEV int shadow_walk(const DScene& sc, const Ray& r, float lightDistance, int mode, int limit, int* stk, int stride, bool* defer = nullptr) {
    const float bound = mode == 0 ? lightDistance : 3.4028233e38f;   /* largest float below FLT_MAX: t < FLT_MAX */
    int visited = 0;
    int sp = 0;
    int link = sc.top_root;
    for (;;) {
        if (link >= 0) {
            const float4* nd = sc.tnodes + 4 * (size_t)link;
            float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
            float tL, tR;
            bool hL = slab(r, a.x, a.y, a.z, a.w, b.x, b.y, tL);
            bool hR = slab(r, b.z, b.w, c.x, c.y, c.z, c.w, tR);
            int left = __float_as_int(d.x), right = __float_as_int(d.y);
            if (hL && hR) { stk[lds_index(sp++, stride)] = right; link = left; continue; }
            if (hL) { link = left; continue; }
            if (hR) { link = right; continue; }
        } else {
            if (mode == 1 && visited >= limit) return 0;
            uint2 lf = sc.tleaves[-1 - link];
            for (uint32_t i = 0; i < lf.y; i++) {
                int ord = (int)sc.titems[lf.x + i];
                bool hit;
                if (ord < sc.n_squares) {
                    const float4* q = sc.squares + 4 * ord;
                    float t, u, v;
                    hit = square_test(r, q[0], q[1], q[2], q[3], t, u, v) && t <= bound;
                } else if (ord < sc.n_squares + sc.n_spheres) {
                    const float4* q = sc.spheres + 2 * (ord - sc.n_squares);
                    float t; F3 n;
                    hit = sphere_test(r, q[0], t, n) && t <= bound;
                } else {
                    DMesh m = sc.meshes[ord - sc.n_squares - sc.n_spheres];
                    if (FRONT) {
                        hit = false;
                        bool may = m.wroot != LINK_NONE;
                        if (m.wroot >= 0) {
                            QuantRay qr;
                            quant_setup(qr, r, m);
                            quant_entry(0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, qr, 100.0f, may);
                        }
                        if (may) { *defer = true; return -2; }
                    } else hit = mesh_any(sc, m, r, bound, mode == 0, stk, stride, sc.top_cap);
                }
                if (hit) return mode == 0 ? visited : 1;
            }
            visited++;
        }
        if (sp == 0) break;
        link = stk[lds_index(--sp, stride)];
    }
    return mode == 0 ? -1 : 0;
}

RT_DEV bool occluded(const DScene& sc, const Ray& r, float lightDistance, int* stk, int stride) {
    int before = shadow_walk<false>(sc, r, lightDistance, 0, 0, stk, stride);
    if (before < 0) return false;
    if (before == 0) return true;
    return shadow_walk<false>(sc, r, lightDistance, 1, before, stk, stride) == 0;
}
/* the same for a ray that stays clear of every mesh; otherwise defer is set and the result means nothing */
RT_DEV bool occluded_front(const DScene& sc, const Ray& r, float lightDistance, int* stk, int stride, bool& defer) {
    defer = false;
    int before = shadow_walk<true>(sc, r, lightDistance, 0, 0, stk, stride, &defer);
    if (defer || before < 0) return false;
    if (before == 0) return true;
    return shadow_walk<true>(sc, r, lightDistance, 1, before, stk, stride, &defer) == 0;
}

/* Scene::isInShadow, Scene.h:160-168 */
RT_DEV bool in_shadow(const DScene& sc, F3 P, F3 L, int* stk, int stride) {
    F3 dir = sub(L, P);
    float length = len(dir);
    dir = normalized(dir);
    Ray r = make_ray(add(P, mul(0.001f, dir)), dir);
    return occluded(sc, r, length, stk, stride);
}

/* Material::reflect, Material.h:60-65 */
RT_DEV F3 reflect_dir(F3 d, F3 n) {
    F3 md = mul(-1.f, d);
    F3 parallel = sub(mul(dot(md, n), n), md);
    return normalized(add(n, parallel));
}

RT_DEV void store_state(PathState* s, F3 o, F3 d, F3 thr, uint32_t slot, uint32_t pixel, uint32_t sample, uint64_t code,
                        int remaining, uint32_t flags) {
    float4* p = reinterpret_cast<float4*>(s);
    p[0] = make_float4(o.x, o.y, o.z, d.x);
    p[1] = make_float4(d.y, d.z, thr.x, thr.y);
    p[2] = make_float4(thr.z, __uint_as_float(slot), __uint_as_float(pixel), __uint_as_float(sample));
    p[3] = make_float4(__uint_as_float((uint32_t)code), __uint_as_float((uint32_t)(code >> 32)), __int_as_float(remaining),
                       __uint_as_float(flags));
}

/* Radiance sums are int64 fixed point (order-independent, so frames are bit-identical for any
 * tile sharding). The reference lets NaN/Inf pass through its float sums (main.cpp:217-220); here
 * they are recorded per slot and channel (bit 3k: NaN, 3k+1: +Inf, 3k+2: -Inf) and re-applied by
 * k_resolve. Finite magnitudes >= 6e7 are treated as infinite, and so is a channel whose running sum leaves +-2^26
 * (6.7e7 before the division by the sample count) -- the accumulator holds +-2^27. Contributions of magnitude >= 64
 * take the returning form of the atomic and look at the sum they leave behind (or wrap); smaller ones do not wait
 * for the result (the wait costs 5 % of a C3 frame if every add takes it): a million of them (2^20 per pixel and
 * channel, each below 64) move the sum by less than 2^26, so between two looks it stays inside the accumulator
 * and cannot wrap unseen. (Unnormalised normals out of a normal map, Scene.h:199-202, drive specular terms that far:
 * the cornell box has a few such pixels; they are saturated in the reference's frame too.) */
RT_DEV void accumulate(long long* acc, uint32_t* nonfinite, Counters* ctr, uint32_t slot, F3 c) {
    float v[3] = {c.x, c.y, c.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float x = v[k];
        if (!(fabsf(x) < 6.0e7f)) {
            uint32_t bit = (x != x) ? 1u : (x > 0.f ? 2u : 4u);
            atomicOr(nonfinite + slot, bit << (3 * k));
            atomicAdd(&ctr->nonfinite, 1ull);
            continue;
        }
        long long q = __double2ll_rn((double)x * RT_FIX_SCALE);
        unsigned long long* cell = reinterpret_cast<unsigned long long*>(acc + 3 * (size_t)slot + k);
        if (fabsf(x) >= 64.0f) {
            const long long before = (long long)atomicAdd(cell, (unsigned long long)q);
            const long long after = (long long)((unsigned long long)before + (unsigned long long)q);
            const bool wrapped = ((before ^ after) & (q ^ after)) < 0;     /* both operands of one sign, the sum of the other */
            const long long lim = 1ll << 62;                               /* 2^26 in units of 2^-36 */
            if (wrapped || after >= lim || after <= -lim) {
                atomicOr(nonfinite + slot, (q > 0 ? 2u : 4u) << (3 * k));
                atomicAdd(&ctr->nonfinite, 1ull);
            }
        } else if (q != 0) atomicAdd(cell, (unsigned long long)q);
    }
}

/* accumulate() for a whole wave whose neighbouring lanes mostly add to the same pixel (k_shade: the diffuse records of a
 * batch of primaries come sample after sample, pixel after pixel, so a wave holds a few runs of up to 64 equal slots):
 * the ordinary contributions (finite, below 64) are summed over each run of lanes with equal slot -- integer sums, so the
 * accumulators end with the same bits -- and the run's first lane adds the total: three atomics per run instead of per
 * lane, and the L2 no longer serialises dozens of adds to one address. Contributions that take accumulate()'s careful
 * paths (not finite, or large) still go one by one. Every lane of the wave must call this (active = false: nothing to add). */
RT_DEV void accumulate_runs(long long* acc, uint32_t* nonfinite, Counters* ctr, uint32_t slot, F3 c, bool active) {
    const float v[3] = {c.x, c.y, c.z};
    long long q[3] = {0, 0, 0};
    if (active) {
        F3 careful = f3s(0.f);
        bool any = false;
        float* cv[3] = {&careful.x, &careful.y, &careful.z};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (fabsf(v[k]) < 64.0f) q[k] = __double2ll_rn((double)v[k] * RT_FIX_SCALE);
            else { *cv[k] = v[k]; any = true; }
        }
        if (any) accumulate(acc, nonfinite, ctr, slot, careful);      /* (its other channels are zeros: nothing added) */
    } else slot = 0xFFFFFFFFu;
    const int lane = __lane_id();
    const uint32_t prev = __shfl_up(slot, 1);
    const bool head = lane == 0 || prev != slot;
    const unsigned long long heads = __ballot(head);
    const unsigned long long later = lane == 63 ? 0ull : heads & ~((2ull << lane) - 1ull);
    const int next_head = later ? __builtin_ctzll(later) : 64;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const long long other = __shfl_down(q[k], off);
            if (lane + off < next_head) q[k] += other;
        }
    }
    if (head && active) {
#pragma unroll
        for (int k = 0; k < 3; k++)
            if (q[k] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(acc + 3 * (size_t)slot + k), (unsigned long long)q[k]);
    }
}

/* Block-aggregated append to the two queues of k_closest at once: every thread of the block calls this (no early
 * exits before it) with its own request counts. All waves of all blocks used to hit the same two counters with
 * returning atomics (plus two for the ray statistics): on scenes with cheap rays that contention, not the
 * traversal, set the kernel's time. Now one 64-bit atomic per block reserves both ranges (n_child in the low word,
 * n_diffuse in the high word of the same 8 bytes). s_tot: 2 * (waves per block) + 2 words of LDS. */
RT_DEV void block_reserve2(Counters* ctr, uint32_t want_child, uint32_t want_diffuse, uint32_t* s_tot, uint32_t& cbase, uint32_t& dbase) {
    const unsigned long long m1 = __ballot(want_child == 1), m2 = __ballot(want_child == 2), md = __ballot(want_diffuse != 0);
    const int lane = __lane_id();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const uint32_t cmine = __popcll(m1 & lt) + 2u * __popcll(m2 & lt), ctot = __popcll(m1) + 2u * __popcll(m2);
    const uint32_t dmine = __popcll(md & lt), dtot = __popcll(md);
    const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    if (nwaves == 1) {                                   /* one wave per block: no LDS, no barrier */
        unsigned long long base = 0ull;
        if (lane == 0 && (ctot | dtot)) base = atomicAdd(reinterpret_cast<unsigned long long*>(&ctr->n_child), (unsigned long long)ctot | ((unsigned long long)dtot << 32));
        base = __shfl(base, 0);
        cbase = (uint32_t)base + cmine;
        dbase = (uint32_t)(base >> 32) + dmine;
        return;
    }
    if (lane == 0) { s_tot[2 * wave] = ctot; s_tot[2 * wave + 1] = dtot; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t c = 0, d = 0;
        for (int w = 0; w < nwaves; w++) { uint32_t a = s_tot[2 * w], b = s_tot[2 * w + 1]; s_tot[2 * w] = c; s_tot[2 * w + 1] = d; c += a; d += b; }
        unsigned long long base = 0ull;
        if (c | d) base = atomicAdd(reinterpret_cast<unsigned long long*>(&ctr->n_child), (unsigned long long)c | ((unsigned long long)d << 32));
        s_tot[2 * nwaves] = (uint32_t)base; s_tot[2 * nwaves + 1] = (uint32_t)(base >> 32);
    }
    __syncthreads();
    cbase = s_tot[2 * nwaves] + s_tot[2 * wave] + cmine;
    dbase = s_tot[2 * nwaves + 1] + s_tot[2 * wave + 1] + dmine;
}

/* ------------------------------------------------------------------ kernels */

/* main.cpp:191-201 + matrixUtilities.h:198-222: one lane per (pixel, sample). */
RT_DEV void primary_ray(const CameraD& cam, const FrameD& fr, int x, int y, int s, F3& pos, F3& dir) {
    float ju = 0.5f, jv = 0.5f;
    uint32_t pixel = (uint32_t)(x + y * fr.img_w);
    if (fr.jitter) {
        ju = (float)(rt_rng_impl(fr.seed, pixel, (uint32_t)s, 0, 0xFFFFFFFFu, 0, 0) & 0x7fffffffu) / 2147483648.0f;
        jv = (float)(rt_rng_impl(fr.seed, pixel, (uint32_t)s, 0, 0xFFFFFFFFu, 0, 1) & 0x7fffffffu) / 2147483648.0f;
    }
    float u = ((float)x + ju) / (float)fr.img_w;
    float v = ((float)y + jv) / (float)fr.img_h;
    double xn = 2.0 * (double)u - 1.0, yn = -(2.0 * (double)v - 1.0);
    const double* pi = cam.pinv;
    double a0 = pi[0] * xn + pi[4] * yn + pi[8] * 0.0 + pi[12] * 1.0;
    double a1 = pi[1] * xn + pi[5] * yn + pi[9] * 0.0 + pi[13] * 1.0;
    double a2 = pi[2] * xn + pi[6] * yn + pi[10] * 0.0 + pi[14] * 1.0;
    double a3 = pi[3] * xn + pi[7] * yn + pi[11] * 0.0 + pi[15] * 1.0;
    const double* mv = cam.mvinv;
    double r0 = mv[0] * a0 + mv[4] * a1 + mv[8] * a2 + mv[12] * a3;
    double r1 = mv[1] * a0 + mv[5] * a1 + mv[9] * a2 + mv[13] * a3;
    double r2 = mv[2] * a0 + mv[6] * a1 + mv[10] * a2 + mv[14] * a3;
    double r3 = mv[3] * a0 + mv[7] * a1 + mv[11] * a2 + mv[15] * a3;
    F3 p = f3((float)(r0 / r3), (float)(r1 / r3), (float)(r2 / r3));
    pos = f3(cam.eye[0], cam.eye[1], cam.eye[2]);
    dir = normalized(sub(p, pos));
}

RT_DEV void locate_sample(const TileD* tiles, int n_tiles, const FrameD& fr, unsigned long long k, int& x, int& y, int& s) {
    int lo = 0, hi = n_tiles - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (tiles[mid].first <= k) lo = mid; else hi = mid - 1;
    }
    TileD t = tiles[lo];
    unsigned long long local = k - t.first;
    /* (a 64-bit division costs about 150 instructions, a 32-bit one 25: the samples of one tile almost always fit 32 bits) */
    if ((local >> 32) == 0ull) {
        const uint32_t l32 = (uint32_t)local, pix = l32 / (uint32_t)fr.spp;
        s = fr.sample_base + (int)(l32 - pix * (uint32_t)fr.spp);
        const uint32_t row = pix / (uint32_t)t.w;
        x = t.x0 + (int)(pix - row * (uint32_t)t.w);
        y = t.y0 + (int)row;
        return;
    }
    unsigned long long pix = local / (unsigned)fr.spp;
    s = fr.sample_base + (int)(local % (unsigned)fr.spp);
    x = t.x0 + (int)(pix % (unsigned)t.w);
    y = t.y0 + (int)(pix / (unsigned)t.w);
}

__global__ void __launch_bounds__(256) k_raygen(CameraD cam, FrameD fr, const TileD* tiles, int n_tiles, unsigned long long first,
                                                uint32_t n, PathState* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int x, y, s;
    locate_sample(tiles, n_tiles, fr, first + i, x, y, s);
    F3 o, d;
    primary_ray(cam, fr, x, y, s, o, d);
    uint32_t pixel = (uint32_t)(x + y * fr.img_w);
    store_state(out + i, o, d, f3s(1.f), pixel, pixel, (uint32_t)s, 1ull, fr.max_depth, s == 0 ? 1u : 0u);
}

__global__ void __launch_bounds__(256) k_camera_rays(CameraD cam, FrameD fr, int x0, int y0, int w, int h, float* org, float* dir) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = (uint32_t)w * h * fr.spp;
    if (i >= n) return;
    uint32_t pix = i / fr.spp;
    int s = i % fr.spp;
    F3 o, d;
    primary_ray(cam, fr, x0 + (int)(pix % w), y0 + (int)(pix / w), s, o, d);
    org[3 * i] = o.x; org[3 * i + 1] = o.y; org[3 * i + 2] = o.z;
    dir[3 * i] = d.x; dir[3 * i + 1] = d.y; dir[3 * i + 2] = d.z;
}

/* seeds path states from explicit rays (rt_trace) */
__global__ void __launch_bounds__(256) k_seed(const float* org, const float* dir, const uint32_t* pixel, const uint32_t* sample,
                                              uint32_t first, uint32_t n, int max_depth, PathState* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t g = first + i;
    store_state(out + i, f3(org[3 * g], org[3 * g + 1], org[3 * g + 2]), f3(dir[3 * g], dir[3 * g + 1], dir[3 * g + 2]), f3s(1.f), g,
                pixel[g], sample[g], 1ull, max_depth, 0u);
}

/* After k_closest, one thread: the two queue lengths (and, while the front pass is on trial, the deferred rays so far) go to the
 * host's pinned counters and are zeroed for the next iteration -- one small launch instead of a fill and one or two copies
 * per iteration of the wavefront loop (what a short iteration costs is its stream operations, not its rays). */
__global__ void k_publish(Counters* ctr, Counters* host, int with_deferred) {
    host->n_child = ctr->n_child;
    host->n_diffuse = ctr->n_diffuse;
    if (with_deferred) host->deferred_sum = ctr->deferred_sum;
    ctr->n_child = 0u;
    ctr->n_diffuse = 0u;
    __threadfence_system();
}

/* Scene::rayTraceRecursive (Scene.h:174-280) for one path state: the recursion is unrolled into
 * child states carrying a throughput (radiance is linear in the child radiance, Scene.h:229,263-271). */
/* FRESH: the states from number fresh.n_old on are primaries of the queued tiles that no kernel has written out: the lane
 * makes its camera ray itself (k_raygen's arithmetic) instead of loading a 64-byte state that k_raygen would have stored a
 * moment before -- 8 GB less traffic on a 1080p frame at 30 spp. */
struct FreshD { CameraD cam; FrameD fr; const TileD* tiles; int n_tiles; unsigned long long first; uint32_t n_old; };
template <bool SUB, int BLOCK, bool FRESH, bool WIDE>      /* BLOCK: the largest block it is launched with (64 or RT_CBLOCK); the one-wave build is the better one for heavy rays */
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 8))) k_closest(DScene sc, const PathState* __restrict__ in, uint32_t n, PathState* child,
                                                      uint32_t child_cap, DiffuseRec* drec, unsigned long long* dcode, uint32_t* dtex,
                                                      uint32_t diffuse_cap, Counters* ctr, long long* acc, uint32_t* nonfinite, uint32_t* prim, FreshD fresh) {
    extern __shared__ __align__(16) int s_stack[];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n;
    if (!FRESH && sc.spread_shift > 0) {         /* (a small batch: one state per 2^shift lanes, see DScene::spread_shift) */
        active = (i & ((1u << sc.spread_shift) - 1u)) == 0u && (i >> sc.spread_shift) < n;
        i >>= sc.spread_shift;
    }
    int* stk = s_stack + threadIdx.x;
    const int stride = blockDim.x;

    uint32_t want_child = 0;
    bool want_diffuse = false, culled = false;
    F3 c_o[2], c_d[2], c_thr[2];
    uint64_t c_code[2];
    DiffuseRec rec;
    uint32_t rec_rgb = 0;
    uint32_t slot = 0, pixel = 0, sample = 0, flags = 0;
    uint64_t code = 0;
    int remaining = 0;
    F3 sky_add = f3s(0.f);        /* a miss adds the sky (below, by runs of lanes on one pixel) */
    bool missed = false;

    if (active) {
        F3 o, d, thr;
        if (FRESH && i >= fresh.n_old) {                           /* main.cpp:191-201, as k_raygen */
            int x, y, sm;
            locate_sample(fresh.tiles, fresh.n_tiles, fresh.fr, fresh.first + (i - fresh.n_old), x, y, sm);
            primary_ray(fresh.cam, fresh.fr, x, y, sm, o, d);
            thr = f3s(1.f);
            slot = pixel = (uint32_t)(x + y * fresh.fr.img_w); sample = (uint32_t)sm;
            code = 1ull; remaining = fresh.fr.max_depth; flags = sm == 0 ? 1u : 0u;
        } else {
            const float4* p = reinterpret_cast<const float4*>(in + i);
            float4 p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3];
            o = f3(p0.x, p0.y, p0.z); d = f3(p0.w, p1.x, p1.y); thr = f3(p1.z, p1.w, p2.x);
            slot = __float_as_uint(p2.y); pixel = __float_as_uint(p2.z); sample = __float_as_uint(p2.w);
            code = (uint64_t)__float_as_uint(p3.x) | ((uint64_t)__float_as_uint(p3.y) << 32);
            remaining = __float_as_int(p3.z); flags = __float_as_uint(p3.w);
        }

        Ray r = make_ray(o, d);
        HitD h = closest_hit<SUB, WIDE>(sc, r, stk, stride);
        if ((flags & 1u) && prim) {
            prim[2 * (size_t)slot] = h.exists ? (uint32_t)h.obj : 0xFFFFFFFFu;
            prim[2 * (size_t)slot + 1] = h.exists ? (uint32_t)h.tri : 0xFFFFFFFFu;
        }
        if (!h.exists) {                                           /* Scene.h:274-278 */
            float t = (float)(0.5 * ((double)r.d.y + 1.0));
            float omt = (float)(1.0 - (double)t);
            F3 sky = add(mul(omt, f3s(1.0f)), mul(t, f3(0.5f, 0.7f, 1.0f)));
            sky_add = comp(thr, sky);
            missed = true;
        } else {
            DMat m = sc.mats[h.mat];
            uint32_t matword = (uint32_t)h.mat;
            if (m.tex >= 0 || m.nmap >= 0) {                       /* image texture / normal map, Scene.h:184-202 */
                float uf = h.u, vf = h.v;
                if (h.obj >= sc.n_squares && h.obj < sc.n_squares + sc.n_spheres) sphere_uv(r.d, uf, vf);
                if (m.tex >= 0) {
                    DTex t = sc.textures[m.tex];
                    int pu = (int)((double)uf * (double)t.w * m.scale) % t.w, pv = (int)((double)vf * (double)t.h * m.scale) % t.h;
                    if (pu < 0) pu += t.w;
                    if (pv < 0) pv += t.h;
                    const uint8_t* px = t.data + ((size_t)pv * (size_t)t.w + (size_t)pu) * (size_t)t.ch;
                    rec_rgb = (uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16);
                    F3 col = f3((float)((double)px[0] / 255.0), (float)((double)px[1] / 255.0), (float)((double)px[2] / 255.0));
                    for (int k = 0; k < 3; k++) { m.kd[k] = comp_at(col, k); m.ks[k] = comp_at(col, k); }
                    matword |= 0x80000000u;                        /* k_shade takes kd = ks from the packed texel */
                }
                if (m.nmap >= 0) {                                 /* Material::applyTBN, Material.h:118-126: not remapped, not normalised */
                    F3 nm = texel_at(sc.textures[m.nmap], (double)uf, (double)vf, m.scale);
                    F3 tg = f3(1.f, 0.f, 0.f);
                    F3 bt = cross(h.n, tg);
                    h.n = f3(tg.x * nm.x + bt.x * nm.y + h.n.x * nm.z,      /* Mat3(t,b,n as columns) * nm, Vec3.h:156-163 */
                             tg.y * nm.x + bt.y * nm.y + h.n.y * nm.z,
                             tg.z * nm.x + bt.z * nm.y + h.n.z * nm.z);
                }
            }
            if (m.type == 3 && m.tex < 0) {                        /* checker: Material.h:153-161, Scene.h:51,192-197 (a texture takes precedence) */
                const float inv = (float)(1. / 3.f);
                int xi = (int)floorf(inv * h.pos.x), yi = (int)floorf(inv * h.pos.y), zi = (int)floorf(inv * h.pos.z);
                bool even = (xi + yi + zi) % 2 == 0;
                matword |= 0x40000000u | (even ? 0u : 0x20000000u);
            }
            if (m.type == 0 || m.type == 3 || remaining <= 0) {   /* Scene.h:204-222: shaded after the shadow pass */
                want_diffuse = true;
                /* Contribution culling. The hit adds thr * f * colour to its pixel, f in [0, 1] the shadow factor and colour the
                 * Phong sum over the lights (k_shade), and the sum goes through accumulate() channel by channel: a value below
                 * half a unit of the fixed-point accumulator (2^-37) adds exactly nothing. |colour| is bounded without tracing a
                 * single shadow ray: per light, Li * (ka + kd * |theta| + ks * pow(max(0, r.V), shininess)) with |theta| <= |N|
                 * (|Ldir| = 1) and |r.V| <= 2 |theta| |N| + 1. If max|thr| times that bound stays below 2^-39 (a factor four for
                 * the float rounding of the products) the hit cannot change a bit of the frame and its light_samples x lights
                 * shadow rays are not traced -- they are counted (rt_stats::rays_shadow_culled). A depth-50 recursion inside a
                 * glass mesh ends in billions of such hits: every internal reflection takes (1 - transparency) * R off the
                 * throughput. NaN and infinite throughputs compare false and are kept. */
                const float tmax = fmaxf(fabsf(thr.x), fmaxf(fabsf(thr.y), fabsf(thr.z)));
                if (sc.light_sum >= 0.f && tmax < 1e-7f) {
                    const float nl = len(h.n) * 1.0001f;
                    float kmax = fmaxf(fmaxf(fabsf(m.kd[0]), fabsf(m.kd[1])), fmaxf(fabsf(m.kd[2]), fmaxf(fabsf(m.ks[0]), fmaxf(fabsf(m.ks[1]), fabsf(m.ks[2])))));
                    if (matword & 0xC0000000u) kmax = fmaxf(kmax, 1.0f);              /* checker colour / texel: at most 1 */
                    const float amax = fmaxf(fabsf(m.ka[0]), fmaxf(fabsf(m.ka[1]), fabsf(m.ka[2])));
                    const double rv = 2.0 * (double)nl * (double)nl + 1.0;
                    const double spec = m.shininess > 0.0 ? pow(rv, m.shininess) : (m.shininess == 0.0 ? 1.0 : (double)INFINITY);   /* (a negative exponent has no bound: pow(0+, -s)) */
                    const double cmax = (double)sc.light_sum * ((double)amax + (double)kmax * ((double)nl + spec)) * 1.01;
                    if ((double)tmax * cmax < 1.8189894035458565e-12) want_diffuse = false;        /* 2^-39 */
                }
                if (!want_diffuse) culled = true;
                F3 V = mul(-1.f, r.d);
                rec.P[0] = h.pos.x; rec.P[1] = h.pos.y; rec.P[2] = h.pos.z;
                rec.N[0] = h.n.x; rec.N[1] = h.n.y; rec.N[2] = h.n.z;
                rec.V[0] = V.x; rec.V[1] = V.y; rec.V[2] = V.z;
                rec.thr[0] = thr.x; rec.thr[1] = thr.y; rec.thr[2] = thr.z;
                rec.mat = matword; rec.slot = slot; rec.pixel = pixel; rec.sample = sample;
            } else if (m.type == 2) {                              /* mirror, Scene.h:225-229 */
                F3 nd = reflect_dir(r.d, h.n);
                c_o[0] = add(h.pos, mul(0.001f, h.n));
                c_d[0] = nd;
                c_thr[0] = comp(thr, mul(m.reflectivity, f3(m.ks[0], m.ks[1], m.ks[2])));
                c_code[0] = code * 2;
                want_child = 1;
            } else {                                               /* glass, Scene.h:230-272 */
                float cosTheta = (float)fmin((double)dot(mul(-1.f, r.d), h.n), 1.);
                float sinTheta = (float)sqrt(1.0 - (double)(cosTheta * cosTheta));
                bool inside = cosTheta < 0.f;
                float ri = inside ? m.ior : (float)(1. / (double)m.ior);
                double r0 = (1 - (double)ri) / (1 + (double)ri);   /* Material.h:79-84 */
                r0 = r0 * r0;
                double R = r0 + (1 - r0) * pow((1 - (double)cosTheta), 5.0);
                F3 kd = f3(m.kd[0], m.kd[1], m.kd[2]);
                F3 ab = mul(-(h.t * m.transparency), sub(f3s(1.f), kd));   /* Scene.h:267-270 */
                ab = f3(expf(ab.x), expf(ab.y), expf(ab.z));
                float fR = (float)R, fT = (float)(1.0 - R);
                float omtr = (float)(1.0 - (double)m.transparency);
                /* color = absorb * ((1-tr)*(R*refl + (1-R)*refr) + tr*refr) */
                c_o[0] = add(h.pos, mul(0.001f, h.n));
                c_d[0] = reflect_dir(r.d, h.n);
                c_thr[0] = comp(thr, mul(omtr * fR, ab));
                c_code[0] = code * 2;
                want_child = 1;
                if (!((double)(ri * sinTheta) > 1.0)) {
                    double off = inside ? 0.00001 : -0.00001;
                    F3 perp = mul(ri, add(r.d, mul(cosTheta, h.n)));         /* Material.h:67-76 */
                    float k = (float)(1.0 - (double)sqlen(perp));
                    if (!(k < 0.0f)) {
                        F3 para = mul(-sqrtf(k), h.n);
                        F3 fd = normalized(add(para, perp));
                        if (sqlen(fd) != 0) {
                            c_o[1] = add(h.pos, mul((float)off, h.n));
                            c_d[1] = fd;
                            c_thr[1] = comp(thr, mul(omtr * fT + m.transparency, ab));
                            c_code[1] = code * 2 + 1;
                            want_child = 2;
                        }
                    }
                }
            }
        }
    }

    if (__ballot(missed) != 0ull) accumulate_runs(acc, nonfinite, ctr, slot, sky_add, missed);
    {   /* culled hits: one atomic per wave that has any */
        const unsigned long long cm = __ballot(culled);
        if (cm && __lane_id() == 0) atomicAdd(&ctr->culled, (unsigned long long)__popcll(cm));
    }
    /* The same argument one level up: a child state whose throughput is so small that nothing below it can reach half a unit
     * of the accumulators is not traced. Per bounce a throughput grows by at most G (mirror: reflectivity * ks; glass: the
     * larger of (1 - transparency) * Rmax and (1 - transparency) * max(1, Rmax - 1) + transparency, Rmax the Schlick term at
     * cos = -|N| -- the reference evaluates (1 - cos)^5 for rays inside the medium too, Material.h:79-84 -- and the absorption
     * factor is at most 1), so over the r bounces it has left, by at most cull_gpow[r], and what a hit or the sky adds per unit
     * of throughput is at most cull_cmax (host: rt_upload_scene; off for scenes with non-finite normals or unbounded
     * materials). The depth-50 recursion inside a glass mesh loses a factor (1 - transparency) * R per internal reflection:
     * its tree is cut where it stops mattering instead of growing 1.43x per level down to depth 50. */
    if (want_child && sc.cull_cmax >= 0.f) {
        const float g = sc.cull_gpow[remaining > 0 ? remaining - 1 : 0] * sc.cull_cmax;
        uint32_t keep = 0;
        for (uint32_t k = 0; k < want_child; k++) {
            const float tm = fmaxf(fabsf(c_thr[k].x), fmaxf(fabsf(c_thr[k].y), fabsf(c_thr[k].z)));
            if (!((double)tm * (double)g < 1.8189894035458565e-12)) {      /* 2^-39; NaN / inf stay */
                if (keep != k) { c_o[keep] = c_o[k]; c_d[keep] = c_d[k]; c_thr[keep] = c_thr[k]; c_code[keep] = c_code[k]; }
                keep++;
            }
        }
        want_child = keep;
    }
    /* queue appends (the ray statistics are the host's: it knows how many states of a batch are fresh primaries) */
    uint32_t cbase, dbase;
    block_reserve2(ctr, want_child, want_diffuse ? 1u : 0u, reinterpret_cast<uint32_t*>(s_stack + (size_t)sc.cstack_cap * blockDim.x), cbase, dbase);
    if (want_child && cbase + want_child <= child_cap) {
        for (uint32_t k = 0; k < want_child; k++)
            store_state(child + cbase + k, c_o[k], c_d[k], c_thr[k], slot, pixel, sample, c_code[k], remaining - 1, 0u);
    }
    if (want_diffuse && dbase < diffuse_cap) {
        float4* q = reinterpret_cast<float4*>(drec + dbase);
        q[0] = make_float4(rec.P[0], rec.P[1], rec.P[2], rec.N[0]);
        q[1] = make_float4(rec.N[1], rec.N[2], rec.V[0], rec.V[1]);
        q[2] = make_float4(rec.V[2], rec.thr[0], rec.thr[1], rec.thr[2]);
        q[3] = make_float4(__uint_as_float(rec.mat), __uint_as_float(rec.slot), __uint_as_float(rec.pixel), __uint_as_float(rec.sample));
        dcode[dbase] = sc.rng_hash ? (unsigned long long)rt_rng_prefix(sc.rng_seed, rec.pixel, rec.sample, code) : code;
        dtex[dbase] = rec_rgb;
    }
}

/* ------------------------------------------------------------------ persistent shadow kernel
 * Scene::sampleSphereLight + Scene::isInShadow (Scene.h:110-134,160-168) for every (diffuse record,
 * light, sample): ray index = (record*n_lights + light)*S + i. cnt[record*n_lights+light] receives
 * the number of occluded samples (shadowCount).
 *
 * Shadow rays end at very different times (an occluder is usually found after a few nodes, an
 * unoccluded ray walks its whole box chain), so one ray per lane leaves most lanes of a wave idle.
 * Here every wave is persistent: lanes whose ray has ended are refilled from a global work counter
 * (one wave-aggregated atomicAdd, lanes ranked with a ballot), and each lane runs a small state
 * machine -- top level (ST_TOP) or inside one mesh tree (ST_MESH) -- whose per-lane stack lives in LDS.
 * The traversal semantics are those of shadow_walk()/mesh_any() above (mode 0 / mode 1). */
#define TRI_NONE 0xFFFFFFFFu
#define ST_IDLE 0
#define ST_TOP 1
#define ST_MESH 2             /* inside a mesh tree with a step to take */
#define ST_BLOCKED 3          /* inside a mesh tree, waiting for its pending leaf to be tested (see can_walk) */
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 16      /* refill once this many lanes are idle (4 / 8 / 16 / 24 / 32: 229.7 / 219.7 / 215.3 / 223.8 / 240.3 ms on the reduced C5 frame; gorilla, cow, chamber -2..3 % at 16) */
#endif
#ifndef RT_REFILL_MIN_SUB
#define RT_REFILL_MIN_SUB 32   /* ... in the variant with sub-trees in fat leaves (short walks; 16 / 24 / 28 / 32 / 40: gorilla 40.4 / 39.5 / 39.2 / 39.2 / 39.5 ms, cow 27.0 / 26.6 / 26.2 / 26.0 / 26.1, mirror chamber 101.1 / 99.0 / 98.9 / 98.5 / 99.5; the many-object variant prefers 16) */
#endif
#ifndef RT_GRAB
#define RT_GRAB 512u           /* rays a wave takes from the global counter at a time */
#endif
#ifndef RT_TRI_MIN
#define RT_TRI_MIN 24         /* pending leaves that make a triangle round worthwhile */
#endif
#ifndef RT_EXACT_MIN
#define RT_EXACT_MIN 8        /* records that passed the plane filter and make a round of the reference's triangle test worthwhile */
#endif
#ifndef RT_FILTER_FROM_MODE
#define RT_FILTER_FROM_MODE 3  /* k_shadow variants (MODE) from this one on run the plane filter in front of the triangle test (measured with 2 / 3 / never, k_shadow ms: default scene 63.4 / 63.4 / 70.2, gorilla 43.8 / 43.6 / 43.5, cow 29.7 / 29.2 / 29.4, chamber 107.2 / 101.9 / 101.7, soup +2 % with it) */
#endif
#ifndef RT_TRI_MORE
#define RT_TRI_MORE 16        /* pending leaves that justify another triangle step in the same round */
#endif
#ifndef RT_CHILD_ORDER
#define RT_CHILD_ORDER 0      /* 4-wide walk: 0 = first hit child in slot order next, 1 = nearest hit child next (see shadow_quant_step) */
#endif
#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 4     /* register budget of k_shadow: 128 VGPRs */
#endif
#ifndef RT_NODE_STEPS
#define RT_NODE_STEPS 12      /* node-phase steps between two refill checks (measured: 8, 12, 16, 24 -> 274, 272, 276, 285 ms) */
#endif
#ifndef RT_TRI_STEPS
#define RT_TRI_STEPS 4        /* triangle-phase steps per round */
#endif
/* the same knobs for the variant with the flat top-level pass (k_shadow<5>: many objects, short walks) */
#ifndef RT_REFILL_MIN_FLAT
#define RT_REFILL_MIN_FLAT 16
#endif
#ifndef RT_TRI_MIN_FLAT
#define RT_TRI_MIN_FLAT 24
#endif
#ifndef RT_NODE_STEPS_FLAT
#define RT_NODE_STEPS_FLAT 12
#endif
#ifndef RT_EXACT_MIN_FLAT
#define RT_EXACT_MIN_FLAT 8
#endif

#define TS_MODE 1u
#define TS_OPEN 2u
#define TS_SP_ONE 4u
#define TS_SP(ts) (((ts) >> 2) & 63u)
#define TS_VISITED_ONE 0x100u
#define TS_VISITED(ts) (((ts) >> 8) & 0xFFFu)
#define TS_LIMIT(ts) ((ts) >> 20)
struct ShadowLane {
    Ray r;
    float bound;        /* lightDistance (mode 0) or just below FLT_MAX (mode 1) */
    float prune_t;      /* a child whose hits all lie beyond this slab time cannot hold one at t <= bound (inf in mode 1;
                         * quantised walk: already clamped to the reference's tMax = 100) */
    float maxinv;       /* exact walk only: max |1/d|, turns a child's hit excursion into a slab-time slack */
    float prune_raw;    /* quantised walk with leaf sub-trees: prune_t without the tMax clamp */
    uint32_t key;       /* record*n_lights + light */
    int st;
    /* the small top-level state shares one register (k_shadow has none to spare): bit 0 mode, bit 1 a leaf is open,
     * bits 2-7 top-level stack pointer, bits 8-19 leaves visited, bits 20-31 limit (mode 1: leaves to look at) */
    uint32_t ts;
    int tlink;          /* top-level cursor */
    uint32_t top_pos, top_end;
    int link, msp;      /* mesh cursor and stack pointer */
    uint32_t tri_pos;   /* next triangle record of the leaf being tested, TRI_NONE if none; quantised walk: bit 31
                         * (TRI_NEW) = the leaf's own box is still to be tested */
    QuantRay q;         /* quantised walk only, set when a mesh is entered */
};

RT_DEV void shadow_restart(const DScene& sc, ShadowLane& s, int mode, int limit) {
    s.ts = (uint32_t)mode | ((uint32_t)limit << 20);
    if (mode == 1) s.bound = 3.4028233e38f;             /* (mode 0: the refill has put lightDistance there) */
    s.tlink = sc.top_root; s.top_pos = s.top_end = 0;
    if (sc.n_flat > 0) {                                /* flat top-level pass (k_shadow<5> only: the host zeroes n_flat for the others): the ray's leaf mask, kept in entry 0 of the lane's LDS stack */
        extern __shared__ __align__(16) int s_stack[];
        s.tlink = s_stack[threadIdx.x];
    }
    s.tri_pos = TRI_NONE;                               /* (at the top level this word only ever holds an object handed over by a leaf's sub-tree) */
    s.st = ST_TOP;
}
RT_DEV void shadow_finish(ShadowLane& s, bool occ, uint32_t* cnt) {
    if (occ) atomicAdd(cnt + s.key, 1u);
    s.st = ST_IDLE;
}
/* an object of the current leaf is hit at t <= bound */
RT_DEV void shadow_on_hit(const DScene& sc, ShadowLane& s, uint32_t* cnt) {
    if (s.ts & TS_MODE) { shadow_finish(s, false, cnt); return; }    /* an earlier leaf returns a result beyond the light */
    const int visited = (int)TS_VISITED(s.ts);
    if (visited == 0) { shadow_finish(s, true, cnt); return; }
    shadow_restart(sc, s, 1, visited);
}
/* the top-level tables as k_shadow reads them: in global memory, or its LDS copy (generic pointers) */
struct TopTables { const float4* tnodes; const uint2* tleaves; const uint32_t* titems; const float4* squares; const float4* spheres; const DMesh* meshes; const TopFrame* tf; const int* tsub; const float4* flat; };
#define RT_TOP_LDS_MAX 2048
struct WideStack { int* lds; int* ovf; int lds_cap; const TopFrame* tf; int ovf_cap; uint32_t* err; };
RT_DEV void wpush(const WideStack& w, int& sp, int link);
/* one object of a top-level leaf: a square or sphere is tested on the spot, a mesh is entered (st = ST_MESH on return).
 * Returns true if the caller has to return (the ray ended, restarted in mode 1, or sits in a mesh now). */
template <bool QUANT, bool IN_SUB>     /* IN_SUB: reached through the sub-tree of its top-level leaf (the lane's s.q is the scene's frame then) */
RT_DEV bool shadow_top_item(const DScene& sc, const TopTables& tt, ShadowLane& s, int ord, const WideStack& w, uint32_t* cnt) {
    /* The caller's loop runs over the objects of a leaf with the ray fixed, and the compiler would hoist every product of
     * ray components out of it (a = d.d, 4a, the plane times of the mesh frame test ...) and keep them in registers across
     * the whole kernel loop, which this kernel cannot afford (k_shadow sits at its register limit: they end up spilled).
     * Values that pass through an empty asm are opaque per call: nothing is hoisted, nothing is computed twice. */
    asm volatile("" : "+v"(s.r.d.x), "+v"(s.r.d.y), "+v"(s.r.d.z), "+v"(s.r.inv.x), "+v"(s.r.inv.y), "+v"(s.r.inv.z));
    if (ord < sc.n_squares) {
        const float4* q = tt.squares + 4 * ord;
        float t, u, v;
        if (square_test(s.r, q[0], q[1], q[2], q[3], t, u, v) && t <= s.bound) { shadow_on_hit(sc, s, cnt); return true; }
        return false;
    }
    if (ord < sc.n_squares + sc.n_spheres) {
        const float4* q = tt.spheres + 2 * (ord - sc.n_squares);
        float t; F3 n;
        if (sphere_test(s.r, q[0], t, n) && t <= s.bound) { shadow_on_hit(sc, s, cnt); return true; }
        return false;
    }
    DMesh m = tt.meshes[ord - sc.n_squares - sc.n_spheres];
    /* beyond(): (tEnter - slack)*0.99999 > bound; the slightly weaker tEnter - slack > bound*1.00002 is
     * still conservative and costs one compare per child */
    s.prune_t = (s.ts & TS_MODE) == 0u ? s.bound * 1.00002f : INFINITY;
    s.prune_raw = s.prune_t;
    if (QUANT) s.prune_t = fminf(100.0f, s.prune_t);     /* tMax of AABB::intersect */
    else s.maxinv = max_abs_inv(s.r);
    if (QUANT) {
        quant_setup(s.q, s.r, m);
        /* The reference tests no box for the mesh as a whole (Mesh.cpp:97-123 starts at the root's
         * children), but every box of the shadow tree lies inside the mesh's plane frame: a ray that
         * misses the whole frame (or reaches it beyond the distance of interest) cannot reach a leaf,
         * and stays at the top level -- scenes with dozens of meshes are mostly such visits. */
        if (m.wroot >= 0) {
            bool hit;
            quant_entry(0xFFFF0000u, 0xFFFF0000u, 0xFFFF0000u, s.q, (m.wroot & WIDE_UNPRUNED) ? 100.0f : s.prune_t, hit);
            if (!hit) {
                if (IN_SUB) quant_setup(s.q, s.r, tt.tf->frame);      /* (rare: the object's box in the sub-tree is this frame, rounded the other way) */
                return false;
            }
        }
    }
    if (IN_SUB) wpush(w, s.msp, LINK_SENTINEL);      /* what lies below on the stack belongs to the walk of the leaf's sub-tree */
    else s.msp = sc.top_cap;
    s.link = QUANT ? m.wroot : m.root; s.tri_pos = TRI_NONE;
    s.st = ST_MESH;
    return true;
}
/* top-level state machine: runs until the ray ends or enters a mesh tree (or, TOPSUB, the sub-tree of a top-level leaf) */
/* Flat top-level pass (k_shadow<5>). The reference walks its top-level tree left first and enters every child whose box passes
 * AABB::intersect (AccelerationStruct.cpp:315-336); its boxes are nested, so a leaf is reached iff its OWN box passes (monotone slab
 * arithmetic, the argument of the quantised mesh trees), and the leaves are met in the tree's left-to-right order. For a tree of at
 * most 32 leaves the walk is therefore a bit mask: flat_reach() runs the reference's slab test on every leaf's own box when the ray
 * is taken on -- all refilled lanes together, outside the divergent loop -- and the top-level state machine then only takes the next
 * set bit: no node steps, no stack. Mode 1 ("does an earlier leaf hold any hit at all") walks the same mask again up to its limit. */
RT_DEV TopTables top_tables_global_flat(const DScene& sc) {      /* (only .flat is read through it) */
    TopTables tt{};
    tt.flat = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sc.top_blob) + sc.top_off[7]);
    return tt;
}
RT_DEV uint32_t flat_reach(const TopTables& tt, int n_flat, const Ray& r) {
    uint32_t mask = 0u;
#pragma unroll 1
    for (int i = 0; i < n_flat; i++) {
        const float4 a = tt.flat[2 * i], b = tt.flat[2 * i + 1];
        float tEnter;
        if (slab(r, a.x, a.y, a.z, a.w, b.x, b.y, tEnter)) mask |= 1u << i;
    }
    return mask;
}
template <bool QUANT>
RT_DEV void shadow_advance_top_flat(const DScene& sc, const TopTables& tt, ShadowLane& s, uint32_t* cnt, const WideStack& w) {
    if (s.tri_pos != TRI_NONE) {
        /* the walk of a leaf's sub-tree has reached an object (shadow_quant_step); afterwards that walk goes on */
        const int ord = (int)(s.tri_pos & ~TOP_ITEM);
        s.tri_pos = TRI_NONE;
        if (shadow_top_item<QUANT, true>(sc, tt, s, ord, w, cnt)) return;
        s.link = LINK_NONE;
        s.st = ST_MESH;
        return;
    }
    for (;;) {
        if (s.top_pos < s.top_end) {
            int ord = (int)tt.titems[s.top_pos++];
            if (shadow_top_item<QUANT, false>(sc, tt, s, ord, w, cnt)) return;
            continue;
        }
        if (s.ts & TS_OPEN) s.ts = (s.ts & ~TS_OPEN) + TS_VISITED_ONE;
        const uint32_t mask = (uint32_t)s.tlink;
        if (mask == 0u) { shadow_finish(s, (s.ts & TS_MODE) != 0u, cnt); return; }     /* mode 0: no occluder; mode 1: nothing hides it */
        if ((s.ts & TS_MODE) && TS_VISITED(s.ts) >= TS_LIMIT(s.ts)) { shadow_finish(s, true, cnt); return; }
        const int bit = __builtin_ctz(mask);
        s.tlink = (int)(mask & (mask - 1u));
        const int leaf = __float_as_int(tt.flat[2 * bit + 1].z);
        s.ts |= TS_OPEN;
        const int sub = tt.tsub[leaf];
        const TopFrame* tf = tt.tf;
        const bool inside = s.r.o.x >= tf->omin[0] && s.r.o.x <= tf->omax[0] && s.r.o.y >= tf->omin[1] && s.r.o.y <= tf->omax[1] &&
                            s.r.o.z >= tf->omin[2] && s.r.o.z <= tf->omax[2];
        if (sub != LINK_NONE && inside) {
            s.top_pos = s.top_end = 0;
            s.prune_raw = (s.ts & TS_MODE) == 0u ? s.bound * 1.00002f : INFINITY;
            quant_setup(s.q, s.r, tf->frame);
            s.link = sub; s.msp = sc.top_cap; s.tri_pos = TRI_NONE;
            s.st = ST_MESH;
            return;
        }
        uint2 lf = tt.tleaves[leaf];
        s.top_pos = lf.x; s.top_end = lf.x + lf.y;
    }
}
template <bool QUANT, bool TOPSUB>
RT_DEV void shadow_advance_top(const DScene& sc, const TopTables& tt, ShadowLane& s, int* stk, int stride, uint32_t* cnt, const WideStack& w) {
    if (TOPSUB && s.tri_pos != TRI_NONE) {
        /* the walk of a leaf's sub-tree has reached an object (shadow_quant_step); afterwards that walk goes on */
        const int ord = (int)(s.tri_pos & ~TOP_ITEM);
        s.tri_pos = TRI_NONE;
        if (shadow_top_item<QUANT, true>(sc, tt, s, ord, w, cnt)) return;
        s.link = LINK_NONE;
        s.st = ST_MESH;
        return;
    }
    for (;;) {
        if (s.top_pos < s.top_end) {
            int ord = (int)tt.titems[s.top_pos++];
            if (shadow_top_item<QUANT, false>(sc, tt, s, ord, w, cnt)) return;
            continue;
        }
        if (s.ts & TS_OPEN) s.ts = (s.ts & ~TS_OPEN) + TS_VISITED_ONE;
        if (s.tlink == LINK_NONE) {
            if (TS_SP(s.ts) == 0u) { shadow_finish(s, (s.ts & TS_MODE) != 0u, cnt); return; }   /* mode 0: no occluder; mode 1: nothing hides it */
            s.ts -= TS_SP_ONE;
            s.tlink = stk[lds_index((int)TS_SP(s.ts), stride)];
        }
        if (s.tlink >= 0) {
            const float4* nd = tt.tnodes + 4 * (size_t)s.tlink;
            float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
            float tL, tR;
            bool hL = slab(s.r, a.x, a.y, a.z, a.w, b.x, b.y, tL);
            bool hR = slab(s.r, b.z, b.w, c.x, c.y, c.z, c.w, tR);
            int left = __float_as_int(d.x), right = __float_as_int(d.y);
            if (hL && hR) { stk[lds_index((int)TS_SP(s.ts), stride)] = right; s.ts += TS_SP_ONE; s.tlink = left; }
            else if (hL) s.tlink = left;
            else if (hR) s.tlink = right;
            else s.tlink = LINK_NONE;
        } else {
            if ((s.ts & TS_MODE) && TS_VISITED(s.ts) >= TS_LIMIT(s.ts)) { shadow_finish(s, true, cnt); return; }
            const int leaf = -1 - s.tlink;
            s.tlink = LINK_NONE;
            s.ts |= TS_OPEN;
            if (TOPSUB) {
                const int sub = tt.tsub[leaf];
                /* the object boxes of the sub-tree assume an origin within the scene's neighbourhood (rtgpu.hip); NaN compares false */
                const TopFrame* tf = tt.tf;
                const bool inside = s.r.o.x >= tf->omin[0] && s.r.o.x <= tf->omax[0] && s.r.o.y >= tf->omin[1] && s.r.o.y <= tf->omax[1] &&
                                    s.r.o.z >= tf->omin[2] && s.r.o.z <= tf->omax[2];
                if (sub != LINK_NONE && inside) {
                    s.top_pos = s.top_end = 0;
                    s.prune_raw = (s.ts & TS_MODE) == 0u ? s.bound * 1.00002f : INFINITY;
                    quant_setup(s.q, s.r, tf->frame);
                    s.link = sub; s.msp = sc.top_cap; s.tri_pos = TRI_NONE;
                    s.st = ST_MESH;
                    return;
                }
            }
            uint2 lf = tt.tleaves[leaf];
            s.top_pos = lf.x; s.top_end = lf.x + lf.y;
        }
    }
}
/* A lane inside a mesh can take a node step unless a pending leaf blocks it: with a leaf pending it may still
 * visit nodes and pop, but not take on a second leaf (the cursor then keeps that leaf's link) and not leave the
 * mesh. Without a pending leaf every cursor value has a step to take (a leaf link -- e.g. a mesh that is one
 * leaf -- becomes pending, an exhausted walk returns to the top level). */
RT_DEV bool can_walk(const ShadowLane& s, int top_cap) {   /* decides between ST_MESH and ST_BLOCKED after every step */
    if (s.tri_pos == TRI_NONE) return true;
    return s.link >= 0 || (s.link == LINK_NONE && s.msp > top_cap);
}
/* one traversal unit inside a mesh tree for a lane with no pending triangle: pop if the cursor is
 * empty, then either fetch a leaf's triangle range or process one inner node (two slab tests) and
 * move to the next link straight away */
RT_DEV void shadow_node_step(const DScene& sc, ShadowLane& s, int* stk, int stride) {
    int link = s.link;
    if (link == LINK_NONE) {
        if (s.msp == sc.top_cap) { s.st = ST_TOP; return; }        /* mesh done without a hit: next object of the leaf */
        link = stk[lds_index(--s.msp, stride)];
    }
    if (link >= 0) {
        const float4* nd = sc.mnodes + 4 * (size_t)link;
        float4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
        float tL, tR;
        bool hL = slab(s.r, a.x, a.y, a.z, a.w, b.x, b.y, tL);
        bool hR = slab(s.r, b.z, b.w, c.x, c.y, c.z, c.w, tR);
        int left = __float_as_int(d.x), right = __float_as_int(d.y);
        bool nearLeft = tL <= tR;
        hL = hL && !(tL - d.z * s.maxinv > s.prune_t);      /* d.z, d.w: how far hits below the child leave its box */
        hR = hR && !(tR - d.w * s.maxinv > s.prune_t);
        if (hL && hR) {
            stk[lds_index(s.msp++, stride)] = nearLeft ? right : left;
            link = nearLeft ? left : right;
        } else if (hL) link = left;
        else if (hR) link = right;
        else if (s.msp > sc.top_cap) link = stk[lds_index(--s.msp, stride)];
        else link = LINK_NONE;
    }
    /* a leaf: its triangles become pending; with one leaf pending already the lane keeps the link and is blocked */
    if (link < 0 && link != LINK_NONE && s.tri_pos == TRI_NONE) {
        s.tri_pos = leaf_first(link);
        link = LINK_NONE;
    }
    s.link = link;
    if (!can_walk(s, sc.top_cap)) s.st = ST_BLOCKED;
}
RT_DEV void shadow_tri_step(const DScene& sc, ShadowLane& s, uint32_t* cnt) {
    const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)s.tri_pos;
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    s.tri_pos = (__float_as_uint(q2.w) & TRI_LEAF_END) != 0u ? TRI_NONE : s.tri_pos + 1;
    if (s.tri_pos == TRI_NONE && s.link < 0 && s.link != LINK_NONE) { s.tri_pos = leaf_first(s.link); s.link = LINK_NONE; }   /* the blocked leaf is next */
    s.st = can_walk(s, sc.top_cap) ? ST_MESH : ST_BLOCKED;
    float t, w0, w1, w2;
    if (tri_test(s.r, q0, q1, q2, q3, t, w0, w1, w2) && t <= s.bound) shadow_on_hit(sc, s, cnt);
}


/* ------------------------------------------------------------------ quantised shadow nodes
 * A shadow query only asks whether SOME triangle the reference reaches is hit at t <= bound (see
 * shadow_on_hit), so neither the visiting order nor the route to a leaf matters -- only which leaves
 * the reference reaches. It reaches a leaf iff AABB::intersect passes for every box on the way down
 * (AccelerationStruct.cpp:255-283). Those boxes are nested (each bounds a subset of its parent's
 * triangles; checked at upload) and the slab arithmetic is monotone in the box planes -- a smaller
 * box gives entry times that are no smaller and exit times that are no larger, and a NaN product only
 * ever drops a constraint -- hence:  the leaf's own box passes  =>  every box above it passes.
 * So the reference reaches exactly the leaves whose OWN box passes its test, and the boxes above the
 * leaves may be replaced by anything that never loses such a leaf.
 *
 * The binary walk over the reference's 64-byte nodes saturates two things at once -- the vector L1 (one
 * 16-byte lane access per clock per CU when the lanes sit at unrelated nodes) and VALU issue (DESIGN.md
 * section 5) -- so the shadow tree cuts both: DScene::wnodes is the reference tree with every other level
 * folded away (half the steps) and the child boxes rounded OUTWARD to 16-bit planes in the mesh's frame
 * (a quarter of the bytes; quantise_mesh and WideFold in rtgpu.hip), x|y|z of a box as lo | hi << 16.
 * The test is conservative: one fma per plane, t = q*(qstep*inv) + ((qmin - o)*inv -/+ slack),
 * slack = 2e-6 * max(|o|,|mesh|) * |1/d| of that axis -- more than the rounding differences between
 * this form on the padded planes and the reference's (plane - o)*inv on the true ones (below
 * 1e-6 * M * |inv| in total). v_max3/v_min3 drop NaN operands, which only loosens the test.
 * A leaf is then entered only if the reference's own test (slab()) passes on the leaf's reference
 * box (slots 4-5 of the leaf's first triangle record), and its triangles are tested exactly as
 * before: bit-identical occlusion. */
#define TRI_NEW 0x80000000u
/* The 4-wide walk can have three links pending per level. Only the first lds_cap entries of a lane's stack
 * live in LDS (what keeps 16 waves on a CU); deeper ones -- rare -- go to a per-lane slice of global memory,
 * entry k of lane l of block b at ovf[(b*ovf_cap + k)*64 + l]. */
/* The upload sizes the slice for the deepest stack the walk can build (wstack_cap, rtgpu.hip); should a walk ever want more, the entry is
 * dropped and the launch marked (Counters::walk_overflow -> RT_ERR_LIMIT) instead of writing into the next block's slice. */
RT_DEV void wpush(const WideStack& w, int& sp, int link) {
    if (sp < w.lds_cap) w.lds[sp * RT_BLOCK] = link;
    else if (sp - w.lds_cap < w.ovf_cap) w.ovf[(sp - w.lds_cap) * RT_BLOCK] = link;
    else *w.err = 1u;
    sp++;
}
RT_DEV int wpop(const WideStack& w, int& sp) {
    --sp;
    if (sp < w.lds_cap) return w.lds[sp * RT_BLOCK];
    return sp - w.lds_cap < w.ovf_cap ? w.ovf[(sp - w.lds_cap) * RT_BLOCK] : LINK_NONE;
}
/* one 4-wide node (64 bytes: twelve packed plane pairs, four links): the nearest hit child next, the others pushed.
 * Stack traffic stays in LDS unless this step could cross lds_cap (then, and only then, the spilling push/pop run). */
template <bool SUB, bool TOPSUB>
RT_DEV void shadow_quant_step(const DScene& sc, ShadowLane& s, const WideStack& w) {
    int link = s.link;
    if (link == LINK_NONE) {                                         /* the one place where the walk pops */
        if (s.msp == sc.top_cap) { s.st = ST_TOP; return; }        /* mesh done without a hit: next object of the leaf */
        --s.msp;
        /* the LDS read is unconditional (of a clamped slot) and the spilled entry a separate, rare load: as the two arms of one
         * branch the compiler merges them into a select of the two addresses and a flat load, which costs a 64-bit address, both
         * memory pipes and both wait counters on every pop */
        link = w.lds[(s.msp < w.lds_cap ? s.msp : w.lds_cap - 1) * RT_BLOCK];
        if (s.msp >= w.lds_cap) link = s.msp - w.lds_cap < w.ovf_cap ? w.ovf[(s.msp - w.lds_cap) * RT_BLOCK] : LINK_NONE;     /* (a dropped entry, see wpush) */
    }
    if (TOPSUB && link == LINK_SENTINEL) {
        /* the mesh entered from the sub-tree of a top-level leaf is done: back to the scene's frame -- once the mesh's
         * last pending leaf has been tested (a fat leaf that passes its box pushes its own sub-tree, which belongs above
         * the sentinel: until then the sentinel stays on the stack and the lane is blocked) */
        if (s.tri_pos != TRI_NONE) { ++s.msp; s.link = LINK_NONE; s.st = ST_BLOCKED; return; }
        quant_setup(s.q, s.r, w.tf->frame);
        s.link = LINK_NONE;
        return;
    }
    if (link >= 0) {
        /* the child boxes of this tree already hold every hit accepted below them (quantise_mesh), so a child entered
         * beyond prune_t cannot matter; a node with a child that has no such bound is marked and only uses tMax */
        float lim = (link & WIDE_UNPRUNED) ? 100.0f : s.prune_t;
        /* below a fat leaf's own box the reference tests triangles, not boxes: no tMax there (a triangle of a leaf
         * entered before t = 100 may be hit beyond it) */
        if (SUB && (link & WIDE_SUB)) lim = (link & WIDE_UNPRUNED) ? INFINITY : s.prune_raw;
        const uint4* nd = sc.wnodes + 4 * (size_t)(link & WIDE_INDEX);
        uint4 a = nd[0], b = nd[1], c = nd[2], d = nd[3];
        const float lo = (TOPSUB && (link & WIDE_LINE)) ? -INFINITY : 0.0f;
        bool h0, h1, h2, h3;
        float t0 = quant_entry(a.x, a.y, a.z, s.q, lim, h0, lo);
        float t1 = quant_entry(a.w, b.x, b.y, s.q, lim, h1, lo);
        float t2 = quant_entry(b.z, b.w, c.x, s.q, lim, h2, lo);
        float t3 = quant_entry(c.y, c.z, c.w, s.q, lim, h3, lo);
        const int l0 = (int)d.x, l1 = (int)d.y, l2 = (int)d.z, l3 = (int)d.w;
        /* nearest hit child first (an occluder is usually found early that way); the order of the rest is free */
#if RT_CHILD_ORDER == 0
        int la = h0 ? l0 : h1 ? l1 : h2 ? l2 : l3, ia = h0 ? 0 : h1 ? 1 : h2 ? 2 : 3;
        (void)t0; (void)t1; (void)t2; (void)t3;
#else
        t0 = h0 ? t0 : INFINITY; t1 = h1 ? t1 : INFINITY; t2 = h2 ? t2 : INFINITY; t3 = h3 ? t3 : INFINITY;
        float ta = t0; int la = l0, ia = 0;
        if (t1 < ta) { ta = t1; la = l1; ia = 1; }
        if (t2 < ta) { ta = t2; la = l2; ia = 2; }
        if (t3 < ta) { ta = t3; la = l3; ia = 3; }
#endif
        const bool p0 = h0 && ia != 0, p1 = h1 && ia != 1, p2 = h2 && ia != 2, p3 = h3 && ia != 3;
        if (h0 || h1 || h2 || h3) {
            if (s.msp + 3 <= w.lds_cap) {                             /* the common case: no entry of this step can leave LDS */
                int* top = w.lds + s.msp * RT_BLOCK;
                if (p0) { *top = l0; top += RT_BLOCK; }
                if (p1) { *top = l1; top += RT_BLOCK; }
                if (p2) { *top = l2; top += RT_BLOCK; }
                if (p3) { *top = l3; }
                s.msp += (int)p0 + (int)p1 + (int)p2 + (int)p3;
            } else {
                if (p0) wpush(w, s.msp, l0);
                if (p1) wpush(w, s.msp, l1);
                if (p2) wpush(w, s.msp, l2);
                if (p3) wpush(w, s.msp, l3);
            }
            link = la;
        } else link = LINK_NONE;                                      /* nothing hit: the next step pops */
    }
    if (TOPSUB && link < 0 && link != LINK_NONE && (leaf_first(link) & TOP_ITEM)) {
        /* an object of a top-level leaf (the walk is in the leaf's sub-tree; no triangle leaf can be pending there):
         * the top-level step tests it, then this walk goes on */
        s.tri_pos = leaf_first(link);
        s.link = LINK_NONE;
        s.st = ST_TOP;
        return;
    }
    if (link < 0 && link != LINK_NONE && s.tri_pos == TRI_NONE) {
        s.tri_pos = leaf_first(link) | TRI_NEW;
        link = LINK_NONE;
    }
    s.link = link;
    if (!can_walk(s, sc.top_cap)) s.st = ST_BLOCKED;
}
/* A filter in front of the reference's triangle test. Triangle::getIntersection costs about 300 instructions (three
 * cross products, three correctly rounded square roots and divisions) and nine in ten of the triangles a shadow ray
 * meets are misses, but the lanes of a wave miss at different times, so an early exit inside the test saves nothing.
 * The filter computes the plane point P exactly as the test does (same operations, same bits) and looks at two affine
 * coordinates of it, u = g1.(P - c0), v = g2.(P - c0) (g1, g2: the dual basis of the triangle's edges, 16-bit floats in
 * slot 7 of the record): a point the reference accepts lies within the record's hit excursion e of the triangle
 * (rtgpu.hip, tri_excursion), hence has u, v >= -m and u + v <= 1 + m with the margin m the upload derives from e, the
 * rounding of g1, g2 to 8 bits and of this evaluation (m = inf where there is no bound: nothing is rejected). NaNs
 * compare false and pass (the reference lets a NaN sum pass, Triangle.h:105). What passes is tested by the reference's
 * arithmetic in a round of its own (TRI_EXACT), with the lanes that also have a candidate. */
RT_DEV bool tri_maybe(const Ray& r, float4 q0, float4 q3, uint4 q7) {
    const F3 c0 = f3(q0.x, q0.y, q0.z), n = f3(q3.x, q3.y, q3.z);
    const float dn = dot(r.d, n);
    if (dn == 0) return false;                                   /* Triangle.h:49-53, as in tri_test */
    const float t = (q0.w - dot(r.o, n)) / dn;                   /* Triangle.h:59 */
    const F3 v = sub(add(r.o, mul(t, r.d)), c0);                 /* Triangle.h:61 */
    const F3 g1 = f3(__uint_as_float(q7.x << 16), __uint_as_float(q7.x & 0xFFFF0000u), __uint_as_float(q7.y << 16));
    const F3 g2 = f3(__uint_as_float(q7.y & 0xFFFF0000u), __uint_as_float(q7.z << 16), __uint_as_float(q7.z & 0xFFFF0000u));
    const float m = __uint_as_float(q7.w << 16);
    const float a = dot(v, g1), b = dot(v, g2);
    return !(a < -m || b < -m || a + b > 1.0f + m);
}
/* one triangle of a leaf of the quantised walk; the first one brings the reference's test of the leaf's own box along.
 * SUB: a fat leaf that passes hands over to its sub-tree (pushed; words 24-25 of the record), whose runs are then
 * reached like leaves but without a box of their own to test. */
template <bool SUB, bool FILTER>
RT_DEV void shadow_quant_tri_step(const DScene& sc, ShadowLane& s, uint32_t* cnt, const WideStack& w) {
    const uint32_t pos = s.tri_pos & (SUB ? ~(TRI_NEW | TRI_SUB) : ~TRI_NEW);
    const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)pos;
    float4 q0 = q[0], q1, q2, q3 = q[3];
    uint4 q7 = make_uint4(0u, 0u, 0u, 0u);
    if (FILTER) q7 = *reinterpret_cast<const uint4*>(q + 7);     /* the filter's planes, its margin and the record's flags */
    else { q1 = q[1]; q2 = q[2]; }
    bool reached = true;
    if ((s.tri_pos & TRI_NEW) && !(SUB && (s.tri_pos & TRI_SUB))) {
        float4 b0 = q[4], b1 = q[5];                               /* the leaf's reference box rides in the same line */
        float tEnter;
        reached = slab(s.r, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, tEnter);
        if (SUB && reached) {
            float4 gate = q[6];
            if (__float_as_uint(gate.y) != 0u) { wpush(w, s.msp, __float_as_int(gate.x)); reached = false; }
        }
    }
    if (FILTER && reached && tri_maybe(s.r, q0, q3, q7)) { s.tri_pos = pos | TRI_EXACT; return; }    /* a candidate: stays pending (the lane may go on walking, see can_walk) */
    s.tri_pos = (!reached || (FILTER ? (q7.w >> 16) : __float_as_uint(q2.w)) != 0u) ? TRI_NONE : pos + 1;
    if (s.tri_pos == TRI_NONE && s.link < 0 && s.link != LINK_NONE) { s.tri_pos = leaf_first(s.link) | TRI_NEW; s.link = LINK_NONE; }
    s.st = can_walk(s, sc.top_cap) ? ST_MESH : ST_BLOCKED;
    if (!FILTER) {
        float t, w0, w1, w2;
        if (reached && tri_test(s.r, q0, q1, q2, q3, t, w0, w1, w2) && t <= s.bound) shadow_on_hit(sc, s, cnt);
    }
}
/* the reference's test on a record that passed the filter */
RT_DEV void shadow_quant_exact_step(const DScene& sc, ShadowLane& s, uint32_t* cnt) {
    const uint32_t pos = s.tri_pos & ~TRI_EXACT;
    const float4* q = sc.tri + RT_TRI_STRIDE * (size_t)pos;
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    s.tri_pos = __float_as_uint(q2.w) != 0u ? TRI_NONE : pos + 1;
    if (s.tri_pos == TRI_NONE && s.link < 0 && s.link != LINK_NONE) { s.tri_pos = leaf_first(s.link) | TRI_NEW; s.link = LINK_NONE; }
    s.st = can_walk(s, sc.top_cap) ? ST_MESH : ST_BLOCKED;
    float t, w0, w1, w2;
    if (tri_test(s.r, q0, q1, q2, q3, t, w0, w1, w2) && t <= s.bound) shadow_on_hit(sc, s, cnt);
}

/* Scene::sampleSphereLight's sample position + Scene::isInShadow's ray (Scene.h:115-126,161-164) for
 * every (record, light, sample), at full lane utilisation, 32 bytes per ray: (origin, lightDistance)
 * (direction, -). The persistent kernel's refill then is two coalesced loads. */
/* the shadow ray of (record, light, sample) number idx */
RT_DEV void shadow_ray_of(const DScene& sc, const FrameD& fr, const DiffuseRec* __restrict__ drec, const unsigned long long* __restrict__ dcode,
                          unsigned long long idx, int S, uint32_t& key_out, F3& o_out, F3& d_out, float& dist_out) {
    uint32_t key, i;
    if ((idx >> 32) == 0ull) {               /* (the rays of one launch almost always fit 32 bits: a quarter of the 64-bit division's instructions) */
        key = (uint32_t)idx / (uint32_t)S;
        i = (uint32_t)idx - key * (uint32_t)S;
    } else {
        key = (uint32_t)(idx / (unsigned)S);
        i = (uint32_t)(idx % (unsigned)S);
    }
    if (sc.group_keys) key = sc.group_keys[key];      /* (the launch holds the pairs k_shadow_group left, in its order) */
    uint32_t rec = key, li = 0u;
    if (sc.n_lights != 1) { rec = key / (uint32_t)sc.n_lights; li = key - rec * (uint32_t)sc.n_lights; }
    const float4* q = reinterpret_cast<const float4*>(drec + rec);
    float4 q0 = q[0];
    F3 P = f3(q0.x, q0.y, q0.z);
    DLight L = sc.lights[li];
    F3 lp = f3(L.pos[0], L.pos[1], L.pos[2]);
    if (fr.shadow_mode == 0) {
        /* rt_rng(seed, pixel, sample, code, light, i, which): the rounds over the first four arguments were run once per
         * record (k_hash_records left them in the low word of the record's code slot), the last three differ per ray */
        const uint32_t h = (uint32_t)dcode[rec];
        uint32_t ky = rt_rng_tail(h, li, i, 0) % 360u;   /* randY first, Scene.h:115 */
        uint32_t kx = rt_rng_tail(h, li, i, 1) % 360u;
        double sinY = sc.sincos[ky], cosY = sc.sincos[360 + ky];
        double sinX = sc.sincos[kx], cosX = sc.sincos[360 + kx];
        float p1 = (float)(-sinY), p2 = (float)cosY;                                 /* Scene.h:119 */
        float p0 = (float)(-sinX * (double)p2);                                      /* Scene.h:121 */
        p2 = (float)(cosX * (double)p2);                                             /* Scene.h:122 */
        lp = add(lp, mul(L.radius, f3(p0, p1, p2)));                                 /* Scene.h:124-126 */
    }
    F3 dir = sub(lp, P);                                                             /* Scene.h:161-164 */
    float lightDistance = len(dir);
    dir = normalized(dir);
    F3 o = add(P, mul(0.001f, dir));
    F3 d = normalized(dir);                                                          /* Line.h:15 */
    key_out = key; o_out = o; d_out = d; dist_out = lightDistance;
}
/* per diffuse record: its path code (8 bytes, written by k_closest) gives way to the state of the counter RNG after the
 * rounds that the record's 35 x lights shadow rays share; runs once per batch, before the kernels that make shadow rays */
__global__ void __launch_bounds__(256) k_hash_records(uint32_t seed, const DiffuseRec* __restrict__ drec, unsigned long long* dcode, uint32_t n_rec) {
    const uint32_t rec = blockIdx.x * blockDim.x + threadIdx.x;
    if (rec >= n_rec) return;
    const float4 q3 = reinterpret_cast<const float4*>(drec + rec)[3];
    dcode[rec] = (unsigned long long)rt_rng_prefix(seed, __float_as_uint(q3.z), __float_as_uint(q3.w), dcode[rec]);
}
/* ---- group culling of shadow rays. The 35 shadow rays of a (diffuse record, light) pair all start within 0.001 of the hit
 * point P and end within 0.001 of the light's sphere (centre C, radius R): every point o + t d, 0 <= t <= lightDistance, of
 * every one of them lies in the swept cube {P + s (C - P) +- h(s)}, s in [0, 1], h(s) = h0 + s (h1 - h0), h0 = 0.002 + slack,
 * h1 = R (1 + 1e-5) + 0.002 + slack. Every hit an object's test accepts lies in the object's box (DScene::obox: the boxes of
 * the sub-trees over top-level objects, rtgpu.hip), so a box the swept cube stays clear of cannot yield a hit at
 * t <= lightDistance on any of the 35 rays; a square is also out when all 35 rays travel along its normal (Square.h:74 returns
 * before anything else); a sphere's tangent root may be negative (Sphere.h:113), so its box must be clear of the rays' lines
 * behind P as well. When every object of the scene is out, Scene::isInShadow is false for all 35 rays -- no object returns a
 * result at t <= lightDistance, so no leaf of BVH_Node::intersectObjects does, whatever their order (AccelerationStruct.cpp:
 * 293-311,326-330) -- and the pair's shadow count is the 0 its counter was cleared to: its rays are not made. Everything is
 * decided by comparisons that are false on NaN; a pair is dropped only on proof. */
RT_DEV bool hull_clear_of_box(F3 A, F3 D, float h0, float dh, float smin, float smax, float hs, float4 b0, float4 b1) {
    const float mn[3] = {b0.x, b0.y, b0.z}, mx[3] = {b0.w, b1.x, b1.y};
    const float a3[3] = {A.x, A.y, A.z}, d3[3] = {D.x, D.y, D.z};
    float lo = smin, hi = smax;
    bool empty = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        /* the cube's upper face reaches the box's lower one: s (D_k + hs dh) >= mn_k - A_k - h0 */
        {
            const float a = d3[k] + hs * dh, b = mn[k] - a3[k] - h0;
            const float q = b / a, m = 1e-5f * (1.0f + fabsf(q));
            if (a > 0.f) lo = fmaxf(lo, q - m);
            else if (a < 0.f) hi = fminf(hi, q + m);
            else if (a == 0.f && b > 0.f) empty = true;
        }
        /* its lower face stays below the box's upper one: s (D_k - hs dh) <= mx_k - A_k + h0 */
        {
            const float a = d3[k] - hs * dh, b = mx[k] - a3[k] + h0;
            const float q = b / a, m = 1e-5f * (1.0f + fabsf(q));
            if (a > 0.f) hi = fminf(hi, q + m);
            else if (a < 0.f) lo = fmaxf(lo, q - m);
            else if (a == 0.f && b < 0.f) empty = true;
        }
    }
    return empty || lo > hi;
}
/* the decision for one pair: hit point P, light centre C, radius R (0 for hard shadows). why: -1 clear, 0 kept for its origin or length,
 * 1 + kind of the object that keeps it */
RT_DEV bool group_pair_clear(const DScene& sc, const F3 P, const F3 C, const float R, int& why_out) {
    const F3 D = sub(C, P);
    const float dist = len(D);
    const float big = fmaxf(fmaxf(fabsf(P.x), fabsf(P.y)), fmaxf(fabsf(P.z), fmaxf(fmaxf(fabsf(C.x), fabsf(C.y)), fabsf(C.z)))) + R;
    const float slack = 64.f * 5.9604645e-8f * (big + dist);              /* the float arithmetic of the sample position, the origin and the hit point */
    const float h0 = 0.002f + slack, h1 = R * 1.00001f + 0.002f + slack, dh = h1 - h0;
    /* origins the object boxes' paddings were derived for (the light may lie outside: what the paddings bound is the rounding of a hit
     * point, which grows with its distance from the origin, and a hit lies in an object's box, within the scene), a ray with a length */
    bool clear = P.x >= sc.gomin[0] && P.x <= sc.gomax[0] && P.y >= sc.gomin[1] && P.y <= sc.gomax[1] && P.z >= sc.gomin[2] && P.z <= sc.gomax[2] &&
                 dist > 4.f * h1 && dist < 1e15f && big < 1e15f;
    /* how far back the rays' lines can matter (spheres): the admitted box's diagonal, in units of |C - P| */
    const float back = clear ? (fabsf(sc.gomax[0] - sc.gomin[0]) + fabsf(sc.gomax[1] - sc.gomin[1]) + fabsf(sc.gomax[2] - sc.gomin[2])) / dist : 0.f;
    int why = clear ? -1 : 0;
    for (int o = 0; o < sc.n_obox && clear; o++) {
        float4 b0 = sc.obox[2 * o], b1 = sc.obox[2 * o + 1];
        const int kind = __float_as_int(b1.z);
        if (kind == 1) {
            /* a sphere's uploaded box is padded for the farthest admitted origin (the rounding of a root grows with the square of the
             * origin's distance: | |hit - centre|^2 - r^2 | <= 96 u (|o - centre| + r)^2, rtgpu.hip) -- a scene with a large floor makes
             * that metres. This pair's origins are within 0.002 of P: the same bound with their distance */
            const float4 sp = sc.spheres[2 * (o - sc.n_squares)];
            const float r = sp.w, oc = len(sub(P, f3(sp.x, sp.y, sp.z))) + 0.002f;
            const float pad = 96.f * 5.9604645e-8f * (oc + r) * (oc + r) / r + 1e-5f * r + slack + 1e-6f;
            if (r > 0.f && pad < 1e15f && b0.x <= b0.w) {
                b0 = make_float4(sp.x - r - pad, sp.y - r - pad, sp.z - r - pad, sp.x + r + pad);
                b1 = make_float4(sp.y + r + pad, sp.z + r + pad, b1.z, 0.f);
            }
        }
        bool out = hull_clear_of_box(P, D, h0, dh, 0.f, 1.f, 1.f, b0, b1);
        if (kind == 1) out = out && hull_clear_of_box(P, D, h0, dh, -back, 0.f, -1.f, b0, b1);
        if (kind == 0 && !out) {
            /* every ray has dot(d, n) > 0: (lp - P).n >= (C - P).n - R |n|, with a margin far above the rounding of the dot */
            const float4 q3 = sc.squares[4 * o + 3];
            const F3 n = f3(q3.x, q3.y, q3.z);
            const float nl = len(n);
            out = dot(D, n) > (R * 1.001f + 1e-4f * (dist + R)) * nl && nl > 0.f;
        }
        clear = out;
        if (!out) why = 1 + kind;
    }
    why_out = clear ? -1 : why;
    return clear;
}
__global__ void __launch_bounds__(256) k_shadow_group(DScene sc, FrameD fr, const DiffuseRec* __restrict__ drec, uint32_t n_rec,
                                                      uint32_t* __restrict__ keys, Counters* ctr) {
    __shared__ uint32_t s_tot[6];
    const uint32_t key = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_pairs = n_rec * (uint32_t)sc.n_lights;
    bool keep = key < n_pairs;
    int why_out = -1;
    if (keep) {
        uint32_t rec = key, li = 0u;
        if (sc.n_lights != 1) { rec = key / (uint32_t)sc.n_lights; li = key - rec * (uint32_t)sc.n_lights; }
        const float4 q0 = reinterpret_cast<const float4*>(drec + rec)[0];
        const DLight L = sc.lights[li];
        keep = !group_pair_clear(sc, f3(q0.x, q0.y, q0.z), f3(L.pos[0], L.pos[1], L.pos[2]), fr.shadow_mode == 0 ? fabsf(L.radius) : 0.f, why_out);
    }
    if (sc.group_debug) {                         /* one atomic per wave and reason */
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot(why_out == k);
            if (m && __lane_id() == 0) atomicAdd(&ctr->group_why[k], (unsigned long long)__popcll(m));
        }
    }
    /* block-aggregated append of the pairs that stay */
    const unsigned long long km = __ballot(keep);
    const int lane = __lane_id(), wave = threadIdx.x >> 6;
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    if (lane == 0) s_tot[wave] = (uint32_t)__popcll(km);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < 4; w++) { const uint32_t a = s_tot[w]; s_tot[w] = sum; sum += a; }
        s_tot[4] = sum ? (uint32_t)atomicAdd(&ctr->n_keys, (unsigned long long)sum) : 0u;
    }
    __syncthreads();
    if (keep) keys[s_tot[4] + s_tot[wave] + (uint32_t)__popcll(km & lt)] = key;
}

/* rt_group_clear: the same decision for pairs given by the caller (diagnostic; tests hold "clear" against Scene::isInShadow on sampled rays) */
__global__ void __launch_bounds__(256) k_group_probe(DScene sc, const float* __restrict__ pts, const float* __restrict__ ctr, const float* __restrict__ rad,
                                                     uint32_t n, int32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int why;
    const bool clear = sc.n_obox > 0 && group_pair_clear(sc, f3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), f3(ctr[3 * i], ctr[3 * i + 1], ctr[3 * i + 2]), fabsf(rad[i]), why);
    out[i] = clear ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_shadow_setup(DScene sc, FrameD fr, const DiffuseRec* __restrict__ drec,
                                                      const unsigned long long* __restrict__ dcode, uint32_t n_rec, int S,
                                                      float4* __restrict__ rays, uint32_t* __restrict__ masks) {
    unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long total = sc.group_keys ? (unsigned long long)sc.group_n * S : (unsigned long long)n_rec * sc.n_lights * S;
    if (idx >= total) return;
    uint32_t key; F3 o, d; float lightDistance;
    shadow_ray_of(sc, fr, drec, dcode, idx, S, key, o, d, lightDistance);
    rays[2 * idx] = make_float4(o.x, o.y, o.z, lightDistance);
    rays[2 * idx + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(key));
    if (masks) {
        /* the flat top-level pass of k_shadow<5> (flat_reach: the reference's slab test on every top-level leaf's own box), here with
         * every lane busy instead of at the persistent kernel's refill, where a quarter to a half of a wave's lanes take part */
        Ray r;
        r.o = o; r.d = d; r.inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);                /* AccelerationStruct.cpp:19, as at the refill */
        TopTables tt = top_tables_global_flat(sc);
        masks[idx] = flat_reach(tt, sc.n_flat, r);
    }
}

/* Scenes without meshes (the refraction and reflexion scenes: a sphere and a wall): a shadow ray is a handful of object
 * tests, and the persistent kernel's machinery -- ray buffer, refill, phases -- costs more than the rays (PMC on the
 * refraction scene: a quarter of the lanes active, the rest waiting for the next rays to arrive). Here every ray is
 * one thread from its diffuse record to its verdict (Scene::isInShadow through occluded()), no ray buffer; the lanes
 * of a wave that share a (record, light) pair add their occluded samples with one atomic. */
__global__ void __launch_bounds__(256) k_shadow_direct(DScene sc, FrameD fr, const DiffuseRec* __restrict__ drec,
                                                       const unsigned long long* __restrict__ dcode, uint32_t n_rec, int S, uint32_t* cnt) {
    extern __shared__ __align__(16) int s_stack[];
    unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long total = sc.group_keys ? (unsigned long long)sc.group_n * S : (unsigned long long)n_rec * sc.n_lights * S;
    const bool active = idx < total;
    uint32_t key = 0xFFFFFFFFu;
    bool occ = false;
    if (active) {
        F3 o, d; float lightDistance;
        shadow_ray_of(sc, fr, drec, dcode, idx, S, key, o, d, lightDistance);
        Ray r;
        r.o = o; r.d = d; r.inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);                /* AccelerationStruct.cpp:19 */
        occ = occluded(sc, r, lightDistance, s_stack + threadIdx.x, blockDim.x);
    }
    /* consecutive lanes hold consecutive samples: the first lane of every run of equal keys adds the run's count */
    const int lane = __lane_id();
    const uint32_t prev = __shfl_up(key, 1);
    const unsigned long long leaders = __ballot(active && (lane == 0 || prev != key));
    const unsigned long long occs = __ballot(occ);
    if (active && (lane == 0 || prev != key)) {
        const unsigned long long after = lane == 63 ? 0ull : leaders & (~0ull << (lane + 1));
        const int stop = after ? __builtin_ctzll(after) : 64;
        const unsigned long long run = (stop == 64 ? ~0ull : ((1ull << stop) - 1ull)) & (~0ull << lane);
        const uint32_t n = (uint32_t)__popcll(occs & run);
        if (n) atomicAdd(cnt + key, n);
    }
}

/* Scenes with meshes that most shadow rays pass by (a gorilla on a floor): the front pass settles every ray that stays
 * clear of all mesh frames the way k_shadow_direct does -- one thread per ray, full occupancy -- and hands the others, as
 * whole rays, to the persistent kernel through a compact ray buffer (one append atomic per 1024-thread block; *n_deferred
 * is the persistent kernel's ray count). LDS: top_cap stack entries per thread + 2 * waves + 2 words. */
#define RT_FBLOCK 1024
__global__ void __launch_bounds__(RT_FBLOCK) k_shadow_front(DScene sc, FrameD fr, const DiffuseRec* __restrict__ drec,
                                                            const unsigned long long* __restrict__ dcode, uint32_t n_rec, int S, uint32_t* cnt,
                                                            float4* __restrict__ rays, unsigned long long* n_deferred, unsigned long long* n_deferred_sum) {
    extern __shared__ __align__(16) int s_stack[];
    uint32_t* s_tot = reinterpret_cast<uint32_t*>(s_stack + (size_t)sc.top_cap * blockDim.x);
    unsigned long long idx = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long total = sc.group_keys ? (unsigned long long)sc.group_n * S : (unsigned long long)n_rec * sc.n_lights * S;
    const bool active = idx < total;
    uint32_t key = 0xFFFFFFFFu;
    bool occ = false, defer = false;
    F3 o = f3s(0.f), d = f3s(0.f);
    float lightDistance = 0.f;
    if (active) {
        shadow_ray_of(sc, fr, drec, dcode, idx, S, key, o, d, lightDistance);
        Ray r;
        r.o = o; r.d = d; r.inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);                /* AccelerationStruct.cpp:19 */
        occ = occluded_front(sc, r, lightDistance, s_stack + threadIdx.x, blockDim.x, defer);
        occ = occ && !defer;
    }
    const int lane = __lane_id();
    {   /* occluded samples: one atomic per run of lanes with the same (record, light) */
        const uint32_t prev = __shfl_up(key, 1);
        const bool leader = active && (lane == 0 || prev != key);
        const unsigned long long leaders = __ballot(leader), occs = __ballot(occ);
        if (leader) {
            const unsigned long long after = lane == 63 ? 0ull : leaders & (~0ull << (lane + 1));
            const int stop = after ? __builtin_ctzll(after) : 64;
            const unsigned long long run = (stop == 64 ? ~0ull : ((1ull << stop) - 1ull)) & (~0ull << lane);
            const uint32_t n = (uint32_t)__popcll(occs & run);
            if (n) atomicAdd(cnt + key, n);
        }
    }
    /* deferred rays: block-aggregated append */
    const unsigned long long dm = __ballot(defer);
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    if (lane == 0) s_tot[wave] = (uint32_t)__popcll(dm);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < nwaves; w++) { uint32_t a = s_tot[w]; s_tot[w] = sum; sum += a; }
        unsigned long long base = 0ull;
        if (sum) { base = atomicAdd(n_deferred, (unsigned long long)sum); atomicAdd(n_deferred_sum, (unsigned long long)sum); }
        s_tot[nwaves] = (uint32_t)base; s_tot[nwaves + 1] = (uint32_t)(base >> 32);
    }
    __syncthreads();
    if (defer) {
        const unsigned long long at = ((unsigned long long)s_tot[nwaves] | ((unsigned long long)s_tot[nwaves + 1] << 32)) + s_tot[wave] + (uint32_t)__popcll(dm & lt);
        rays[2 * at] = make_float4(o.x, o.y, o.z, lightDistance);
        rays[2 * at + 1] = make_float4(d.x, d.y, d.z, __uint_as_float(key));
    }
}

/* Work is phased so that the lanes of a wave execute the same kind of step together: a node phase
 * (lanes with a pending triangle wait) and a triangle phase (postponed tests run side by side). */
template <int MODE, bool COUNT>        /* MODE 0: exact 64-byte nodes, 1: quantised 4-wide tree, 2: ... with sub-trees in fat leaves and batched top-level rounds, 3: ... and sub-trees over the objects of top-level leaves, 5: mode 3 with the flat top-level pass. COUNT: step and clock diagnostics (RT_COUNT=1); they cost registers, so they are compiled out otherwise */
__global__ void __launch_bounds__(RT_BLOCK) __attribute__((amdgpu_waves_per_eu(RT_WAVES_PER_EU, 8))) k_shadow(DScene sc, const float4* __restrict__ rays, unsigned long long total, int S,
                                                        uint32_t* cnt, Counters* ctr, unsigned long long* next, int top_lds,
                                                        int* ovf, int lds_cap, int ovf_cap, int top_min, const unsigned long long* total_ptr,
                                                        const uint32_t* __restrict__ masks) {
    constexpr bool count_steps = COUNT;
    if (total_ptr) total = *total_ptr;       /* the rays the front pass deferred (k_shadow_front) */
    constexpr bool QUANT = MODE != 0, SUB = MODE >= 2, TOPSUB = MODE == 3 || MODE == 5, FLAT = MODE == 5;
    /* the plane filter in front of the triangle test pays where triangle rounds are a large part of a short walk (the
     * default scene: 14 node and 5 triangle steps per ray, -10 %); on the soup, where a ray takes 13 node steps per
     * triangle, the lanes that wait with a candidate thin out the node rounds (+2 %), and the single-mesh scenes are
     * indifferent: it is part of the variant for scenes with many top-level objects only */
    constexpr bool FILTER = MODE >= RT_FILTER_FROM_MODE;
    extern __shared__ __align__(16) int s_stack[];
    unsigned long long c_node = 0, c_tri = 0, c_other = 0, c_iter = 0, c_wn = 0, c_wt = 0, c_wtop = 0, c_ltop = 0, c_m1n = 0, c_m1t = 0;
    int* stk = s_stack + threadIdx.x;
    constexpr int stride = RT_BLOCK;        /* the host always launches RT_BLOCK-lane blocks */
    const int lane = __lane_id();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    WideStack wst;
    wst.lds = stk; wst.lds_cap = lds_cap; wst.ovf_cap = ovf_cap; wst.err = &ctr->walk_overflow;
    wst.ovf = ovf + ((size_t)blockIdx.x * ovf_cap) * RT_BLOCK + threadIdx.x;
    TopTables tt;
    {
        const char* base = reinterpret_cast<const char*>(sc.top_blob);
        if (top_lds) {                                  /* one wave per block: copy, then everybody reads the LDS copy */
            float4* dst = reinterpret_cast<float4*>(s_stack + lds_cap * RT_BLOCK);
            for (int i = lane; i < sc.top_blob_bytes / 16; i += RT_BLOCK) dst[i] = sc.top_blob[i];
            __syncthreads();
            base = reinterpret_cast<const char*>(dst);
        }
        tt.tnodes = reinterpret_cast<const float4*>(base + sc.top_off[0]);
        tt.tleaves = reinterpret_cast<const uint2*>(base + sc.top_off[1]);
        tt.titems = reinterpret_cast<const uint32_t*>(base + sc.top_off[2]);
        tt.squares = reinterpret_cast<const float4*>(base + sc.top_off[3]);
        tt.spheres = reinterpret_cast<const float4*>(base + sc.top_off[4]);
        tt.meshes = reinterpret_cast<const DMesh*>(base + sc.top_off[5]);
        tt.tf = reinterpret_cast<const TopFrame*>(base + sc.top_off[6]);
        tt.tsub = reinterpret_cast<const int*>(base + sc.top_off[6] + sizeof(TopFrame));
        tt.flat = reinterpret_cast<const float4*>(base + sc.top_off[7]);
        wst.tf = tt.tf;
    }
    /* (a launch with fewer rays than the grid could take RT_GRAB at a time would leave most waves without any) */
    const unsigned long long grab = total >= (unsigned long long)gridDim.x * RT_GRAB ? (unsigned long long)RT_GRAB
                                    : (total / gridDim.x < 64ull ? 64ull : (total / gridDim.x) & ~63ull);
    ShadowLane s;
    s.st = ST_IDLE;
    s.tri_pos = TRI_NONE;
    bool exhausted = false;             /* the global counter has run out; the wave-local range may still hold rays */
    unsigned long long wpos = 0, wend = 0;
    unsigned long long traced = 0;      /* lane 0 only */
    unsigned long long k_refill = 0, k_top = 0, k_nodeph = 0, k_triph = 0, k_t0 = 0, k_t1;
    for (;;) {
        if (count_steps) k_t0 = __builtin_readcyclecounter();
        unsigned long long idle = __builtin_amdgcn_ballot_w64(s.st == ST_IDLE);
        if ((!exhausted || wpos < wend) && (__popcll(idle) >= (MODE == 2 ? RT_REFILL_MIN_SUB : FLAT ? RT_REFILL_MIN_FLAT : RT_REFILL_MIN))) {
            /* ray indices come from a wave-local range [wpos, wend) that is topped up RT_GRAB rays at a time: one
             * atomic on the single global counter per RT_GRAB rays instead of one per refill (same-address
             * device-scope atomics serialise) */
            const uint32_t n = __popcll(idle);
            const uint32_t rank = __popcll(idle & lt);
            const unsigned long long avail = wend - wpos;      /* wave-uniform */
            unsigned long long idx = wpos + rank;
            if (avail >= n) wpos += n;
            else if (exhausted) { if (rank >= avail) idx = total; wpos = wend; }
            else {
                unsigned long long nb = 0;
                if (lane == 0) nb = atomicAdd(next, grab);
                nb = __shfl(nb, 0);
                if (rank >= avail) idx = nb + (rank - avail);
                wpos = nb + (n - avail);
                wend = nb + grab;
                if (wend >= total) {
                    exhausted = true;
                    wend = nb < total ? total : nb;
                    if (wpos > wend) wpos = wend;
                }
            }
            const unsigned long long got = __builtin_amdgcn_ballot_w64(s.st == ST_IDLE && idx < total);
            if (lane == 0) traced += __popcll(got);
            if (s.st == ST_IDLE && idx < total) {
                float4 ro = rays[2 * idx], rd = rays[2 * idx + 1];
                s.key = __float_as_uint(rd.w);
                s.bound = ro.w;                                                       /* lightDistance */
                s.r.o = f3(ro.x, ro.y, ro.z);
                s.r.d = f3(rd.x, rd.y, rd.z);
                s.r.inv = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);                 /* AccelerationStruct.cpp:19 */
                s.tri_pos = TRI_NONE;
                if (FLAT) stk[0] = (int)masks[idx];        /* which top-level leaves the ray reaches (k_shadow_setup: flat_reach); before the restart, which reads it back */
                shadow_restart(sc, s, 0, 0);
            }
        }
        if (__builtin_amdgcn_ballot_w64(s.st != ST_IDLE) == 0ull) {
            if (exhausted && wpos >= wend) break;
            continue;
        }
        if (count_steps) { if (s.st == ST_TOP) c_ltop++; k_t1 = __builtin_readcyclecounter(); k_refill += k_t1 - k_t0; k_t0 = k_t1; }
        /* top-level work runs lane by lane to its next mesh (or the end of the ray), so in a scene with many objects a
         * round for one or two lanes stalls the wave: there the host asks to wait until top_min lanes want one,
         * unless nothing else can move */
        bool top_round = true;
        if (SUB && top_min > 1) {       /* (MODE 2 only: the host picks that kernel for scenes with many top-level objects too) */
            const int ntop = __popcll(__builtin_amdgcn_ballot_w64(s.st == ST_TOP));
            top_round = ntop >= top_min || __builtin_amdgcn_ballot_w64(s.st >= ST_MESH) == 0ull;
        }
        if (top_round && s.st == ST_TOP) { if (FLAT) shadow_advance_top_flat<QUANT>(sc, tt, s, cnt, wst); else shadow_advance_top<QUANT, TOPSUB>(sc, tt, s, stk, stride, cnt, wst); }
        if (count_steps) { k_t1 = __builtin_readcyclecounter(); k_top += k_t1 - k_t0; k_t0 = k_t1; }
        /* node phase */
#pragma unroll 1
        for (int k = 0; k < (FLAT ? RT_NODE_STEPS_FLAT : RT_NODE_STEPS); k++) {
            /* a lane walks while it has a node to visit or something to pop, also with one leaf pending; it is
             * blocked only by a second leaf or when nothing is left but the pending leaf */
            const bool walking = s.st == ST_MESH, waiting = s.st == ST_BLOCKED;
            unsigned long long mw = __builtin_amdgcn_ballot_w64(walking), mt = __builtin_amdgcn_ballot_w64(waiting);
            if (mw == 0ull || __popcll(mw) < __popcll(mt)) break;
            if (count_steps) { c_iter++; if (threadIdx.x == 0) c_wn++; if (walking) { if (s.link >= 0 || s.msp > sc.top_cap) c_node++; else c_other++; if (s.ts & TS_MODE) c_m1n++; } }
            if (walking) { if (QUANT) shadow_quant_step<SUB, TOPSUB>(sc, s, wst); else shadow_node_step(sc, s, stk, stride); }
        }
        if (count_steps) { k_t1 = __builtin_readcyclecounter(); k_nodeph += k_t1 - k_t0; k_t0 = k_t1; }
        /* triangle phase: worth a round once enough lanes have a leaf pending, or when the walkers are the minority.
         * Quantised walk: the rounds run the leaf's box test and the plane filter (tri_maybe); the records that pass
         * wait (TRI_EXACT) for a round of the reference's test, which starts once RT_EXACT_MIN lanes hold one or the
         * walkers are the minority. */
        {
            const bool pend = s.st >= ST_MESH && s.tri_pos != TRI_NONE;
            const int np = __popcll(__builtin_amdgcn_ballot_w64(pend)), nw = __popcll(__builtin_amdgcn_ballot_w64(s.st == ST_MESH));
            const int nb = __popcll(__builtin_amdgcn_ballot_w64(s.st == ST_BLOCKED));
            if (np < (FLAT ? RT_TRI_MIN_FLAT : RT_TRI_MIN) && nw > nb && nw > 0) continue;
        }
#pragma unroll 1
        for (int k = 0; k < RT_TRI_STEPS; k++) {
            bool pending = s.st >= ST_MESH && s.tri_pos != TRI_NONE && !(FILTER && (s.tri_pos & TRI_EXACT));
            const int np = __popcll(__builtin_amdgcn_ballot_w64(pending));
            if (np == 0) break;
            if (k > 0 && np < RT_TRI_MORE && __builtin_amdgcn_ballot_w64(s.st == ST_MESH) != 0ull) break;   /* a thin follow-up round only if nobody could walk instead */
            if (count_steps) { c_iter++; if (threadIdx.x == 0) c_wt++; if (pending) { c_tri++; if (s.ts & TS_MODE) c_m1t++; } }
            if (pending) { if (QUANT) shadow_quant_tri_step<SUB, FILTER>(sc, s, cnt, wst); else shadow_tri_step(sc, s, cnt); }
        }
        if (FILTER) {
            const bool cand = s.st >= ST_MESH && s.tri_pos != TRI_NONE && (s.tri_pos & TRI_EXACT);
            const int nc = __popcll(__builtin_amdgcn_ballot_w64(cand));
            const int nw = __popcll(__builtin_amdgcn_ballot_w64(s.st == ST_MESH)), nb = __popcll(__builtin_amdgcn_ballot_w64(s.st == ST_BLOCKED));
            if (nc > 0 && (nc >= (FLAT ? RT_EXACT_MIN_FLAT : RT_EXACT_MIN) || nw <= nb)) {
                if (count_steps) { c_iter++; if (threadIdx.x == 0) c_wtop++; if (cand) c_other++; }
                if (cand) shadow_quant_exact_step(sc, s, cnt);
            }
        }
        if (count_steps) { k_t1 = __builtin_readcyclecounter(); k_triph += k_t1 - k_t0; }
    }
    if (lane == 0 && traced) atomicAdd(&ctr->rays_shadow, traced);
    if (count_steps) {
        atomicAdd(&ctr->sh_node_steps, c_node); atomicAdd(&ctr->sh_tri_steps, c_tri);
        atomicAdd(&ctr->sh_other_steps, c_other); atomicAdd(&ctr->sh_loop_iters, c_iter);
        atomicAdd(&ctr->sh_top_lanes, c_ltop); atomicAdd(&ctr->sh_m1_node, c_m1n); atomicAdd(&ctr->sh_m1_tri, c_m1t);
        if (threadIdx.x == 0) { atomicAdd(&ctr->sh_clk_refill, k_refill); atomicAdd(&ctr->sh_clk_top, k_top); atomicAdd(&ctr->sh_clk_node, k_nodeph); atomicAdd(&ctr->sh_clk_tri, k_triph); }
        if (threadIdx.x == 0) { atomicAdd(&ctr->sh_node_iters, c_wn); atomicAdd(&ctr->sh_tri_iters, c_wt); atomicAdd(&ctr->sh_top_iters, c_wtop); }
    }
}

/* Scene::phong (Scene.h:90-108) + the shadow factor of Scene.h:210-222. */
__global__ void __launch_bounds__(256) k_shade(DScene sc, const DiffuseRec* __restrict__ drec, const uint32_t* __restrict__ dtex, uint32_t n_rec, int S,
                                               const uint32_t* __restrict__ cnt, long long* acc, uint32_t* nonfinite, Counters* ctr) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = i < n_rec;
    if (!active) i = n_rec - 1;                      /* (a valid record to read; nothing of it is added) */
    const float4* q = reinterpret_cast<const float4*>(drec + i);
    float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    F3 P = f3(q0.x, q0.y, q0.z), N = f3(q0.w, q1.x, q1.y), V = f3(q1.z, q1.w, q2.x), thr = f3(q2.y, q2.z, q2.w);
    uint32_t matword = __float_as_uint(q3.x), slot = __float_as_uint(q3.y);
    DMat m = sc.mats[matword & 0x00FFFFFFu];
    F3 ka = f3(m.ka[0], m.ka[1], m.ka[2]), kd = f3(m.kd[0], m.kd[1], m.kd[2]), ks = f3(m.ks[0], m.ks[1], m.ks[2]);
    if (matword & 0x40000000u) { float c = (matword & 0x20000000u) ? 1.f : 0.f; kd = f3s(c); ks = f3s(c); }
    if (matword & 0x80000000u) {                                   /* texel of Material::getColorAt, Scene.h:189-191 */
        uint32_t p = dtex[i];
        kd = f3((float)((double)(p & 255u) / 255.0), (float)((double)((p >> 8) & 255u) / 255.0), (float)((double)((p >> 16) & 255u) / 255.0));
        ks = kd;
    }
    F3 color = f3s(0.f);
    float avg = 0.f;
    for (int li = 0; li < sc.n_lights; li++) {
        DLight L = sc.lights[li];
        float strength = (float)cnt[(size_t)i * sc.n_lights + li] / (float)S;            /* Scene.h:133 */
        avg += strength;
        F3 Li = f3(L.color[0], L.color[1], L.color[2]);
        F3 Ld = normalized(sub(f3(L.pos[0], L.pos[1], L.pos[2]), P));
        float theta = dot(N, Ld);
        F3 ambient = comp(Li, ka);
        F3 diffuse = mul((theta < 0.f) ? 0.f : theta, comp(Li, kd));
        F3 rr = sub(mul((float)(2. * (double)theta), N), Ld);
        float rv = dot(rr, V);
        float rDotV = (0.0f < rv) ? rv : 0.0f;
        F3 specular = mul((float)pow((double)rDotV, m.shininess), comp(Li, ks));
        color = add(color, add(add(ambient, diffuse), specular));
    }
    float f = (float)(1. - (double)(avg / (float)sc.n_lights));
    color = mul(f, color);
    accumulate_runs(acc, nonfinite, ctr, slot, comp(thr, color), active);
}

/* The displayable frame: the bytes the reference writes into its PPM (main.cpp:251-253: (int)(255.f * std::min<float>(1.f, c)),
 * where std::min keeps its first argument for a NaN), as RGBA8 with alpha 255; what the text file can hold and a byte
 * cannot (negative values) is clamped to 0. Rows top to bottom like the PPM, or bottom-up for glDrawPixels / GL textures. */
__global__ void __launch_bounds__(256) k_display(const float* __restrict__ rgb, int w, int h, int bottom_up, uint32_t* __restrict__ rgba) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)w * (uint32_t)h) return;
    uint32_t px = 0xFF000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float c = rgb[3 * (size_t)i + k];
        const float m = c < 1.f ? c : 1.f;                   /* std::min<float>(1.f, c): (c < 1.f) ? c : 1.f */
        const float v = 255.f * m;
        const uint32_t b = v >= 0.f ? (uint32_t)(int)v : 0u;
        px |= b << (8 * k);
    }
    const uint32_t x = i % (uint32_t)w, y = i / (uint32_t)w;
    rgba[(size_t)(bottom_up ? (uint32_t)h - 1u - y : y) * (uint32_t)w + x] = px;
}

__global__ void __launch_bounds__(256) k_resolve(const long long* acc, const uint32_t* nonfinite, uint32_t n_pix, int spp, float* rgb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pix * 3) return;
    float sum = (float)((double)acc[i] * (1.0 / RT_FIX_SCALE));
    uint32_t f = (nonfinite[i / 3] >> (3 * (i % 3))) & 7u;
    if (f) {
        if ((f & 1u) || (f & 6u) == 6u) sum = __int_as_float(0x7fc00000);
        else sum = (f & 2u) ? __int_as_float(0x7f800000) : __int_as_float(0xff800000);
    }
    rgb[i] = sum / (float)spp;                                                           /* main.cpp:219 */
}

/* Scene::computeIntersection on explicit rays (rt_intersect). */
struct HitOut { int32_t exists, obj, tri; float t; float pos[3]; float nrm[3]; float u, v; };
template <bool SUB, bool WIDE>
__global__ void __launch_bounds__(RT_BLOCK) k_intersect(DScene sc, const float* org, const float* dir, uint32_t n, HitOut* out) {
    extern __shared__ __align__(16) int s_stack[];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray r = make_ray(f3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), f3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]));
    HitD h = closest_hit<SUB, WIDE>(sc, r, s_stack + threadIdx.x, blockDim.x);
    HitOut o;
    o.exists = h.exists; o.obj = h.exists ? h.obj : -1; o.tri = h.exists ? h.tri : 0; o.t = h.exists ? h.t : FLT_MAX;
    o.pos[0] = h.pos.x; o.pos[1] = h.pos.y; o.pos[2] = h.pos.z;
    o.nrm[0] = h.n.x; o.nrm[1] = h.n.y; o.nrm[2] = h.n.z;
    o.u = h.u; o.v = h.v;
    if (h.exists && h.obj >= sc.n_squares && h.obj < sc.n_squares + sc.n_spheres) sphere_uv(r.d, o.u, o.v);
    out[i] = o;
}
/* Scene::isInShadow on explicit (point, light) pairs (rt_occluded). */
__global__ void __launch_bounds__(RT_BLOCK) k_occluded(DScene sc, const float* pts, const float* lp, uint32_t n, int32_t* out) {
    extern __shared__ __align__(16) int s_stack[];
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool o = in_shadow(sc, f3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), f3(lp[3 * i], lp[3 * i + 1], lp[3 * i + 2]),
                       s_stack + threadIdx.x, blockDim.x);
    out[i] = o ? 1 : 0;
}
