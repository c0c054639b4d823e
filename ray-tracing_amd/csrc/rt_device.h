/*
 * rt_device.h -- device-side arithmetic of the render loop for gfx950 (wave64).
 *
 * Every function restates one reference routine in IEEE float32 with the
 * reference's operation order (the translation unit is compiled with
 * -ffp-contract=off; hipcc's default correctly-rounded f32 divide/sqrt is kept),
 * so that hit identities and geometry are bit-identical to the CPU path.
 * Places where the reference promotes to double are double here as well.
 */
#pragma once
#ifndef RT_HOST_EMU          /* tools/emu compiles these headers with g++ behind a shim of the HIP names (a memory checker for the walks) */
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <float.h>

#define RT_DEV __device__ __forceinline__

struct F3 { float x, y, z; };

RT_DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV F3 f3s(float s) { return f3(s, s, s); }
RT_DEV F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }            /* Vec3.h:97 */
RT_DEV F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }            /* Vec3.h:100 */
RT_DEV F3 mul(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }               /* Vec3.h:103,106 */
RT_DEV F3 divs(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }              /* Vec3.h:109 */
RT_DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           /* Vec3.h:38 */
RT_DEV F3 cross(F3 a, F3 b) {                                                        /* Vec3.h:41 */
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RT_DEV F3 comp(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }           /* Vec3.h:66 */
RT_DEV float sqlen(F3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }               /* Vec3.h:29 */
/* Vec3.h:32 computes sqrt in double and narrows; for a float argument that equals the
 * correctly rounded float sqrt (53 >= 2*24+2 bits); sqrtf lowers to the correctly rounded
 * sequence on gfx950 (__fsqrt_rn would be the 1-ulp v_sqrt_f32). */
RT_DEV float len(F3 a) { return sqrtf(sqlen(a)); }
RT_DEV F3 normalized(F3 a) { float L = len(a); return f3(a.x / L, a.y / L, a.z / L); }   /* Vec3.h:37 */
RT_DEV float comp_at(F3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

/* Counter RNG in place of random()/rand() (Scene.h:115-116, main.cpp:195-196). */
__host__ __device__ inline uint32_t rt_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
/* the rounds that only depend on the path (seed, pixel, sample, path code): one word per diffuse record, k_hash_records */
__host__ __device__ inline uint32_t rt_rng_prefix(uint32_t seed, uint32_t pixel, uint32_t sample, uint64_t code) {
    uint32_t h = rt_mix32(seed + 0x9e3779b9U);
    h = rt_mix32((h ^ pixel) + 0x9e3779b9U);
    h = rt_mix32((h ^ sample) + 0x9e3779b9U);
    h = rt_mix32((h ^ (uint32_t)code) + 0x9e3779b9U);
    h = rt_mix32((h ^ (uint32_t)(code >> 32)) + 0x9e3779b9U);
    return h;
}
__host__ __device__ inline uint32_t rt_rng_tail(uint32_t h, uint32_t light, uint32_t i, uint32_t which) {
    h = rt_mix32((h ^ light) + 0x9e3779b9U);
    h = rt_mix32((h ^ i) + 0x9e3779b9U);
    h = rt_mix32((h ^ which) + 0x9e3779b9U);
    return h;
}
__host__ __device__ inline uint32_t rt_rng_impl(uint32_t seed, uint32_t pixel, uint32_t sample, uint64_t code,
                                                uint32_t light, uint32_t i, uint32_t which) {
    return rt_rng_tail(rt_rng_prefix(seed, pixel, sample, code), light, i, which);
}

/* ---------------------------------------------------------------- device scene */
struct DMat {              /* 80 B */
    float ka[3], kd[3], ks[3];
    float ior;
    double shininess;
    int32_t type;
    float transparency, reflectivity;
    int32_t tex, nmap;     /* index into DScene::textures, -1 = none (hasTexture / hasNormalMap, Material.h:29-30) */
    int32_t pad;
    double scale;          /* materialScale, Material.h:31 */
};
struct DTex { const uint8_t* data; int32_t w, h, ch, pad; };   /* stbi_load layout, Material.h:24-27 */
struct DLight { float pos[3]; float radius; float color[3]; float pad; };
struct DMesh {                 /* 48 B */
    int32_t root;              /* link of the tree's root */
    int32_t material;
    float delta;               /* largest finite hit excursion of a child of this mesh (rtgpu.hip, HitBox); informational */
    float amax;                /* largest |coordinate| of the mesh's vertices */
    float qmin[3], qstep[3];   /* frame of the 16-bit box planes in DScene::wnodes: plane = qmin + q*qstep */
    int32_t wroot;             /* link of the 4-wide shadow tree's root */
    int32_t pad;
};

#define RT_TRI_STRIDE 8
struct DScene {
    const float4* tri;          /* RT_TRI_STRIDE x float4 (one 128-byte line) per triangle, leaf order: (c0,D) (c1,area) (c2,last-in-leaf
                                 * flag) (n,-), then for the first record of a leaf the reference box of that leaf (min.xyz,max.x)
                                 * (max.yz,-,-) -- a shadow ray's leaf visit is one cache line -- and two spare slots */
    const uint4* tri_ids;       /* (v0, v1, v2 global vertex ids, index into Mesh::triangles), leaf order */
    const float* normals;       /* 3 per vertex */
    const float4* mnodes;       /* 4 x float4 per inner node; child links encoded as in rt_kernels.h (LINK_*) */
    const uint4* wnodes;        /* shadow walk: 4-wide nodes, 64 bytes: 4 x (x,y,z plane pairs of a child box rounded outward to 16 bits), 4 links */
    const DMesh* meshes;
    const float4* tnodes;
    const uint2* tleaves;
    const uint32_t* titems;
    const float4* squares;      /* 4 x float4: (bl,D) (right,|right|^2) (up,|up|^2) (n,material) */
    const float4* spheres;      /* 2 x float4: (center,radius) (material,-,-,-) */
    const DMat* mats;
    const DTex* textures;
    const DLight* lights;
    const double* sincos;       /* sin[360], cos[360] of (float)k * degToRad, host libm (Scene.h:112-119) */
    int32_t n_lights, n_squares, n_spheres, n_meshes;
    int32_t top_root;
    int32_t top_cap;            /* stack entries reserved for the top level */
    int32_t stack_cap;          /* total per-lane stack entries */
    int32_t cstack_cap;         /* ... of the closest-hit walks (stack_cap + the sub-tree of a fat leaf) */
    int32_t wstack_cap;         /* per-lane stack entries the 4-wide shadow walk can need (three pending links per level) */
    int32_t quant_ok;           /* all mesh boxes are finite and nested: the 32-byte shadow nodes may be used */
    int32_t wide_closest;       /* closest hits walk the 4-wide quantised tree too (mesh_closest_wide) */
    int32_t n_top_items;        /* entries of the top-level item lists */
    int32_t has_sub;            /* some leaves carry a sub-tree (rtgpu.hip, LeafSplitter) */
    /* the top-level tables once more, packed into one blob that k_shadow copies into LDS when it is small:
     * byte offsets of tnodes, tleaves, titems, squares, spheres, meshes inside top_blob */
    const float4* top_blob;
    int32_t top_blob_bytes;
    int32_t top_off[8];         /* ... and tsub: per top-level leaf the root of the sub-tree over its objects, or LINK_NONE; [7]: the flat leaf table (n_flat) */
    int32_t n_flat;             /* k_shadow with the flat top-level pass: number of top-level leaves with a box, in the order the reference's walk meets them
                                 * (2 x float4 each: box min.xyz max.x | max.yz, leaf index, -); 0 = no flat pass (rtgpu.hip) */
    int32_t has_topsub;         /* some top-level leaves carry a sub-tree over their objects (rtgpu.hip) */
    float light_sum;            /* sum over the lights of their largest colour component; < 0: contribution culling off (k_closest) */
    const float* cull_gpow;     /* [RT_MAX_PATH_DEPTH + 2]: entry r bounds how much a path's throughput can grow over r more bounces (k_closest) */
    float cull_cmax;            /* what one hit (or the sky) can add per unit of throughput, any material of the scene; < 0: no culling of path states */
    /* per launch (run_wavefront): k_closest leaves, in a diffuse record's code slot, the state of the counter RNG after the rounds its shadow
     * rays share (rt_rng_prefix) instead of the path code, so that no pass over the records (k_hash_records) is needed for it */
    int32_t rng_hash;
    uint32_t rng_seed;
    /* per launch: k_closest takes one state per 2^spread_shift lanes (the others stay idle). A small batch of bounced rays is
     * latency-bound, and the lanes of a wave serialise their divergent walks: spread over more waves the batch ends sooner */
    int32_t spread_shift;
    /* group culling of shadow rays (k_shadow_group): per top-level object ordinal the box every accepted hit on it stays inside
     * (2 x float4: min.xyz max.x | max.yz kind -; kind 0 square, 1 sphere, 2 mesh; an object that cannot be hit has min = +inf,
     * one without a bound min = -inf, max = +inf), and the box of admitted ray origins the paddings assume. n_obox = 0: none. */
    const float4* obox;
    int32_t n_obox;
    float gomin[3], gomax[3];
    /* per launch: the (record, light) pairs whose rays are traced -- pair k of the launch is key group_keys[k] (nullptr: all
     * n_rec * n_lights pairs in order) */
    const uint32_t* group_keys;
    uint32_t group_n;
    int32_t group_debug;        /* RT_DEBUG: k_shadow_group counts why pairs stay (Counters::group_why) */
};
/* head of section 6 of top_blob, in front of tsub */
struct TopFrame {
    DMesh frame;                /* frame (qmin, qstep, amax) of the 16-bit planes of the sub-trees of top-level leaves: the whole scene */
    float omin[3], omax[3];     /* rays that start outside this box take the plain item scan (the object boxes assume origins inside) */
    float pad[2];
};

struct Ray { F3 o, d, inv; };
/* Ray/Line constructor (Line.h:13-16): the direction is normalised again. */
RT_DEV Ray make_ray(F3 o, F3 d) {
    Ray r;
    r.o = o;
    r.d = normalized(d);
    r.inv = f3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);   /* AccelerationStruct.cpp:19, hoisted: same value every call */
    return r;
}

/* AABB::intersect, AccelerationStruct.cpp:13-30 (tMin = 0, tMax = 100), written without branches so
 * that a node's loads are issued together instead of one memory round trip per axis:
 *  - std::max(tMin,t0) = (tMin<t0)?t0:tMin and std::min(tMax,t1) = (t1<tMax)?t1:tMax ignore a NaN
 *    t0/t1 (tMin/tMax are never NaN), which is exactly fmaxf/fminf (v_max_f32/v_min_f32);
 *  - tMin only grows and tMax only shrinks, so the per-axis `if (tMin > tMax) return false` of the
 *    reference is equivalent to one test after the third axis;
 *  - the swap on invD < 0 selects the same two products either way. */
RT_DEV bool slab(const Ray& r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float& tEnter) {
    float ax = (mnx - r.o.x) * r.inv.x, bx = (mxx - r.o.x) * r.inv.x;
    float ay = (mny - r.o.y) * r.inv.y, by = (mxy - r.o.y) * r.inv.y;
    float az = (mnz - r.o.z) * r.inv.z, bz = (mxz - r.o.z) * r.inv.z;
    bool sx = r.inv.x < 0.0f, sy = r.inv.y < 0.0f, sz = r.inv.z < 0.0f;
    float tMin = fmaxf(0.0f, sx ? bx : ax);
    float tMax = fminf(100.0f, sx ? ax : bx);
    tMin = fmaxf(tMin, sy ? by : ay);
    tMax = fminf(tMax, sy ? ay : by);
    tMin = fmaxf(tMin, sz ? bz : az);
    tMax = fminf(tMax, sz ? az : bz);
    tEnter = tMin;
    return !(tMin > tMax);
}

/* Triangle::getIntersection (Triangle.h:86-121) on a record precomputed at upload:
 * c0,c1,c2 already scaled by 1.0001f (Mesh.cpp:107-109), n/area per Triangle.h:28-33,
 * D = dot(c0,n) per Triangle.h:58. Returns exists; t,w0,w1,w2 out. */
RT_DEV bool tri_test(const Ray& r, float4 q0, float4 q1, float4 q2, float4 q3, float& t_out, float& w0, float& w1, float& w2) {
    F3 c0 = f3(q0.x, q0.y, q0.z), c1 = f3(q1.x, q1.y, q1.z), c2 = f3(q2.x, q2.y, q2.z), n = f3(q3.x, q3.y, q3.z);
    float D = q0.w, area = q1.w;
    float dn = dot(r.d, n);
    if (dn == 0) return false;                                   /* Triangle.h:49-53 */
    float t = (D - dot(r.o, n)) / dn;                            /* Triangle.h:59 */
    F3 P = add(r.o, mul(t, r.d));                                /* Triangle.h:61 */
    /* The reference tests "in front" (Triangle.h:94-96) before the barycentric sum (Triangle.h:101-107);
     * both must pass and neither has side effects, so the one that rejects most goes first. */
    /* Triangle.h:65-84: (|C| / 2.) / area in double, narrowed; equal to the float quotient below */
    float u0 = (len(cross(sub(c2, c1), sub(P, c1))) * 0.5f) / area;
    float u1 = (len(cross(sub(c0, c2), sub(P, c2))) * 0.5f) / area;
    float u2 = (len(cross(sub(c1, c0), sub(P, c0))) * 0.5f) / area;
    float sum = u0 + u1 + u2;
    if ((double)sum < 0.9999 || (double)sum > 1.00001) return false;   /* Triangle.h:105 */
    /* Triangle.h:94-96: reject if dot(normalize(P - o), d) < 0. The normalisation (a correctly rounded sqrt and three
     * divisions) only matters when that cosine is within rounding of zero: the computed value differs from the true
     * cosine c of the float vectors by less than 4e-7 |d|, and the plain product s = (P - o).d equals c |P - o| up to
     * 2e-7 |P - o| |d|. So for |s| > 1e-4 |P - o| (|d| is 1 within rounding: Line.h:15) the sign of s IS the sign the
     * reference computes, and the exact evaluation is kept for the rest (and for lengths whose square leaves the
     * normal float range, where the reference's own arithmetic degenerates: NaN and overflow compare false here). */
    F3 v = sub(P, r.o);
    float sd = dot(v, r.d), vv = sqlen(v);
    if (vv > 1e-30f && vv < 1e30f && sd * sd > 1e-8f * vv) {
        if (sd < 0) return false;
    } else {
        F3 cd = normalized(v);                                   /* Triangle.h:94-95 */
        if (dot(cd, r.d) < 0) return false;                      /* Triangle.h:96 */
    }
    t_out = len(sub(r.o, P));                                    /* Triangle.h:110 */
    w0 = u0; w1 = u1; w2 = u2;
    return true;
}

/* Material::getColorAt / getNormalAt, Material.h:86-104: nearest texel, % wrap, byte / 255.0 narrowed to float.
 * u, v are never negative on the reference's primitives (there a negative index reads out of bounds); a negative
 * remainder is wrapped into the image. */
RT_DEV F3 texel_at(const DTex& t, double u, double v, double scale) {
    int pu = (int)(u * (double)t.w * scale) % t.w;
    int pv = (int)(v * (double)t.h * scale) % t.h;
    if (pu < 0) pu += t.w;
    if (pv < 0) pv += t.h;
    const uint8_t* px = t.data + ((size_t)pv * (size_t)t.w + (size_t)pu) * (size_t)t.ch;
    return f3((float)((double)px[0] / 255.0), (float)((double)px[1] / 255.0), (float)((double)px[2] / 255.0));
}
/* u, v of a sphere hit, Sphere.h:141-142: from the ray direction, double libm, narrowed where Scene.h:184 builds a Vec3 */
RT_DEV void sphere_uv(F3 d, float& u, float& v) {
    u = (float)(0.5 + (atan2((double)d.z, (double)d.x) / (2. * M_PI)));
    v = (float)(0.5 + (asin((double)d.y) / M_PI));
}

/* Sphere::intersect, Sphere.h:86-146. */
RT_DEV bool sphere_test(const Ray& r, float4 s, float& t_out, F3& n_out) {
    F3 c = f3(s.x, s.y, s.z);
    float radius = s.w;
    F3 oc = sub(r.o, c);
    float a = dot(r.d, r.d);
    float b = 2.0f * dot(oc, r.d);
    float cc = dot(oc, oc) - (radius * radius);
    float disc = (b * b) - (4.f * a * cc);
    float t = 0;
    if (disc > 0) {
        float sq = sqrtf(disc);
        float t1 = (-b - sq) / (2.0f * a);
        float t2 = (-b + sq) / (2.0f * a);
        if (t1 > 0 && t2 > 0) t = (t2 < t1) ? t2 : t1;
        else if (t1 > 0) t = t1;
        else if (t2 > 0) t = t2;
        else return false;
    } else if (disc == 0) {
        t = (float)((double)(-b) / (2. * (double)a));            /* Sphere.h:113 */
    }
    if (t != 0.f) {
        F3 P = add(r.o, mul(t, r.d));
        F3 n = normalized(sub(P, c));
        if (dot(r.d, n) > 0.f) return false;                     /* Sphere.h:122 */
        t_out = t;
        n_out = n;
        return true;
    }
    return false;
}

/* Square::intersect, Square.h:62-111, on a record precomputed at upload. */
RT_DEV bool square_test(const Ray& r, float4 q0, float4 q1, float4 q2, float4 q3, float& t_out, float& u_out, float& v_out) {
    F3 bl = f3(q0.x, q0.y, q0.z), right = f3(q1.x, q1.y, q1.z), up = f3(q2.x, q2.y, q2.z), n = f3(q3.x, q3.y, q3.z);
    float dn = dot(r.d, n);
    if (dn > 0) return false;                                    /* Square.h:74 */
    float t = (q0.w - dot(r.o, n)) / dn;                         /* Square.h:77-78 */
    if (t > 0 && !isinf(t)) {
        F3 P = add(r.o, mul(t, r.d));
        F3 d = sub(P, bl);
        float u = dot(d, right) / q1.w;
        float v = dot(d, up) / q2.w;
        if (u >= 0 && u <= 1.0f && v >= 0 && v <= 1.0f) {
            t_out = t; u_out = u; v_out = v;
            return true;
        }
    }
    return false;
}
