/*
 * ref_shim.cpp -- C entry points over the GL-free part of the reference,
 * compiled UNMODIFIED from /root/reference (include/Vec3.h, include/Line.h,
 * src/Trackball.cpp, include/stb_image.h) into oracle/_ref/libref.so.  Test
 * infrastructure only: it pins the oracle's vector primitives, Ray/Line
 * normalisation, build_rotmatrix and the image reader (the texels
 * Material::loadTexture hands to the render loop, Material.h:128-138) against the
 * real reference code.  Everything else on the path includes <GL/glut.h> and is
 * unbuildable in this image.
 */
#include <cstring>
#include "Vec3.h"
#include "Line.h"
#include "Trackball.h"
#define STB_IMAGE_IMPLEMENTATION          /* as main.cpp:34-35 */
#include "stb_image.h"

extern "C" {
/* out[0]=dot, out[1..3]=cross, out[4]=length(a), out[5..7]=normalize(a),
 * out[8..10]=a*s, out[11..13]=a/s, out[14..16]=compProduct, out[17..19]=Line(a,b).direction() */
void ref_vec3_ops(const float* a, const float* b, float s, float* out) {
    Vec3 A(a[0], a[1], a[2]), B(b[0], b[1], b[2]);
    out[0] = Vec3::dot(A, B);
    Vec3 c = Vec3::cross(A, B);
    out[1] = c[0]; out[2] = c[1]; out[3] = c[2];
    out[4] = A.length();
    Vec3 n = A; n.normalize();
    out[5] = n[0]; out[6] = n[1]; out[7] = n[2];
    Vec3 m = A * s;
    out[8] = m[0]; out[9] = m[1]; out[10] = m[2];
    Vec3 d = A / s;
    out[11] = d[0]; out[12] = d[1]; out[13] = d[2];
    Vec3 p = Vec3::compProduct(A, B);
    out[14] = p[0]; out[15] = p[1]; out[16] = p[2];
    Line L(A, B);
    out[17] = L.direction()[0]; out[18] = L.direction()[1]; out[19] = L.direction()[2];
}
void ref_mat3_mul(const float* m9, const float* v, float* out) {
    Mat3 M(m9[0], m9[1], m9[2], m9[3], m9[4], m9[5], m9[6], m9[7], m9[8]);
    Vec3 r = M * Vec3(v[0], v[1], v[2]);
    out[0] = r[0]; out[1] = r[1]; out[2] = r[2];
}
void ref_build_rotmatrix(const float* q, float* m16) {
    float m[4][4];
    float qq[4] = {q[0], q[1], q[2], q[3]};
    build_rotmatrix(m, qq);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m16[4 * i + j] = m[i][j];
}
void ref_trackball(float* q, float p1x, float p1y, float p2x, float p2y) { trackball(q, p1x, p1y, p2x, p2y); }
/* Material::loadTexture (Material.h:129): stbi_load(file, &w, &h, &channels, 0). Copies up to cap bytes into out
 * (may be null to query the size); returns 0, or -1 if the file does not decode. */
int ref_stbi_load(const char* path, int* w, int* h, int* channels, unsigned char* out, long cap) {
    unsigned char* px = stbi_load(path, w, h, channels, 0);
    if (!px) return -1;
    long n = (long)*w * *h * *channels;
    if (out) memcpy(out, px, (size_t)(n < cap ? n : cap));
    stbi_image_free(px);
    return 0;
}
}
