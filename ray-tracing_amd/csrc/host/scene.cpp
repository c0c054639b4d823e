/*
 * scene.cpp -- host-side mirror of the reference's Scene / Mesh / Square / Sphere /
 * Camera types, its two BVH builders and its render entry point (see rthost.h).
 * Product code: never includes or calls anything under oracle/.
 *
 * Float arithmetic follows the reference's operation order (compile with
 * -ffp-contract=off) so that the flattened scene is bit-identical to what the CPU
 * path would hold in memory.
 */
#include "rthost.h"
#include "../lbvh.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <fstream>
#include <iterator>
#include <new>
#include <sstream>
#include <thread>

#include <zlib.h>      /* inflate, for load_png only */

namespace rth {

static std::string g_asset_dir = "assets/models";
void set_asset_dir(const std::string& dir) { g_asset_dir = dir; }
static std::string g_image_dir = "assets/img";
void set_image_dir(const std::string& dir) { g_image_dir = dir; }

/* Binary PPM: "P6", whitespace / # comments, width, height, maxval 255, one whitespace byte, then rgb bytes.
 * This is what stbi_load returns for the reference's .ppm assets (3 channels, rows top to bottom). */
bool load_ppm(const std::string& filename, Image& out) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return false;
    int c = std::fgetc(f);
    auto next = [&](std::string& tok) {
        tok.clear();
        for (;;) {
            while (c == ' ' || c == '\t' || c == '\r' || c == '\n') c = std::fgetc(f);
            if (c != '#') break;
            while (c != '\n' && c != EOF) c = std::fgetc(f);
        }
        while (c != EOF && c != ' ' && c != '\t' && c != '\r' && c != '\n') { tok.push_back((char)c); c = std::fgetc(f); }
        return !tok.empty();
    };
    std::string tok;
    bool ok = next(tok) && tok == "P6";
    ok = ok && next(tok) && (out.width = std::atoi(tok.c_str())) > 0;
    ok = ok && next(tok) && (out.height = std::atoi(tok.c_str())) > 0;
    ok = ok && next(tok) && std::atoi(tok.c_str()) == 255;     /* `c` is the single whitespace byte after maxval, already consumed */
    if (ok) {
        out.channels = 3;
        out.data.resize((size_t)out.width * out.height * 3);
        ok = std::fread(out.data.data(), 1, out.data.size(), f) == out.data.size();
    }
    std::fclose(f);
    return ok;
}

/* PNG, for the .png assets of setup_cornell_box: the texels stbi_load(file, .., 0) returns for an 8-bit, non-interlaced
 * grey / grey+alpha / RGB / RGBA file (PNG is lossless, so any conforming decoder agrees; checked against the reference's
 * vendored decoder in tests/test_oracle_pins.py). Palette, 16-bit, interlaced and tRNS files are refused. */
namespace {
struct PngReader {
    const std::vector<uint8_t>& f;
    size_t at = 8;
    uint32_t word(size_t o) const { return (uint32_t)f[o] << 24 | (uint32_t)f[o + 1] << 16 | (uint32_t)f[o + 2] << 8 | (uint32_t)f[o + 3]; }
    /* next chunk: false at the end of the file or on a length that does not fit */
    bool next(std::string& type, size_t& begin, size_t& len) {
        if (at + 12 > f.size()) return false;
        len = word(at);
        if (len > f.size() - at - 12) return false;
        type.assign((const char*)&f[at + 4], 4);
        begin = at + 8;
        at = begin + len + 4;
        return true;
    }
};
inline int paeth(int left, int up, int corner) {
    const int est = left + up - corner;
    const int dl = std::abs(est - left), du = std::abs(est - up), dc = std::abs(est - corner);
    if (dl <= du && dl <= dc) return left;
    return du <= dc ? up : corner;
}
}  // namespace
bool load_png(const std::string& filename, Image& out) {
    std::ifstream in(filename.c_str(), std::ios::binary);
    if (!in) return false;
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    if (file.size() < 8 || std::memcmp(file.data(), "\x89PNG\r\n\x1a\n", 8) != 0) return false;
    PngReader rd{file};
    std::string type;
    size_t begin = 0, len = 0;
    int channels = 0;
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    bool inflating = false, done = false, ok = true;
    std::vector<uint8_t> rows;
    while (ok && !done && rd.next(type, begin, len)) {
        if (type == "IHDR") {
            if (len != 13 || channels) { ok = false; break; }
            const uint32_t w = rd.word(begin), h = rd.word(begin + 4);
            const int depth = file[begin + 8], colour = file[begin + 9], interlace = file[begin + 12];
            static const int per_colour[7] = {1, 0, 3, 0, 2, 0, 4};
            channels = colour <= 6 ? per_colour[colour] : 0;
            if (!channels || depth != 8 || interlace || !w || !h || w > 16384 || h > 16384) { ok = false; break; }
            out.width = (int)w; out.height = (int)h; out.channels = channels;
            rows.resize(((size_t)w * channels + 1) * h);
            ok = inflateInit(&zs) == Z_OK;
            inflating = ok;
            zs.next_out = rows.data();
            zs.avail_out = (uInt)rows.size();
        } else if (type == "IDAT") {
            if (!inflating) { ok = false; break; }
            zs.next_in = &file[begin];
            zs.avail_in = (uInt)len;
            const int rc = inflate(&zs, Z_NO_FLUSH);
            if (rc == Z_STREAM_END) done = true;
            else if (rc != Z_OK || zs.avail_in) ok = false;          /* more data than the header promised */
        } else if (type == "tRNS") ok = false;
        else if (type == "IEND") break;
    }
    if (inflating) inflateEnd(&zs);
    if (!ok || !done || zs.avail_out) return false;
    const size_t stride = (size_t)out.width * channels;
    out.data.resize(stride * out.height);
    std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < out.height; y++) {
        const uint8_t* src = &rows[(stride + 1) * y];
        uint8_t* dst = &out.data[stride * y];
        const uint8_t* above = y ? dst - stride : zero.data();
        const uint8_t kind = *src++;
        if (kind > 4) return false;
        for (size_t i = 0; i < stride; i++) {
            const bool first = i < (size_t)channels;
            const int left = first ? 0 : dst[i - channels], up = above[i], corner = first ? 0 : above[i - channels];
            const int guess = kind == 0 ? 0 : kind == 1 ? left : kind == 2 ? up : kind == 3 ? (left + up) / 2 : paeth(left, up, corner);
            dst[i] = (uint8_t)(src[i] + guess);
        }
    }
    return true;
}
/* What Material::loadTexture / loadNormalMap get from stbi_load (Material.h:128-138), by file type: PPM and PNG are read
 * here; JPEG decoding is the caller's (its output depends on the decoder): for x.jpg the decoded texels are expected as a
 * binary PPM x.ppm beside it (assets/img/textures/sand.ppm holds what the reference's decoder returns for its sand.jpg). */
bool load_image(const std::string& filename, Image& out) {
    const size_t dot = filename.rfind('.');
    std::string ext = dot == std::string::npos ? std::string() : filename.substr(dot);
    for (char& ch : ext) ch = (char)std::tolower((unsigned char)ch);
    if (ext == ".png") return load_png(filename, out);
    if (ext == ".jpg" || ext == ".jpeg") return load_ppm(filename.substr(0, dot) + ".ppm", out);
    return load_ppm(filename, out);
}

/* ------------------------------------------------------------------ Vec3.h */
float Vec3::squareLength() const { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }          /* Vec3.h:29 */
float Vec3::length() const { return (float)::sqrt((double)squareLength()); }                  /* Vec3.h:32 (double sqrt) */
void Vec3::normalize() { float L = length(); v[0] /= L; v[1] /= L; v[2] /= L; }               /* Vec3.h:37 */
float Vec3::dot(const Vec3& a, const Vec3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
Vec3 Vec3::cross(const Vec3& a, const Vec3& b) {
    return Vec3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
Vec3 operator*(float s, const Vec3& a) { return Vec3(s * a[0], s * a[1], s * a[2]); }
Vec3 operator*(const Vec3& a, float s) { return Vec3(s * a[0], s * a[1], s * a[2]); }
Vec3 operator/(const Vec3& a, float s) { return Vec3(a[0] / s, a[1] / s, a[2] / s); }
Vec3 Mat3::operator*(const Vec3& p) const {                                                    /* Vec3.h:156-163 */
    return Vec3(m[0] * p[0] + m[1] * p[1] + m[2] * p[2],
                m[3] * p[0] + m[4] * p[1] + m[5] * p[2],
                m[6] * p[0] + m[7] * p[1] + m[8] * p[2]);
}

AABB::AABB() : min(INFINITY), max(-INFINITY) {}                                                /* AccelerationStruct.h:29-32 */
Vec3 AABB::getCenter() const { return (min + max) / 2.f; }                                     /* AccelerationStruct.cpp:32-34 */

/* ------------------------------------------------------------------ Mesh */
bool Mesh::loadOFF(const std::string& filename) {
    std::ifstream in(filename.c_str());
    if (!in) return false;                 /* the reference exits the process here (Mesh.cpp:7-8) */
    std::string header;
    unsigned int sizeV = 0, sizeT = 0, tmp = 0;
    in >> header >> sizeV >> sizeT >> tmp;
    /* same read order as Mesh.cpp:10-21; unlike the reference, a file that ends early, does not parse or names a vertex
     * that does not exist is refused (recomputeNormals would write out of bounds) instead of being used as it comes */
    if (!in || sizeV > (1u << 31) || sizeT > (1u << 29)) { positions.clear(); triangles.clear(); return false; }
    {   /* every vertex needs at least 6 bytes of text, every triangle 8: a header that promises more than the file holds is refused before the resize */
        const std::streampos here = in.tellg();
        in.seekg(0, std::ios::end);
        const std::streampos end = in.tellg();
        in.seekg(here);
        if (here < 0 || end < here || (uint64_t)(end - here) < 5ull * sizeV + 7ull * sizeT) return false;
    }
    positions.assign(sizeV, Vec3());
    triangles.assign(3 * (size_t)sizeT, 0u);
    for (unsigned int i = 0; i < sizeV; i++) in >> positions[i][0] >> positions[i][1] >> positions[i][2];
    for (unsigned int i = 0; i < sizeT; i++) {
        int s;
        in >> s;
        for (unsigned int j = 0; j < 3; j++) in >> triangles[3 * (size_t)i + j];
    }
    bool ok = (bool)in;
    for (size_t i = 0; ok && i < triangles.size(); i++) ok = triangles[i] < sizeV;
    if (!ok) { positions.clear(); triangles.clear(); }
    return ok;
}

void Mesh::recomputeNormals() {
    normals.assign(positions.size(), Vec3(0.f, 0.f, 0.f));
    const size_t nt = triangleCount();
    for (size_t t = 0; t < nt; t++) {
        const uint32_t i0 = triangles[3 * t], i1 = triangles[3 * t + 1], i2 = triangles[3 * t + 2];
        Vec3 n = Vec3::cross(positions[i1] - positions[i0], positions[i2] - positions[i0]);
        n.normalize();
        normals[i0] = normals[i0] + n;
        normals[i1] = normals[i1] + n;
        normals[i2] = normals[i2] + n;
    }
    for (Vec3& n : normals) n.normalize();
}

void Mesh::centerAndScaleToUnit() {
    Vec3 c(0.f, 0.f, 0.f);
    for (const Vec3& p : positions) c = c + p;
    c = c / (float)positions.size();
    float maxD = (positions[0] - c).length();
    for (const Vec3& p : positions) {
        float m = (p - c).length();
        if (m > maxD) maxD = m;
    }
    for (Vec3& p : positions) p = (p - c) / maxD;
}

void Mesh::translate(const Vec3& t) { for (Vec3& p : positions) p = p + t; }
void Mesh::apply_transformation_matrix(const Mat3& t) { for (Vec3& p : positions) p = t * p; }
static Mat3 scaleMatrix(const Vec3& s) { return Mat3{{s[0], 0.f, 0.f, 0.f, s[1], 0.f, 0.f, 0.f, s[2]}}; }
/* angle * M_PI / 180. is evaluated in double and narrowed; cos/sin are the double versions (Mesh.h:151-173). */
static void angleCosSin(float angle, float& c, float& s) {
    float a = (float)((double)angle * M_PI / 180.);
    c = (float)::cos((double)a);
    s = (float)::sin((double)a);
}
static Mat3 rotX(float angle) { float c, s; angleCosSin(angle, c, s); return Mat3{{1.f, 0.f, 0.f, 0.f, c, -s, 0.f, s, c}}; }
static Mat3 rotY(float angle) { float c, s; angleCosSin(angle, c, s); return Mat3{{c, 0.f, s, 0.f, 1.f, 0.f, -s, 0.f, c}}; }
static Mat3 rotZ(float angle) { float c, s; angleCosSin(angle, c, s); return Mat3{{c, -s, 0.f, s, c, 0.f, 0.f, 0.f, 1.f}}; }
void Mesh::scale(const Vec3& s) { apply_transformation_matrix(scaleMatrix(s)); }
void Mesh::rotate_x(float angle) { apply_transformation_matrix(rotX(angle)); }
void Mesh::rotate_y(float angle) { apply_transformation_matrix(rotY(angle)); }
void Mesh::rotate_z(float angle) { apply_transformation_matrix(rotZ(angle)); }
void Mesh::buildTree(unsigned nb, int tree_mode) { buildBVH(positions, triangles, (int)nb, triangleTree, tree_mode); }

/* ------------------------------------------------------------------ Square / Sphere */
void Square::setQuad(const Vec3& bottomLeft, const Vec3& rightVector, const Vec3& upVector, float width, float height) {
    Vec3 r = rightVector, u = upVector;          /* Square.h:33-43 */
    r.normalize();
    u.normalize();
    r = r * width;
    u = u * height;
    vertices[0] = bottomLeft;                    /* Square.h:47-50 */
    vertices[1] = bottomLeft + r;
    vertices[2] = bottomLeft + r + u;
    vertices[3] = bottomLeft + u;
}
void Square::translate(const Vec3& t) { for (Vec3& p : vertices) p = p + t; }
void Square::scale(const Vec3& s) { Mat3 m = scaleMatrix(s); for (Vec3& p : vertices) p = m * p; }
void Square::rotate_x(float angle) { Mat3 m = rotX(angle); for (Vec3& p : vertices) p = m * p; }
void Square::rotate_y(float angle) { Mat3 m = rotY(angle); for (Vec3& p : vertices) p = m * p; }
AABB Square::getBoundingBox() const {
    Vec3 mn = vertices[0], mx = vertices[0];
    for (const Vec3& p : vertices)
        for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); }
    return AABB(mn, mx);
}
AABB Sphere::getBoundingBox() const { Vec3 d(m_radius); return AABB(m_center - d, m_center + d); }

rt_camera Camera::pack() const {
    rt_camera c;
    c.x = x; c.y = y; c.z = z; c.zoom = _zoom;
    for (int i = 0; i < 4; i++) c.quat[i] = curquat[i];
    c.fovy_deg = fovAngle; c.aspect = aspectRatio; c.znear = nearPlane; c.zfar = farPlane;
    return c;
}

/* ------------------------------------------------------------------ BVH builders
 * Both builders work in place on one array of build records and emit the flattened
 * tree directly. The reference partitions with std::partition on a random-access
 * range and then copies both halves (AccelerationStruct.cpp:164-170,210-216); the
 * element order after that call is what fixes leaf order, so `split` below performs
 * the same sequence of swaps (scan from the left for the first element failing the
 * predicate, from the right for the last one passing it, swap, repeat). */
template <class Rec, class Pred>
static size_t split(Rec* a, size_t lo, size_t hi, Pred pred) {
    size_t first = lo, last = hi;
    for (;;) {
        for (;;) {
            if (first == last) return first;
            if (pred(a[first])) ++first; else break;
        }
        --last;
        for (;;) {
            if (first == last) return first;
            if (!pred(a[last])) --last; else break;
        }
        std::swap(a[first], a[last]);
        ++first;
    }
}

struct BuildRec { uint32_t index; float c[3]; float mn[3], mx[3]; };

static AABB unionBox(const BuildRec* a, size_t lo, size_t hi) {      /* getBoundingBox, AccelerationStruct.cpp:57-69,77-89 */
    AABB g(Vec3(a[lo].mn[0], a[lo].mn[1], a[lo].mn[2]), Vec3(a[lo].mx[0], a[lo].mx[1], a[lo].mx[2]));
    for (size_t i = lo; i < hi; i++)
        for (int k = 0; k < 3; k++) {
            g.min[k] = std::min(g.min[k], a[i].mn[k]);
            g.max[k] = std::max(g.max[k], a[i].mx[k]);
        }
    return g;
}

static int32_t emitLeaf(FlatTree& t, const BuildRec* a, size_t lo, size_t hi) {
    rt_bvh_leaf lf;
    lf.first = (uint32_t)t.items.size();
    lf.count = (uint32_t)(hi - lo);
    for (size_t i = lo; i < hi; i++) t.items.push_back(a[i].index);
    t.leaves.push_back(lf);
    return -(int32_t)t.leaves.size();            /* link = -1 - leafIndex */
}
static void storeBoxes(rt_bvh_node& n, const AABB& l, const AABB& r) {
    for (int k = 0; k < 3; k++) { n.lmin[k] = l.min[k]; n.lmax[k] = l.max[k]; n.rmin[k] = r.min[k]; n.rmax[k] = r.max[k]; }
}

/* Mode 2, NOT the reference's tree (a free-topology builder): binned surface-area heuristic over the
 * centroids, 16 bins per axis, the cheapest of the 45 candidate planes; same record array, same flattened output, leaf
 * when n <= 2. The render loop walks such a tree like any other, but which triangles a ray can REACH then differs from
 * the reference's (a triangle is only found along its own box chain, and the records are the triangles scaled by
 * 1.0001 with a tolerance on top), so hit identities may differ in a few grazing pixels: a non-parity mode, off by default. */
static bool sahSplit(const BuildRec* a, size_t lo, size_t hi, int& axis_out, float& pivot_out) {
    constexpr int NB = 16;
    float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = lo; i < hi; i++)
        for (int k = 0; k < 3; k++) { cmn[k] = std::min(cmn[k], a[i].c[k]); cmx[k] = std::max(cmx[k], a[i].c[k]); }
    double best = INFINITY;
    bool found = false;
    for (int k = 0; k < 3; k++) {
        const float ext = cmx[k] - cmn[k];
        if (!(ext > 0.f) || !(ext < 1e30f)) continue;
        float bmn[NB][3], bmx[NB][3];
        size_t cnt[NB];
        for (int b = 0; b < NB; b++) { cnt[b] = 0; for (int j = 0; j < 3; j++) { bmn[b][j] = INFINITY; bmx[b][j] = -INFINITY; } }
        const float scale = (float)NB / ext;
        for (size_t i = lo; i < hi; i++) {
            int b = (int)((a[i].c[k] - cmn[k]) * scale);
            b = std::min(std::max(b, 0), NB - 1);
            cnt[b]++;
            for (int j = 0; j < 3; j++) { bmn[b][j] = std::min(bmn[b][j], a[i].mn[j]); bmx[b][j] = std::max(bmx[b][j], a[i].mx[j]); }
        }
        auto area = [](const float* mn, const float* mx) {
            if (!(mn[0] <= mx[0])) return 0.0;
            const double x = (double)mx[0] - mn[0], y = (double)mx[1] - mn[1], z = (double)mx[2] - mn[2];
            return x * y + y * z + z * x;
        };
        double rarea[NB]; size_t rcnt[NB];
        float amn[3] = {INFINITY, INFINITY, INFINITY}, amx[3] = {-INFINITY, -INFINITY, -INFINITY};
        size_t n = 0;
        for (int b = NB - 1; b >= 1; b--) {
            for (int j = 0; j < 3; j++) { amn[j] = std::min(amn[j], bmn[b][j]); amx[j] = std::max(amx[j], bmx[b][j]); }
            n += cnt[b];
            rarea[b] = area(amn, amx); rcnt[b] = n;
        }
        for (int j = 0; j < 3; j++) { amn[j] = INFINITY; amx[j] = -INFINITY; }
        n = 0;
        for (int b = 0; b + 1 < NB; b++) {
            for (int j = 0; j < 3; j++) { amn[j] = std::min(amn[j], bmn[b][j]); amx[j] = std::max(amx[j], bmx[b][j]); }
            n += cnt[b];
            if (n == 0 || rcnt[b + 1] == 0) continue;
            const double cost = area(amn, amx) * (double)n + rarea[b + 1] * (double)rcnt[b + 1];
            if (cost < best) { best = cost; axis_out = k; pivot_out = cmn[k] + ext * (float)(b + 1) / (float)NB; found = true; }
        }
    }
    return found;
}

/* mode 0: triangle tree (longest axis, leaf when n<=1); mode 1: object tree (shared axis
 * counter, leaf when n<=2); mode 2: triangle tree by binned SAH (sahSplit). Returns the link of the subtree. */
static int32_t buildNode(FlatTree& t, BuildRec* a, size_t lo, size_t hi, int deepLimit, const AABB& box, int mode,
                         int& axisCounter, int level) {
    const size_t n = hi - lo;
    const size_t leafMax = mode == 0 ? 1 : 2;
    if (!(deepLimit > 0 && n > leafMax)) {
        t.depth = std::max(t.depth, level);
        return emitLeaf(t, a, lo, hi);
    }
    Vec3 diag = box.max - box.min;
    int axis = 0;
    float pivot = 0.f;
    if (mode == 2) {
        if (!sahSplit(a, lo, hi, axis, pivot)) {                                       /* all centroids in one point: halve by index */
            const size_t half = lo + n / 2;
            const int32_t me2 = (int32_t)t.nodes.size();
            t.nodes.emplace_back();
            std::memset(&t.nodes[me2], 0, sizeof(rt_bvh_node));
            AABB lb2 = unionBox(a, lo, half), rb2 = unionBox(a, half, hi);
            const int32_t l2 = buildNode(t, a, lo, half, deepLimit - 1, lb2, mode, axisCounter, level + 1);
            const int32_t r2 = buildNode(t, a, half, hi, deepLimit - 1, rb2, mode, axisCounter, level + 1);
            storeBoxes(t.nodes[me2], lb2, rb2);
            t.nodes[me2].left = l2; t.nodes[me2].right = r2;
            return me2;
        }
    } else {
        if (mode == 0) {
            axis = diag[0] > std::max(diag[1], diag[2]) ? 0 : diag[1] > diag[2] ? 1 : 2;   /* AccelerationStruct.cpp:160 */
        } else {
            axisCounter %= 3;                                                              /* AccelerationStruct.cpp:206 */
            axis = axisCounter;
        }
        Vec3 center = box.min + diag / 2.f;
        pivot = center[axis];
    }
    size_t mid = split(a, lo, hi, [axis, pivot](const BuildRec& r) { return r.c[axis] < pivot; });
    if (mode == 1) axisCounter++;                                                      /* AccelerationStruct.cpp:222 */
    const int32_t me = (int32_t)t.nodes.size();
    t.nodes.emplace_back();
    std::memset(&t.nodes[me], 0, sizeof(rt_bvh_node));
    AABB lb, rb;                                  /* empty halves keep the inverted +-INF box */
    int32_t l, r;
    if (mid - lo >= 1) { lb = unionBox(a, lo, mid); l = buildNode(t, a, lo, mid, deepLimit - 1, lb, mode, axisCounter, level + 1); }
    else l = buildNode(t, a, lo, lo, 0, lb, mode, axisCounter, level + 1);
    if (hi - mid >= 1) { rb = unionBox(a, mid, hi); r = buildNode(t, a, mid, hi, deepLimit - 1, rb, mode, axisCounter, level + 1); }
    else r = buildNode(t, a, mid, mid, 0, rb, mode, axisCounter, level + 1);
    storeBoxes(t.nodes[me], lb, rb);
    t.nodes[me].left = l;
    t.nodes[me].right = r;
    return me;
}

/* The triangle tree of a big mesh on several threads, same result: the two halves of a node are independent once the
 * node's partition is done, so large halves are built into fragments of their own (left half on a new thread) and
 * appended in the order the sequential recursion emits them -- node, left subtree, right subtree -- with the links
 * moved by the fragment's place. */
static void appendFragment(FlatTree& t, const FlatTree& f, int32_t& link) {
    const int32_t node0 = (int32_t)t.nodes.size(), leaf0 = (int32_t)t.leaves.size();
    const uint32_t item0 = (uint32_t)t.items.size();
    auto moved = [&](int32_t l) { return l >= 0 ? l + node0 : l - leaf0; };
    for (rt_bvh_node n : f.nodes) { n.left = moved(n.left); n.right = moved(n.right); t.nodes.push_back(n); }
    for (rt_bvh_leaf lf : f.leaves) { lf.first += item0; t.leaves.push_back(lf); }
    t.items.insert(t.items.end(), f.items.begin(), f.items.end());
    t.depth = std::max(t.depth, f.depth);
    link = moved(link);
}
static int32_t buildNodeParallel(FlatTree& t, BuildRec* a, size_t lo, size_t hi, int deepLimit, const AABB& box, int level, int threadsLeft, int mode = 0) {
    const size_t n = hi - lo;
    int unused = 0;
    if (threadsLeft <= 1 || n < 200000 || !(deepLimit > 0 && n > 1)) return buildNode(t, a, lo, hi, deepLimit, box, mode, unused, level);
    Vec3 diag = box.max - box.min;
    int axis = diag[0] > std::max(diag[1], diag[2]) ? 0 : diag[1] > diag[2] ? 1 : 2;       /* AccelerationStruct.cpp:160 */
    Vec3 center = box.min + diag / 2.f;
    float pivot = center[axis];
    if (mode == 2 && !sahSplit(a, lo, hi, axis, pivot)) return buildNode(t, a, lo, hi, deepLimit, box, mode, unused, level);
    const size_t mid = split(a, lo, hi, [axis, pivot](const BuildRec& r) { return r.c[axis] < pivot; });
    const int32_t me = (int32_t)t.nodes.size();
    t.nodes.emplace_back();
    std::memset(&t.nodes[me], 0, sizeof(rt_bvh_node));
    AABB lb, rb;
    FlatTree lf, rf;
    for (FlatTree* f : {&lf, &rf}) {        /* one allocation each (growing a vector of this size step by step is mostly page faults) */
        const size_t cnt = f == &lf ? mid - lo : hi - mid;
        f->nodes.reserve(cnt); f->leaves.reserve(cnt + 1); f->items.reserve(cnt);
    }
    int32_t l = 0, r = 0;
    std::thread left([&] {
        if (mid - lo >= 1) { lb = unionBox(a, lo, mid); l = buildNodeParallel(lf, a, lo, mid, deepLimit - 1, lb, level + 1, threadsLeft / 2, mode); }
        else { int u = 0; l = buildNode(lf, a, lo, lo, 0, lb, 0, u, level + 1); }
    });
    if (hi - mid >= 1) { rb = unionBox(a, mid, hi); r = buildNodeParallel(rf, a, mid, hi, deepLimit - 1, rb, level + 1, threadsLeft - threadsLeft / 2, mode); }
    else { int u = 0; r = buildNode(rf, a, mid, mid, 0, rb, 0, u, level + 1); }
    left.join();
    appendFragment(t, lf, l);
    appendFragment(t, rf, r);
    storeBoxes(t.nodes[me], lb, rb);
    t.nodes[me].left = l;
    t.nodes[me].right = r;
    return me;
}

void buildBVH(const std::vector<Vec3>& positions, const std::vector<uint32_t>& triangles, int deepLimit, FlatTree& out, int tree_mode) {
    out = FlatTree();
    const size_t nt = triangles.size() / 3;
    std::vector<BuildRec> recs(nt);
    for (size_t i = 0; i < nt; i++) {             /* BuildingTriangle, AccelerationStruct.cpp:36-55 */
        const Vec3 &a = positions[triangles[3 * i]], &b = positions[triangles[3 * i + 1]], &c = positions[triangles[3 * i + 2]];
        Vec3 avg = (a + b + c) / 3.f;
        BuildRec& r = recs[i];
        r.index = (uint32_t)i;
        for (int k = 0; k < 3; k++) {
            r.c[k] = avg[k];
            float mn = a[k], mx = a[k];
            mn = std::min(b[k], mn); mx = std::max(b[k], mx);
            mn = std::min(c[k], mn); mx = std::max(c[k], mx);
            r.mn[k] = mn; r.mx[k] = mx;
        }
    }
    AABB box;                                     /* AccelerationStruct.cpp:103-110 */
    for (const Vec3& p : positions)
        for (int k = 0; k < 3; k++) {
            box.min[k] = p[k] < box.min[k] ? p[k] : box.min[k];
            box.max[k] = p[k] > box.max[k] ? p[k] : box.max[k];
        }
    out.rootBox = box;
    if (tree_mode == 2 && nt >= 2) {
        /* Mode 2 of Settings::tree_mode, NOT the reference's tree either: a linear BVH built on the GPU (lbvh.hip; SURVEY
         * 8(f).2, "BVH construction on GPU"), one triangle per leaf, same flattened layout. A non-parity mode, off by default. */
        static_assert(sizeof(Vec3) == 12, "positions are handed to the device as packed floats");
        out.nodes.resize(nt - 1); out.leaves.resize(nt); out.items.resize(nt);
        char err[256];
        double dev_s = 0.0;
        const auto T0 = std::chrono::steady_clock::now();
        const int rc = rtgpu_lbvh_build(reinterpret_cast<const float*>(positions.data()), (int64_t)positions.size(), triangles.data(), (int64_t)nt, out.nodes.data(),
                                        out.leaves.data(), out.items.data(), &out.root, &out.depth, &dev_s, err, sizeof err);
        if (rc != RT_OK) throw BuildError{rc, err};
        if (getenv("RTH_TIMING")) fprintf(stderr, "[rthost] GPU-built tree of %zu triangles: %.3f s on the device, %.3f s with copies; depth %d\n", nt, dev_s,
                                          std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count(), out.depth);
        return;
    }
    out.items.reserve(nt); out.nodes.reserve(nt); out.leaves.reserve(nt + 1);
    const auto T0 = std::chrono::steady_clock::now();
    int threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    if (const char* e = getenv("OMP_NUM_THREADS")) threads = std::max(1, std::min(threads, atoi(e)));
    out.root = buildNodeParallel(out, recs.data(), 0, nt, deepLimit, box, 0, threads, tree_mode == 1 ? 2 : 0);
    if (getenv("RTH_TIMING")) fprintf(stderr, "[rthost] tree of %zu triangles, depth limit %d, %d threads: %.3f s\n", nt, deepLimit, threads, std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count());
}

void buildBVH(const std::vector<AABB>& objectBoxes, int deepLimit, FlatTree& out) {
    out = FlatTree();
    std::vector<BuildRec> recs(objectBoxes.size());
    AABB box;
    for (size_t i = 0; i < objectBoxes.size(); i++) {
        const AABB& b = objectBoxes[i];
        Vec3 c = b.getCenter();                   /* predicate of AccelerationStruct.cpp:212 */
        recs[i].index = (uint32_t)i;
        for (int k = 0; k < 3; k++) {
            recs[i].c[k] = c[k]; recs[i].mn[k] = b.min[k]; recs[i].mx[k] = b.max[k];
            box.min[k] = std::min(box.min[k], b.min[k]);                               /* AccelerationStruct.cpp:138-145 */
            box.max[k] = std::max(box.max[k], b.max[k]);
        }
    }
    out.rootBox = box;
    int axisCounter = 0;                          /* comparisonIdx = 0, AccelerationStruct.h:122 */
    out.root = buildNode(out, recs.data(), 0, recs.size(), deepLimit, box, 1, axisCounter, 0);
}

/* ------------------------------------------------------------------ Scene */
void Scene::clear() { meshes.clear(); spheres.clear(); squares.clear(); lights.clear(); textures.clear(); sceneBvh = FlatTree(); }
int Scene::loadImage(const std::string& relative_path) {
    Image img;
    if (!load_image(g_image_dir + "/" + relative_path, img)) return -1;
    textures.push_back(std::move(img));
    return (int)textures.size() - 1;
}
int Scene::addImage(int width, int height, int channels, const uint8_t* texels) {
    if (width <= 0 || height <= 0 || channels < 3 || !texels) return -1;
    Image img;
    img.width = width; img.height = height; img.channels = channels;
    img.data.assign(texels, texels + (size_t)width * height * channels);
    textures.push_back(std::move(img));
    return (int)textures.size() - 1;
}
void Scene::applySettings(Settings set) { for (Mesh& m : meshes) m.buildTree(set.tree_subdivide, set.tree_mode); }
void Scene::buildSceneBvh() {
    std::vector<AABB> boxes;                      /* squares, spheres, meshes: AccelerationStruct.cpp:123-134 */
    for (const Square& s : squares) boxes.push_back(s.getBoundingBox());
    for (const Sphere& s : spheres) boxes.push_back(s.getBoundingBox());
    for (const Mesh& m : meshes) boxes.push_back(m.getBoundingBox());
    buildBVH(boxes, topDepthLimit, sceneBvh);
}

static Light pointLight(float x, float y, float z, float radius) {
    Light l; l.pos = Vec3(x, y, z); l.radius = radius; l.material = Vec3(1.f, 1.f, 1.f); return l;
}
/* The floor quad every scene uses: setQuad, translate(0,0,-2), scale(sx,sy,1), rotate_x(-90). */
static Square floorQuad(float sx, float sy) {
    Square s;
    s.setQuad(Vec3(-1.f, -1.f, 0.f), Vec3(1.f, 0.f, 0.f), Vec3(0.f, 1.f, 0.f), 2.f, 2.f);
    s.translate(Vec3(0.f, 0.f, -2.f));
    s.scale(Vec3(sx, sy, 1.f));
    s.rotate_x(-90.f);
    return s;
}
static void setColour(Material& m, float r, float g, float b, double shininess) {
    m.diffuse_material = Vec3(r, g, b); m.specular_material = Vec3(r, g, b); m.shininess = shininess;
}

bool Scene::setup_refraction_scene() {
    clear();
    lights.push_back(pointLight(0.0f, 3.5f, 0.0f, .5f));
    Square s = floorQuad(80.f, 80.f);
    setColour(s.material, 0.75f, 1.f, 0.75f, 16);
    s.material.type = RT_MAT_CHECKER;
    squares.push_back(s);
    Sphere g;
    g.m_center = Vec3(0.f, 0.f, 0.5f); g.m_radius = 0.75f;
    g.material.type = RT_MAT_GLASS;
    setColour(g.material, 1.f, 0.f, 0.f, 16);
    g.material.transparency = 0.1f; g.material.index_medium = 1.4f;
    spheres.push_back(g);
    buildSceneBvh();
    return true;
}
bool Scene::setup_reflexion_scene() {
    clear();
    lights.push_back(pointLight(0.0f, 3.5f, 0.0f, .5f));
    Square s = floorQuad(80.f, 80.f);
    setColour(s.material, 0.75f, 1.f, 0.75f, 16);
    s.material.type = RT_MAT_CHECKER;
    squares.push_back(s);
    Sphere g;
    g.m_center = Vec3(0.f, -1.25f, 0.5f); g.m_radius = 0.75f;
    g.material.type = RT_MAT_MIRROR;
    setColour(g.material, 0.75f, 1.f, 1.f, 16);
    g.material.transparency = 1.0f; g.material.index_medium = 1.4f; g.material.reflectivity = 0.5f;
    spheres.push_back(g);
    buildSceneBvh();
    return true;
}
bool Scene::setup_mesh_scene(const std::string& off, bool unit) {
    clear();
    lights.push_back(pointLight(1.0f, 2.5f, 0.0f, .5f));           /* Scene.h:722-731 */
    Square f = floorQuad(2.f, 2.f);                                 /* Scene.h:733-744 */
    setColour(f.material, 1.f, 1.f, 1.f, 16);
    squares.push_back(f);
    Square w;                                                       /* Scene.h:746-756 */
    w.setQuad(Vec3(-1.f, -1.f, 0.f), Vec3(1.f, 0.f, 0.f), Vec3(0.f, 1.f, 0.f), 2.f, 2.f);
    w.scale(Vec3(2.f, 2.f, 1.f));
    w.translate(Vec3(0.f, 0.f, -2.f));
    setColour(w.material, 0.5f, 0.f, 0.5f, 16);
    squares.push_back(w);
    meshes.emplace_back();                                          /* Scene.h:768-781 */
    Mesh& m = meshes.back();
    if (!m.loadOFF(g_asset_dir + "/" + off)) return false;
    if (unit) m.centerAndScaleToUnit(); else m.scale(Vec3(0.0125f));
    m.translate(Vec3(0.f, -1.05f, 0.f));
    m.build_arrays();
    setColour(m.material, 1.f, 1.f, 1.f, 16);
    m.buildTree(3);
    buildSceneBvh();
    return true;
}

/* The scenes of the reference's stage images etapes/normal_interpolation/{with,without}.ppm and etapes/mesh/gorilla_1.ppm:
 * the room of setup_test_scene (Scene.h:733-756) around one white mesh. Light position (that of Scene.h:477), hard shadows
 * and the blob's transformation are not in the reference any more; they were recovered from the images
 * (tests/golden/make_fixtures.py). `flat`: per-triangle vertices, i.e. face normals (without.ppm predates Mesh.cpp:117-130). */
void Mesh::unweld() {
    std::vector<Vec3> pos(triangles.size());
    for (size_t i = 0; i < triangles.size(); i++) { pos[i] = positions[triangles[i]]; triangles[i] = (uint32_t)i; }
    positions.swap(pos);
}
bool Scene::setup_stage_scene(const std::string& off, bool flat, float scale, float rotx, const Vec3& move) {
    if (!setup_mesh_scene(off, false)) return false;        /* (the room; the mesh is replaced below) */
    lights[0] = pointLight(0.0f, 3.5f, 0.0f, 0.f);
    Mesh& m = meshes.back();
    if (!m.loadOFF(g_asset_dir + "/" + off)) return false;
    if (scale != 1.f) m.scale(Vec3(scale));
    if (rotx != 0.f) m.rotate_x(rotx);
    m.translate(move);
    if (flat) m.unweld();
    m.build_arrays();
    m.buildTree(3);
    buildSceneBvh();
    return true;
}

/* Scene::setup_reflexion_chamber, Scene.h:601-714 */
bool Scene::setup_reflexion_chamber() {
    clear();
    lights.push_back(pointLight(1.f, 1.5f, 1.f, .1f));
    lights.push_back(pointLight(-1.f, 1.5f, -1.f, .1f));
    for (int i = 0; i < 6; i++) {                                   /* Scene.h:630-639 */
        Square s;
        s.setQuad(Vec3(-1.f, -1.f, 0.f), Vec3(1.f, 0.f, 0.f), Vec3(0.f, 1.f, 0.f), 2.f, 2.f);
        s.scale(Vec3(2.f, 2.f, 1.f));
        s.translate(Vec3(0.f, 0.f, -2.f));
        setColour(s.material, 1.f, 1.f, 1.f, 16);
        squares.push_back(s);
    }
    auto mirror = [](Square& s) { s.material.type = RT_MAT_MIRROR; s.material.reflectivity = 0.9f; };
    { Square& s = squares[0]; s.material.materialScale = 4; mirror(s); }                                              /* back wall */
    { Square& s = squares[1]; s.rotate_y(90); setColour(s.material, 1.f, .75f, .75f, 16); mirror(s); }                /* left wall */
    { Square& s = squares[2]; s.rotate_y(-90); setColour(s.material, .75f, 1.f, .75f, 16); mirror(s); }               /* right wall */
    { Square& s = squares[3]; s.rotate_x(-90); s.material.type = RT_MAT_CHECKER;                                      /* floor */
      s.material.normalMap = loadImage("normalMaps/n3.ppm"); if (s.material.normalMap < 0) return false; }
    { Square& s = squares[4]; s.rotate_x(90); mirror(s); }                                                            /* ceiling */
    { Square& s = squares[5]; s.rotate_y(180); mirror(s); }                                                           /* front wall */
    meshes.emplace_back();                                          /* Scene.h:698-711 */
    Mesh& m = meshes.back();
    if (!m.loadOFF(g_asset_dir + "/flamingo.off")) return false;
    m.scale(Vec3(0.0125f));
    m.translate(Vec3(0.f, -1.55f, 0.f));
    m.build_arrays();
    setColour(m.material, 1.f, 0.753f, 0.796f, 16);
    m.material.transparency = 1.0f; m.material.index_medium = 1.2f;
    m.buildTree(3);
    buildSceneBvh();
    return true;
}

/* Scene::setup_cornell_box, Scene.h:288-466: scene 2 of main.cpp:390 and the scene of the reference's committed rendu.ppm.
 * textures/cliff.jpg, textures/stone_floor.jpg and normalMaps/sand.png are not in the reference checkout: there those
 * loads fail and the material stays plain (Material.h:128-138), so a missing one is not an error here either. */
bool Scene::setup_cornell_box() {
    clear();
    { Light l = pointLight(-1.5f, -0.75f, 1.25f, .5f); l.material = Vec3(1.f, 0.5f, 0.75f); lights.push_back(l); }   /* Scene.h:305-314 */
    auto quad = [](bool scaleThenTranslate) {
        Square s;
        s.setQuad(Vec3(-1.f, -1.f, 0.f), Vec3(1.f, 0.f, 0.f), Vec3(0.f, 1.f, 0.f), 2.f, 2.f);
        if (scaleThenTranslate) { s.scale(Vec3(2.f, 2.f, 1.f)); s.translate(Vec3(0.f, 0.f, -2.f)); }
        else { s.translate(Vec3(0.f, 0.f, -2.f)); s.scale(Vec3(2.f, 2.f, 1.f)); }
        setColour(s.material, 1.f, 1.f, 1.f, 16);
        return s;
    };
    bool found = true;
    auto must = [&](const char* rel) { const int t = loadImage(rel); found = found && t >= 0; return t; };
    { Square s = quad(true);                                             /* back wall, Scene.h:316-328 */
      s.material.normalMap = must("normalMaps/n4.ppm"); s.material.materialScale = 4; squares.push_back(s); }
    { Square s = quad(true); s.rotate_y(90); setColour(s.material, 1.f, 0.f, 0.f, 16);                     /* left wall, 330-344 */
      s.material.normalMap = must("normalMaps/cliff.png"); s.material.texture = loadImage("textures/cliff.jpg"); squares.push_back(s); }
    { Square s = quad(false); s.rotate_y(-90); setColour(s.material, 0.f, 1.f, 0.f, 16);                   /* right wall, 346-359 */
      s.material.texture = must("sphereTextures/s5.ppm"); squares.push_back(s); }
    { Square s = quad(false); s.rotate_x(-90);                                                             /* floor, 361-375 */
      s.material.normalMap = must("normalMaps/stone_floor.png"); s.material.texture = loadImage("textures/stone_floor.jpg"); squares.push_back(s); }
    { Square s = quad(false); s.rotate_x(90);                                                              /* ceiling, 377-391 */
      s.material.normalMap = loadImage("normalMaps/sand.png"); s.material.texture = must("textures/sand.jpg"); squares.push_back(s); }
    { Square s = quad(false); s.rotate_y(180); s.material.texture = must("sphereTextures/s3.ppm"); squares.push_back(s); }   /* front wall, 393-405 */
    { Sphere g; g.m_center = Vec3(1.0f, -1.25f, 0.5f); g.m_radius = 0.75f; g.material.type = RT_MAT_GLASS;                  /* 408-420 */
      setColour(g.material, 1.f, 0.f, 0.f, 16); g.material.transparency = 0.1f; g.material.index_medium = 1.2f; spheres.push_back(g); }
    { Sphere g; g.m_center = Vec3(-1.0f, -1.25f, -0.5f); g.m_radius = 0.75f; g.material.type = RT_MAT_MIRROR;               /* 422-435 */
      setColour(g.material, 0.75f, 1.f, 1.f, 16); g.material.transparency = 1.0f; g.material.index_medium = 1.4f;
      g.material.reflectivity = 0.5f; spheres.push_back(g); }
    { Sphere g; g.m_center = Vec3(-0.75f, 0.f, -0.75f); g.m_radius = 0.5f; setColour(g.material, 1.f, 1.f, 1.f, 10);         /* 437-448 */
      g.material.texture = must("sphereTextures/s4.ppm"); g.material.materialScale = 2; spheres.push_back(g); }
    if (!found) return false;
    buildSceneBvh();
    return true;
}

static float rand01() { return (float)rand() / (float)RAND_MAX; }

bool Scene::setup_final_scene() {
    clear();
    srand(1);     /* main() sets this scene up first, from glibc's default seed (main.cpp:387-393) */
    static const float nice[10][2][3] = {
        {{1.0f, 0.0f, 0.0f}, {0.5f, 0.0f, 0.0f}}, {{0.0f, 1.0f, 0.0f}, {0.0f, 0.5f, 0.0f}},
        {{0.0f, 0.0f, 1.0f}, {0.0f, 0.0f, 0.5f}}, {{1.0f, 1.0f, 0.0f}, {0.5f, 0.5f, 0.0f}},
        {{1.0f, 0.5f, 0.0f}, {0.5f, 0.25f, 0.0f}}, {{0.5f, 0.0f, 1.0f}, {0.25f, 0.0f, 0.5f}},
        {{0.0f, 1.0f, 1.0f}, {0.0f, 0.5f, 0.5f}}, {{1.0f, 0.0f, 1.0f}, {0.5f, 0.0f, 0.5f}},
        {{0.5f, 0.5f, 0.5f}, {0.25f, 0.25f, 0.25f}}, {{1.0f, 1.0f, 1.0f}, {0.5f, 0.5f, 0.5f}}};
    lights.push_back(pointLight(4.0f, 0.f, 10.f, .5f));
    lights.push_back(pointLight(-6.f, 2.f, 1.25f, 10.75f));
    {
        Square s = floorQuad(20.f, 20.f);
        setColour(s.material, 1.f, 1.f, 1.f, 16);
        s.material.type = RT_MAT_CHECKER;
        squares.push_back(s);
    }
    {
        meshes.emplace_back();
        Mesh& m = meshes.back();
        if (!m.loadOFF(g_asset_dir + "/gorilla18_fixed.obj.off")) return false;
        m.scale(Vec3(0.04f));
        m.translate(Vec3(-5.f, -1.9f, 0.f));
        m.build_arrays();
        m.material.type = RT_MAT_GLASS;
        setColour(m.material, 1.f, 0.f, 0.f, 16);
        m.material.transparency = 1.0f; m.material.index_medium = 1.4f;
    }
    {
        Sphere s;
        s.m_center = Vec3(-4.f, 0.f, 7.f); s.m_radius = 1.75f;
        s.material.diffuse_material = Vec3(nice[0][0][0], nice[0][0][1], nice[0][0][2]);
        s.material.specular_material = Vec3(nice[0][1][0], nice[0][1][1], nice[0][1][2]);
        s.material.type = RT_MAT_GLASS; s.material.transparency = 0.8f; s.material.index_medium = 1.3f;
        spheres.push_back(s);
    }
    {
        Sphere s;
        s.m_center = Vec3(0.f); s.m_radius = 3.75f;
        s.material.type = RT_MAT_MIRROR;
        s.material.diffuse_material = Vec3(1.f, 1.f, 1.f); s.material.specular_material = Vec3(1.f, 1.f, 1.f);
        s.material.reflectivity = 0.9f;
        spheres.push_back(s);
    }
    Mesh flamingo;
    if (!flamingo.loadOFF(g_asset_dir + "/flamingo.off")) return false;
    for (int i = 0; i < 30; i++) {
        meshes.push_back(flamingo);
        Mesh& m = meshes.back();
        float randZ = (float)((double)rand01() * 50. - 25.);
        m.scale(Vec3(0.0125f));
        m.translate(Vec3(0.f, -2.f, randZ));
        m.rotate_y((float)((double)rand01() * 360.));
        m.build_arrays();
        setColour(m.material, 1.f, 0.753f, 0.796f, 16);
    }
    for (int i = 0; i < 50; i++) {
        Sphere s;
        s.m_radius = (float)((double)rand01() + 0.5);
        float randZ = (float)((double)rand01() * 50. - 25.);
        float randX = (float)((double)rand01() * 50. - 25.);
        float randY = (float)((double)rand01() * 20. - 2.);
        s.m_center = Vec3(randX, randY, randZ);
        rand(); rand(); rand();                                     /* the unused randColor, Scene.h:947 */
        int c = rand() % 10;
        s.material.diffuse_material = Vec3(nice[c][0][0], nice[c][0][1], nice[c][0][2]);
        s.material.specular_material = Vec3(nice[c][1][0], nice[c][1][1], nice[c][1][2]);
        s.material.shininess = rand() % 16;
        int randType = rand() % 3;
        if (randType == 0) {
            s.material.type = RT_MAT_GLASS;
            s.material.transparency = (float)(rand() / RAND_MAX);   /* integer division as written, Scene.h:959 */
        } else if (randType == 1) {
            s.material.type = RT_MAT_MIRROR;
            s.material.reflectivity = (float)(rand() / RAND_MAX * 0.5 + 0.5);
            s.material.transparency = (float)(rand() / RAND_MAX * 0.5 + 0.5);
        }
        spheres.push_back(s);
    }
    for (Mesh& m : meshes) m.buildTree(3);                          /* build_arrays -> buildTree(), Mesh.h:216-219 */
    buildSceneBvh();
    return true;
}

/* ------------------------------------------------------------------ scene description files (rthost.h) */
bool Mesh::loadOFFColoured(const std::string& filename) {
    std::ifstream in(filename.c_str());
    if (!in) return false;
    std::string header;
    size_t nv = 0, nf = 0, ne = 0;
    in >> header >> nv >> nf >> ne;
    if (!in || (header != "OFF" && header != "COFF")) return false;
    const bool coloured = header == "COFF";
    positions.assign(nv, Vec3());
    triangles.clear();
    for (size_t i = 0; i < nv; i++) {
        in >> positions[i][0] >> positions[i][1] >> positions[i][2];
        if (coloured) { float c; in >> c >> c >> c >> c; }
    }
    for (size_t f = 0; f < nf; f++) {
        size_t k = 0;
        in >> k;
        if (!in || k < 3) return false;
        std::vector<uint32_t> idx(k);
        for (size_t j = 0; j < k; j++) { in >> idx[j]; if (idx[j] >= nv) return false; }
        for (size_t j = 1; j + 1 < k; j++) { triangles.push_back(idx[0]); triangles.push_back(idx[j]); triangles.push_back(idx[j + 1]); }
        std::string rest;
        std::getline(in, rest);                    /* optional face colour */
    }
    return (bool)in || in.eof();
}

static void writeMaterial(FILE* f, const Material& m) {
    fprintf(f, " material %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g  %.17g %d %.9g %.9g %.9g %d %d %.17g\n",
            m.ambient_material[0], m.ambient_material[1], m.ambient_material[2], m.diffuse_material[0], m.diffuse_material[1], m.diffuse_material[2],
            m.specular_material[0], m.specular_material[1], m.specular_material[2], m.shininess, m.type, m.index_medium, m.transparency,
            m.reflectivity, m.texture, m.normalMap, m.materialScale);
}
static bool readMaterial(std::istringstream& in, Material& m, size_t n_textures) {
    std::string word;
    in >> word;
    if (word != "material") return false;
    for (int k = 0; k < 3; k++) in >> m.ambient_material[k];
    for (int k = 0; k < 3; k++) in >> m.diffuse_material[k];
    for (int k = 0; k < 3; k++) in >> m.specular_material[k];
    in >> m.shininess >> m.type >> m.index_medium >> m.transparency >> m.reflectivity >> m.texture >> m.normalMap >> m.materialScale;
    if (!in) return false;
    return m.texture >= -1 && m.texture < (int)n_textures && m.normalMap >= -1 && m.normalMap < (int)n_textures && m.type >= 0 && m.type <= 3;
}

bool Scene::save(const std::string& filename) const {
    FILE* f = fopen(filename.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "rtscene 1\n");
    for (const Image& im : textures) {             /* first: materials refer to them by index */
        fprintf(f, "texture %d %d %d\n", im.width, im.height, im.channels);
        fwrite(im.data.data(), 1, im.data.size(), f);
        fputc('\n', f);
    }
    for (const Light& l : lights)
        fprintf(f, "light %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", l.pos[0], l.pos[1], l.pos[2], l.radius, l.material[0], l.material[1], l.material[2]);
    for (const Square& q : squares) {
        fprintf(f, "square");
        for (int v = 0; v < 4; v++) fprintf(f, " %.9g %.9g %.9g", q.vertices[v][0], q.vertices[v][1], q.vertices[v][2]);
        writeMaterial(f, q.material);
    }
    for (const Sphere& sp : spheres) {
        fprintf(f, "sphere %.9g %.9g %.9g %.9g", sp.m_center[0], sp.m_center[1], sp.m_center[2], sp.m_radius);
        writeMaterial(f, sp.material);
    }
    for (const Mesh& m : meshes) {
        fprintf(f, "mesh %zu %zu", m.positions.size(), m.triangleCount());
        writeMaterial(f, m.material);
        fwrite(m.positions.data(), sizeof(Vec3), m.positions.size(), f);
        fwrite(m.triangles.data(), sizeof(uint32_t), m.triangles.size(), f);
        fputc('\n', f);
    }
    fprintf(f, "end\n");
    return fclose(f) == 0;
}

bool Scene::load(const std::string& filename, std::string* why) {
    auto fail = [&](const std::string& what) { if (why) *why = what; clear(); return false; };
    std::ifstream in(filename.c_str(), std::ios::binary);
    if (!in) return fail("cannot open " + filename);
    clear();
    std::string line;
    if (!std::getline(in, line) || line.rfind("rtscene 1", 0) != 0) return fail("not an rtscene 1 file");
    Mesh* last = nullptr;                          /* the mesh the transformation lines apply to */
    bool ended = false;
    int lineno = 1;
    auto block = [&](void* dst, size_t bytes) {    /* a binary block follows its line and ends with a newline */
        in.read(static_cast<char*>(dst), (std::streamsize)bytes);
        return (size_t)in.gcount() == bytes && in.get() == '\n';
    };
    while (!ended && std::getline(in, line)) {
        lineno++;
        std::istringstream ls(line);
        std::string kind;
        ls >> kind;
        const std::string at = " (line " + std::to_string(lineno) + ")";
        if (kind.empty() || kind[0] == '#') continue;
        if (kind == "end") ended = true;
        else if (kind == "texture") {
            Image im;
            std::string path;
            ls >> path;
            if (!path.empty() && (path[0] < '0' || path[0] > '9')) {          /* texture <file.ppm under the image directory> */
                if (loadImage(path) < 0) return fail("cannot read image " + path + at);
                continue;
            }
            std::istringstream dims(line);
            dims >> kind >> im.width >> im.height >> im.channels;
            if (!dims || im.width <= 0 || im.height <= 0 || im.channels < 3 || im.channels > 4 || (int64_t)im.width * im.height > (1ll << 28)) return fail("bad texture header" + at);
            {
                const std::streampos here = in.tellg();
                in.seekg(0, std::ios::end);
                const std::streampos end = in.tellg();
                in.seekg(here);
                if (here < 0 || end < here || (uint64_t)(end - here) < (uint64_t)im.width * im.height * im.channels) return fail("truncated texture" + at);
            }
            im.data.resize((size_t)im.width * im.height * im.channels);
            if (!block(im.data.data(), im.data.size())) return fail("truncated texture" + at);
            textures.push_back(std::move(im));
        } else if (kind == "light") {
            Light l;
            ls >> l.pos[0] >> l.pos[1] >> l.pos[2] >> l.radius >> l.material[0] >> l.material[1] >> l.material[2];
            if (!ls) return fail("bad light" + at);
            lights.push_back(l);
        } else if (kind == "square") {
            Square q;
            for (int v = 0; v < 4; v++) ls >> q.vertices[v][0] >> q.vertices[v][1] >> q.vertices[v][2];
            if (!ls || !readMaterial(ls, q.material, textures.size())) return fail("bad square" + at);
            squares.push_back(q);
        } else if (kind == "sphere") {
            Sphere sp;
            ls >> sp.m_center[0] >> sp.m_center[1] >> sp.m_center[2] >> sp.m_radius;
            if (!ls || !readMaterial(ls, sp.material, textures.size())) return fail("bad sphere" + at);
            spheres.push_back(sp);
        } else if (kind == "mesh") {
            size_t nv = 0, nt = 0;
            ls >> nv >> nt;
            Material mat;
            if (!ls || nv == 0 || nt == 0 || nv > (1ull << 31) || nt > (1ull << 29) || !readMaterial(ls, mat, textures.size())) return fail("bad mesh header" + at);
            {   /* the block must be in the file before anything is sized by an untrusted header */
                const std::streampos here = in.tellg();
                in.seekg(0, std::ios::end);
                const std::streampos end = in.tellg();
                in.seekg(here);
                if (here < 0 || end < here || (uint64_t)(end - here) < nv * sizeof(Vec3) + 3 * nt * sizeof(uint32_t)) return fail("truncated mesh" + at);
            }
            meshes.emplace_back();
            Mesh& m = meshes.back();
            m.material = mat;
            m.positions.resize(nv);
            m.triangles.resize(3 * nt);
            in.read(reinterpret_cast<char*>(m.positions.data()), (std::streamsize)(nv * sizeof(Vec3)));      /* positions and triangles are one block */
            if ((size_t)in.gcount() != nv * sizeof(Vec3) || !block(m.triangles.data(), 3 * nt * sizeof(uint32_t))) return fail("truncated mesh" + at);
            for (uint32_t i : m.triangles) if (i >= nv) return fail("triangle index out of range" + at);
            m.build_arrays();
            last = &m;
        } else if (kind == "mesh_off") {
            std::string path;
            ls >> path;
            Material mat;
            if (!ls || !readMaterial(ls, mat, textures.size())) return fail("bad mesh_off" + at);
            meshes.emplace_back();
            Mesh& m = meshes.back();
            m.material = mat;
            if (!m.loadOFFColoured(g_asset_dir + "/" + path) || m.positions.empty() || m.triangles.empty()) return fail("cannot read OFF file " + path + at);
            m.build_arrays();
            last = &m;
        } else if (kind == "center_unit" || kind == "scale" || kind == "translate" || kind == "rotate_x" || kind == "rotate_y" || kind == "rotate_z") {
            if (!last) return fail(kind + " without a mesh" + at);
            float a = 0.f, b = 0.f, c = 0.f;
            if (kind == "center_unit") last->centerAndScaleToUnit();
            else if (kind == "scale" || kind == "translate") {
                ls >> a >> b >> c;
                if (!ls) return fail("bad " + kind + at);
                if (kind == "scale") last->scale(Vec3(a, b, c)); else last->translate(Vec3(a, b, c));
            } else {
                ls >> a;
                if (!ls) return fail("bad " + kind + at);
                if (kind == "rotate_x") last->rotate_x(a); else if (kind == "rotate_y") last->rotate_y(a); else last->rotate_z(a);
            }
            last->build_arrays();                   /* normals follow the positions, as after every setup_* block */
        } else return fail("unknown entry '" + kind + "'" + at);
    }
    if (!ended) return fail("no end line: truncated file");
    return true;
}

/* Counter RNG of rtgpu (rt_rng) drives the synthetic soup so that both sides of the
 * boundary can generate it without files. */
static float unit24(uint32_t h) { return (float)(h >> 8) * (1.0f / 16777216.0f); }
bool Scene::setup_soup_scene(int64_t ntris) {
    clear();
    lights.push_back(pointLight(0.f, 8.f, 0.f, .5f));
    Square f = floorQuad(2.f, 2.f);
    setColour(f.material, 1.f, 1.f, 1.f, 16);
    squares.push_back(f);
    meshes.emplace_back();
    Mesh& m = meshes.back();
    m.positions.resize((size_t)ntris * 3);
    m.triangles.resize((size_t)ntris * 3);
    for (int64_t k = 0; k < ntris; k++) {
        float r[9];
        for (uint32_t j = 0; j < 9; j++) r[j] = unit24(rt_rng(1u, (uint32_t)k, (uint32_t)(k >> 32), 0, 0xFFFFFFFEu, j, 0));
        Vec3 c(-8.f + 16.f * r[0], -2.f + 8.f * r[1], -8.f + 16.f * r[2]);
        Vec3 e1(-0.02f + 0.04f * r[3], -0.02f + 0.04f * r[4], -0.02f + 0.04f * r[5]);
        Vec3 e2(-0.02f + 0.04f * r[6], -0.02f + 0.04f * r[7], -0.02f + 0.04f * r[8]);
        m.positions[3 * k] = c;
        m.positions[3 * k + 1] = c + e1;
        m.positions[3 * k + 2] = c + e2;
        for (int j = 0; j < 3; j++) m.triangles[3 * k + j] = (uint32_t)(3 * k + j);
    }
    m.build_arrays();
    setColour(m.material, 0.8f, 0.8f, 0.8f, 16);
    m.buildTree(3);
    buildSceneBvh();
    return true;
}

bool Scene::setup(const std::string& name) {
    if (name == "final") return setup_final_scene();
    if (name == "refraction") return setup_refraction_scene();
    if (name == "reflexion") return setup_reflexion_scene();
    if (name == "mesh_gorilla") return setup_mesh_scene("gorilla18_fixed.obj.off", false);
    if (name == "mesh_cow") return setup_mesh_scene("cow.off", true);
    if (name == "stage_blob") return setup_stage_scene("blob.off", false, 1.f, 180.f, Vec3(-0.125f, -1.05f, -1.25f));
    if (name == "stage_blob_flat") return setup_stage_scene("blob.off", true, 1.f, 180.f, Vec3(-0.125f, -1.05f, -1.25f));
    if (name == "stage_gorilla") return setup_stage_scene("gorilla18_fixed.obj.off", false, 0.0125f, 0.f, Vec3(0.f, -1.05f, 0.f));
    if (name == "reflexion_chamber") return setup_reflexion_chamber();
    if (name == "cornell_box") return setup_cornell_box();
    if (name.rfind("soup:", 0) == 0) return setup_soup_scene(atoll(name.c_str() + 5));
    return false;
}

static int32_t addMaterial(std::vector<rt_material>& out, const Material& m) {
    rt_material r;
    std::memset(&r, 0, sizeof(r));
    for (int k = 0; k < 3; k++) { r.ka[k] = m.ambient_material[k]; r.kd[k] = m.diffuse_material[k]; r.ks[k] = m.specular_material[k]; }
    r.shininess = m.shininess; r.type = m.type; r.ior = m.index_medium; r.transparency = m.transparency; r.reflectivity = m.reflectivity;
    r.texture = m.texture + 1; r.normal_map = m.normalMap + 1; r.material_scale = m.materialScale;
    out.push_back(r);
    return (int32_t)out.size() - 1;
}

const rt_scene_desc& Scene::flatten() {
    f_materials.clear(); f_lights.clear(); f_squares.clear(); f_spheres.clear(); f_meshes.clear();
    f_positions.clear(); f_normals.clear(); f_tris.clear(); f_items.clear(); f_nodes.clear(); f_leaves.clear();
    for (const Light& l : lights) {
        rt_light r;
        std::memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) { r.pos[k] = l.pos[k]; r.color[k] = l.material[k]; }
        r.radius = l.radius;
        f_lights.push_back(r);
    }
    for (const Square& s : squares) {
        rt_square r;
        for (int k = 0; k < 3; k++) { r.v0[k] = s.vertices[0][k]; r.v1[k] = s.vertices[1][k]; r.v3[k] = s.vertices[3][k]; }
        r.material = addMaterial(f_materials, s.material);
        f_squares.push_back(r);
    }
    for (const Sphere& s : spheres) {
        rt_sphere r;
        std::memset(&r, 0, sizeof(r));
        for (int k = 0; k < 3; k++) r.center[k] = s.m_center[k];
        r.radius = s.m_radius;
        r.material = addMaterial(f_materials, s.material);
        f_spheres.push_back(r);
    }
    size_t nv = 0, nt = 0, nn = 0, nl = 0;
    for (const Mesh& m : meshes) { nv += m.positions.size(); nt += m.triangleCount(); nn += m.triangleTree.nodes.size(); nl += m.triangleTree.leaves.size(); }
    f_positions.reserve(3 * nv); f_normals.reserve(3 * nv); f_tris.reserve(3 * nt); f_items.reserve(nt);
    f_nodes.reserve(nn); f_leaves.reserve(nl);
    for (const Mesh& m : meshes) {
        rt_mesh r;
        std::memset(&r, 0, sizeof(r));
        r.vert_first = (int64_t)f_positions.size() / 3; r.vert_count = (int64_t)m.positions.size();
        r.tri_first = (int64_t)f_tris.size() / 3; r.tri_count = (int64_t)m.triangleCount();
        r.node_first = (int64_t)f_nodes.size(); r.node_count = (int64_t)m.triangleTree.nodes.size();
        r.leaf_first = (int64_t)f_leaves.size(); r.leaf_count = (int64_t)m.triangleTree.leaves.size();
        r.root = m.triangleTree.root; r.depth = m.triangleTree.depth;
        r.material = addMaterial(f_materials, m.material);
        for (size_t i = 0; i < m.positions.size(); i++)
            for (int k = 0; k < 3; k++) { f_positions.push_back(m.positions[i][k]); f_normals.push_back(m.normals[i][k]); }
        f_tris.insert(f_tris.end(), m.triangles.begin(), m.triangles.end());
        f_items.insert(f_items.end(), m.triangleTree.items.begin(), m.triangleTree.items.end());
        f_nodes.insert(f_nodes.end(), m.triangleTree.nodes.begin(), m.triangleTree.nodes.end());
        f_leaves.insert(f_leaves.end(), m.triangleTree.leaves.begin(), m.triangleTree.leaves.end());
        f_meshes.push_back(r);
    }
    rt_scene_desc& d = f_desc;
    std::memset(&d, 0, sizeof(d));
    f_textures.clear();
    for (const Image& img : textures) {
        rt_texture t;
        std::memset(&t, 0, sizeof(t));
        t.width = img.width; t.height = img.height; t.channels = img.channels; t.texels = img.data.data();
        f_textures.push_back(t);
    }
    d.n_textures = (int32_t)f_textures.size(); d.textures = f_textures.empty() ? nullptr : f_textures.data();
    d.n_materials = (int32_t)f_materials.size(); d.materials = f_materials.data();
    d.n_lights = (int32_t)f_lights.size(); d.lights = f_lights.data();
    d.n_squares = (int32_t)f_squares.size(); d.squares = f_squares.data();
    d.n_spheres = (int32_t)f_spheres.size(); d.spheres = f_spheres.data();
    d.n_meshes = (int32_t)f_meshes.size(); d.meshes = f_meshes.data();
    d.n_verts = (int64_t)nv; d.positions = f_positions.data(); d.normals = f_normals.data();
    d.n_tris = (int64_t)nt; d.tri_indices = f_tris.data(); d.mesh_items = f_items.data();
    d.n_mesh_nodes = (int64_t)f_nodes.size(); d.mesh_nodes = f_nodes.data();
    d.n_mesh_leaves = (int64_t)f_leaves.size(); d.mesh_leaves = f_leaves.data();
    d.n_top_nodes = (int32_t)sceneBvh.nodes.size(); d.top_nodes = sceneBvh.nodes.data();
    d.n_top_leaves = (int32_t)sceneBvh.leaves.size(); d.top_leaves = sceneBvh.leaves.data();
    d.n_top_items = (int32_t)sceneBvh.items.size(); d.top_items = sceneBvh.items.data();
    d.top_root = sceneBvh.root; d.top_depth = sceneBvh.depth;
    return d;
}

/* ------------------------------------------------------------------ render entry point */
int write_ppm(const char* filename, const float* rgb, int w, int h) {
    std::ofstream f(filename, std::ios::binary);
    if (f.fail()) {
        std::printf("Could not open file: %s\n", filename);
        return RT_ERR_ARG;
    }
    f << "P3" << std::endl << w << " " << h << std::endl << 255 << std::endl;
    for (int i = 0; i < w * h; i++)
        f << (int)(255.f * std::min<float>(1.f, rgb[3 * i])) << " " << (int)(255.f * std::min<float>(1.f, rgb[3 * i + 1])) << " "
          << (int)(255.f * std::min<float>(1.f, rgb[3 * i + 2])) << " ";
    f << std::endl;
    f.close();
    return RT_OK;
}

int ray_trace_from_camera(rt_ctx* ctx, Scene& scene, const Camera& camera, int w, int h, const rt_render_params* prm,
                          const char* filename, double* seconds) {
    (void)scene;   /* the scene was handed over with rt_upload_scene(scene.flatten()) */
    const auto startTime = std::chrono::high_resolution_clock::now();
    std::printf("Ray tracing a %d x %d image on the GPU\n", w, h);
    rt_render_params p;
    if (prm) p = *prm;
    else { p.img_w = w; p.img_h = h; p.spp = 30; p.max_depth = 50; p.light_samples = 35; p.shadow_mode = 0; p.jitter = 1; p.seed = 0; }
    p.img_w = w; p.img_h = h;
    rt_camera cam = camera.pack();
    int rc;
    if ((rc = rt_set_camera(ctx, &cam)) != RT_OK) return rc;
    if ((rc = rt_begin_frame(ctx, &p)) != RT_OK) return rc;
    const int T = 32;
    for (int y = 0; y < h; y += T)
        for (int x = 0; x < w; x += T)
            if ((rc = rt_render_tile(ctx, x, y, std::min(T, w - x), std::min(T, h - y))) != RT_OK) return rc;
    std::vector<float> image((size_t)w * h * 3);
    if ((rc = rt_download_framebuffer(ctx, image.data(), nullptr, nullptr)) != RT_OK) return rc;
    const auto stopTime = std::chrono::high_resolution_clock::now();
    double secs = std::chrono::duration<double>(stopTime - startTime).count();
    if (seconds) *seconds = secs;
    std::printf("Done in: %.3fs\n", secs);
    if (filename) return write_ppm(filename, image.data(), w, h);
    return RT_OK;
}

}  // namespace rth

/* ====================================================================== C facade */
struct rth_scene { rth::Scene scene; rth::Settings settings; };

/* No exception crosses the C-ABI (include/rtgpu.h): every body that can throw runs under this guard. */
template <typename F>
static int guarded(F body) {
    try { return body(); }
    catch (const rth::BuildError& e) { fprintf(stderr, "[rthost] %s\n", e.message.c_str()); return e.code; }
    catch (const std::bad_alloc&) { return RT_ERR_NOMEM; }
    catch (...) { return RT_ERR_ARG; }
}

extern "C" {
rth_scene* rth_scene_new(void) { try { return new rth_scene(); } catch (...) { return nullptr; } }
void rth_scene_free(rth_scene* s) { delete s; }
void rth_set_asset_dir(const char* dir) { rth::set_asset_dir(dir); }
void rth_set_image_dir(const char* dir) { rth::set_image_dir(dir); }
int rth_add_texture(rth_scene* s, int width, int height, int channels, const uint8_t* texels) {
    if (!s || !texels) return RT_ERR_ARG;
    return guarded([&] {
        int idx = s->scene.addImage(width, height, channels, texels);
        return idx < 0 ? (int)RT_ERR_ARG : idx + 1;
    });
}
int rth_scene_setup(rth_scene* s, const char* name) {
    if (!s || !name) return RT_ERR_ARG;
    return guarded([&] { return s->scene.setup(name) ? (int)RT_OK : (int)RT_ERR_ARG; });
}
int rth_add_light(rth_scene* s, const float pos[3], float radius, const float color[3]) {
    if (!s || !pos || !color) return RT_ERR_ARG;
    return guarded([&] {
        rth::Light l; l.pos = rth::Vec3(pos[0], pos[1], pos[2]); l.radius = radius; l.material = rth::Vec3(color[0], color[1], color[2]);
        s->scene.lights.push_back(l);
        return (int)RT_OK;
    });
}
static rth::Material fromC(const rt_material* m) {
    rth::Material r;
    if (!m) return r;
    r.ambient_material = rth::Vec3(m->ka[0], m->ka[1], m->ka[2]);
    r.diffuse_material = rth::Vec3(m->kd[0], m->kd[1], m->kd[2]);
    r.specular_material = rth::Vec3(m->ks[0], m->ks[1], m->ks[2]);
    r.shininess = m->shininess; r.type = m->type; r.index_medium = m->ior; r.transparency = m->transparency; r.reflectivity = m->reflectivity;
    r.texture = m->texture - 1; r.normalMap = m->normal_map - 1; r.materialScale = m->material_scale == 0.0 ? 1.0 : m->material_scale;
    return r;
}
static bool imagesOk(const rth_scene* s, const rth::Material& m) {
    const int n = (int)s->scene.textures.size();
    return m.texture >= -1 && m.texture < n && m.normalMap >= -1 && m.normalMap < n;
}
int rth_add_square(rth_scene* s, const float verts[12], const rt_material* m) {
    if (!s || !verts || !imagesOk(s, fromC(m))) return RT_ERR_ARG;
    return guarded([&] {
        rth::Square q;
        for (int j = 0; j < 4; j++) q.vertices[j] = rth::Vec3(verts[3 * j], verts[3 * j + 1], verts[3 * j + 2]);
        q.material = fromC(m);
        s->scene.squares.push_back(q);
        return (int)RT_OK;
    });
}
int rth_add_sphere(rth_scene* s, const float c[3], float radius, const rt_material* m) {
    if (!s || !c || !imagesOk(s, fromC(m))) return RT_ERR_ARG;
    return guarded([&] {
        rth::Sphere sp; sp.m_center = rth::Vec3(c[0], c[1], c[2]); sp.m_radius = radius; sp.material = fromC(m);
        s->scene.spheres.push_back(sp);
        return (int)RT_OK;
    });
}
int rth_add_mesh(rth_scene* s, const float* verts, int64_t nv, const uint32_t* tris, int64_t nt, const rt_material* m) {
    if (!s || !verts || !tris || nv <= 0 || nt <= 0 || nv > (1ll << 31) || nt > (1ll << 29) || !imagesOk(s, fromC(m))) return RT_ERR_ARG;
    for (int64_t i = 0; i < 3 * nt; i++) if (tris[i] >= (uint64_t)nv) return RT_ERR_ARG;
    return guarded([&] {
        rth::Mesh ms;
        ms.positions.resize((size_t)nv);
        for (int64_t i = 0; i < nv; i++) ms.positions[i] = rth::Vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
        ms.triangles.assign(tris, tris + 3 * nt);
        ms.build_arrays();
        ms.material = fromC(m);
        s->scene.meshes.push_back(std::move(ms));
        return (int)RT_OK;
    });
}
int rth_build(rth_scene* s, int top_depth, int mesh_depth) { return rth_build_mode(s, top_depth, mesh_depth, 0); }
int rth_build_mode(rth_scene* s, int top_depth, int mesh_depth, int tree_mode) {
    if (!s || top_depth < 0 || mesh_depth < 0 || tree_mode < 0 || tree_mode > 2) return RT_ERR_ARG;
    if (top_depth > RT_MAX_TOP_DEPTH || mesh_depth > RT_MAX_TREE_DEPTH) return RT_ERR_LIMIT;
    return guarded([&] {
        s->settings.tree_subdivide = (unsigned)mesh_depth;
        s->settings.tree_mode = tree_mode;
        s->scene.applySettings(s->settings);
        s->scene.topDepthLimit = top_depth;
        s->scene.buildSceneBvh();
        s->scene.flatten();
        return (int)RT_OK;
    });
}
const rt_scene_desc* rth_scene_desc(rth_scene* s) {
    if (!s) return nullptr;
    try { return &s->scene.flatten(); } catch (...) { return nullptr; }
}
int rth_ray_trace_from_camera(rt_ctx* ctx, rth_scene* s, const rt_camera* cam, const rt_render_params* p, const char* path, double* seconds) {
    if (!ctx || !s || !cam || !p) return RT_ERR_ARG;
    rth::Camera c;
    c.x = cam->x; c.y = cam->y; c.z = cam->z; c._zoom = cam->zoom;
    for (int i = 0; i < 4; i++) c.curquat[i] = cam->quat[i];
    c.fovAngle = cam->fovy_deg; c.aspectRatio = cam->aspect; c.nearPlane = cam->znear; c.farPlane = cam->zfar;
    return guarded([&] { return rth::ray_trace_from_camera(ctx, s->scene, c, p->img_w, p->img_h, p, path, seconds); });
}
int rth_write_ppm(const char* path, const float* rgb, int w, int h) {
    if (!path || !rgb || w <= 0 || h <= 0) return RT_ERR_ARG;
    return guarded([&] { return rth::write_ppm(path, rgb, w, h); });
}
int rth_scene_save(rth_scene* s, const char* path) {
    if (!s || !path) return RT_ERR_ARG;
    return guarded([&] { return s->scene.save(path) ? (int)RT_OK : (int)RT_ERR_ARG; });
}
int rth_scene_load(rth_scene* s, const char* path) {
    if (!s || !path) return RT_ERR_ARG;
    return guarded([&] {
        std::string why;
        if (!s->scene.load(path, &why)) { fprintf(stderr, "[rthost] %s\n", why.c_str()); return (int)RT_ERR_ARG; }
        return (int)RT_OK;
    });
}
int rth_load_off(const char* path, float* verts, int64_t max_v, uint32_t* tris, int64_t max_t, int64_t* nv, int64_t* nt) {
    if (!path) return RT_ERR_ARG;
    return guarded([&] {
        rth::Mesh m;
        if (!m.loadOFF(path)) return (int)RT_ERR_ARG;
        if (nv) *nv = (int64_t)m.positions.size();
        if (nt) *nt = (int64_t)m.triangleCount();
        if (verts) for (int64_t i = 0; i < (int64_t)m.positions.size() && i < max_v; i++) for (int k = 0; k < 3; k++) verts[3 * i + k] = m.positions[i][k];
        if (tris) for (int64_t i = 0; i < (int64_t)m.triangleCount() && i < max_t; i++) for (int k = 0; k < 3; k++) tris[3 * i + k] = m.triangles[3 * i + k];
        return (int)RT_OK;
    });
}
}
