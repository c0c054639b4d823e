/*
 * ref_like.h -- TEST SCAFFOLDING for integration/rt_from_reference.h: stand-in types that carry the member names the
 * binding reads from the reference's classes (Scene.h:43-49, Mesh.h:26-63,103-111,210-213, Sphere.h:39-40, Square.h:47-50,
 * Scene.h:20-34, Material.h:18-38, AccelerationStruct.h:11-18,88-106, Camera.h:52-64) and nothing else of them -- no methods,
 * no algorithms. The private members are private here too, with the same `friend struct rtref::Access;` line the patch
 * adds to the reference, so the test also proves that this one line is enough. Filled from an rth::Scene by the test.
 */
#ifndef REF_LIKE_H
#define REF_LIKE_H
#include <cstddef>
#include <memory>
#include <vector>

namespace rtref { struct Access; }

namespace reflike {

struct Vec3 {
    float mVals[3] = {0.f, 0.f, 0.f};
    float& operator[](unsigned i) { return mVals[i]; }
    float operator[](unsigned i) const { return mVals[i]; }
};
enum MaterialType { Material_Diffuse_Blinn_Phong, Material_Glass, Material_Mirror, Material_Checker };
struct Material {
    Vec3 ambient_material, diffuse_material, specular_material;
    double shininess = 0.0;
    unsigned char* textureData = nullptr;
    unsigned char* normalMapData = nullptr;
    int textureWidth = 0, textureHeight = 0, textureChannels = 0;
    int normalMapWidth = 0, normalMapHeight = 0, normalMapChannels = 0;
    bool hasTexture = false, hasNormalMap = false;
    double materialScale = 1.;
    float index_medium = 1.f, transparency = 0.f, reflectivity = 0.f;
    MaterialType type = Material_Diffuse_Blinn_Phong;
};
struct MeshVertex { Vec3 position, normal; float u = 0.f, v = 0.f; };
struct MeshTriangle { unsigned int v[3] = {0, 0, 0}; };
struct AABB { Vec3 min, max; };
struct BuildingMesh { AABB boundingBox; void* ref = nullptr; int type = 0; };
struct BVH_Node {
    bool leaf = true;
    std::unique_ptr<BVH_Node> l_child, r_child;
    std::vector<size_t> trianglesIdx;
    std::vector<BuildingMesh> meshes;
    AABB boundingBox;
};
class Mesh {
public:
    std::vector<MeshVertex> vertices;
    std::vector<MeshTriangle> triangles;
    Material material;
};
class RayTraceMesh : public Mesh {
    std::unique_ptr<BVH_Node> triangleTree;
    friend struct rtref::Access;
public:
    void set_tree(std::unique_ptr<BVH_Node> t) { triangleTree = std::move(t); }      /* (the reference: buildTree) */
};
class Sphere : public Mesh { public: Vec3 m_center; float m_radius = 0.f; };
class Square : public Mesh {};
struct Light { Vec3 material; Vec3 pos; float radius = 0.f; };
class Scene {
    std::vector<RayTraceMesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<Square> squares;
    BVH_Node sceneBvh;
    std::vector<Light> lights;
    friend struct rtref::Access;
    friend struct Filler;                                /* (the reference: its setup_* member functions) */
};
class Camera {
    float fovAngle = 45.f, aspectRatio = 1.f, nearPlane = 4.1f, farPlane = 10000.f;
    float curquat[4] = {0.f, 0.f, 0.f, 1.f};
    float x = 0.f, y = 0.f, z = 0.f;
    float _zoom = 3.f;
    friend struct rtref::Access;
    friend struct Filler;
};

}  // namespace reflike
#endif
