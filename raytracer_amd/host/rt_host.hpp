// rt_host.hpp -- C++ host layer above the C ABI (include/rtx_capi.h), shaped like the
// reference's Raytracer.cpp so that its driver code ports line by line:
//
//   reference                                   here
//   Material / Surface (Material.h:4-15)        rt::Material / rt::Surface (same layout)
//   Triangle / Mesh (Obj.h:96-200)              rt::Triangle / rt::Mesh (add_triangle, add_material)
//   Sphere (Obj.h:249-256)                      rt::Sphere
//   Box (commented, Obj.h:35-93), Plane         rt::Box, rt::Plane
//   Scene::load_mesh / addMesh / commit         rt::Scene::load_mesh / addMesh / commit
//     (Scene.h:55-142)
//   Img (Raytracer.cpp:47-144)                  rt::Img (pixels, sample_count, clear/get/set, save*)
//   render(img, pathTrace) (Raytracer.cpp:236)  rt::render(img, scene, spp)
//
// Plain C++17, no HIP headers: all device work happens inside librtx_hip.so.
// Errors are reported by exceptions carrying rtx_last_error() (the reference
// calls exit(1), Scene.h:66).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtx_capi.h"

namespace rt {

struct Vec3 {
  float x = 0, y = 0, z = 0;
  Vec3() = default;
  explicit Vec3(float s) : x(s), y(s), z(s) {}
  Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};

enum Surface { diffuse = RTX_DIFFUSE, specular = RTX_SPECULAR, reflective = RTX_REFLECTIVE, varnish = RTX_VARNISH };

struct Material {  // Material.h:11-15, identical to rtx_material
  Vec3 albedo;
  float emission = 0;
  int32_t surface = diffuse;
};
static_assert(sizeof(Material) == sizeof(rtx_material), "Material must match rtx_material");

struct Triangle {
  Vec3 a, b, c;
};

struct Sphere {  // Obj.h:249-256
  Vec3 c;
  float r = 0;
  Material material;
  Sphere(Vec3 c_, float r_) : c(c_), r(r_) {}
};
static_assert(sizeof(Sphere) == sizeof(rtx_sphere), "Sphere must match rtx_sphere");

struct Box {  // the reference's commented-out class Box (Obj.h:35-93) + its material
  Vec3 min, max;
  Material material;
  Box(Vec3 mn, Vec3 mx) : min(mn), max(mx) {}
};
static_assert(sizeof(Box) == sizeof(rtx_box), "Box must match rtx_box");

struct Plane {  // Plane(point, normal) of the reference's scenes (Raytracer.cpp:184-221)
  Vec3 point, normal;
  Material material;
  Plane(Vec3 p, Vec3 n) : point(p), normal(n) {}
};
static_assert(sizeof(Plane) == sizeof(rtx_plane), "Plane must match rtx_plane");

struct Mesh {  // Obj.h:163-182
  std::vector<Triangle> triangles;
  std::vector<unsigned> material_idxs;
  std::vector<Material> materials;
  void add_triangle(Triangle t, unsigned material_id) {
    triangles.push_back(t);
    material_idxs.push_back(material_id);
  }
  void add_material(Material m) { materials.push_back(m); }
};

class Error : public std::runtime_error {
 public:
  explicit Error(const std::string& what) : std::runtime_error(what) {}
};
inline void check(int status, const char* what) {
  if (status != RTX_OK) throw Error(std::string(what) + ": " + rtx_last_error());
}

// The constants of globals.h / Integrator.h:12 / Raytracer.cpp:145 as defaults.
inline rtx_config default_config(uint32_t width = 800, uint32_t height = 800) {
  rtx_config c{};
  c.width = width;
  c.height = height;
  c.fov = 50.0f;
  c.cam_origin[0] = 0.0f;
  c.cam_origin[1] = 1.0f;
  c.cam_origin[2] = 3.0f;
  c.eps = 1.0e-3f;
  c.nee_threshold = 0.05f;
  c.rr_cap = 0.99f;
  c.focal_length = 2.4f;
  c.lens_radius = 0.09f;
  c.stretch_x = c.stretch_y = 1.2f;
  c.device = -1;
  return c;
}

namespace detail {
// Index of an OBJ "v" reference: 1-based, negative = relative to the end.
inline bool obj_index(const std::string& tok, size_t nverts, size_t& out) {
  long v = std::strtol(tok.c_str(), nullptr, 10);  // stops at '/', so v, v/vt, v//vn, v/vt/vn all work
  if (v > 0 && (size_t)v <= nverts) {
    out = (size_t)v - 1;
    return true;
  }
  if (v < 0 && (size_t)(-v) <= nverts) {
    out = nverts - (size_t)(-v);
    return true;
  }
  return false;
}
inline std::vector<std::string> mtl_names(const std::string& path) {
  std::vector<std::string> names;
  std::ifstream f(path);
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream ss(line);
    std::string key, name;
    if ((ss >> key >> name) && key == "newmtl") names.push_back(name);
  }
  return names;
}
}  // namespace detail

// The OBJ front end of Scene::load_mesh (Scene.h:55-115): reads a Wavefront OBJ; one Material per material of
// its MTL library, initialised to the reference's placeholder (0.65,0.05,0.05) diffuse
// (Scene.h:78) for the caller to override by slot; every face is fan-tessellated around
// its first vertex (Scene.h:88-109).  Faces without a material (the reference would index
// with -1 cast to unsigned, Scene.h:107) get one extra placeholder slot at the end.
inline Mesh* load_obj(const std::string& file_name) {
  std::ifstream f(file_name);
  if (!f) throw Error("load_mesh: cannot open " + file_name);
  Mesh* mesh = new Mesh();
  std::vector<Vec3> verts;
  std::map<std::string, unsigned> slot_of;
  std::string dir = file_name.substr(0, file_name.find_last_of("/\\") + 1);
  const Material placeholder{Vec3(0.65f, 0.05f, 0.05f), 0.0f, diffuse};
  int current = -1;
  bool need_default = false;
  std::vector<std::pair<size_t, int>> tri_slots;  // (triangle, material slot or -1)
  std::string line;
  size_t lineno = 0;
  while (std::getline(f, line)) {
    lineno++;
    std::istringstream ss(line);
    std::string key;
    if (!(ss >> key) || key[0] == '#') continue;
    if (key == "v") {
      Vec3 v;
      if (!(ss >> v.x >> v.y >> v.z)) {
        delete mesh;
        throw Error("load_mesh: bad vertex at line " + std::to_string(lineno));
      }
      verts.push_back(v);
    } else if (key == "mtllib") {
      std::string name;
      ss >> name;
      for (const std::string& n : detail::mtl_names(dir + name))
        if (!slot_of.count(n)) {
          slot_of[n] = (unsigned)mesh->materials.size();
          mesh->add_material(placeholder);
        }
    } else if (key == "usemtl") {
      std::string name;
      ss >> name;
      auto it = slot_of.find(name);
      current = it == slot_of.end() ? -1 : (int)it->second;
    } else if (key == "f") {
      std::vector<size_t> idx;
      std::string tok;
      while (ss >> tok) {
        size_t i;
        if (!detail::obj_index(tok, verts.size(), i)) {
          delete mesh;
          throw Error("load_mesh: bad face index at line " + std::to_string(lineno));
        }
        idx.push_back(i);
      }
      for (size_t v = 2; v < idx.size(); v++) {  // fan around the first vertex, b = previous c
        tri_slots.emplace_back(mesh->triangles.size(), current);
        mesh->add_triangle(Triangle{verts[idx[0]], verts[idx[v - 1]], verts[idx[v]]}, 0);
        if (current < 0) need_default = true;
      }
    }
  }
  unsigned default_slot = 0;
  if (need_default) {
    default_slot = (unsigned)mesh->materials.size();
    mesh->add_material(placeholder);
  }
  for (auto& ts : tri_slots) mesh->material_idxs[ts.first] = ts.second < 0 ? default_slot : (unsigned)ts.second;
  return mesh;
}


class Scene {
 public:
  std::vector<Mesh*> meshes;
  std::vector<Sphere> spheres;
  std::vector<Box> boxes;
  std::vector<Plane> planes;

  explicit Scene(const rtx_config& cfg) : cfg_(cfg) { check(rtx_create(&cfg_, &ctx_), "rtx_create"); }
  ~Scene() {
    if (accum_) rtx_accum_free(ctx_, accum_);
    rtx_destroy(ctx_);
    for (Mesh* m : meshes) delete m;
  }
  Scene(const Scene&) = delete;
  Scene& operator=(const Scene&) = delete;

  // Scene::load_mesh (Scene.h:55-115); see load_obj.
  Mesh* load_mesh(const std::string& file_name) { return load_obj(file_name); }

  void addMesh(Mesh* mesh) { meshes.push_back(mesh); }  // Scene.h:117-120

  // Scene::commit (Scene.h:122-142): hands the geometry to the library, which builds the
  // BVH and the light list.  Several meshes are concatenated in order.
  void commit() {
    std::vector<float> tris;
    std::vector<uint32_t> midx;
    std::vector<Material> mats;
    for (const Mesh* m : meshes) {
      const uint32_t base = (uint32_t)mats.size();
      mats.insert(mats.end(), m->materials.begin(), m->materials.end());
      for (size_t i = 0; i < m->triangles.size(); i++) {
        const Triangle& t = m->triangles[i];
        const float v[9] = {t.a.x, t.a.y, t.a.z, t.b.x, t.b.y, t.b.z, t.c.x, t.c.y, t.c.z};
        tris.insert(tris.end(), v, v + 9);
        midx.push_back(base + m->material_idxs[i]);
      }
    }
    if (mats.empty()) mats.push_back(Material{});
    rtx_scene s{};
    s.triangles = tris.data();
    s.num_triangles = midx.size();
    s.material_index = midx.data();
    s.materials = reinterpret_cast<const rtx_material*>(mats.data());
    s.num_materials = mats.size();
    s.spheres = reinterpret_cast<const rtx_sphere*>(spheres.data());
    s.num_spheres = spheres.size();
    s.boxes = reinterpret_cast<const rtx_box*>(boxes.data());
    s.num_boxes = boxes.size();
    s.planes = reinterpret_cast<const rtx_plane*>(planes.data());
    s.num_planes = planes.size();
    check(rtx_set_scene(ctx_, &s), "rtx_set_scene");
  }

  rtx_ctx* ctx() const { return ctx_; }
  const rtx_config& config() const { return cfg_; }
  float* accum() {
    if (!accum_) check(rtx_accum_alloc(ctx_, &accum_), "rtx_accum_alloc");
    return accum_;
  }

 private:
  rtx_config cfg_;
  rtx_ctx* ctx_ = nullptr;
  float* accum_ = nullptr;
};

// Raytracer.cpp:47-144.  pixels holds the running mean, idx = px + py*width.
class Img {
 public:
  std::vector<Vec3> pixels;
  unsigned int sample_count = 0;
  uint32_t width, height;

  Img(uint32_t w, uint32_t h) : width(w), height(h) { pixels.resize((size_t)w * h); }
  size_t idx(int px, int py) const { return (size_t)px + (size_t)py * width; }
  Vec3 get(int px, int py) const { return pixels[idx(px, py)]; }
  void set(int px, int py, Vec3 v) { pixels[idx(px, py)] = v; }

  // Float image, bottom row first (PFM convention), little endian.
  bool savePFM(const char* path) const {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    for (uint32_t y = height; y-- > 0;) std::fwrite(&pixels[(size_t)y * width], sizeof(Vec3), width, f);
    return std::fclose(f) == 0;
  }
  // 8-bit sRGB with the reference's transfer function (Vec3.cpp:39-53): clamp to 1, then
  // 12.92x below 0.0031308, 1.055 x^(1/2.4) - 0.055 above.
  bool savePPM(const char* path) const {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    auto enc = [](float x) {
      x = x < 1.0f ? x : 1.0f;
      float g = x <= 0.0031308f ? 12.92f * x : 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f;
      g = g < 0 ? 0 : g;
      return (unsigned char)(255.0f * g);
    };
    for (const Vec3& p : pixels) {
      unsigned char rgb[3] = {enc(p.x), enc(p.y), enc(p.z)};
      std::fwrite(rgb, 1, 3, f);
    }
    return std::fclose(f) == 0;
  }
};

// Img::clear (Raytracer.cpp:61-68)
inline void clear(Img& img, Scene& scene) {
  check(rtx_accum_clear(scene.ctx(), scene.accum(), nullptr), "rtx_accum_clear");
  for (Vec3& p : img.pixels) p = Vec3();
  img.sample_count = 0;
}

// render(img, pathTrace) + img.sample_count++ (Raytracer.cpp:236-277,319-320) for `spp`
// passes at once: adds spp samples per pixel and refreshes img.pixels with the mean.
inline void render(Img& img, Scene& scene, uint32_t spp = 1, uint64_t seed = 0x5EED, Vec3 pos = Vec3()) {
  const float off[3] = {pos.x, pos.y, pos.z};
  check(rtx_render(scene.ctx(), scene.accum(), spp, img.sample_count, seed, off, 0, 0, nullptr), "rtx_render");
  img.sample_count += spp;
  check(rtx_resolve(scene.ctx(), scene.accum(), img.sample_count, &img.pixels[0].x, nullptr), "rtx_resolve");
}

}  // namespace rt
