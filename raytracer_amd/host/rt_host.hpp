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
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <limits>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
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
inline bool blank(char c) { return c == ' ' || c == '\t'; }
inline bool digit(char c) { return (unsigned)(c - '0') < 10u; }
inline const char* skip_blanks(const char* p) { return p + std::strspn(p, " \t"); }

// Lines of a text file; "\n", "\r\n" and a lone "\r" all end a line.
inline bool read_lines(const std::string& path, std::vector<std::string>& lines) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  size_t start = 0;
  for (size_t i = 0; i < text.size(); i++) {
    if (text[i] != '\n' && text[i] != '\r') continue;
    lines.emplace_back(text, start, i - start);
    if (text[i] == '\r' && i + 1 < text.size() && text[i + 1] == '\n') i++;
    start = i + 1;
  }
  if (start < text.size()) lines.emplace_back(text, start, text.size() - start);
  return true;
}

// Decimal text in [p, end) to double with the arithmetic of the OBJ library the reference
// parses with (tinyobjloader 2.0's number reader, behind Scene::load_mesh, reference
// Scene.h:57-62).  That reader is not correctly rounded: it sums the integer digits as
// m*10+d, adds each fraction digit times a power of ten (a literal for the first seven
// places, pow(10,-k) after), and applies a decimal exponent e as ldexp(m*5^e, e).  Doing
// the same sums in the same order gives the same double, hence the same float vertex.
// Text after the number is ignored; false when there is no number.
inline bool obj_number(const char* p, const char* end, double& out) {
  static const double place[8] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
  if (p >= end) return false;
  int sign = 1, e10 = 0;
  bool dot_first = false;
  if (*p == '+' || *p == '-') {
    sign = *p == '-' ? -1 : 1;
    p++;
    dot_first = p != end && *p == '.';
  } else if (*p == '.') {
    dot_first = true;
  } else if (!digit(*p)) {
    return false;
  }
  double m = 0.0;
  if (!dot_first) {
    const char* first = p;
    for (; p != end && digit(*p); p++) {
      m *= 10;
      m += (int)(*p - '0');
    }
    if (p == first) return false;
  }
  if (p != end && *p == '.') {
    p++;
    for (int k = 1; p != end && digit(*p); p++, k++) m += (int)(*p - '0') * (k < 8 ? place[k] : std::pow(10.0, -k));
  }
  if (p != end && (*p == 'e' || *p == 'E')) {
    p++;
    int esign = 1;
    if (p != end && (*p == '+' || *p == '-')) {
      esign = *p == '-' ? -1 : 1;
      p++;
    } else if (p == end || !digit(*p)) {
      return false;
    }
    const char* first = p;
    for (; p != end && digit(*p); p++) e10 = e10 * 10 + (int)(*p - '0');
    if (p == first) return false;
    e10 *= esign;
  }
  out = sign * (e10 ? std::ldexp(m * std::pow(5.0, e10), e10) : m);
  return true;
}
// The next blank-separated number of a line as float; a missing or malformed one is 0, as in
// that library.
inline float obj_real(const char*& p) {
  p = skip_blanks(p);
  const char* end = p + std::strcspn(p, " \t\r");
  double v = 0.0;
  obj_number(p, end, v);
  p = end;
  return (float)v;
}

// "newmtl" entries of an MTL stream in the order the library numbers them: an entry is
// stored when the next one starts (unless its name is empty) and once more at the end of
// the file, whatever the name; the name is the rest of the line; the first slot of a name is
// the one "usemtl" refers to.
inline void mtl_slots(const std::vector<std::string>& lines, std::map<std::string, int>& slot_of, size_t& count) {
  std::string name;
  auto store = [&]() {
    slot_of.insert(std::make_pair(name, (int)count));
    count++;
  };
  for (std::string line : lines) {
    line = line.substr(0, line.find_last_not_of(" \t") + 1);
    const char* p = skip_blanks(line.c_str());
    if (std::strncmp(p, "newmtl", 6) != 0 || !blank(p[6])) continue;
    if (!name.empty()) store();
    name = p + 7;
  }
  store();
}

struct ObjFace {
  std::vector<int> v;  // zero-based vertex indices
  int material;        // slot, or -1
};

// One polygon to triangles the way the library does before load_mesh sees it (its default
// configuration triangulates): ear clipping in the plane of the two axes along which the
// polygon's first non-degenerate corner is widest.  Starting at vertex 0, the corner
// (g, g+1, g+2) is cut off when it turns the way the polygon's signed area does and none of
// the other vertices lies inside it; otherwise g moves on.  A convex polygon therefore
// comes out as the fan (0,1,2), (0,2,3), ...; a polygon on which no ear is found in a full
// round keeps only the triangles cut so far.
inline void obj_triangulate(const std::vector<int>& face, const std::vector<float>& xyz, std::vector<int>& out) {
  const size_t n = face.size(), nfloats = xyz.size();
  auto coord = [&](int vi, size_t axis) { return xyz[(size_t)vi * 3 + axis]; };
  size_t ax0 = 1, ax1 = 2;
  for (size_t k = 0; k < n; k++) {
    const int i0 = face[k], i1 = face[(k + 1) % n], i2 = face[(k + 2) % n];
    const float e0x = coord(i1, 0) - coord(i0, 0), e0y = coord(i1, 1) - coord(i0, 1), e0z = coord(i1, 2) - coord(i0, 2);
    const float e1x = coord(i2, 0) - coord(i1, 0), e1y = coord(i2, 1) - coord(i1, 1), e1z = coord(i2, 2) - coord(i1, 2);
    const float cx = std::fabs(e0y * e1z - e0z * e1y);
    const float cy = std::fabs(e0z * e1x - e0x * e1z);
    const float cz = std::fabs(e0x * e1y - e0y * e1x);
    const float tiny = std::numeric_limits<float>::epsilon();
    if (cx > tiny || cy > tiny || cz > tiny) {
      if (!(cx > cy && cx > cz)) {
        ax0 = 0;
        if (cz > cx && cz > cy) ax1 = 1;
      }
      break;
    }
  }
  float area = 0;
  for (size_t k = 0; k < n; k++) {
    const int i0 = face[k], i1 = face[(k + 1) % n];
    area += (coord(i0, ax0) * coord(i1, ax1) - coord(i0, ax1) * coord(i1, ax0)) * 0.5f;
  }
  (void)nfloats;
  std::vector<int> rest = face;
  size_t g = 0, tries = n, last_size = n;
  while (rest.size() > 3 && tries > 0) {
    const size_t m = rest.size();
    if (g >= m) g -= m;
    if (last_size != m) {
      last_size = m;
      tries = m;
    } else {
      tries--;
    }
    int ind[3];
    float px[3], py[3];
    for (size_t k = 0; k < 3; k++) {
      ind[k] = rest[(g + k) % m];
      px[k] = coord(ind[k], ax0);
      py[k] = coord(ind[k], ax1);
    }
    const float cross = (px[1] - px[0]) * (py[2] - py[1]) - (py[1] - py[0]) * (px[2] - px[1]);
    if (cross * area < 0.0f) {
      g++;
      continue;
    }
    bool inside = false;
    for (size_t other = 3; other < m && !inside; other++) {
      const int vi = rest[(g + other) % m];
      const float tx = coord(vi, ax0), ty = coord(vi, ax1);
      // crossing-number test of (tx,ty) against the triangle's three edges
      for (int i = 0, j = 2; i < 3; j = i++)
        if (((py[i] > ty) != (py[j] > ty)) && (tx < (px[j] - px[i]) * (ty - py[i]) / (py[j] - py[i]) + px[i]))
          inside = !inside;
    }
    if (inside) {
      g++;
      continue;
    }
    out.insert(out.end(), ind, ind + 3);
    rest.erase(rest.begin() + (ptrdiff_t)((g + 1) % m));
  }
  if (rest.size() == 3) out.insert(out.end(), rest.begin(), rest.end());
}
}  // namespace detail

// The OBJ front end of Scene::load_mesh (Scene.h:55-115), which reads the file with
// tinyobjloader in its default configuration and walks the result.  Same rules here:
//  - "v x y z": numbers as detail::obj_number reads them, missing ones 0;
//  - "f": vertex references v, v/vt, v//vn, v/vt/vn, 1-based or negative = relative to the
//    vertices read so far; 0 or text is an error; faces of fewer than 3 vertices are dropped;
//    larger ones go through detail::obj_triangulate, and load_mesh's fan over the resulting
//    triangles (Scene.h:88-109) changes nothing;
//  - "mtllib a b ...": the first of the names that opens (next to the OBJ file) is read;
//    one Material per entry (detail::mtl_slots), initialised to the reference's placeholder
//    (0.65,0.05,0.05) diffuse (Scene.h:78) for the caller to override by slot;
//  - "usemtl name": the slot for the faces that follow, none for an unknown name.
// Two deliberate differences: faces without a material (the reference would index its
// material list with -1, Scene.h:107) get one extra placeholder slot at the end, and a face
// that refers to a vertex the file never defines is an error instead of a read outside the
// vertex array.  Polygons are triangulated once the whole file is read, so a face may refer
// to a vertex defined further down.
inline Mesh* load_obj(const std::string& file_name) {
  std::vector<std::string> lines;
  if (!detail::read_lines(file_name, lines)) throw Error("load_mesh: cannot open " + file_name);
  const std::string dir = file_name.substr(0, file_name.find_last_of("/\\") + 1);
  std::vector<float> xyz;
  std::vector<detail::ObjFace> faces;
  std::map<std::string, int> slot_of;
  size_t num_materials = 0;
  int current = -1;
  for (size_t lineno = 1; lineno <= lines.size(); lineno++) {
    const char* p = detail::skip_blanks(lines[lineno - 1].c_str());
    if (p[0] == '\0' || p[0] == '#') continue;
    if (p[0] == 'v' && detail::blank(p[1])) {
      p += 2;
      for (int c = 0; c < 3; c++) xyz.push_back(detail::obj_real(p));
    } else if (p[0] == 'f' && detail::blank(p[1])) {
      p = detail::skip_blanks(p + 2);
      detail::ObjFace face;
      face.material = current;
      while (*p != '\0' && *p != '\r' && *p != '\n') {
        // v, v/vt, v//vn or v/vt/vn; every index that is present must be a non-zero number
        const int raw = std::atoi(p);
        bool ok = raw != 0;
        p += std::strcspn(p, "/ \t\r");
        for (int part = 0; ok && part < 2 && *p == '/'; part++) {
          p++;
          if (part == 0 && *p == '/') continue;  // no texture index
          ok = std::atoi(p) != 0;
          p += std::strcspn(p, "/ \t\r");
        }
        if (!ok) throw Error("load_mesh: bad face index at line " + std::to_string(lineno));
        face.v.push_back(raw > 0 ? raw - 1 : (int)(xyz.size() / 3) + raw);
        p += std::strspn(p, " \t\r");
      }
      faces.push_back(std::move(face));
    } else if (!std::strncmp(p, "usemtl", 6)) {
      p = detail::skip_blanks(p + 6);
      auto it = slot_of.find(std::string(p, std::strcspn(p, " \t\r")));
      current = it == slot_of.end() ? -1 : it->second;
    } else if (!std::strncmp(p, "mtllib", 6) && detail::blank(p[6])) {
      // names separated by single spaces, a backslash keeps the next character
      std::vector<std::string> names(1);
      for (p += 7; *p; p++) {
        if (*p == '\\' && p[1]) names.back() += *++p;
        else if (*p == '\\') break;
        else if (*p == ' ') {
          if (names.back().empty()) continue;
          names.emplace_back();
        } else names.back() += *p;
      }
      for (const std::string& name : names) {
        std::vector<std::string> mtl;
        if (!detail::read_lines(dir + name, mtl)) continue;
        detail::mtl_slots(mtl, slot_of, num_materials);
        break;
      }
    }
  }
  Mesh* mesh = new Mesh();
  const Material placeholder{Vec3(0.65f, 0.05f, 0.05f), 0.0f, diffuse};
  for (size_t i = 0; i < num_materials; i++) mesh->add_material(placeholder);
  bool need_default = false;
  std::vector<int> tri;
  const int nverts = (int)(xyz.size() / 3);
  for (const detail::ObjFace& face : faces) {
    if (face.v.size() < 3) continue;
    for (int vi : face.v)
      if (vi < 0 || vi >= nverts) {
        delete mesh;
        throw Error("load_mesh: bad face index (vertex " + std::to_string(vi + 1) + " is not in the file)");
      }
    tri.clear();
    detail::obj_triangulate(face.v, xyz, tri);
    auto vertex = [&](int vi) { return Vec3(xyz[3 * (size_t)vi], xyz[3 * (size_t)vi + 1], xyz[3 * (size_t)vi + 2]); };
    for (size_t t = 0; t + 2 < tri.size(); t += 3) {
      mesh->add_triangle(Triangle{vertex(tri[t]), vertex(tri[t + 1]), vertex(tri[t + 2])},
                         face.material < 0 ? (unsigned)num_materials : (unsigned)face.material);
      if (face.material < 0) need_default = true;
    }
  }
  if (need_default) mesh->add_material(placeholder);
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
    // Identity of what is being rendered, for checkpoints: FNV-1a over the configuration fields that
    // shape the image and over every scene array.
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n) {
      const unsigned char* b = static_cast<const unsigned char*>(p);
      for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    };
    mix(&cfg_.width, sizeof(cfg_.width));
    mix(&cfg_.height, sizeof(cfg_.height));
    mix(&cfg_.fov, sizeof(cfg_.fov));
    mix(cfg_.cam_origin, sizeof(cfg_.cam_origin));
    mix(&cfg_.eps, sizeof(cfg_.eps));
    mix(&cfg_.nee_threshold, sizeof(cfg_.nee_threshold));
    mix(&cfg_.rr_cap, sizeof(cfg_.rr_cap));
    mix(&cfg_.sky, sizeof(cfg_.sky));
    mix(&cfg_.max_bounces, sizeof(cfg_.max_bounces));
    mix(&cfg_.camera, sizeof(cfg_.camera));
    mix(&cfg_.integrator, sizeof(cfg_.integrator));
    mix(&cfg_.focal_length, sizeof(cfg_.focal_length));
    mix(&cfg_.lens_radius, sizeof(cfg_.lens_radius));
    mix(&cfg_.stretch_x, sizeof(cfg_.stretch_x));
    mix(&cfg_.stretch_y, sizeof(cfg_.stretch_y));
    mix(tris.data(), tris.size() * sizeof(float));
    mix(midx.data(), midx.size() * sizeof(uint32_t));
    mix(mats.data(), mats.size() * sizeof(Material));
    mix(spheres.data(), spheres.size() * sizeof(Sphere));
    mix(boxes.data(), boxes.size() * sizeof(Box));
    mix(planes.data(), planes.size() * sizeof(Plane));
    identity_ = h;
  }
  uint64_t identity() const { return identity_; }  // of the configuration and the committed scene

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
  uint64_t identity_ = 0;
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
  // `scale` is the magnitude of the PFM scale line (its sign says little endian); checkpoints
  // keep the sample count there.
  bool savePFM(const char* path, float scale = 1.0f) const {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%u %u\n-%.9g\n", width, height, (double)scale);
    for (uint32_t y = height; y-- > 0;) std::fwrite(&pixels[(size_t)y * width], sizeof(Vec3), width, f);
    return std::fclose(f) == 0;
  }
  // OpenEXR file as Img::saveEXR writes it through tinyexr (reference Raytracer.cpp:83-143):
  // one scanline part, no compression (InitEXRHeader leaves the type at 0), three HALF
  // channels named B, G, R, data and display window the whole image, one chunk per line
  // holding the line's B values, then G, then R.  Written here directly; the bytes are the
  // ones tinyexr produces (checked against the reference's vendored copy in the tests).
  // Float to half as tinyexr converts: the mantissa is cut to ten bits and rounded up when
  // the first dropped bit is set; below the half range the value is shifted into a
  // denormal the same way, float denormals become signed zero, overflow becomes infinity.
  static uint16_t half_bits(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    const uint32_t sign = (u >> 16) & 0x8000u, exp = (u >> 23) & 0xffu, man = u & 0x7fffffu;
    uint32_t h = 0;
    if (exp == 0) {
      h = 0;
    } else if (exp == 255) {
      h = 0x7c00u | (man ? 0x200u : 0u);
    } else {
      const int e = (int)exp - 127 + 15;
      if (e >= 31) {
        h = 0x7c00u;
      } else if (e <= 0) {
        if (14 - e <= 24) {
          const uint32_t m = man | 0x800000u;
          h = m >> (14 - e);
          if ((m >> (13 - e)) & 1u) h++;
        }
      } else {
        h = ((uint32_t)e << 10) | (man >> 13);
        if (man & 0x1000u) h++;
      }
    }
    return (uint16_t)(sign | h);
  }
  bool saveEXR(const char* path) const {
    std::vector<unsigned char> out;
    auto put = [&](const void* p, size_t n) {
      const unsigned char* b = static_cast<const unsigned char*>(p);
      out.insert(out.end(), b, b + n);
    };
    auto put_u32 = [&](uint32_t v) { put(&v, 4); };  // the file is little endian, as is the host
    auto attribute = [&](const char* name, const char* type, const void* data, uint32_t n) {
      put(name, std::strlen(name) + 1);
      put(type, std::strlen(type) + 1);
      put_u32(n);
      put(data, n);
    };
    const unsigned char magic_version[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
    put(magic_version, 8);
    {
      std::vector<unsigned char> ch;
      for (const char* name : {"B", "G", "R"}) {
        ch.push_back((unsigned char)name[0]);
        ch.push_back(0);
        const uint32_t fields[4] = {1u /* HALF */, 0u /* pLinear + 3 reserved */, 1u, 1u /* sampling */};
        const unsigned char* b = reinterpret_cast<const unsigned char*>(fields);
        ch.insert(ch.end(), b, b + 16);
      }
      ch.push_back(0);
      attribute("channels", "chlist", ch.data(), (uint32_t)ch.size());
    }
    const unsigned char none = 0;
    attribute("compression", "compression", &none, 1);
    const int32_t window[4] = {0, 0, (int32_t)width - 1, (int32_t)height - 1};
    attribute("dataWindow", "box2i", window, 16);
    attribute("displayWindow", "box2i", window, 16);
    attribute("lineOrder", "lineOrder", &none, 1);  // increasing y
    const float one = 1.0f, centre[2] = {0.0f, 0.0f};
    attribute("pixelAspectRatio", "float", &one, 4);
    attribute("screenWindowCenter", "v2f", centre, 8);
    attribute("screenWindowWidth", "float", &one, 4);
    out.push_back(0);  // end of header
    const uint64_t line_bytes = 8 + (uint64_t)width * 6;
    uint64_t offset = out.size() + (uint64_t)height * 8;
    for (uint32_t y = 0; y < height; y++, offset += line_bytes) put(&offset, 8);
    std::vector<uint16_t> line((size_t)width * 3);
    for (uint32_t y = 0; y < height; y++) {
      put_u32(y);
      put_u32(width * 6u);
      const Vec3* row = &pixels[(size_t)y * width];
      for (uint32_t x = 0; x < width; x++) {
        line[x] = half_bits(row[x].z);
        line[width + x] = half_bits(row[x].y);
        line[2 * (size_t)width + x] = half_bits(row[x].x);
      }
      put(line.data(), line.size() * 2);
    }
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return (std::fclose(f) == 0) && ok;
  }
  // Reads what savePFM wrote (or any little-endian RGB PFM) back into pixels.
  bool loadPFM(const char* path, float* scale_out = nullptr) {
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    unsigned w = 0, h = 0;
    float scale = 0;
    char tag[3] = {0, 0, 0};
    bool ok = std::fscanf(f, "%2s %u %u %f", tag, &w, &h, &scale) == 4 && !std::strcmp(tag, "PF") && scale < 0 &&
              std::fgetc(f) == '\n';
    if (ok) {
      width = w;
      height = h;
      pixels.assign((size_t)w * h, Vec3());
      for (uint32_t y = height; ok && y-- > 0;)
        ok = std::fread(&pixels[(size_t)y * width], sizeof(Vec3), width, f) == width;
    }
    std::fclose(f);
    if (ok && scale_out) *scale_out = -scale;
    return ok;
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

// The same without waiting for the pass (rtx_render_async: two passes in flight, the second one's
// work fills the GPU while the last long paths of the first run out); img.pixels is not refreshed
// until refresh() is called.  Same sums as render().
inline void render_in_flight(Img& img, Scene& scene, uint32_t spp = 1, uint64_t seed = 0x5EED, Vec3 pos = Vec3()) {
  const float off[3] = {pos.x, pos.y, pos.z};
  check(rtx_render_async(scene.ctx(), scene.accum(), spp, img.sample_count, seed, off, 0, 0, nullptr),
        "rtx_render_async");
  img.sample_count += spp;
}
inline void refresh(Img& img, Scene& scene) {  // waits for the passes in flight
  if (img.sample_count == 0) return;
  check(rtx_resolve(scene.ctx(), scene.accum(), img.sample_count, &img.pixels[0].x, nullptr), "rtx_resolve");
}

// Checkpoint of an accumulation in progress -- what Img holds between passes (Img::pixels as sums,
// Img::sample_count; Raytracer.cpp:49-59) plus what makes it resumable without surprises:
//   "RTXCKPT1" | u32 width | u32 height | u64 sample count | u64 seed | u64 scene identity |
//   width*height*3 floats (per-pixel sums, top row first)
// all little endian.  The file is written under another name and renamed, so a crash while
// writing leaves the previous checkpoint intact.  A render resumed from it continues with the
// next sample index and ends with the bits of an uninterrupted one; a checkpoint of another
// image size, seed, configuration or scene is refused.
inline void save_checkpoint(const Img& img, Scene& scene, const char* path, uint64_t seed = 0x5EED) {
  std::vector<float> sums((size_t)img.width * img.height * 3);
  check(rtx_resolve(scene.ctx(), scene.accum(), 1, sums.data(), nullptr), "rtx_resolve");
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) throw Error(std::string("cannot write ") + tmp);
  const uint32_t dims[2] = {img.width, img.height};
  const uint64_t meta[3] = {img.sample_count, seed, scene.identity()};
  bool ok = std::fwrite("RTXCKPT1", 8, 1, f) == 1 && std::fwrite(dims, sizeof(dims), 1, f) == 1 &&
            std::fwrite(meta, sizeof(meta), 1, f) == 1 && std::fwrite(sums.data(), sizeof(float), sums.size(), f) == sums.size();
  ok = (std::fflush(f) == 0) && ok;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path) != 0) {
    std::remove(tmp.c_str());
    throw Error(std::string("cannot write ") + path);
  }
}
inline void load_checkpoint(Img& img, Scene& scene, const char* path, uint64_t seed = 0x5EED) {
  FILE* f = std::fopen(path, "rb");
  if (!f) throw Error(std::string("cannot read ") + path);
  char magic[8];
  uint32_t dims[2] = {0, 0};
  uint64_t meta[3] = {0, 0, 0};
  std::vector<float> sums;
  bool ok = std::fread(magic, 8, 1, f) == 1 && !std::memcmp(magic, "RTXCKPT1", 8) &&
            std::fread(dims, sizeof(dims), 1, f) == 1 && std::fread(meta, sizeof(meta), 1, f) == 1;
  if (ok && (dims[0] != img.width || dims[1] != img.height)) {
    std::fclose(f);
    throw Error(std::string("checkpoint ") + path + " does not fit this image");
  }
  if (ok) {
    sums.resize((size_t)dims[0] * dims[1] * 3);
    ok = std::fread(sums.data(), sizeof(float), sums.size(), f) == sums.size() && std::fgetc(f) == EOF;
  }
  std::fclose(f);
  if (!ok) throw Error(std::string("cannot read ") + path + " (not a checkpoint, or cut short)");
  if (meta[0] < 1 || meta[0] > 0xffffffffull) throw Error(std::string("checkpoint ") + path + " has a bad sample count");
  if (meta[1] != seed) throw Error(std::string("checkpoint ") + path + " was rendered with another seed");
  if (meta[2] != scene.identity())
    throw Error(std::string("checkpoint ") + path + " was rendered from another scene or configuration");
  check(rtx_accum_upload(scene.ctx(), scene.accum(), sums.data(), nullptr), "rtx_accum_upload");
  img.sample_count = (uint32_t)meta[0];
  check(rtx_resolve(scene.ctx(), scene.accum(), img.sample_count, &img.pixels[0].x, nullptr), "rtx_resolve");
}

// ---- several GPUs: one process per GPU, one exchange step ----------------------------------------
// The reference renders with one OpenMP loop (Raytracer.cpp:242); here every rank (process) owns
// one GPU and one Scene, takes its share of each pass's sample indices, and the per-GPU sums
// are added onto rank 0 by the library's RCCL reduce (rtx_comm_reduce).  Rank 0 keeps the
// running total in its buffer; the other ranks' buffers hold one pass at a time.
struct Ranks {
  int rank = 0, world = 1;
};

// Share `index` of n items cut into `parts` near-equal contiguous shares (the first n % parts
// get one more): {first, count}.
inline void split_range(uint32_t n, int parts, int index, uint32_t& first, uint32_t& count) {
  const uint32_t q = n / (uint32_t)parts, r = n % (uint32_t)parts;
  first = (uint32_t)index * q + ((uint32_t)index < r ? (uint32_t)index : r);
  count = q + ((uint32_t)index < r ? 1u : 0u);
}

// Joins the ranks' communicator.  Rank 0 makes the id and publishes it as `id_file` (written
// under another name and renamed, so a reader never sees half of it); the others wait for the file.
inline void comm_join(Scene& scene, const Ranks& rk, const std::string& id_file, double timeout_s = 120.0) {
  char id[RTX_COMM_ID_BYTES];
  if (rk.rank == 0) {
    check(rtx_comm_create_id(id), "rtx_comm_create_id");
    const std::string tmp = id_file + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id) || std::fclose(f) != 0 ||
        std::rename(tmp.c_str(), id_file.c_str()) != 0)
      throw Error("cannot write " + id_file);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      FILE* f = std::fopen(id_file.c_str(), "rb");
      if (f) {
        const size_t got = std::fread(id, 1, sizeof(id), f);
        std::fclose(f);
        if (got == sizeof(id)) break;
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        throw Error("timed out waiting for " + id_file);
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  }
  check(rtx_comm_init(scene.ctx(), id, rk.rank, rk.world), "rtx_comm_init");
}

// One pass of `spp` samples per pixel spread over the ranks: sample indices
// img.sample_count .. +spp-1 are cut into contiguous shares, every rank renders its own, and the
// sums meet on rank 0, whose img.pixels gets the new mean.  The union of the shares is what
// render() renders on one GPU; only the order of the float additions differs.
inline void render_ranks(Img& img, Scene& scene, const Ranks& rk, uint32_t spp = 1, uint64_t seed = 0x5EED,
                         Vec3 pos = Vec3()) {
  const float off[3] = {pos.x, pos.y, pos.z};
  uint32_t first, count;
  split_range(spp, rk.world, rk.rank, first, count);
  if (rk.rank != 0) check(rtx_accum_clear(scene.ctx(), scene.accum(), nullptr), "rtx_accum_clear");
  if (count)
    check(rtx_render(scene.ctx(), scene.accum(), count, img.sample_count + first, seed, off, 0, 0, nullptr),
          "rtx_render");
  check(rtx_comm_reduce(scene.ctx(), scene.accum(), 0, nullptr), "rtx_comm_reduce");
  check(rtx_stream_synchronize(scene.ctx(), nullptr), "rtx_stream_synchronize");
  img.sample_count += spp;
  if (rk.rank == 0)
    check(rtx_resolve(scene.ctx(), scene.accum(), img.sample_count, &img.pixels[0].x, nullptr), "rtx_resolve");
}

}  // namespace rt
