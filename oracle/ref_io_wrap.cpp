// ref_io_wrap.cpp -- C entry points around the third-party file-format libraries the
// REFERENCE vendors and uses for its scene input and image output: tiny_obj_loader.h
// (Scene::load_mesh, reference Scene.h:55-115) and tinyexr.h + miniz (Img::saveEXR,
// reference Raytracer.cpp:83-143).
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes those
// headers where they lie (the Makefile passes -I$(REFERENCE_DIR) and compiles
// $(REFERENCE_DIR)/miniz.cpp in place) and calls them the way the reference's two
// functions do.  The result, oracle/_ref/libref_io.so, pins the product's OBJ front end
// and EXR writer (raytracer_amd/host/rt_host.hpp) to what the reference's own loader
// parses and what its own writer emits, byte for byte.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"
#define TINYEXR_IMPLEMENTATION
#include "tinyexr.h"

extern "C" {

// Parses `path` with tinyobj::ObjReader and default ObjReaderConfig (Scene.h:57-62) and
// walks the shapes as load_mesh does (Scene.h:84-111): per face, vertex 0 and a running
// second vertex, one triangle per further vertex, the face's material id as parsed (-1 when
// the face has no known material).  Returns 0 on success; the arrays are malloc'ed.
int ref_obj_triangles(const char* path, float** tris_out, int** mat_out, size_t* ntri_out, size_t* nmat_out) {
  tinyobj::ObjReaderConfig reader_config;
  tinyobj::ObjReader reader;
  if (!reader.ParseFromFile(path, reader_config)) return 1;
  const tinyobj::attrib_t& attrib = reader.GetAttrib();
  const std::vector<tinyobj::shape_t>& shapes = reader.GetShapes();
  std::vector<float> tris;
  std::vector<int> mats;
  for (size_t s = 0; s < shapes.size(); s++) {
    size_t index_offset = 0;
    for (size_t f = 0; f < shapes[s].mesh.num_face_vertices.size(); f++) {
      const int fv = shapes[s].mesh.num_face_vertices[f];
      auto vertex = [&](size_t k, float out[3]) {
        const tinyobj::index_t idx = shapes[s].mesh.indices[index_offset + k];
        for (int c = 0; c < 3; c++) out[c] = attrib.vertices[3 * idx.vertex_index + c];
      };
      float a[3], b[3], c[3];
      vertex(0, a);
      vertex(1, b);
      for (int v = 2; v < fv; v++) {
        vertex((size_t)v, c);
        tris.insert(tris.end(), a, a + 3);
        tris.insert(tris.end(), b, b + 3);
        tris.insert(tris.end(), c, c + 3);
        mats.push_back(shapes[s].mesh.material_ids[f]);
        std::memcpy(b, c, sizeof(b));
      }
      index_offset += (size_t)fv;
    }
  }
  *ntri_out = mats.size();
  *nmat_out = reader.GetMaterials().size();
  *tris_out = (float*)std::malloc(sizeof(float) * (tris.size() + 1));
  *mat_out = (int*)std::malloc(sizeof(int) * (mats.size() + 1));
  std::memcpy(*tris_out, tris.data(), sizeof(float) * tris.size());
  std::memcpy(*mat_out, mats.data(), sizeof(int) * mats.size());
  return 0;
}
void ref_free(void* p) { std::free(p); }

// Writes rgb (w*h*3 floats, row-major, pixel = px + py*w: the layout of Img::pixels) the
// way Img::saveEXR does: three FLOAT planes handed over in the order B, G, R, stored as
// HALF, header from InitEXRHeader (no compression), SaveEXRImageToFile.
int ref_exr_save(const float* rgb, int w, int h, const char* path) {
  EXRHeader header;
  InitEXRHeader(&header);
  EXRImage image;
  InitEXRImage(&image);
  image.num_channels = 3;
  std::vector<float> planes[3];
  for (int c = 0; c < 3; c++) planes[c].resize((size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++)
    for (int c = 0; c < 3; c++) planes[c][i] = rgb[3 * i + c];
  float* image_ptr[3] = {planes[2].data(), planes[1].data(), planes[0].data()};  // B, G, R
  image.images = (unsigned char**)image_ptr;
  image.width = w;
  image.height = h;
  header.num_channels = 3;
  header.channels = (EXRChannelInfo*)std::malloc(sizeof(EXRChannelInfo) * 3);
  const char* names[3] = {"B", "G", "R"};
  for (int c = 0; c < 3; c++) {
    std::strncpy(header.channels[c].name, names[c], 255);
    header.channels[c].name[1] = '\0';
  }
  header.pixel_types = (int*)std::malloc(sizeof(int) * 3);
  header.requested_pixel_types = (int*)std::malloc(sizeof(int) * 3);
  for (int c = 0; c < 3; c++) {
    header.pixel_types[c] = TINYEXR_PIXELTYPE_FLOAT;
    header.requested_pixel_types[c] = TINYEXR_PIXELTYPE_HALF;
  }
  const char* err = NULL;
  const int ret = SaveEXRImageToFile(&image, &header, path, &err);
  if (err) FreeEXRErrorMessage(err);
  std::free(header.channels);
  std::free(header.pixel_types);
  std::free(header.requested_pixel_types);
  return ret;
}

// Reads an EXR back with the same library (LoadEXR: RGBA float, row-major).  out_rgba must
// hold w*h*4 floats; returns 0 on success and the size through w/h.
int ref_exr_load(const char* path, float* out_rgba, size_t capacity_floats, int* w, int* h) {
  float* rgba = NULL;
  const char* err = NULL;
  const int ret = LoadEXR(&rgba, w, h, path, &err);
  if (err) FreeEXRErrorMessage(err);
  if (ret != TINYEXR_SUCCESS) return ret;
  const size_t n = (size_t)(*w) * (size_t)(*h) * 4;
  if (n <= capacity_floats) std::memcpy(out_rgba, rgba, n * sizeof(float));
  std::free(rgba);
  return n <= capacity_floats ? 0 : -100;
}
}
