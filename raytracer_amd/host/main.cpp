// main.cpp -- a headless driver in the shape of the reference's Raytracer.cpp main():
// build the scene (buildScene, Raytracer.cpp:150-230), then the progressive loop of
// render_loop (Raytracer.cpp:307-331), saving the image instead of showing a window.
//
//   rt_host [--obj FILE] [--width W] [--height H] [--spp N] [--passes P] [--sky]
//           [--out PREFIX] [--checkpoint FILE] [--checkpoint-every P] [--resume FILE] [--parse-only]
//           [--pfm-to-exr IN OUT] [--dump-mesh FILE] [--gpus N] [--in-flight]
//
// --in-flight leaves every pass in flight (rtx_render_async, two at a time) instead of waiting for
// it and refreshing the image: the next pass's work fills the GPU while the last long paths of this
// one run out.  The image is read back for checkpoints and at the end; same bits.
//
// --gpus N renders on N GPUs of this host: the program starts N copies of itself, one rank per
// GPU (before it has touched a GPU itself), every rank renders its share of each pass's samples,
// the sums meet on rank 0 through the library's RCCL reduce, and rank 0 writes the image.
// --comm goes through the same start-up, communicator and reduce with a single rank as well
// (a one-GPU host can rehearse everything but the wire: RCCL refuses two ranks on one GPU).
//
// --dump-mesh (with --obj, no device needed) writes what load_mesh produced as a binary file:
// two little-endian uint64 (triangles, materials), then 9 floats per triangle, then one uint32
// material slot per triangle -- the arrays rtx_set_scene takes.
//
// --checkpoint writes the per-pixel sums, the sample count, the seed and the scene's identity
// after the last pass (and after every P-th pass with --checkpoint-every P); --resume
// starts from such a file and goes on with the next sample index, so that two runs of P passes
// end with the bits of one run of 2P.
//
// Without --obj the live scene's roles are built procedurally (the reference's
// geometry/CornellBox-Original.obj is not part of its repository): red/green/white walls,
// a 2-triangle light of emission 10 and the glass sphere of Raytracer.cpp:174-175.
// With --obj, material slots are overridden the way Raytracer.cpp:163-171 does when the
// file has at least 8 materials.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include "rt_host.hpp"

using namespace rt;

static void quad(Mesh* m, Vec3 p0, Vec3 p1, Vec3 p2, Vec3 p3, unsigned mat) {
  m->add_triangle(Triangle{p0, p1, p2}, mat);
  m->add_triangle(Triangle{p0, p2, p3}, mat);
}

static Mesh* procedural_cornell() {
  Mesh* m = new Mesh();
  for (int i = 0; i < 8; i++) m->add_material(Material{Vec3(0.65f, 0.05f, 0.05f), 0.0f, diffuse});
  quad(m, {-1, 0, 1}, {-1, 0, -1}, {-1, 2, -1}, {-1, 2, 1}, 0);   // left
  quad(m, {1, 0, -1}, {1, 0, 1}, {1, 2, 1}, {1, 2, -1}, 1);       // right
  quad(m, {-1, 0, 1}, {1, 0, 1}, {1, 0, -1}, {-1, 0, -1}, 2);     // floor
  quad(m, {-1, 2, -1}, {1, 2, -1}, {1, 2, 1}, {-1, 2, 1}, 3);     // ceiling
  quad(m, {-1, 0, -1}, {1, 0, -1}, {1, 2, -1}, {-1, 2, -1}, 4);   // back
  quad(m, {-0.25f, 1.98f, -0.25f}, {0.25f, 1.98f, -0.25f}, {0.25f, 1.98f, 0.25f}, {-0.25f, 1.98f, 0.25f}, 7);
  return m;
}

extern char** environ;

// Starts `world` copies of this program, rank r with "--rank r --world N --rendezvous FILE"
// appended, and waits for them; a failing rank ends the others.  No GPU call is made here.
static int spawn_ranks(int argc, char** argv, int world, const std::string& id_file) {
  std::remove(id_file.c_str());
  std::vector<pid_t> pids;
  for (int r = 0; r < world; r++) {
    std::vector<std::string> args(argv, argv + argc);
    args.insert(args.end(), {"--rank", std::to_string(r), "--world", std::to_string(world), "--rendezvous", id_file});
    std::vector<char*> av;
    for (std::string& a : args) av.push_back(&a[0]);
    av.push_back(nullptr);
    pid_t pid = 0;
    if (posix_spawn(&pid, "/proc/self/exe", nullptr, nullptr, av.data(), environ) != 0) {
      std::fprintf(stderr, "error: cannot start rank %d\n", r);
      for (pid_t p : pids) kill(p, SIGTERM);
      return 1;
    }
    pids.push_back(pid);
  }
  int rc = 0;
  size_t left = pids.size();
  while (left > 0) {
    int status = 0;
    const pid_t p = wait(&status);
    if (p < 0) break;
    left--;
    const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + WTERMSIG(status);
    if (code != 0 && rc == 0) {
      rc = code;
      for (pid_t q : pids)
        if (q != p) kill(q, SIGTERM);  // the others would wait in the collective for ever
    }
  }
  std::remove(id_file.c_str());
  return rc;
}

int main(int argc, char** argv) {
  std::string obj, out = "render";
  int gpus = 1;
  bool force_comm = false;
  Ranks rk;
  std::string rendezvous;
  uint32_t width = 800, height = 800, spp = 16, passes = 1, checkpoint_every = 0;
  bool sky = false, parse_only = false, in_flight = false;
  std::string pfm_in, exr_out, resume, checkpoint, dump_mesh;
  for (int i = 1; i < argc; i++) {
    auto next = [&](const char* name) -> const char* {
      if (i + 1 >= argc) {
        std::cerr << name << " needs a value\n";
        std::exit(2);
      }
      return argv[++i];
    };
    if (!std::strcmp(argv[i], "--obj")) obj = next("--obj");
    else if (!std::strcmp(argv[i], "--width")) width = (uint32_t)std::atoi(next("--width"));
    else if (!std::strcmp(argv[i], "--height")) height = (uint32_t)std::atoi(next("--height"));
    else if (!std::strcmp(argv[i], "--spp")) spp = (uint32_t)std::atoi(next("--spp"));
    else if (!std::strcmp(argv[i], "--passes")) passes = (uint32_t)std::atoi(next("--passes"));
    else if (!std::strcmp(argv[i], "--out")) out = next("--out");
    else if (!std::strcmp(argv[i], "--sky")) sky = true;
    else if (!std::strcmp(argv[i], "--parse-only")) parse_only = true;
    else if (!std::strcmp(argv[i], "--resume")) resume = next("--resume");
    else if (!std::strcmp(argv[i], "--checkpoint")) checkpoint = next("--checkpoint");
    else if (!std::strcmp(argv[i], "--in-flight")) in_flight = true;
    else if (!std::strcmp(argv[i], "--checkpoint-every")) checkpoint_every = (uint32_t)std::atoi(next("--checkpoint-every"));
    else if (!std::strcmp(argv[i], "--dump-mesh")) dump_mesh = next("--dump-mesh");
    else if (!std::strcmp(argv[i], "--gpus")) gpus = std::atoi(next("--gpus"));
    else if (!std::strcmp(argv[i], "--comm")) force_comm = true;
    else if (!std::strcmp(argv[i], "--rank")) rk.rank = std::atoi(next("--rank"));
    else if (!std::strcmp(argv[i], "--world")) rk.world = std::atoi(next("--world"));
    else if (!std::strcmp(argv[i], "--rendezvous")) rendezvous = next("--rendezvous");
    else if (!std::strcmp(argv[i], "--pfm-to-exr")) {
      pfm_in = next("--pfm-to-exr");
      exr_out = next("--pfm-to-exr");
    }
    else {
      std::cerr << "unknown argument " << argv[i] << "\n";
      return 2;
    }
  }
  if (gpus < 1 || rk.world < 1 || rk.rank < 0 || rk.rank >= rk.world) {
    std::cerr << "bad --gpus / --rank / --world\n";
    return 2;
  }
  if ((gpus > 1 || force_comm) && rendezvous.empty())  // the parent: it only starts the ranks
    return spawn_ranks(argc, argv, gpus, out + ".commid." + std::to_string((long)getpid()));
  const bool root = rk.rank == 0;
  try {
    if (!pfm_in.empty()) {
      // Image writer only (no device needed): a float image in, the saveEXR file out.
      Img img(1, 1);
      if (!img.loadPFM(pfm_in.c_str())) throw Error("cannot read " + pfm_in);
      if (!img.saveEXR(exr_out.c_str())) throw Error("cannot write " + exr_out);
      return 0;
    }
    if (!dump_mesh.empty()) {
      if (obj.empty()) throw Error("--dump-mesh needs --obj");
      Mesh* m = load_obj(obj);
      FILE* f = std::fopen(dump_mesh.c_str(), "wb");
      if (!f) throw Error("cannot write " + dump_mesh);
      const uint64_t head[2] = {m->triangles.size(), m->materials.size()};
      bool ok = std::fwrite(head, sizeof(head), 1, f) == 1;
      for (const Triangle& t : m->triangles) {
        const float v[9] = {t.a.x, t.a.y, t.a.z, t.b.x, t.b.y, t.b.z, t.c.x, t.c.y, t.c.z};
        ok = ok && std::fwrite(v, sizeof(v), 1, f) == 1;
      }
      for (unsigned idx : m->material_idxs) {
        const uint32_t v = idx;
        ok = ok && std::fwrite(&v, 4, 1, f) == 1;
      }
      ok = (std::fclose(f) == 0) && ok;
      if (!ok) throw Error("cannot write " + dump_mesh);
      std::printf("triangles %zu materials %zu\n", m->triangles.size(), m->materials.size());
      delete m;
      return 0;
    }
    if (parse_only) {
      // OBJ front end only (no device needed): prints what load_mesh produced.
      if (obj.empty()) throw Error("--parse-only needs --obj");
      Mesh* m = load_obj(obj);
      std::printf("triangles %zu materials %zu\n", m->triangles.size(), m->materials.size());
      for (size_t i = 0; i < m->triangles.size(); i++) {
        const Triangle& t = m->triangles[i];
        std::printf("%u %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", m->material_idxs[i], t.a.x, t.a.y, t.a.z,
                    t.b.x, t.b.y, t.b.z, t.c.x, t.c.y, t.c.z);
      }
      delete m;
      return 0;
    }
    rtx_config cfg = default_config(width, height);
    cfg.sky = sky ? 1 : 0;
    const bool ranks = !rendezvous.empty();  // started by spawn_ranks: one GPU per rank, sums meet on rank 0
    if (ranks) cfg.device = rk.rank;

    auto t1 = std::chrono::high_resolution_clock::now();
    Scene scene(cfg);
    Mesh* mesh = obj.empty() ? procedural_cornell() : scene.load_mesh(obj);
    // buildScene's overrides (Raytracer.cpp:151-171)
    Material diffuseMat{Vec3(0.93f, 0.93f, 0.93f), 0.0f, diffuse};
    Material redMat{Vec3(0.55f, 0.09f, 0.09f), 0.0f, diffuse};
    Material greenMat{Vec3(0.16f, 0.55f, 0.15f), 0.0f, diffuse};
    Material specularMat{Vec3(1.0f), 0.0f, specular};
    if (mesh->materials.size() >= 8) {
      mesh->materials[0] = redMat;
      mesh->materials[1] = greenMat;
      for (int k = 2; k <= 6; k++) mesh->materials[k] = diffuseMat;
      mesh->materials[7].emission = 10.0f;
    } else {
      for (Material& m : mesh->materials) m = diffuseMat;
    }
    scene.addMesh(mesh);
    scene.spheres.push_back(Sphere(Vec3(-0.5f, 0.302f, 0.55f), 0.3f));
    scene.spheres[0].material = specularMat;
    scene.commit();
    if (ranks) comm_join(scene, rk, rendezvous);
    float seconds = std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - t1).count();
    if (root)
      std::printf("Scene built in %f seconds (%zu triangles, %zu materials%s)\n", seconds, mesh->triangles.size(),
                  mesh->materials.size(), ranks ? (", " + std::to_string(rk.world) + " ranks").c_str() : "");

    Img img(width, height);
    clear(img, scene);
    if (!resume.empty()) {
      // every rank learns the sample count; only rank 0's buffer carries the sums on
      load_checkpoint(img, scene, resume.c_str());
      if (root) std::printf("Resumed %s at %u samples per pixel\n", resume.c_str(), img.sample_count);
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    for (uint32_t p = 0; p < passes; p++) {
      auto tp = std::chrono::high_resolution_clock::now();
      if (ranks)
        render_ranks(img, scene, rk, spp);
      else if (in_flight)
        render_in_flight(img, scene, spp);
      else
        render(img, scene, spp);
      if (!root) continue;
      float s = std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - tp).count();
      // the reference's per-pass report (Raytracer.cpp:281-285, 322-323)
      std::printf("Took %f seconds\nCast %llu rays\n%f samples per second \n", s,
                  (unsigned long long)spp * width * height, (double)spp * width * height / s);
      float total = std::chrono::duration<float>(std::chrono::high_resolution_clock::now() - t0).count();
      std::printf("Accumulated %u samples per pixel over %f seconds\n\n", img.sample_count, total);
      if (!checkpoint.empty() && checkpoint_every && (p + 1) % checkpoint_every == 0 && p + 1 < passes) {
        if (in_flight) refresh(img, scene);
        save_checkpoint(img, scene, checkpoint.c_str());
      }
    }
    if (in_flight && !ranks) refresh(img, scene);
    rtx_stats st;
    check(rtx_get_stats(scene.ctx(), &st), "rtx_get_stats");
    if (!root) {
      std::printf("rank %d: paths %llu, path-bounces %llu, shadow rays %llu\n", rk.rank, (unsigned long long)st.paths,
                  (unsigned long long)st.bounces, (unsigned long long)st.shadow_rays);
      return 0;
    }
    if (!checkpoint.empty()) save_checkpoint(img, scene, checkpoint.c_str());
    std::printf("paths %llu, path-bounces %llu, shadow rays %llu\n", (unsigned long long)st.paths,
                (unsigned long long)st.bounces, (unsigned long long)st.shadow_rays);
    if (!img.savePFM((out + ".pfm").c_str()) || !img.savePPM((out + ".ppm").c_str()) ||
        !img.saveEXR((out + ".exr").c_str()))
      throw Error("cannot write " + out + ".pfm/.ppm/.exr");
    std::printf("Saved %s.pfm, %s.ppm and %s.exr\n", out.c_str(), out.c_str(), out.c_str());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
