// render_scene.cpp — what the reference's main() (src/main.rs:28-67) looks like on top of the MI355X backend:
// ImageSpecBuilder -> CameraBuilder -> scene builder -> Scene::render with a tile receiver -> image.ppm.
// The SDL window (src/ui.rs) is replaced by a tile counter; everything else is the reference's host API.
//
//   render_scene <scene> [width] [spp] [depth] [out.ppm] [passes]
//     scene: composition | book_cover | two_spheres | something_blocky | cornell_box | cornell_smoke
//     passes > 1 uses Camera::render_progressive (tiles re-sent after every sample pass, as the preview expects)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../raytracer_amd/host/raytracer.hpp"

using namespace raytracer;

int main(int argc, char** argv) {
  const std::string name = argc > 1 ? argv[1] : "book_cover";
  const size_t width = argc > 2 ? (size_t)std::atol(argv[2]) : 3840 / 3;  // main.rs:30
  const size_t spp = argc > 3 ? (size_t)std::atol(argv[3]) : 36;         // main.rs:39: uniform_sampler(6^2)
  const size_t depth = argc > 4 ? (size_t)std::atol(argv[4]) : 16;       // main.rs:40
  const std::string out = argc > 5 ? argv[5] : "image.ppm";              // mod.rs:115
  const size_t passes = argc > 6 ? (size_t)std::atol(argv[6]) : 1;
  rt_ctx* ctx = nullptr;
  if (rt_init(0, &ctx) != RT_OK) {
    std::fprintf(stderr, "rt_init failed: %s\n", rt_last_error(ctx));
    return 2;
  }
  int rc = 0;
  try {
    ImageSpec image_spec = ImageSpecBuilder().width(width).aspect_ratio(16.0 / 9.0).build();  // main.rs:29-34
    CameraBuilder camera;
    camera.image_spec(image_spec).uniform_sampler(spp).max_ray_depth(depth).defocus_angle(0.2);  // main.rs:36-44
    Scene scene = name == "composition"        ? composition(camera)
                  : name == "book_cover"       ? book_cover(camera)
                  : name == "two_spheres"      ? two_spheres(camera)
                  : name == "something_blocky" ? something_blocky(camera)
                  : name == "cornell_box"      ? cornell_box(camera)
                  : name == "cornell_smoke"    ? cornell_smoke(camera)
                                               : throw Error("unknown scene: " + name);
    size_t tiles = 0;
    TileSender sender = [&](Tile&&) { tiles++; return true; };  // ui.rs:56-77 would paint the tile here
    RenderTarget target;
    target.ctx = ctx;
    target.ppm_path = out;
    const auto start_time = std::chrono::steady_clock::now();  // main.rs:52
    if (passes > 1) scene.camera.render_progressive(*scene.world, sender, target, passes);
    else scene.render(sender, target);
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - start_time).count();
    std::printf("Done in %.3f seconds\n", elapsed);  // main.rs:58
    std::printf("%zu x %zu, %zu spp, depth %zu, %zu tile messages -> %s\n", scene.camera.image_width,
                scene.camera.image_height, scene.camera.samples_per_pixel(), depth, tiles, out.c_str());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    rc = 1;
  }
  rt_shutdown(ctx);
  return rc;
}
