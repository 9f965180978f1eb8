// host_capi.cpp — C facade over the C++ host layer (raytracer.hpp) so that Python (ctypes) tests, bench.py and
// smoke() drive the same objects a C++ application would.  Exceptions are caught here; nothing unwinds
// across the C boundary.
#include <cstring>
#include <string>

#include "raytracer.hpp"

using namespace raytracer;

struct rth_scene {
  Scene scene;
  FlatScene flat;
  rt_scene_desc desc;
  rt_camera cam;
  std::string name;
};

static thread_local std::string g_err;

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

// name: composition | book_cover | two_spheres | earth | something_blocky
// sampler_kind: 0 uniform_sampler(spp) / 1 random_sampler(spp); defocus_override < 0 => not set on the builder
// (the reference's main.rs:41 sets it before the scene builder, which may override it).
// texels (RGBA8, tw x th) feed `earth`.  Returns NULL on error (see rth_last_error).
rth_scene* rth_scene_create(const char* name, uint64_t width, double aspect, int sampler_kind, uint64_t spp,
                            uint64_t depth, double defocus_override, const uint8_t* texels, uint32_t tw, uint32_t th) {
  try {
    ImageSpec spec = ImageSpecBuilder().width((size_t)width).aspect_ratio(aspect).build();
    CameraBuilder cb;
    cb.image_spec(spec).max_ray_depth((size_t)depth);
    if (sampler_kind == 0) cb.uniform_sampler((size_t)spp);
    else cb.random_sampler((size_t)spp);
    if (defocus_override >= 0) cb.defocus_angle(defocus_override);
    auto h = std::make_unique<rth_scene>();
    std::string n(name ? name : "");
    h->name = n;
    if (n == "composition") h->scene = composition(cb);
    else if (n == "book_cover") h->scene = book_cover(cb);
    else if (n == "two_spheres") h->scene = two_spheres(cb);
    else if (n == "something_blocky") h->scene = something_blocky(cb);
    else if (n == "cornell_box") h->scene = cornell_box(cb);
    else if (n == "cornell_smoke") h->scene = cornell_smoke(cb);
    else if (n == "earth" || n == "final_scene" || n == "final_scene_small") {
      RgbaImage img;
      if (texels) {
        img.width = tw;
        img.height = th;
        img.data.assign(texels, texels + (size_t)4 * tw * th);
      }
      if (n == "earth") h->scene = earth(cb, std::move(img));
      else if (n == "final_scene") h->scene = final_scene(cb, std::move(img), 20, 1000);
      else h->scene = final_scene(cb, std::move(img), 6, 60);
    } else {
      throw Error("unknown scene: " + n);
    }
    h->scene.world->describe(h->flat);
    h->desc = h->flat.desc();
    h->cam = h->scene.camera.abi_camera();
    return h.release();
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  } catch (...) {
    g_err = "unknown exception";
    return nullptr;
  }
}
void rth_scene_free(rth_scene* s) { delete s; }
const rt_scene_desc* rth_scene_desc(rth_scene* s) { return &s->desc; }
const rt_camera* rth_scene_camera(rth_scene* s) { return &s->cam; }
uint64_t rth_scene_spp(rth_scene* s) { return s->scene.camera.samples_per_pixel(); }

typedef int (*rth_tile_cb)(void* user, uint64_t top, uint64_t left, uint64_t h, uint64_t w, const double* rgb);

// Scene::render (scene.rs:30-32) -> Camera::render on the GPU.  `cb` receives every tile message
// ((top,left),(h,w), h*w*3 gamma-corrected f64) and returns 0 to hang up (=> "cancelled").  Returns 0 on success.
int rth_scene_render(rth_scene* s, rt_ctx* ctx, uint64_t seed, const char* ppm_path, rth_tile_cb cb, void* user) {
  try {
    RenderTarget t;
    t.ctx = ctx;
    t.seed = seed;
    if (ppm_path) t.ppm_path = ppm_path;
    TileSender sender = [&](Tile&& tile) -> bool {
      if (!cb) return true;
      auto& [tl, rect, colors] = tile;
      static_assert(sizeof(Color) == 3 * sizeof(double), "Color must be 3 packed doubles");
      return cb(user, tl.first, tl.second, rect.first, rect.second, reinterpret_cast<const double*>(colors.data())) != 0;
    };
    s->scene.render(sender, t);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  } catch (...) {
    g_err = "unknown exception";
    return -1;
  }
}

// Camera::render_progressive: tiles after every pass, optional checkpoint/resume.  *passes_rendered (may be NULL)
// receives the number of passes rendered by this call.
int rth_scene_render_progressive(rth_scene* s, rt_ctx* ctx, uint64_t seed, uint64_t passes, const char* checkpoint_path,
                                 const char* ppm_path, rth_tile_cb cb, void* user, uint64_t* passes_rendered) {
  try {
    RenderTarget t;
    t.ctx = ctx;
    t.seed = seed;
    if (ppm_path) t.ppm_path = ppm_path;
    TileSender sender = [&](Tile&& tile) -> bool {
      if (!cb) return true;
      auto& [tl, rect, colors] = tile;
      return cb(user, tl.first, tl.second, rect.first, rect.second, reinterpret_cast<const double*>(colors.data())) != 0;
    };
    size_t n = s->scene.camera.render_progressive(*s->scene.world, sender, t, (size_t)passes,
                                                  checkpoint_path ? checkpoint_path : "");
    if (passes_rendered) *passes_rendered = n;
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  } catch (...) {
    g_err = "unknown exception";
    return -1;
  }
}

// un-normalised linear sums of rank's sample subset (W*H*3 f32) for an external reduce
int rth_scene_render_partial(rth_scene* s, rt_ctx* ctx, uint64_t seed, uint32_t rank, uint32_t world_size, float* out) {
  try {
    RenderTarget t;
    t.ctx = ctx;
    t.seed = seed;
    t.rank = rank;
    t.world_size = world_size;
    std::vector<float> sums = s->scene.camera.render_partial_sums(*s->scene.world, t);
    std::memcpy(out, sums.data(), sums.size() * sizeof(float));
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  } catch (...) {
    g_err = "unknown exception";
    return -1;
  }
}

// host Rng exposed for the KAT tests of the product-side scene generator (random.rs)
void rth_rng_next_f64_seq(uint64_t a, uint64_t b, int short_jumps, int n, double* out, uint64_t* state_out) {
  Rng r = Rng::from_seed(a, b);
  for (int i = 0; i < short_jumps; i++) r.short_jump();
  if (state_out) { state_out[0] = r.state()[0]; state_out[1] = r.state()[1]; }
  for (int i = 0; i < n; i++) out[i] = r.next_f64();
}

// Writes W x H gamma-corrected colours (row-major, 3 f64 each: what the tile messages carry) with the camera's
// writer: plain-text PPM as in the reference, or PNG when the path ends in ".png".  Needs no GPU.
int rth_write_image(const char* path, const double* rgb, uint64_t width, uint64_t height) {
  try {
    if (!path || !rgb) throw Error("rth_write_image: NULL argument");
    std::vector<Color> buf((size_t)(width * height));
    for (size_t i = 0; i < buf.size(); i++) buf[i] = Color(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
    Camera::write_buffer_to_file(path, buf, (size_t)width, (size_t)height);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  } catch (...) {
    g_err = "unknown exception";
    return -1;
  }
}

}  // extern "C"
