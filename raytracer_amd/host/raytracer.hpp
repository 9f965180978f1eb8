// raytracer.hpp — host-side mirror of the reference's scene / camera / hittable / material API (C++17),
// sitting above the C ABI of include/rt_abi.h.  Same names, argument meaning and error behaviour as the
// reference so that scene builders read like src/scene.rs; the only behavioural change is that
// `Camera::render` (src/camera/mod.rs:54-117) runs on the MI355X through librt_mi355x instead of on CPU threads.
//
// This is PRODUCT code: it never includes or links anything under oracle/.
//
// Reference types are opaque trait objects (src/hittable/mod.rs:26-29, materials.rs:24-26, texture.rs:15-17);
// flattening needs one extra method per trait — `describe()` — which is what a Rust maintainer would add
// (INTEGRATION.md).  Panics of the reference (`expect`, `panic!`) become `raytracer::Error` exceptions here;
// they are caught at the C facade (host_capi.cpp) and never cross a C boundary.
#pragma once
#include <cmath>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/rt_abi.h"

namespace raytracer {

struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ------------------------------------------------------------------ random.rs (needed by book_cover's layout)
class Rng {  // src/random.rs:3-62
 public:
  Rng() : s_{0x853c49e6748fea9bULL, 0xda3e39cb94b95bdbULL} {}
  static Rng from_seed(uint64_t a, uint64_t b) { Rng r; r.s_[0] = a; r.s_[1] = b; return r; }
  uint64_t next_u64();
  double next_f64() { return (double)next_u64() / 18446744073709551616.0; }
  double next_f64_range(double start, double end) { return next_f64() * (end - start) + start; }
  Rng& short_jump();
  Rng& long_jump();
  const uint64_t* state() const { return s_; }

 private:
  Rng& jump_impl(const uint64_t jumper[2]);
  uint64_t s_[2];
};

// ------------------------------------------------------------------ vec3.rs / color.rs (what scenes need)
struct Vec3 {
  double x = 0, y = 0, z = 0;
  Vec3() = default;
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double dot(const Vec3& r) const { return x * r.x + y * r.y + z * r.z; }
  Vec3 cross(const Vec3& r) const { return {y * r.z - z * r.y, z * r.x - x * r.z, x * r.y - y * r.x}; }
  double length_squared() const { return x * x + y * y + z * z; }
  double length() const { return std::sqrt(length_squared()); }
  Vec3 unit_vector() const { double l = length(); return {x / l, y / l, z / l}; }
  double distance(const Vec3& o) const { return Vec3(x - o.x, y - o.y, z - o.z).length(); }
};
using Point3 = Vec3;
inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3 operator/(Vec3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }

struct Color {
  double r = 0, g = 0, b = 0;
  Color() = default;
  Color(double r_, double g_, double b_) : r(r_), g(g_), b(b_) {}
  static Color gray(double v) { return {v, v, v}; }
  static Color from_hex(uint32_t hex) {
    return {((hex >> 16) & 0xff) / 255.0, ((hex >> 8) & 0xff) / 255.0, (hex & 0xff) / 255.0};
  }
  static Color white() { return {1, 1, 1}; }
  static Color black() { return {0, 0, 0}; }
  static Color random(Rng& rng) {
    double a = rng.next_f64(), b = rng.next_f64(), c = rng.next_f64();
    return {a, b, c};
  }
  std::tuple<uint8_t, uint8_t, uint8_t> into_u8() const;  // color.rs:68-73
};
inline Color operator*(Color a, Color b) { return {a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Color operator/(Color a, double s) { return {a.r / s, a.g / s, a.b / s}; }
inline Color operator+(Color a, double s) { return {a.r + s, a.g + s, a.b + s}; }

// ------------------------------------------------------------------ flattening target
struct Xform {  // p_world = R p + T (row-major R); instances are baked into the primitives while flattening
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double T[3] = {0, 0, 0};
  Vec3 point(const Vec3& p) const {
    return {R[0] * p.x + R[1] * p.y + R[2] * p.z + T[0], R[3] * p.x + R[4] * p.y + R[5] * p.z + T[1],
            R[6] * p.x + R[7] * p.y + R[8] * p.z + T[2]};
  }
  Vec3 vec(const Vec3& v) const {
    return {R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z,
            R[6] * v.x + R[7] * v.y + R[8] * v.z};
  }
};
struct FlatScene {
  Xform xf;  // current instance transform (identity at top level)
  std::vector<rt_prim> prims;
  std::vector<rt_material> materials;
  std::vector<rt_texture> textures;
  std::vector<uint8_t> texels;
  std::map<const void*, int32_t> material_ids, texture_ids;  // dedupe shared Arc<>s
  rt_scene_desc desc() const;
};

// ------------------------------------------------------------------ hittable/texture.rs
class Texture {  // texture.rs:15-17 + describe()
 public:
  virtual ~Texture() = default;
  virtual int32_t describe(FlatScene& out) const = 0;
};
class SolidColor : public Texture {  // texture.rs:20-36
 public:
  explicit SolidColor(Color c) : color_(c) {}
  int32_t describe(FlatScene& out) const override;

 private:
  Color color_;
};
class CheckerTexture : public Texture {  // texture.rs:39-66 — new(scale, odd, even)
 public:
  CheckerTexture(double scale, std::shared_ptr<Texture> odd, std::shared_ptr<Texture> even)
      : inv_scale_(1.0 / scale), odd_(std::move(odd)), even_(std::move(even)) {}
  int32_t describe(FlatScene& out) const override;

 private:
  double inv_scale_;
  std::shared_ptr<Texture> odd_, even_;
};
struct RgbaImage {  // image::RgbaImage: what image::open(..).to_rgba8() yields (scene.rs:250)
  uint32_t width = 0, height = 0;
  std::vector<uint8_t> data;
};
class ImageTexture : public Texture {  // texture.rs:69-93
 public:
  explicit ImageTexture(RgbaImage image) : image_(std::move(image)) {}
  int32_t describe(FlatScene& out) const override;

 private:
  RgbaImage image_;
};
class NoiseTexture : public Texture {  // texture.rs:96-146
 public:
  explicit NoiseTexture(double scale) : inv_scale_(1.0 / scale) {}
  int32_t describe(FlatScene& out) const override;

 private:
  double inv_scale_;
};

// ------------------------------------------------------------------ hittable/materials.rs
class Material {  // materials.rs:24-26 + describe()
 public:
  virtual ~Material() = default;
  virtual int32_t describe(FlatScene& out) const = 0;
};
class Lambertian : public Material {  // materials.rs:14-16,45-56
 public:
  explicit Lambertian(std::shared_ptr<Texture> albedo) : albedo(std::move(albedo)) {}
  explicit Lambertian(Color c) : albedo(std::make_shared<SolidColor>(c)) {}
  int32_t describe(FlatScene& out) const override;
  std::shared_ptr<Texture> albedo;
};
class Metal : public Material {  // materials.rs:59-86
 public:
  Metal(Color albedo, double fuzz) : albedo_(albedo), fuzz_(fuzz) {}
  explicit Metal(Color albedo) : Metal(albedo, 0.) {}
  int32_t describe(FlatScene& out) const override;

 private:
  Color albedo_;
  double fuzz_;
};
class DiffuseLight : public Material {  // EXT (SURVEY aX4): emits, never scatters
 public:
  explicit DiffuseLight(Color emit) : emit_(std::make_shared<SolidColor>(emit)) {}
  explicit DiffuseLight(std::shared_ptr<Texture> emit) : emit_(std::move(emit)) {}
  int32_t describe(FlatScene& out) const override;

 private:
  std::shared_ptr<Texture> emit_;
};
class Isotropic : public Material {  // EXT (SURVEY aX3): phase function of a constant medium
 public:
  explicit Isotropic(Color albedo) : albedo_(std::make_shared<SolidColor>(albedo)) {}
  int32_t describe(FlatScene& out) const override;

 private:
  std::shared_ptr<Texture> albedo_;
};
class Dielectric : public Material {  // materials.rs:89-102
 public:
  explicit Dielectric(double index_of_refraction) : ior_(index_of_refraction) {}
  int32_t describe(FlatScene& out) const override;

 private:
  double ior_;
};

// ------------------------------------------------------------------ hittable/{mod,geometry,containers}.rs
class Hittable {  // hittable/mod.rs:26-29 + describe()
 public:
  virtual ~Hittable() = default;
  virtual void describe(FlatScene& out) const = 0;
  // EXT: describe this object as the convex boundary of a constant medium (sphere or box, possibly instanced)
  virtual void describe_boundary(FlatScene&, double /*density*/, int32_t /*material*/) const {
    throw Error("ConstantMedium: boundary must be a Sphere or a box (optionally under Translate / RotateY)");
  }
};
class Sphere : public Hittable {  // geometry.rs:12-28
 public:
  Sphere(Point3 center, double radius, std::shared_ptr<Material> material)
      : center(center), radius(radius), material(std::move(material)) {}
  void describe(FlatScene& out) const override;
  void describe_boundary(FlatScene& out, double density, int32_t material) const override;
  Point3 center;
  double radius;
  std::shared_ptr<Material> material;
};
// ---- EXT geometry (absent from the reference; "The Next Week" definitions in the reference's conventions)
class Quad : public Hittable {  // SURVEY aX1
 public:
  Quad(Point3 Q, Vec3 u, Vec3 v, std::shared_ptr<Material> material) : Q(Q), u(u), v(v), material(std::move(material)) {}
  void describe(FlatScene& out) const override;
  Point3 Q;
  Vec3 u, v;
  std::shared_ptr<Material> material;
};
class Translate : public Hittable {  // SURVEY aX2
 public:
  Translate(std::unique_ptr<Hittable> object, Vec3 offset) : object_(std::move(object)), offset_(offset) {}
  void describe(FlatScene& out) const override;
  void describe_boundary(FlatScene& out, double density, int32_t material) const override;

 private:
  std::unique_ptr<Hittable> object_;
  Vec3 offset_;
};
class RotateY : public Hittable {  // SURVEY aX2
 public:
  RotateY(std::unique_ptr<Hittable> object, double degrees) : object_(std::move(object)), degrees_(degrees) {}
  void describe(FlatScene& out) const override;
  void describe_boundary(FlatScene& out, double density, int32_t material) const override;

 private:
  std::unique_ptr<Hittable> object_;
  double degrees_;
};
class ConstantMedium : public Hittable {  // SURVEY aX3
 public:
  ConstantMedium(std::unique_ptr<Hittable> boundary, double density, Color albedo)
      : boundary_(std::move(boundary)), density_(density), phase_(std::make_shared<Isotropic>(albedo)) {}
  void describe(FlatScene& out) const override;

 private:
  std::unique_ptr<Hittable> boundary_;
  double density_;
  std::shared_ptr<Material> phase_;
};
class MovingSphere : public Hittable {  // geometry.rs:84-100
 public:
  MovingSphere(Sphere sphere, Point3 destination) : sphere(std::move(sphere)), destination(destination) {}
  void describe(FlatScene& out) const override;
  Sphere sphere;
  Point3 destination;
};
class HittableList : public Hittable {  // containers.rs:13-26
 public:
  void add(std::unique_ptr<Hittable> object) { objects.push_back(std::move(object)); }
  // containers.rs:23-25.  The reference builds its mean-split BVH here; closest-hit results do not depend on
  // topology, so the GPU library builds its own SAH tree at upload and this only marks the world as BVH-backed.
  std::unique_ptr<Hittable> into_bvh();
  void describe(FlatScene& out) const override;
  std::vector<std::unique_ptr<Hittable>> objects;
};
class BoxList;
std::unique_ptr<BoxList> box(Point3 a, Point3 b, std::shared_ptr<Material> material);  // EXT: six quads
class BoxList : public HittableList {
 public:
  void describe_boundary(FlatScene& out, double density, int32_t material) const override;
  Point3 min_corner, max_corner;
};
class BVHNode : public Hittable {  // containers.rs:48-52 (opaque on the host; see into_bvh)
 public:
  explicit BVHNode(std::vector<std::unique_ptr<Hittable>> objects) : objects_(std::move(objects)) {}
  void describe(FlatScene& out) const override;

 private:
  std::vector<std::unique_ptr<Hittable>> objects_;
};

// ------------------------------------------------------------------ camera (image.rs [missing], builder.rs, mod.rs)
struct ImageSpec {  // src/camera/image.rs is missing upstream; contract inferred from its uses
  size_t width = 0, height = 0;
  double aspect_ratio = 1.0;
};
class ImageSpecBuilder {  // main.rs:28-33
 public:
  ImageSpecBuilder& width(size_t w) { width_ = w; return *this; }
  ImageSpecBuilder& aspect_ratio(double a) { aspect_ = a; return *this; }
  ImageSpec build() const;  // height = trunc(width / aspect), at least 1

 private:
  std::optional<size_t> width_;
  std::optional<double> aspect_;
};

enum class PixelSamplerKind { Uniform, Random };
struct PixelSampler {  // mod.rs:21-24
  PixelSamplerKind kind;
  size_t n;
};

using Tile = std::tuple<std::pair<size_t, size_t>, std::pair<size_t, size_t>, std::vector<Color>>;
// Stand-in for SyncSender<((usize,usize),(usize,usize),Vec<Color>)> (mod.rs:57): returns false when the
// receiver has hung up (send() == Err), which cancels the render (mod.rs:110-113).
using TileSender = std::function<bool(Tile&&)>;

class Camera;
class CameraBuilder {  // builder.rs:8-54
 public:
  CameraBuilder& image_spec(ImageSpec v) { image_spec_ = v; return *this; }
  CameraBuilder& max_ray_depth(size_t v) { max_ray_depth_ = v; return *this; }
  CameraBuilder& field_of_view(double v) { field_of_view_ = v; return *this; }
  CameraBuilder& lookfrom(Point3 v) { lookfrom_ = v; return *this; }
  CameraBuilder& lookat(Point3 v) { lookat_ = v; return *this; }
  CameraBuilder& up_vector(Vec3 v) { up_vector_ = v; return *this; }
  CameraBuilder& defocus_angle(double v) { defocus_angle_ = v; return *this; }
  CameraBuilder& focus_distance(double v) { focus_distance_ = v; return *this; }
  CameraBuilder& uniform_sampler(size_t spp) { sampler_ = PixelSampler{PixelSamplerKind::Uniform, spp}; return *this; }
  CameraBuilder& random_sampler(size_t spp) { sampler_ = PixelSampler{PixelSamplerKind::Random, spp}; return *this; }
  CameraBuilder& background(Color c) { background_ = c; return *this; }  // EXT (aX4): solid background instead of the sky
  Camera build() const;  // builder.rs:55-134; throws Error where the reference panics

 private:
  std::optional<ImageSpec> image_spec_;
  std::optional<PixelSampler> sampler_;
  std::optional<size_t> max_ray_depth_;
  std::optional<double> field_of_view_, defocus_angle_, focus_distance_;
  std::optional<Point3> lookfrom_, lookat_;
  std::optional<Vec3> up_vector_;
  std::optional<Color> background_;
};

// Binding of a Camera to a GPU context: which ctx renders, the RNG seed, where the `image.ppm` goes
// (mod.rs:115,216-227; empty = do not write) and which residue class of the sample indices this
// process renders (multi-GPU: rank r of n renders samples s % n == r; the caller reduces the partial sums).
struct RenderTarget {
  rt_ctx* ctx = nullptr;
  uint64_t seed = 0x853c49e6748fea9bULL;  // the reference's default RNG seed word (random.rs:11)
  std::string ppm_path;
  uint32_t rank = 0, world_size = 1;
};

class Camera {  // mod.rs:26-51
 public:
  size_t image_width = 0, image_height = 0;
  // mod.rs:54-117: flatten -> rt_scene_upload -> rt_render -> rt_resolve -> the same 32x32 tile tuples
  // (gamma-corrected f64, row-major) in index order -> optional image.ppm.  Prints "cancelled" and returns
  // without writing the file when the sender reports a hang-up.
  void render(const Hittable& world, const TileSender& sender, const RenderTarget& target) const;
  // un-normalised linear sums of this rank's sample subset (for an external multi-GPU reduce)
  std::vector<float> render_partial_sums(const Hittable& world, const RenderTarget& target) const;
  // Progressive form of render() for the live preview (the reference streams tiles to ui.rs:56-77 as they
  // finish; a GPU frame finishes all tiles at once, so the preview is refreshed per PASS instead): the samples
  // are split into `passes` residue classes, and after each pass every tile is sent with the mean of the samples
  // rendered so far (same tuples as render()).  A hang-up prints "cancelled" and returns.  When
  // `checkpoint_path` is non-empty the accumulated sums are written there after every pass (atomically) and a
  // matching checkpoint found at start is resumed — the 10 000-spp runs of C5 survive a restart.  The final
  // frame equals render()'s up to f32 summation order (counter-based sample keys).  Returns passes rendered now.
  size_t render_progressive(const Hittable& world, const TileSender& sender, const RenderTarget& target, size_t passes,
                            const std::string& checkpoint_path = "") const;
  rt_camera abi_camera() const;
  size_t samples_per_pixel() const;
  PixelSampler pixel_sampler{PixelSamplerKind::Uniform, 1};
  size_t depth = 0;
  bool solid_background = false;  // EXT: false = the reference's sky gradient (mod.rs:205-207)
  Color background;
  double aspect_ratio = 1, field_of_view = 90, defocus_angle = 0, focus_distance = 1;
  Point3 lookfrom, lookat, center, pixel00_loc;
  Vec3 up_vector, pixel_delta_u, pixel_delta_v, u, v, w, defocus_disk_u, defocus_disk_v;
  static void write_buffer_to_file(const std::string& path, const std::vector<Color>& image_buffer, size_t width,
                                   size_t height);  // mod.rs:216-227
};

// ------------------------------------------------------------------ scene.rs
struct Scene {  // scene.rs:21-33
  Camera camera;
  std::unique_ptr<Hittable> world;
  void render(const TileSender& sender, const RenderTarget& target) const { camera.render(*world, sender, target); }
};
Scene composition(CameraBuilder camera_builder);       // scene.rs:35-87
Scene book_cover(CameraBuilder camera_builder);        // scene.rs:89-164
Scene two_spheres(CameraBuilder camera_builder);       // scene.rs:216-241
Scene earth(CameraBuilder camera_builder, RgbaImage earth_image);  // scene.rs:243-260 (image decoded by the caller)
Scene something_blocky(CameraBuilder camera_builder);  // scene.rs:262-285
// EXT scenes (not in the reference; book scenes built with the reference's types + the EXT types above)
Scene cornell_box(CameraBuilder camera_builder);                       // quads + two rotated boxes + light
Scene cornell_smoke(CameraBuilder camera_builder);                     // the boxes as constant media
Scene final_scene(CameraBuilder camera_builder, RgbaImage earth_image, int boxes_per_side, int cluster_spheres);

}  // namespace raytracer
