// raytracer.cpp — host-side mirror of the reference API over the C ABI (see raytracer.hpp).
#include "raytracer.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace raytracer {

// ------------------------------------------------------------------ random.rs
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t Rng::next_u64() {  // random.rs:18-28
  const uint64_t a = s_[0], b = s_[1];
  const uint64_t sum = a + b;
  const uint64_t c = a ^ b;
  s_[0] = rotl64(a, 24) ^ c ^ (c << 16);
  s_[1] = rotl64(c, 37);
  return sum;
}
Rng& Rng::jump_impl(const uint64_t jumper[2]) {  // random.rs:37-53
  uint64_t acc0 = 0, acc1 = 0;
  for (int word = 0; word < 2; ++word)
    for (int bit = 0; bit < 64; ++bit) {
      if ((jumper[word] >> bit) & 1ULL) { acc0 ^= s_[0]; acc1 ^= s_[1]; }
      next_u64();
    }
  s_[0] = acc0;
  s_[1] = acc1;
  return *this;
}
Rng& Rng::short_jump() {  // random.rs:54-57
  static const uint64_t k[2] = {0xdf900294d8f554a5ULL, 0x170865df4b3201fcULL};
  return jump_impl(k);
}
Rng& Rng::long_jump() {  // random.rs:58-61
  static const uint64_t k[2] = {0xd2a98b26625eee7bULL, 0xdddf9b1090aa7ac1ULL};
  return jump_impl(k);
}

static uint8_t sat_u8(double v) {  // Rust `as u8`
  if (!(v == v) || v <= 0.0) return 0;
  if (v >= 255.0) return 255;
  return (uint8_t)v;
}
std::tuple<uint8_t, uint8_t, uint8_t> Color::into_u8() const {
  return {sat_u8(256.0 * r), sat_u8(256.0 * g), sat_u8(256.0 * b)};
}

// ------------------------------------------------------------------ describe(): flatten the trait objects
rt_scene_desc FlatScene::desc() const {
  rt_scene_desc d;
  std::memset(&d, 0, sizeof(d));
  d.prims = prims.data();
  d.n_prims = prims.size();
  d.materials = materials.data();
  d.n_materials = materials.size();
  d.textures = textures.data();
  d.n_textures = textures.size();
  d.texels = texels.empty() ? nullptr : texels.data();
  d.n_texel_bytes = texels.size();
  return d;
}
static rt_texture blank_texture(uint32_t kind) {
  rt_texture t;
  std::memset(&t, 0, sizeof(t));
  t.kind = kind;
  t.odd = t.even = -1;
  return t;
}
template <class F>
static int32_t intern(std::map<const void*, int32_t>& ids, const void* key, F&& make) {
  auto it = ids.find(key);
  if (it != ids.end()) return it->second;
  int32_t id = make();
  ids[key] = id;
  return id;
}
int32_t SolidColor::describe(FlatScene& out) const {
  return intern(out.texture_ids, this, [&] {
    rt_texture t = blank_texture(RT_TEX_SOLID);
    t.rgb[0] = color_.r; t.rgb[1] = color_.g; t.rgb[2] = color_.b;
    out.textures.push_back(t);
    return (int32_t)out.textures.size() - 1;
  });
}
int32_t CheckerTexture::describe(FlatScene& out) const {
  return intern(out.texture_ids, this, [&] {
    // children first: the ABI requires checker children to precede the checker (acyclic by construction)
    int32_t o = odd_->describe(out), e = even_->describe(out);
    rt_texture t = blank_texture(RT_TEX_CHECKER);
    t.inv_scale = inv_scale_;
    t.odd = o;
    t.even = e;
    out.textures.push_back(t);
    return (int32_t)out.textures.size() - 1;
  });
}
int32_t ImageTexture::describe(FlatScene& out) const {
  return intern(out.texture_ids, this, [&] {
    rt_texture t = blank_texture(RT_TEX_IMAGE);
    t.width = image_.width;
    t.height = image_.height;
    t.texel_offset = out.texels.size();
    out.texels.insert(out.texels.end(), image_.data.begin(), image_.data.end());
    out.textures.push_back(t);
    return (int32_t)out.textures.size() - 1;
  });
}
int32_t NoiseTexture::describe(FlatScene& out) const {
  return intern(out.texture_ids, this, [&] {
    rt_texture t = blank_texture(RT_TEX_NOISE);
    t.inv_scale = inv_scale_;
    out.textures.push_back(t);
    return (int32_t)out.textures.size() - 1;
  });
}
static rt_material blank_material(uint32_t kind) {
  rt_material m;
  std::memset(&m, 0, sizeof(m));
  m.kind = kind;
  m.tex = -1;
  return m;
}
int32_t Lambertian::describe(FlatScene& out) const {
  return intern(out.material_ids, this, [&] {
    rt_material m = blank_material(RT_MAT_LAMBERTIAN);
    m.tex = albedo->describe(out);
    out.materials.push_back(m);
    return (int32_t)out.materials.size() - 1;
  });
}
int32_t Metal::describe(FlatScene& out) const {
  return intern(out.material_ids, this, [&] {
    rt_material m = blank_material(RT_MAT_METAL);
    m.albedo[0] = albedo_.r; m.albedo[1] = albedo_.g; m.albedo[2] = albedo_.b;
    m.fuzz = fuzz_;
    out.materials.push_back(m);
    return (int32_t)out.materials.size() - 1;
  });
}
int32_t DiffuseLight::describe(FlatScene& out) const {
  return intern(out.material_ids, this, [&] {
    rt_material m = blank_material(RT_MAT_DIFFUSE_LIGHT);
    m.tex = emit_->describe(out);
    out.materials.push_back(m);
    return (int32_t)out.materials.size() - 1;
  });
}
int32_t Isotropic::describe(FlatScene& out) const {
  return intern(out.material_ids, this, [&] {
    rt_material m = blank_material(RT_MAT_ISOTROPIC);
    m.tex = albedo_->describe(out);
    out.materials.push_back(m);
    return (int32_t)out.materials.size() - 1;
  });
}
int32_t Dielectric::describe(FlatScene& out) const {
  return intern(out.material_ids, this, [&] {
    rt_material m = blank_material(RT_MAT_DIELECTRIC);
    m.ior = ior_;
    out.materials.push_back(m);
    return (int32_t)out.materials.size() - 1;
  });
}
static rt_prim blank_prim(uint32_t type, int32_t object_id, int32_t material) {
  rt_prim p;
  std::memset(&p, 0, sizeof(p));
  p.type = type;
  p.object_id = object_id;
  p.material_id = (uint32_t)material;
  return p;
}
static void put3(double* dst, const Vec3& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }
void Sphere::describe(FlatScene& out) const {
  rt_prim p = blank_prim(RT_PRIM_SPHERE, (int32_t)out.prims.size(), material->describe(out));
  put3(p.a, out.xf.point(center));
  p.radius = radius;
  out.prims.push_back(p);
}
void Sphere::describe_boundary(FlatScene& out, double density, int32_t mat) const {
  rt_prim p = blank_prim(RT_PRIM_MEDIUM_SPHERE, (int32_t)out.prims.size(), mat);
  put3(p.a, out.xf.point(center));
  p.radius = radius;
  p.density = density;
  out.prims.push_back(p);
}
void MovingSphere::describe(FlatScene& out) const {
  rt_prim p = blank_prim(RT_PRIM_MOVING_SPHERE, (int32_t)out.prims.size(), sphere.material->describe(out));
  put3(p.a, out.xf.point(sphere.center));
  put3(p.b, out.xf.point(destination));
  p.radius = sphere.radius;
  out.prims.push_back(p);
}
void Quad::describe(FlatScene& out) const {
  rt_prim p = blank_prim(RT_PRIM_QUAD, (int32_t)out.prims.size(), material->describe(out));
  put3(p.a, out.xf.point(Q));
  put3(p.b, out.xf.vec(u));
  put3(p.c, out.xf.vec(v));
  out.prims.push_back(p);
}
namespace {
struct XformScope {  // push a child transform for the duration of a describe()
  FlatScene& out;
  Xform saved;
  XformScope(FlatScene& o, const Xform& child) : out(o), saved(o.xf) { out.xf = child; }
  ~XformScope() { out.xf = saved; }
};
Xform with_translate(const Xform& m, const Vec3& off) {  // m o translate(off)
  Xform r = m;
  Vec3 t = m.vec(off);
  r.T[0] += t.x; r.T[1] += t.y; r.T[2] += t.z;
  return r;
}
Xform with_rotate_y(const Xform& m, double degrees) {  // m o rotate_y(theta): x' = cos x + sin z, z' = -sin x + cos z
  const double pi = 3.14159265358979323846264338327950288;
  const double rad = degrees * (pi / 180.0), c = std::cos(rad), s = std::sin(rad);
  const double Q[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
  Xform r = m;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += m.R[3 * i + k] * Q[3 * k + j];
      r.R[3 * i + j] = acc;
    }
  return r;
}
}  // namespace
void Translate::describe(FlatScene& out) const {
  XformScope sc(out, with_translate(out.xf, offset_));
  object_->describe(out);
}
void Translate::describe_boundary(FlatScene& out, double density, int32_t mat) const {
  XformScope sc(out, with_translate(out.xf, offset_));
  object_->describe_boundary(out, density, mat);
}
void RotateY::describe(FlatScene& out) const {
  XformScope sc(out, with_rotate_y(out.xf, degrees_));
  object_->describe(out);
}
void RotateY::describe_boundary(FlatScene& out, double density, int32_t mat) const {
  XformScope sc(out, with_rotate_y(out.xf, degrees_));
  object_->describe_boundary(out, density, mat);
}
void ConstantMedium::describe(FlatScene& out) const {
  boundary_->describe_boundary(out, density_, phase_->describe(out));
}
std::unique_ptr<BoxList> box(Point3 a, Point3 b, std::shared_ptr<Material> m) {
  auto sides = std::make_unique<BoxList>();
  Point3 mn(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
  Point3 mx(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
  sides->min_corner = mn;
  sides->max_corner = mx;
  Vec3 dx(mx.x - mn.x, 0, 0), dy(0, mx.y - mn.y, 0), dz(0, 0, mx.z - mn.z);
  sides->add(std::make_unique<Quad>(Point3(mn.x, mn.y, mx.z), dx, dy, m));    // front
  sides->add(std::make_unique<Quad>(Point3(mx.x, mn.y, mx.z), -dz, dy, m));   // right
  sides->add(std::make_unique<Quad>(Point3(mx.x, mn.y, mn.z), -dx, dy, m));   // back
  sides->add(std::make_unique<Quad>(Point3(mn.x, mn.y, mn.z), dz, dy, m));    // left
  sides->add(std::make_unique<Quad>(Point3(mn.x, mx.y, mx.z), dx, -dz, m));   // top
  sides->add(std::make_unique<Quad>(Point3(mn.x, mn.y, mn.z), dx, dz, m));    // bottom
  return sides;
}
void BoxList::describe_boundary(FlatScene& out, double density, int32_t mat) const {
  rt_prim p = blank_prim(RT_PRIM_MEDIUM_BOX, (int32_t)out.prims.size(), mat);
  put3(p.a, out.xf.point(min_corner));
  put3(p.b, out.xf.vec(Vec3(max_corner.x - min_corner.x, 0, 0)));
  put3(p.c, out.xf.vec(Vec3(0, max_corner.y - min_corner.y, 0)));
  put3(p.d, out.xf.vec(Vec3(0, 0, max_corner.z - min_corner.z)));
  p.density = density;
  out.prims.push_back(p);
}
void HittableList::describe(FlatScene& out) const {
  for (const auto& o : objects) o->describe(out);
}
std::unique_ptr<Hittable> HittableList::into_bvh() { return std::make_unique<BVHNode>(std::move(objects)); }
void BVHNode::describe(FlatScene& out) const {
  for (const auto& o : objects_) o->describe(out);
}

// ------------------------------------------------------------------ camera
ImageSpec ImageSpecBuilder::build() const {
  if (!width_) throw Error("The image width must be set");
  ImageSpec s;
  s.width = *width_;
  s.aspect_ratio = aspect_.value_or(1.0);
  size_t h = (size_t)((double)s.width / s.aspect_ratio);
  s.height = h < 1 ? 1 : h;
  return s;
}

Camera CameraBuilder::build() const {  // builder.rs:55-134
  if (!image_spec_) throw Error("The image specifications must be set");
  if (!sampler_) throw Error("The samples per pixel must be set");
  Camera c;
  if (sampler_->kind == PixelSamplerKind::Uniform) {
    const double root = std::sqrt((double)sampler_->n);
    if (root != std::floor(root))
      throw Error("samples_per_pixel in the grid sampler must be a square number, current value: " +
                  std::to_string(sampler_->n));
    c.pixel_sampler = PixelSampler{PixelSamplerKind::Uniform, (size_t)root};
  } else {
    c.pixel_sampler = *sampler_;
  }
  if (!max_ray_depth_) throw Error("The depth must be set");
  c.depth = *max_ray_depth_;
  c.field_of_view = field_of_view_.value_or(90.0);
  c.lookfrom = lookfrom_.value_or(Point3(0., 0., 0.));
  c.lookat = lookat_.value_or(Point3(0., 0., -1.));
  c.up_vector = up_vector_.value_or(Vec3(0., 1., 0.));
  c.defocus_angle = defocus_angle_.value_or(0.0);
  c.focus_distance = focus_distance_.value_or(c.lookfrom.distance(c.lookat));
  c.aspect_ratio = image_spec_->aspect_ratio;
  c.image_width = image_spec_->width;
  c.image_height = image_spec_->height;

  const double pi = 3.14159265358979323846264338327950288;
  c.center = c.lookfrom;
  const double h = std::tan(c.field_of_view * (pi / 180.0) / 2.0);
  const double viewport_height = 2.0 * h * c.focus_distance;
  const double viewport_width = viewport_height * (double)c.image_width / (double)c.image_height;
  c.w = (c.lookfrom - c.lookat).unit_vector();
  c.u = c.up_vector.cross(c.w).unit_vector();
  c.v = c.w.cross(c.u);
  const Vec3 viewport_u = viewport_width * c.u;
  const Vec3 viewport_v = viewport_height * (-c.v);
  c.pixel_delta_u = viewport_u / (double)c.image_width;
  c.pixel_delta_v = viewport_v / (double)c.image_height;
  const Vec3 upper_left = c.center - (c.focus_distance * c.w) - viewport_u / 2. - viewport_v / 2.;
  c.pixel00_loc = upper_left + 0.5 * (c.pixel_delta_u + c.pixel_delta_v);
  const double defocus_radius = c.focus_distance * std::tan((c.defocus_angle / 2.0) * (pi / 180.0));
  c.defocus_disk_u = defocus_radius * c.u;
  c.defocus_disk_v = defocus_radius * c.v;
  if (background_) {
    c.solid_background = true;
    c.background = *background_;
  }
  return c;
}

size_t Camera::samples_per_pixel() const {
  return pixel_sampler.kind == PixelSamplerKind::Uniform ? pixel_sampler.n * pixel_sampler.n : pixel_sampler.n;
}

rt_camera Camera::abi_camera() const {
  rt_camera c;
  std::memset(&c, 0, sizeof(c));
  c.width = (uint32_t)image_width;
  c.height = (uint32_t)image_height;
  c.sampler_kind = pixel_sampler.kind == PixelSamplerKind::Uniform ? RT_SAMPLER_UNIFORM : RT_SAMPLER_RANDOM;
  c.sampler_n = (uint32_t)pixel_sampler.n;
  c.max_depth = (uint32_t)depth;
  c.background_mode = solid_background ? 1 : 0;
  c.background[0] = background.r; c.background[1] = background.g; c.background[2] = background.b;
  auto put = [](double* dst, const Vec3& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
  put(c.center, center);
  put(c.pixel00, pixel00_loc);
  put(c.du, pixel_delta_u);
  put(c.dv, pixel_delta_v);
  put(c.defocus_u, defocus_disk_u);
  put(c.defocus_v, defocus_disk_v);
  c.defocus_angle = defocus_angle;
  return c;
}

namespace {
struct SceneGuard {  // frees the device scene on every exit path
  rt_ctx* ctx;
  rt_scene* sc = nullptr;
  ~SceneGuard() { if (sc) rt_scene_free(ctx, sc); }
};
void check(rt_ctx* ctx, int rc, const char* what) {
  if (rc != RT_OK) throw Error(std::string(what) + " failed (" + std::to_string(rc) + "): " + rt_last_error(ctx));
}
}  // namespace

std::vector<float> Camera::render_partial_sums(const Hittable& world, const RenderTarget& target) const {
  if (!target.ctx) throw Error("RenderTarget.ctx is null: the GPU backend is required (no CPU fallback)");
  if (target.world_size == 0 || target.rank >= target.world_size) throw Error("RenderTarget: bad rank/world_size");
  FlatScene flat;
  world.describe(flat);
  rt_scene_desc desc = flat.desc();
  SceneGuard g{target.ctx};
  check(target.ctx, rt_scene_upload(target.ctx, &desc, &g.sc), "rt_scene_upload");
  rt_camera cam = abi_camera();
  std::vector<float> sums(image_width * image_height * 3);
  check(target.ctx,
        rt_render(target.ctx, g.sc, &cam, target.seed, target.rank, (uint32_t)samples_per_pixel(), target.world_size,
                  sums.data()),
        "rt_render");
  return sums;
}

// mod.rs:64-78,102-114: resolve the sums (mean, gamma 2.2) and send the same 32x32 tile tuples, in tile-index order.
// Returns false when the receiver hung up.
static bool send_tiles(const Camera& cam, rt_ctx* ctx, const std::vector<float>& sums, size_t samples,
                       const TileSender& sender, std::vector<Color>& image_buffer) {
  const size_t W = cam.image_width, H = cam.image_height, pixel_count = W * H;
  std::vector<double> rgb(pixel_count * 3);
  check(ctx, rt_resolve(ctx, sums.data(), (uint32_t)W, (uint32_t)H, (uint32_t)samples, 2.2, rgb.data(), nullptr),
        "rt_resolve");
  image_buffer.resize(pixel_count);
  for (size_t i = 0; i < pixel_count; i++) image_buffer[i] = Color(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]);
  const size_t rect = 32;
  const size_t by = (H + rect - 1) / rect, bx = (W + rect - 1) / rect;
  for (size_t index = 0; index < by * bx; index++) {
    const size_t top = (index / bx) * rect, left = (index % bx) * rect;
    const size_t h = std::min(rect, H - top), w = std::min(rect, W - left);
    std::vector<Color> result(h * w);
    for (size_t dy = 0; dy < h; dy++)
      for (size_t dx = 0; dx < w; dx++) result[dy * w + dx] = image_buffer[(top + dy) * W + (left + dx)];
    if (sender && !sender(Tile{{top, left}, {h, w}, std::move(result)})) {
      std::printf("cancelled\n");  // mod.rs:110-113
      return false;
    }
  }
  return true;
}

void Camera::render(const Hittable& world, const TileSender& sender, const RenderTarget& target) const {
  if (target.world_size != 1)
    throw Error("Camera::render renders whole frames; use render_partial_sums + a reduce for multi-GPU");
  std::vector<float> sums = render_partial_sums(world, target);
  std::vector<Color> image_buffer;
  if (!send_tiles(*this, target.ctx, sums, samples_per_pixel(), sender, image_buffer)) return;
  if (!target.ppm_path.empty()) write_buffer_to_file(target.ppm_path, image_buffer, image_width, image_height);
}

// ---- checkpoint file of render_progressive: header + W*H*3 f32 sums, little-endian, written to <path>.tmp and renamed
namespace {
struct CheckpointHeader {
  char magic[8];  // "RTCKPT2\0"
  uint64_t width, height, spp, passes, next_pass, seed, sampler_kind;
  uint64_t scene_hash;  // FNV-1a over the flattened scene arrays and the whole rt_camera block (pose, fov, defocus,
                        // max_depth, background): sums of another scene or camera are never resumed
};
uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = static_cast<const uint8_t*>(p);
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ull;
  return h;
}
uint64_t scene_camera_hash(const rt_scene_desc& d, const rt_camera& cam) {
  uint64_t h = 0xCBF29CE484222325ull;
  const uint64_t counts[4] = {d.n_prims, d.n_materials, d.n_textures, d.n_texel_bytes};
  h = fnv1a(h, counts, sizeof(counts));
  if (d.n_prims) h = fnv1a(h, d.prims, d.n_prims * sizeof(rt_prim));
  if (d.n_materials) h = fnv1a(h, d.materials, d.n_materials * sizeof(rt_material));
  if (d.n_textures) h = fnv1a(h, d.textures, d.n_textures * sizeof(rt_texture));
  if (d.n_texel_bytes) h = fnv1a(h, d.texels, d.n_texel_bytes);
  return fnv1a(h, &cam, sizeof(cam));
}
bool checkpoint_load(const std::string& path, const CheckpointHeader& want, CheckpointHeader& got, std::vector<float>& sums) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = std::fread(&got, sizeof(got), 1, f) == 1 && std::memcmp(got.magic, want.magic, 8) == 0 &&
            got.width == want.width && got.height == want.height && got.spp == want.spp && got.passes == want.passes &&
            got.seed == want.seed && got.sampler_kind == want.sampler_kind && got.scene_hash == want.scene_hash &&
            got.next_pass <= got.passes;
  if (ok) ok = std::fread(sums.data(), sizeof(float), sums.size(), f) == sums.size();
  std::fclose(f);
  return ok;
}
void checkpoint_store(const std::string& path, const CheckpointHeader& h, const std::vector<float>& sums) {
  const std::string tmp = path + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) throw Error("cannot create " + tmp);
  bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(sums.data(), sizeof(float), sums.size(), f) == sums.size();
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) throw Error("write failed: " + path);
}
}  // namespace

size_t Camera::render_progressive(const Hittable& world, const TileSender& sender, const RenderTarget& target,
                                  size_t passes, const std::string& checkpoint_path) const {
  if (!target.ctx) throw Error("RenderTarget.ctx is null: the GPU backend is required (no CPU fallback)");
  if (target.world_size != 1) throw Error("Camera::render_progressive renders whole frames on one GPU");
  const size_t spp = samples_per_pixel();
  if (passes == 0 || passes > spp) throw Error("render_progressive: passes must be in 1..samples_per_pixel");
  FlatScene flat;
  world.describe(flat);
  rt_scene_desc desc = flat.desc();
  SceneGuard g{target.ctx};
  check(target.ctx, rt_scene_upload(target.ctx, &desc, &g.sc), "rt_scene_upload");
  rt_camera cam = abi_camera();
  const size_t n_val = image_width * image_height * 3;
  std::vector<float> sums(n_val, 0.f), pass_sums(n_val);
  CheckpointHeader hdr{};
  std::memcpy(hdr.magic, "RTCKPT2", 8);
  hdr.width = image_width; hdr.height = image_height; hdr.spp = spp; hdr.passes = passes; hdr.next_pass = 0;
  hdr.seed = target.seed; hdr.sampler_kind = (uint64_t)cam.sampler_kind;
  hdr.scene_hash = scene_camera_hash(desc, cam);
  if (!checkpoint_path.empty()) {
    CheckpointHeader got{};
    std::vector<float> saved(n_val);
    if (checkpoint_load(checkpoint_path, hdr, got, saved)) {
      hdr.next_pass = got.next_pass;
      sums.swap(saved);
    }
  }
  std::vector<Color> image_buffer;
  size_t rendered = 0;
  for (size_t p = (size_t)hdr.next_pass; p < passes; p++) {
    // pass p = sample indices {p, p + passes, ...} (Uniform: a sub-lattice of the n x n grid, so every pass
    // already covers the pixel footprint evenly)
    check(target.ctx,
          rt_render(target.ctx, g.sc, &cam, target.seed, (uint32_t)p, (uint32_t)spp, (uint32_t)passes, pass_sums.data()),
          "rt_render");
    for (size_t i = 0; i < n_val; i++) sums[i] += pass_sums[i];
    rendered++;
    hdr.next_pass = p + 1;
    if (!checkpoint_path.empty()) checkpoint_store(checkpoint_path, hdr, sums);
    size_t done = 0;  // samples accumulated so far: |{s < spp : s % passes <= p}|
    for (size_t q = 0; q <= p; q++) done += (spp - q + passes - 1) / passes;
    if (!send_tiles(*this, target.ctx, sums, done, sender, image_buffer)) return rendered;
  }
  if (hdr.next_pass == passes && rendered == 0) {  // resumed a finished checkpoint: still deliver the frame
    if (!send_tiles(*this, target.ctx, sums, spp, sender, image_buffer)) return rendered;
  }
  if (!target.ppm_path.empty()) write_buffer_to_file(target.ppm_path, image_buffer, image_width, image_height);
  return rendered;
}

// ---- PNG (8-bit RGB, no filtering, deflate "stored" blocks: no compression library needed; RFC 1950/1951/2083)
namespace {
uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n) {
  static uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (uint32_t i = 0; i < 256; i++) {
      uint32_t v = i;
      for (int k = 0; k < 8; k++) v = (v & 1u) ? 0xEDB88320u ^ (v >> 1) : v >> 1;
      table[i] = v;
    }
    ready = true;
  }
  for (size_t i = 0; i < n; i++) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
  return c;
}
void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  for (int k = 3; k >= 0; k--) v.push_back((uint8_t)(x >> (8 * k)));
}
void png_chunk(FILE* f, const char type[4], const std::vector<uint8_t>& data) {
  std::vector<uint8_t> head;
  put_be32(head, (uint32_t)data.size());
  head.insert(head.end(), type, type + 4);
  uint32_t crc = crc32_update(0xFFFFFFFFu, head.data() + 4, 4);
  crc = crc32_update(crc, data.data(), data.size()) ^ 0xFFFFFFFFu;
  std::vector<uint8_t> tail;
  put_be32(tail, crc);
  std::fwrite(head.data(), 1, head.size(), f);
  if (!data.empty()) std::fwrite(data.data(), 1, data.size(), f);
  std::fwrite(tail.data(), 1, tail.size(), f);
}
void write_png(const std::string& path, const std::vector<Color>& image_buffer, size_t width, size_t height) {
  if (width == 0 || height == 0 || width > 0x7FFFFFFFu || height > 0x7FFFFFFFu) throw Error("png: bad image size");
  std::vector<uint8_t> raw;  // scanlines, each prefixed with filter type 0
  raw.reserve(height * (1 + 3 * width));
  for (size_t j = 0; j < height; j++) {
    raw.push_back(0);
    for (size_t i = 0; i < width; i++) {
      auto [r, g, b] = image_buffer[j * width + i].into_u8();
      raw.push_back(r); raw.push_back(g); raw.push_back(b);
    }
  }
  std::vector<uint8_t> z = {0x78, 0x01};  // zlib header: deflate, 32 K window, no preset dictionary
  uint32_t s1 = 1, s2 = 0;                // Adler-32 of the raw data
  for (size_t off = 0; off < raw.size();) {
    const size_t n = std::min<size_t>(65535, raw.size() - off);
    z.push_back(off + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
    z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
    z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + (ptrdiff_t)off, raw.begin() + (ptrdiff_t)(off + n));
    for (size_t i = off; i < off + n; i++) { s1 = (s1 + raw[i]) % 65521u; s2 = (s2 + s1) % 65521u; }
    off += n;
  }
  put_be32(z, (s2 << 16) | s1);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw Error("cannot create " + path);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::fwrite(sig, 1, 8, f);
  std::vector<uint8_t> ihdr;
  put_be32(ihdr, (uint32_t)width);
  put_be32(ihdr, (uint32_t)height);
  ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});  // 8 bits per channel, colour type 2 (RGB), deflate, no filter, no interlace
  png_chunk(f, "IHDR", ihdr);
  png_chunk(f, "IDAT", z);
  png_chunk(f, "IEND", {});
  const bool bad = std::ferror(f) != 0;
  if (std::fclose(f) != 0 || bad) throw Error("write failed: " + path);
}
bool ends_with(const std::string& s, const char* suffix) {
  const size_t n = std::strlen(suffix);
  return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}
}  // namespace

// The reference writes plain-text PPM (mod.rs:216-227, color.rs:74-81); a path ending in ".png" selects PNG with
// the same 8-bit quantisation.
void Camera::write_buffer_to_file(const std::string& path, const std::vector<Color>& image_buffer, size_t width,
                                  size_t height) {
  if (image_buffer.size() != width * height) throw Error("write_buffer_to_file: buffer does not match the image size");
  if (ends_with(path, ".png") || ends_with(path, ".PNG")) return write_png(path, image_buffer, width, height);
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) throw Error("cannot create " + path);
  std::fprintf(f, "P3\n%zu %zu\n255\n", width, height);
  for (const Color& c : image_buffer) {
    auto [r, g, b] = c.into_u8();
    std::fprintf(f, "%u %u %u\n", r, g, b);
  }
  if (std::fclose(f) != 0) throw Error("write failed: " + path);
}

// ------------------------------------------------------------------ scene.rs
Scene composition(CameraBuilder camera_builder) {  // scene.rs:35-87
  Scene sc;
  sc.camera = camera_builder.field_of_view(55.0)
                  .lookfrom(Point3(0.0, 0.5, 1.0) * 1.5)
                  .lookat(Point3(0.0, 0.3, 0.0))
                  .up_vector(Vec3(0.0, 1.0, 0.0))
                  .defocus_angle(0.0)
                  .build();
  auto world = std::make_unique<HittableList>();
  // Ground
  world->add(std::make_unique<Sphere>(Point3(0., -40000000.5, 0.), 40000000.,
                                      std::make_shared<Lambertian>(Color(0.05, 0.20, 0.07))));
  auto blue_lamb = std::make_shared<Lambertian>(Color(0.1, 0.1, 0.8));
  auto red_lamb = std::make_shared<Lambertian>(Color(0.8, 0.1, 0.1));
  world->add(std::make_unique<Sphere>(Point3(0., 0., -1.), 0.5, red_lamb));
  world->add(std::make_unique<Sphere>(Point3(1.3, 0., -1.7), 0.5, blue_lamb));
  world->add(std::make_unique<Sphere>(Point3(-0.25 - 0.125, -0.25, -0.5), 0.25, std::make_shared<Dielectric>(1.5)));
  world->add(std::make_unique<Sphere>(Point3(-0.25 - 0.125, -0.25, -0.5), -0.20, std::make_shared<Dielectric>(1.5)));
  world->add(std::make_unique<Sphere>(Point3(0.6, 0.1, -0.4), 0.3, std::make_shared<Dielectric>(1.5)));
  world->add(std::make_unique<Sphere>(Point3(-1.0, 0., -1.0), 0.5, std::make_shared<Metal>(Color::gray(0.7), 0.0)));
  sc.world = std::move(world);  // plain list (scene.rs:86)
  return sc;
}

Scene book_cover(CameraBuilder camera_builder) {  // scene.rs:89-164
  Scene sc;
  sc.camera = camera_builder.field_of_view(20.0)
                  .lookfrom(Point3(13.0, 2.0, 3.0))
                  .lookat(Point3(0.0, 0.0, 0.0))
                  .up_vector(Vec3(0.0, 1.0, 0.0))
                  .defocus_angle(0.3)
                  .focus_distance(10.0)
                  .build();
  Rng rng = Rng::from_seed(42, 1337);
  auto world = std::make_unique<HittableList>();
  std::shared_ptr<Texture> checker_texture =
      std::make_shared<CheckerTexture>(0.32, std::make_shared<SolidColor>(Color(0.2, 0.3, 0.1)),
                                       std::make_shared<SolidColor>(Color(0.9, 0.9, 0.9)));
  world->add(std::make_unique<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, std::make_shared<Lambertian>(checker_texture)));
  for (int a = -11; a < 11; a++) {
    for (int b = -11; b < 11; b++) {
      const double choose_mat = rng.next_f64();
      const double cx = (double)a + 0.9 * rng.next_f64();
      const double cz = (double)b + 0.9 * rng.next_f64();
      const Point3 center(cx, 0.2, cz);
      if ((center - Point3(4.0, 0.2, 0.0)).length() > 0.9) {
        std::shared_ptr<Material> sphere_material;
        if (choose_mat < 0.7) {  // diffuse
          const Color c1 = Color::random(rng);
          const Color c2 = Color::random(rng);
          sphere_material = std::make_shared<Lambertian>(c1 * c2);
        } else if (choose_mat < 0.9) {  // metal
          const Color albedo = Color::random(rng) / 2.0 + 0.5;
          const double fuzz = rng.next_f64_range(0.0, 0.5);
          sphere_material = std::make_shared<Metal>(albedo, fuzz);
        } else {  // glass
          sphere_material = std::make_shared<Dielectric>(1.5);
        }
        world->add(std::make_unique<MovingSphere>(Sphere(center, 0.2, sphere_material),
                                                  center + Point3(0.0, 0.5 * (1. - choose_mat), 0.0)));
      }
    }
  }
  world->add(std::make_unique<Sphere>(Point3(0.0, 1.0, 0.0), 1.0, std::make_shared<Dielectric>(1.5)));
  world->add(std::make_unique<Sphere>(Point3(-4.0, 1.0, 0.0), 1.0, std::make_shared<Lambertian>(Color::from_hex(0xffca3a))));
  world->add(std::make_unique<Sphere>(Point3(4.0, 1.0, 0.0), 1.0, std::make_shared<Metal>(Color(0.8, 0.8, 0.8), 0.0)));
  sc.world = world->into_bvh();
  return sc;
}

Scene two_spheres(CameraBuilder camera_builder) {  // scene.rs:216-241
  Scene sc;
  sc.camera = camera_builder.field_of_view(20.0).lookfrom(Point3(13.0, 2.0, 3.0)).lookat(Point3(0.0, 0.0, 0.0)).build();
  auto world = std::make_unique<HittableList>();
  std::shared_ptr<Texture> checker_texture =
      std::make_shared<CheckerTexture>(0.3, std::make_shared<SolidColor>(Color(0.2, 0.3, 0.1)),
                                       std::make_shared<SolidColor>(Color(0.9, 0.9, 0.9)));
  auto material = std::make_shared<Lambertian>(checker_texture);
  world->add(std::make_unique<Sphere>(Point3(0.0, -10.0, 0.0), 10.0, material));
  world->add(std::make_unique<Sphere>(Point3(0.0, 10.0, 0.0), 10.0, material));
  sc.world = world->into_bvh();
  return sc;
}

Scene earth(CameraBuilder camera_builder, RgbaImage earth_image) {  // scene.rs:243-260
  Scene sc;
  sc.camera = camera_builder.field_of_view(20.0).lookfrom(Point3(12.0, 0.0, 0.0)).lookat(Point3(0.0, 0.0, 0.0)).build();
  auto world = std::make_unique<HittableList>();
  std::shared_ptr<Texture> earth_texture = std::make_shared<ImageTexture>(std::move(earth_image));
  auto material = std::make_shared<Lambertian>(earth_texture);
  world->add(std::make_unique<Sphere>(Point3(0.0, 0.0, 0.0), 2.0, material));
  sc.world = world->into_bvh();
  return sc;
}

Scene something_blocky(CameraBuilder camera_builder) {  // scene.rs:262-285
  Scene sc;
  sc.camera = camera_builder.field_of_view(20.0).lookfrom(Point3(13.0, 2.0, 3.0)).lookat(Point3(0.0, 0.0, 0.0)).build();
  auto world = std::make_unique<HittableList>();
  std::shared_ptr<Texture> noise_texture = std::make_shared<NoiseTexture>(0.2);
  auto material = std::make_shared<Lambertian>(noise_texture);
  world->add(std::make_unique<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, material));
  world->add(std::make_unique<Sphere>(Point3(0.0, 2.0, 0.0), 2.0, material));
  sc.world = world->into_bvh();
  return sc;
}

// ------------------------------------------------------------------ EXT scenes (book scenes; not in the reference)
static void cornell_walls(HittableList& world) {
  auto red = std::make_shared<Lambertian>(Color(.65, .05, .05));
  auto white = std::make_shared<Lambertian>(Color(.73, .73, .73));
  auto green = std::make_shared<Lambertian>(Color(.12, .45, .15));
  auto light = std::make_shared<DiffuseLight>(Color(7, 7, 7));
  world.add(std::make_unique<Quad>(Point3(555, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), green));
  world.add(std::make_unique<Quad>(Point3(0, 0, 0), Vec3(0, 555, 0), Vec3(0, 0, 555), red));
  world.add(std::make_unique<Quad>(Point3(113, 554, 127), Vec3(330, 0, 0), Vec3(0, 0, 305), light));
  world.add(std::make_unique<Quad>(Point3(0, 555, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
  world.add(std::make_unique<Quad>(Point3(0, 0, 0), Vec3(555, 0, 0), Vec3(0, 0, 555), white));
  world.add(std::make_unique<Quad>(Point3(0, 0, 555), Vec3(555, 0, 0), Vec3(0, 555, 0), white));
}
static CameraBuilder cornell_camera(CameraBuilder cb) {
  return cb.background(Color(0, 0, 0)).field_of_view(40.0).lookfrom(Point3(278, 278, -800)).lookat(Point3(278, 278, 0))
      .up_vector(Vec3(0, 1, 0)).defocus_angle(0.0);
}
Scene cornell_box(CameraBuilder camera_builder) {
  Scene sc;
  sc.camera = cornell_camera(camera_builder).build();
  auto world = std::make_unique<HittableList>();
  cornell_walls(*world);
  auto white = std::make_shared<Lambertian>(Color(.73, .73, .73));
  world->add(std::make_unique<Translate>(
      std::make_unique<RotateY>(box(Point3(0, 0, 0), Point3(165, 330, 165), white), 15.0), Vec3(265, 0, 295)));
  world->add(std::make_unique<Translate>(
      std::make_unique<RotateY>(box(Point3(0, 0, 0), Point3(165, 165, 165), white), -18.0), Vec3(130, 0, 65)));
  sc.world = world->into_bvh();
  return sc;
}
Scene cornell_smoke(CameraBuilder camera_builder) {
  Scene sc;
  sc.camera = cornell_camera(camera_builder).build();
  auto world = std::make_unique<HittableList>();
  cornell_walls(*world);
  auto white = std::make_shared<Lambertian>(Color(.73, .73, .73));
  world->add(std::make_unique<ConstantMedium>(
      std::make_unique<Translate>(std::make_unique<RotateY>(box(Point3(0, 0, 0), Point3(165, 330, 165), white), 15.0),
                                  Vec3(265, 0, 295)),
      0.01, Color(0, 0, 0)));
  world->add(std::make_unique<ConstantMedium>(
      std::make_unique<Translate>(std::make_unique<RotateY>(box(Point3(0, 0, 0), Point3(165, 165, 165), white), -18.0),
                                  Vec3(130, 0, 65)),
      0.01, Color(1, 1, 1)));
  sc.world = world->into_bvh();
  return sc;
}
Scene final_scene(CameraBuilder camera_builder, RgbaImage earth_image, int boxes_per_side, int cluster_spheres) {
  Scene sc;
  sc.camera = camera_builder.background(Color(0, 0, 0)).field_of_view(40.0).lookfrom(Point3(478, 278, -600))
                  .lookat(Point3(278, 278, 0)).up_vector(Vec3(0, 1, 0)).defocus_angle(0.0).build();
  Rng rng = Rng::from_seed(2024, 7);  // the reference's generator, fixed seed (host and oracle agree)
  auto world = std::make_unique<HittableList>();
  auto ground = std::make_shared<Lambertian>(Color(0.48, 0.83, 0.53));
  const double w = 2000.0 / (double)boxes_per_side;
  for (int i = 0; i < boxes_per_side; i++)
    for (int j = 0; j < boxes_per_side; j++) {
      const double x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
      const double x1 = x0 + w, y1 = rng.next_f64_range(1.0, 101.0), z1 = z0 + w;
      world->add(box(Point3(x0, y0, z0), Point3(x1, y1, z1), ground));
    }
  world->add(std::make_unique<Quad>(Point3(123, 554, 147), Vec3(300, 0, 0), Vec3(0, 0, 265),
                                    std::make_shared<DiffuseLight>(Color(7, 7, 7))));
  const Point3 center1(400, 400, 200);
  world->add(std::make_unique<MovingSphere>(Sphere(center1, 50, std::make_shared<Lambertian>(Color(0.7, 0.3, 0.1))),
                                            center1 + Vec3(30, 0, 0)));
  world->add(std::make_unique<Sphere>(Point3(260, 150, 45), 50, std::make_shared<Dielectric>(1.5)));
  world->add(std::make_unique<Sphere>(Point3(0, 150, 145), 50, std::make_shared<Metal>(Color(0.8, 0.8, 0.9), 1.0)));
  world->add(std::make_unique<Sphere>(Point3(360, 150, 145), 70, std::make_shared<Dielectric>(1.5)));
  world->add(std::make_unique<ConstantMedium>(
      std::make_unique<Sphere>(Point3(360, 150, 145), 70, std::make_shared<Dielectric>(1.5)), 0.2, Color(0.2, 0.4, 0.9)));
  world->add(std::make_unique<ConstantMedium>(
      std::make_unique<Sphere>(Point3(0, 0, 0), 5000, std::make_shared<Dielectric>(1.5)), .0001, Color(1, 1, 1)));
  world->add(std::make_unique<Sphere>(Point3(400, 200, 400), 100,
                                      std::make_shared<Lambertian>(std::make_shared<ImageTexture>(std::move(earth_image)))));
  world->add(std::make_unique<Sphere>(Point3(220, 280, 300), 80,
                                      std::make_shared<Lambertian>(std::make_shared<NoiseTexture>(0.2))));
  auto white = std::make_shared<Lambertian>(Color(.73, .73, .73));
  auto cluster = std::make_unique<HittableList>();
  for (int j = 0; j < cluster_spheres; j++) {
    const double x = rng.next_f64_range(0.0, 165.0), y = rng.next_f64_range(0.0, 165.0), z = rng.next_f64_range(0.0, 165.0);
    cluster->add(std::make_unique<Sphere>(Point3(x, y, z), 10, white));
  }
  world->add(std::make_unique<Translate>(std::make_unique<RotateY>(cluster->into_bvh(), 15.0), Vec3(-100, 270, 395)));
  sc.world = world->into_bvh();
  return sc;
}

}  // namespace raytracer
