// The new body of Camera::render (src/camera/mod.rs:54-117).  Signature, tile protocol, cancellation and the PPM
// write are the reference's; the tile loop that spawned CPU workers (mod.rs:63-100) becomes flatten -> upload ->
// render -> resolve.  Uncompiled: no Rust toolchain in the build image; raytracer_amd/host/raytracer.cpp
// (Camera::render) is the compiled, GPU-tested statement of the same sequence.
use std::sync::mpsc::SyncSender;

use crate::color::Color;
use crate::ffi::*;
use crate::flatten::FlatScene;
use crate::hittable::Hittable;

/// The process's GPU backend: one rt_ctx (resolve, single-GPU render) and, when RT_GPUS lists several devices, an
/// rt_multi over them.  Lives until the process ends (the reference renders from one thread for the process's lifetime).
struct Backend { ctx: *mut rt_ctx, multi: *mut rt_multi }
unsafe impl Send for Backend {}
unsafe impl Sync for Backend {}
static BACKEND: std::sync::OnceLock<Backend> = std::sync::OnceLock::new();
fn backend() -> &'static Backend {
    BACKEND.get_or_init(|| unsafe {
        let gpus: Vec<i32> = std::env::var("RT_GPUS").ok()
            .map(|s| s.split(',').filter_map(|t| t.trim().parse().ok()).collect()).unwrap_or_default();
        let mut ctx = std::ptr::null_mut();
        check(rt_init(*gpus.first().unwrap_or(&0), &mut ctx), std::ptr::null_mut());
        let mut multi = std::ptr::null_mut();
        if gpus.len() > 1 {
            check_multi(rt_multi_init(gpus.as_ptr(), gpus.len() as i32, &mut multi), std::ptr::null_mut());
        }
        Backend { ctx, multi }
    })
}

impl Camera {
    /// The private fields of mod.rs:26-51 as the library's camera block.
    fn abi_camera(&self) -> rt_camera {
        let v = |p: &crate::vec3::Vec3| [p.x, p.y, p.z];
        let (sampler_kind, sampler_n) = match self.pixel_sampler {
            PixelSampler::Uniform(side) => (RT_SAMPLER_UNIFORM, side as u32), // already the grid side (builder.rs:63-69)
            PixelSampler::Random(n) => (RT_SAMPLER_RANDOM, n as u32),
        };
        rt_camera {
            width: self.image_width as u32, height: self.image_height as u32, sampler_kind, sampler_n,
            max_depth: self.depth as u32, background_mode: 0, // the sky gradient of mod.rs:205-207
            center: v(&self.center), pixel00: v(&self.pixel00_loc), du: v(&self.pixel_delta_u), dv: v(&self.pixel_delta_v),
            defocus_u: v(&self.defocus_disk_u), defocus_v: v(&self.defocus_disk_v), defocus_angle: self.defocus_angle,
            background: [0.0; 3],
        }
    }

    fn samples_per_pixel(&self) -> usize {
        match self.pixel_sampler {
            PixelSampler::Uniform(side) => side * side,
            PixelSampler::Random(n) => n,
        }
    }

    pub fn render(
        &self,
        world: &Box<dyn Hittable>,
        sender: SyncSender<((usize, usize), (usize, usize), Vec<Color>)>,
    ) {
        let mut flat = FlatScene::default();
        world.describe(&mut flat);
        let desc = flat.desc();
        let cam = self.abi_camera();
        let (w, h, spp) = (self.image_width, self.image_height, self.samples_per_pixel());
        let seed: u64 = 0x853c49e6748fea9b; // the reference's default RNG seed (random.rs:11)
        let mut sums = vec![0f32; w * h * 3];
        let mut rgb = vec![0f64; w * h * 3];
        unsafe {
            // One backend per process, created on the first frame and kept (a context owns a stream, events, staging
            // buffers and per-kernel launch state: creating one per frame would pay that again for every frame of an
            // animation or a progressive render).  Every GPU of the node when RT_GPUS lists several
            // ("0,1,2,3,4,5,6,7"), else one context on GPU 0.  Camera::render is called from one dedicated render
            // thread (main.rs:50-58), which is what the library's "one ctx per calling thread" rule asks for.
            let be = backend();
            if !be.multi.is_null() {
                check_multi(rt_multi_scene_upload(be.multi, &desc), be.multi);
                check_multi(rt_multi_render(be.multi, &cam, seed, 0, spp as u32, 1, sums.as_mut_ptr()), be.multi);
            } else {
                let mut scene = std::ptr::null_mut();
                check(rt_scene_upload(be.ctx, &desc, &mut scene), be.ctx);
                // blocking; returns RT_E_INTERNAL (-> panic, the reference's error style) for an incomplete frame
                check(rt_render(be.ctx, scene, &cam, seed, 0, spp as u32, 1, sums.as_mut_ptr()), be.ctx);
                rt_scene_free(be.ctx, scene);
            }
            // mean, gamma 2.2 (mod.rs:134, color.rs:58-61): what the tile messages carry
            check(rt_resolve(be.ctx, sums.as_ptr(), w as u32, h as u32, spp as u32, 2.2, rgb.as_mut_ptr(), std::ptr::null_mut()), be.ctx);
        }
        let image_buffer: Vec<Color> = rgb.chunks_exact(3).map(|c| Color::new(c[0], c[1], c[2])).collect();

        // the same tile tuples, in tile-index order, as mod.rs:64-78,102-114
        let rect = (32usize, 32usize);
        let (rows, cols) = ((h + rect.0 - 1) / rect.0, (w + rect.1 - 1) / rect.1);
        for index in 0..rows * cols {
            let top_left = ((index / cols) * rect.0, (index % cols) * rect.1);
            let size = (rect.0.min(h - top_left.0), rect.1.min(w - top_left.1));
            let mut tile = Vec::with_capacity(size.0 * size.1);
            for j in 0..size.0 {
                let row = (top_left.0 + j) * w + top_left.1;
                tile.extend_from_slice(&image_buffer[row..row + size.1]);
            }
            if sender.send((top_left, size, tile)).is_err() {
                println!("cancelled"); // mod.rs:110-113: the preview hung up; no file is written
                return;
            }
        }
        self.write_buffer_to_file(&image_buffer).unwrap(); // unchanged (mod.rs:216-227)
    }
}
