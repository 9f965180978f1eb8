// src/flatten.rs — flattening of the world into the POD description librt_mi355x takes (include/rt_abi.h).
//
// Hittable, Material and Texture are opaque trait objects (src/hittable/mod.rs:26-29, materials.rs:24-26,
// texture.rs:15-17) whose fields are crate-private (geometry.rs:12-17,84-87; containers.rs:13-16,48-52;
// materials.rs:14-16,59-62,89-91; texture.rs:20-22,39-43,69-71,96-98), so each trait gains ONE method, implemented
// in-crate; the scene builders of src/scene.rs stay byte for byte.  Containers are not described (closest hit does not
// depend on the tree: containers.rs:29-40,177-191) — the library builds its own BVH from the flat list.
// The C++ host mirror has exactly these methods (raytracer_amd/host/raytracer.hpp: describe()).
// Uncompiled: no Rust toolchain in the build image.
use std::collections::HashMap;

use crate::ffi::*;

#[derive(Default)]
pub struct FlatScene {
    pub prims: Vec<rt_prim>,
    pub materials: Vec<rt_material>,
    pub textures: Vec<rt_texture>,
    pub texels: Vec<u8>,
    /// Arc-shared materials / textures are described once: interned by the address of the trait object's data
    seen: HashMap<*const (), i32>,
}

impl FlatScene {
    /// Pointers into the Vecs above: valid while `self` is neither moved nor modified.
    pub fn desc(&self) -> rt_scene_desc {
        rt_scene_desc {
            prims: self.prims.as_ptr(), n_prims: self.prims.len() as u64,
            materials: self.materials.as_ptr(), n_materials: self.materials.len() as u64,
            textures: self.textures.as_ptr(), n_textures: self.textures.len() as u64,
            texels: if self.texels.is_empty() { std::ptr::null() } else { self.texels.as_ptr() },
            n_texel_bytes: self.texels.len() as u64,
        }
    }
    pub fn intern<T: ?Sized>(&mut self, key: &T, make: impl FnOnce(&mut FlatScene) -> i32) -> i32 {
        let k = key as *const T as *const ();
        if let Some(&id) = self.seen.get(&k) {
            return id;
        }
        let id = make(self);
        self.seen.insert(k, id);
        id
    }
}

// ---- the three traits grow one method each --------------------------------------------------------------
//   pub trait Hittable: Sync + Debug { fn hit(..); fn bounding_box(..); fn describe(&self, out: &mut FlatScene); }
//   pub trait Material: Sync + Send + Debug { fn scatter(..); fn describe(&self, out: &mut FlatScene) -> i32; }
//   pub trait Texture: Send + Sync + Debug { fn value(..); fn describe(&self, out: &mut FlatScene) -> i32; }

// ---- src/hittable/geometry.rs ----------------------------------------------------------------------------
impl Sphere {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) {
        let material_id = self.material.describe(out) as u32;
        out.prims.push(rt_prim {
            type_: RT_PRIM_SPHERE, material_id, object_id: out.prims.len() as i32,
            a: [self.center.x, self.center.y, self.center.z],
            radius: self.radius, // negative radii stay negative: they flip the normal (scene.rs:73)
            ..Default::default()
        });
    }
}
impl MovingSphere {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) {
        let material_id = self.sphere.material.describe(out) as u32;
        let (c, d) = (self.sphere.center, self.destination);
        out.prims.push(rt_prim {
            type_: RT_PRIM_MOVING_SPHERE, material_id, object_id: out.prims.len() as i32,
            a: [c.x, c.y, c.z], b: [d.x, d.y, d.z], radius: self.sphere.radius,
            ..Default::default()
        });
    }
}

// ---- src/hittable/containers.rs --------------------------------------------------------------------------
impl HittableList {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) {
        for object in &self.objects {
            object.describe(out);
        }
    }
}
impl BVHNode {
    // tree order, not world.add order: object ids only matter to rt_intersect (take a list snapshot before
    // into_bvh() if insertion-order ids are wanted)
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) {
        self.left.describe(out);
        self.right.describe(out);
    }
}

// ---- src/hittable/materials.rs ---------------------------------------------------------------------------
impl Lambertian {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) -> i32 {
        out.intern(self, |out| {
            let tex = self.albedo.describe(out);
            out.materials.push(rt_material { kind: RT_MAT_LAMBERTIAN, tex, ..Default::default() });
            out.materials.len() as i32 - 1
        })
    }
}
impl Metal {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) -> i32 {
        out.intern(self, |out| {
            out.materials.push(rt_material {
                kind: RT_MAT_METAL, tex: -1, albedo: [self.albedo.r, self.albedo.g, self.albedo.b], fuzz: self.fuzz,
                ..Default::default()
            });
            out.materials.len() as i32 - 1
        })
    }
}
impl Dielectric {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) -> i32 {
        out.intern(self, |out| {
            out.materials.push(rt_material {
                kind: RT_MAT_DIELECTRIC, tex: -1, albedo: [1.0, 1.0, 1.0], ior: self.index_of_refraction,
                ..Default::default()
            });
            out.materials.len() as i32 - 1
        })
    }
}

// ---- src/hittable/texture.rs -----------------------------------------------------------------------------
impl SolidColor {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) -> i32 {
        out.intern(self, |out| {
            out.textures.push(rt_texture {
                kind: RT_TEX_SOLID, odd: -1, even: -1, rgb: [self.color.r, self.color.g, self.color.b],
                ..Default::default()
            });
            out.textures.len() as i32 - 1
        })
    }
}
impl CheckerTexture {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) -> i32 {
        out.intern(self, |out| {
            // children first: the library wants a checker's children earlier in the table (acyclic by construction)
            let odd = self.odd.describe(out);
            let even = self.even.describe(out);
            out.textures.push(rt_texture { kind: RT_TEX_CHECKER, odd, even, inv_scale: self.inv_scale, ..Default::default() });
            out.textures.len() as i32 - 1
        })
    }
}
impl ImageTexture {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) -> i32 {
        out.intern(self, |out| {
            while out.texels.len() % 4 != 0 {
                out.texels.push(0);
            }
            let texel_offset = out.texels.len() as u64;
            out.texels.extend_from_slice(self.image.as_raw()); // RgbaImage: RGBA8, row-major (texture.rs:81-91)
            out.textures.push(rt_texture {
                kind: RT_TEX_IMAGE, odd: -1, even: -1, width: self.image.width(), height: self.image.height(),
                texel_offset, ..Default::default()
            });
            out.textures.len() as i32 - 1
        })
    }
}
impl NoiseTexture {
    pub(crate) fn describe_impl(&self, out: &mut FlatScene) -> i32 {
        out.intern(self, |out| {
            out.textures.push(rt_texture { kind: RT_TEX_NOISE, odd: -1, even: -1, inv_scale: self.inv_scale, ..Default::default() });
            out.textures.len() as i32 - 1
        })
    }
}
// Each `impl Hittable for X` / `impl Material for X` / `impl Texture for X` block forwards:
//     fn describe(&self, out: &mut FlatScene) { self.describe_impl(out) }
