// src/camera/image.rs — the module `pub mod image;` (src/camera/mod.rs:19) names but the tree lacks.  Reconstructed from
// its uses: ImageSpecBuilder::default().width(..).aspect_ratio(..).build() (src/main.rs:28-33); fields width, height,
// aspect_ratio read by CameraBuilder::build (src/camera/builder.rs:56-58,90,99-100,109-110,122); Clone (main.rs:36).
// Height rule inferred from the committed renders: height = trunc(width / aspect_ratio), at least 1
// (900 wide @ 16:10 -> 562 px: renders/weekend/1-normal-sphere.png; 900 @ 16:9 -> 506 px: renders/weekend/4-materials.png).
// Uncompiled: no Rust toolchain in the build image.

#[derive(Clone, Debug)]
pub struct ImageSpec {
    pub width: usize,
    pub height: usize,
    pub aspect_ratio: f64,
}

#[derive(Default)]
pub struct ImageSpecBuilder {
    width: Option<usize>,
    aspect_ratio: Option<f64>,
}

impl ImageSpecBuilder {
    pub fn width(self, width: usize) -> Self {
        Self { width: Some(width), ..self }
    }
    pub fn aspect_ratio(self, aspect_ratio: f64) -> Self {
        Self { aspect_ratio: Some(aspect_ratio), ..self }
    }
    pub fn build(self) -> ImageSpec {
        let width = self.width.expect("The image width must be set");
        let aspect_ratio = self.aspect_ratio.unwrap_or(16.0 / 9.0);
        let height = ((width as f64 / aspect_ratio) as usize).max(1);
        ImageSpec { width, height, aspect_ratio }
    }
}
