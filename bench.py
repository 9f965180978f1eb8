#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the GPU `Camera::render` replacement on BASELINE.json's
configs[1] (book_cover: bouncing spheres + motion blur + checker ground, 1200x675, 100 spp, depth 50).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one render of THE CONFIG'S FRAME — 1200 x 675 pixels x 100 samples = 81 M samples — by all ranks
together: rank r of N renders the residue class r (mod N) of the 100 per-pixel sample indices for the whole frame
(north_star: "splitting samples-per-pixel into decorrelated RNG streams per GPU"; a path is a function of its sample
index, so the N-rank frame IS the one-rank frame up to f32 summation order — asserted on rank 0 after the timed
region), the partial f32 frames are summed on rank 0 with one RCCL reduce, and rank 0 resolves (mean, gamma 2.2,
8-bit).  Scene and camera are resident in HBM before the timed region; outputs stay on the device.

Scaling is therefore STRONG (total work fixed; at 8 GPUs each renders 12-13 spp, ~1.2 ms of kernel).  value = 81 M x
steps / max-over-ranks time.  Secondary fields: `reduce_ms` (the RCCL reduce alone), `weak` (every rank renders all
100 spp with its own key: per-GPU work fixed).

The JSON line also carries
  roofline      FP32-VALU roofline of the render kernel (algorithmic flops/sample from the instrumented
                oracle on the reference-topology BVH x samples per launch / hipEvent kernel time), the share of those
                flops that run in f64, and — from the committed PMC profile of the same command — the EXECUTED figure
                (VALU lane-operations per second against the lane peak)
  cpu_baseline  the CPU oracle (C++ restatement of the reference) timed on this host on a bounded sample
  workloads     (N = 1) every other BASELINE config — C1, C3a (earth), C3b (noise), C4 (Cornell smoke), C5 (Next Week
                final scene, 4K: north_star's target workload) — with the same measurements, its own roofline block and
                a bounded CPU leg each; and `c5_full`: BASELINE config 5 at its stated size, 3840 x 2160 x 10 000 spp,
                rendered once as 10 launches of 1000 sample indices each (residue classes mod 10, summed on the device)
  first_frame_ms  the first rt_render_device call of the process (cold: lazy code-object load, LDS opt-in, occupancy
                query, no tile-order feedback) next to the steady-state `ms_per_step`; `ms_per_step_feedback_off` is the
                steady state without the tile order learned from the previous frame of the same view
  strong_workloads  (every N) the configs BASELINE quotes "at 1/2/4/8 GPUs" — Cornell smoke at 1024 spp, a 256-spp slice
                of the final scene at 4K — split N ways like the headline frame
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 16 lanes x 2 (packed) x 2 flop x 2.4 GHz
VALU_FP64_PEAK_TFLOPS = 78.6   # vector f64 FMA: one per lane and clock (the unpacked f32 rate)
VALU_LANE_PEAK_TOPS = 39.3     # lane-operations per second: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz

WORKLOADS = {
    # name: (scene, width, sampler, spp rendered per step, depth, workload fixture, config spp)
    "c1": ("book_cover", 400, "random", 10, 50, "workload_c1.json", 10),
    "c2": ("book_cover", 1200, "uniform", 100, 50, "workload_c2.json", 100),
    "c3a": ("earth", 1920, "uniform", 256, 50, "workload_c3a.json", 256),
    "c3b": ("something_blocky", 1920, "uniform", 256, 50, "workload_c3b.json", 256),
    "c4": ("cornell_smoke", 1920, "uniform", 64, 50, "workload_c4.json", 1024),    # step = 64-spp slice
    "c5": ("final_scene", 3840, "uniform", 16, 50, "workload_c5.json", 10000),     # step = 16-spp slice
    # BASELINE configs[3] / [4], the ones quoted "at 1/2/4/8 GPUs": split N ways like the headline frame (every N line)
    "c4_1024spp": ("cornell_smoke", 1920, "uniform", 1024, 50, "workload_c4.json", 1024),
    "c5_256spp": ("final_scene", 3840, "uniform", 256, 50, "workload_c5.json", 10000),
    # BASELINE configs[4] at its stated size: 100 x 100 grid = 10 000 spp at 4K (one step = 10 launches of 1000 spp)
    "c5_full": ("final_scene", 3840, "uniform", 10000, 50, "workload_c5.json", 10000),
}
C5_FULL_PASSES = 10
# flops of one merged-box test (aabox_hit: six quads of an axis-aligned box as one slab test): 6 sub + 6 mul + 4 min/max +
# 1 compare + the face's division and range test, on the scale of SURVEY §8(d)'s F_box = 18
F_AABOX = 20
# integer work the flop count leaves out (reported separately): the noise texture's lattice hash
NOISE_INT_OPS_REFERENCE = 8 * (128 * 8 + 64)   # texture.rs:148-162 as written: 8 corners x (128 xoroshiro steps + jump XORs)
NOISE_INT_OPS_GPU = 40 * 6 + 8 * 12            # 40 GF(2) table lookups (load + 4 xor each) + 8 x (add, convert)


def cpu_baseline(scene, width, depth, texels=None, cspp=None, sampler="uniform"):
    """Oracle (kind 'port') on a bounded sample of the same workload (about 10-20 s of wall time on 16 threads):
    the same frame at the config's own 100 spp for C2 (width 1200), 16 spp for the 1920-wide frames, 1 spp at 4K."""
    from oracle_lib import OracleScene
    threads = min(16, os.cpu_count() or 1)  # the GPU box's CPU share is 16
    if cspp is None:
        cspp = 100 if width <= 1200 else (16 if width <= 1920 else 1)
    sc = OracleScene(scene, width, 16.0 / 9.0, sampler, cspp, depth, texels=texels)
    r = sc.render(threads=threads)
    n = sc.width * sc.height * cspp
    # 1-thread figure on a sub-window (the reference's Arc refcount contention makes it scale sub-linearly)
    win = (0, sc.height // 2 // 32 * 32, sc.width, 32)
    r1 = sc.render(threads=1, window=win)
    n1 = min(32, sc.height - win[1]) * sc.width * cspp
    return {
        "value": n / r["seconds"] / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{scene} {sc.width}x{sc.height}, {sampler} {cspp} spp, depth {depth}: {n} samples, "
                  f"{r['seconds']:.2f} s",
        "one_thread_value": n1 / r1["seconds"] / 1e6,
        "one_thread_sample": f"rows {win[1]}..{win[1] + 32}, {n1} samples, {r1['seconds']:.2f} s",
    }


def f64_share(wl):
    """Share of the algorithmic flops that this implementation executes in f64: the camera ray, the sphere
    / quad / medium tests, the hit point, checker / noise lattice coordinates.  Boxes, scatter, sky, throughput: f32."""
    p, F = wl["per_sample"], wl["flop_constants"]
    f64 = (F["cam"] + F["sph"] * p["sphere_tests"] + F["sph_root"] * p["sphere_disc_pos"] + F["mov"] * p["moving_tests"] +
           F["quad"] * p["quad_tests"] + F["medium"] * p["medium_tests"] + 6 * p["bounce_hits"] +
           F["tex_chk"] * p["tex_checker"] + F["tex_noise"] * p["tex_noise"])
    return f64 / wl["flops_per_sample"]


def load_pmc(workload):
    """Committed rocprofv3 PMC summary of `bench.py --workload <w>` (scripts/gpu_profile.sh + summarize_profile.py)."""
    for name in (f"pmc_{workload}_latest.json", "pmc_latest.json" if workload == "c2" else None):
        if name and os.path.exists(os.path.join(ROOT, "profiles", name)):
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return json.load(f), "profiles/" + name
    return None, None


def own_tree_flops(wl, own):
    """Algorithmic flops per sample counted on the tree THIS kernel walks (VERDICT r2 #3a): the per-operation constants
    of SURVEY §8(d) applied to the box tests, primitive tests by type and rays the instrumented build of the same kernel
    instance executed (rt_stats.dbg_counts; primitives kept outside the tree included), plus the terms that do not depend
    on the tree (camera ray, hit record, scatter, textures, sky), which are the fixture's.  The reference-topology figure
    (`algorithmic_flops_per_sample`) counts 6-quad box lists and a median-split tree the GPU never walks."""
    p, F = wl["per_sample"], wl["flop_constants"]
    sph = own["sphere_tests"] + own["moving_tests"]
    tree = (F["ray"] * own["rays"] + F["box"] * own["box_tests"] + F["sph"] * sph + F["mov"] * own["moving_tests"] +
            F["sph_root"] * min(p["sphere_disc_pos"], sph) + F["quad"] * own["quad_tests"] + F["medium"] * own["medium_tests"] +
            F_AABOX * own.get("aabox_tests", 0.0))
    shared = (F["cam"] + F["hit"] * p["bounce_hits"] + F["lam"] * p["scat_lambert"] + F["met"] * p["scat_metal"] +
              F["die"] * p["scat_dielectric"] + F["iso"] * p["scat_isotropic"] + F["tex_chk"] * p["tex_checker"] +
              F["tex_img"] * p["tex_image"] + F["tex_noise"] * p["tex_noise"] + F["mul"] * p["bounce_hits"] + F["sky"] * p["misses"])
    return tree + shared


def measure_own_tree(ctx, rig, stream, d_ptr, spp_slice=None):
    """One frame through the instrumented build of the production instance (not timed): operation counts per sample."""
    ctx.set_option("fused_stats", 1)
    try:
        kw = {}
        if spp_slice:  # a residue class of the sample indices is enough for per-sample averages
            kw = dict(sample_begin=0, sample_stride=spp_slice)
        rig.dev.render_device(rig.cam, d_ptr, seed=0x1234, stream=stream, **kw)
        import torch
        torch.cuda.synchronize()
        st = ctx.stats()
    finally:
        ctx.set_option("fused_stats", 0)
    n = max(st["paths"], 1)
    c = st["dbg_counts"]
    if c[0] == 0 and c[1] + c[2] + c[3] + c[4] + c[5] == 0:
        return None  # this instance has no instrumented build
    return {"rays": st["rays"] / n, "box_tests": c[0] / n, "sphere_tests": c[1] / n, "moving_tests": c[2] / n,
            "quad_tests": c[3] / n, "medium_tests": c[4] / n, "aabox_tests": c[5] / n,
            "node_loop_lane_slots": {"walking": st["dbg"][0], "at_leaf": c[6],
                                     "finished": c[7], "no_path": c[8]}}


def roofline_block(workload, wl, samples_per_launch, kernel_ms, stats, own=None, frame_samples=None):
    fps = wl["flops_per_sample"]
    achieved = samples_per_launch * fps / (kernel_ms * 1e-3) / 1e12
    share = f64_share(wl)
    # a kernel whose flops are `share` f64 and the rest packed f32 cannot beat this mixed peak
    mixed_peak = 1.0 / ((1.0 - share) / VALU_FP32_PEAK_TFLOPS + share / VALU_FP64_PEAK_TFLOPS)
    out = {"bound": "valu_fp32", "achieved": achieved, "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / VALU_FP32_PEAK_TFLOPS, "traffic": None,
           "kernel": "k_fused (persistent fused path-tracing kernel), per rt_render_device call",
           "kernel_ms_per_step": kernel_ms, "algorithmic_flops_per_sample": fps,
           "f64_share_of_algorithmic_flops": share, "mixed_f32_f64_peak": mixed_peak, "frac_of_mixed_peak": achieved / mixed_peak,
           "rays_per_sample": stats["rays"] / max(stats["paths"], 1)}
    if wl["per_sample"].get("tex_noise", 0) > 0:
        out["integer_ops_per_sample_not_in_flops"] = {
            "noise_lookups_per_sample": wl["per_sample"]["tex_noise"],
            "reference_int_ops_per_lookup": NOISE_INT_OPS_REFERENCE, "gpu_int_ops_per_lookup": NOISE_INT_OPS_GPU,
            "reference_int_ops_per_sample": wl["per_sample"]["tex_noise"] * NOISE_INT_OPS_REFERENCE,
            "gpu_int_ops_per_sample": wl["per_sample"]["tex_noise"] * NOISE_INT_OPS_GPU}
    if own:
        fo = own_tree_flops(wl, own)
        ach = samples_per_launch * fo / (kernel_ms * 1e-3) / 1e12
        slots = own.pop("node_loop_lane_slots", None)
        out["own_tree"] = {"flops_per_sample": fo, "achieved": ach, "frac_own_tree": ach / VALU_FP32_PEAK_TFLOPS,
                           "per_sample": own,
                           "note": "the same constants applied to what the instrumented build of this kernel instance executes on "
                                   "the library's own tree (binned-SAH, huge primitives outside the tree); tree-independent terms "
                                   "from the fixture"}
        out["frac_own_tree"] = ach / VALU_FP32_PEAK_TFLOPS
        if slots and sum(slots.values()) > 0:
            tot = float(sum(slots.values()))
            out["own_tree"]["node_loop_lane_share"] = {k: v / tot for k, v in slots.items()}
    if out["frac_of_mixed_peak"] > 1.0:
        out["note_frac"] = ("frac_of_mixed_peak > 1: the kernel does NOT execute the counted work — `achieved` prices the "
                            "reference algorithm on the reference-topology tree (6-quad box lists, median splits), which "
                            "needs far more primitive and box tests than the library's tree; `frac` is therefore not a "
                            "measure of this kernel's efficiency — use `frac_own_tree` and `executed`")
    pmc, src = load_pmc(workload)
    # the committed PMC counts are those of the WHOLE frame: scale them to the samples this launch rendered
    scale = (samples_per_launch / frame_samples) if frame_samples else 1.0
    if pmc:
        der, cnt = pmc.get("derived", {}), pmc.get("counters", {})
        if "hbm_write_bytes" in der and "hbm_fetch_bytes_x2_upper" in der:
            out["traffic"] = (der["hbm_fetch_bytes_x2_upper"] + der["hbm_write_bytes"]) * scale
            out["traffic_note"] = ("HBM bytes per launch (rocprofv3 FETCH_SIZE x2 upper bound + WRITE_SIZE, " + src +
                                   "); algorithmic output is W*H*12 B; scene and path state live in LDS / L2 / registers")
        if "SQ_INSTS_VALU" in cnt and der.get("valu_active_lane_fraction"):
            lane_ops = cnt["SQ_INSTS_VALU"] * scale * 64 * der["valu_active_lane_fraction"]
            out["executed"] = {
                "valu_wave_instructions_per_launch": cnt["SQ_INSTS_VALU"] * scale,
                "scaled_to_share_of_frame": scale,
                "valu_active_lane_fraction": der["valu_active_lane_fraction"],
                "lane_ops_per_s_tera": lane_ops / (kernel_ms * 1e-3) / 1e12, "lane_peak_tera": VALU_LANE_PEAK_TOPS,
                "lane_utilisation": lane_ops / (kernel_ms * 1e-3) / 1e12 / VALU_LANE_PEAK_TOPS,
                "l2_hit_rate": der.get("l2_hit_rate"),
                "note": "committed profile of the same command (" + src + "), collected with kernels serialised; "
                        "lane_utilisation = VALU lane-operations actually executed per second / (256 CU x 64 lanes x "
                        "2.4 GHz): what the hardware did, as opposed to `frac`, which prices the REFERENCE algorithm's "
                        "flops (on the reference's tree) at this kernel's speed"}
    return out


class Rig:
    """One workload resident on this rank's GPU."""

    def __init__(self, ra, ctx, workload):
        self.name = workload
        scene_name, width, sampler, spp, depth, fixture, config_spp = WORKLOADS[workload]
        self.scene_name, self.width_req, self.sampler, self.depth, self.config_spp = scene_name, width, sampler, depth, config_spp
        self.texels = None
        if scene_name in ("earth", "final_scene"):
            from golden_data import load_earth_rgba
            self.texels = load_earth_rgba()
        self.host = ra.Scene(scene_name, width, 16.0 / 9.0, sampler, spp, depth, texels=self.texels)
        self.dev = self.host.upload(ctx)
        self.cam = self.host.camera_copy()
        self.W, self.H, self.spp = self.host.width, self.host.height, self.host.spp
        with open(os.path.join(ROOT, "tests", "golden", fixture)) as f:
            self.wl = json.load(f)

    def label(self):
        return (f"{self.name}: {self.scene_name} {self.W}x{self.H}, {self.sampler}_sampler({self.spp}), depth {self.depth}, "
                f"{self.host.desc.n_prims} primitives" + (f" ({self.spp}-spp slice of the config's {self.config_spp})"
                                                           if self.spp != self.config_spp else ""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-strong-workloads", action="store_true",
                    help="skip the split-frame runs of the multi-GPU configs C4 (1024 spp) and C5 (256-spp slice)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the block of the other BASELINE configs on the N = 1 line")
    ap.add_argument("--no-c5-full", action="store_true", help="skip BASELINE config 5 at its full 10 000 spp (about half a minute)")
    ap.add_argument("--lean-run", action="store_true",
                    help="profiling runs (scripts/gpu_profile.sh): only launches of the production kernel — no instrumented frame, "
                         "no feedback-off leg, no batched leg")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="rt_set_option override (profiling runs: fused_partner=0, rocprofv3 serialises kernels)")
    ap.add_argument("--rehearse", action="store_true",
                    help="rehearsal of the N>1 code path on a one-GPU box: every rank renders on GPU 0 and the "
                         "partial frames are reduced with gloo through host memory (not a measurement)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch  # first: its bundled HIP runtime is the one the process uses
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback)")
    if args.rehearse or os.environ.get("BENCH_SAME_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # How partial frames reach rank 0: RCCL over xGMI, on a process group of its own.  The default group is gloo (timing
    # exchange, barriers, and the channel over which the ranks AGREE on what follows): the RCCL group is created and
    # probed with one all-reduce inside a try on every rank, and if that fails anywhere all ranks stage their frames
    # through host memory instead — a slower line that says "reduce": "host" rather than no line at all.
    # (--rehearse: host memory from the start, every rank on GPU 0.  BENCH_SAME_GPU=1: test hook, the real path with every
    # rank on GPU 0, where RCCL must refuse and the fallback must carry the run.)
    rccl_group = None
    use_host_reduce = bool(args.rehearse)
    reduce_note = "gloo through host memory (rehearsal)" if args.rehearse else "rccl"
    if world > 1:
        dist.init_process_group("gloo")
        if not args.rehearse:
            ok, why = 1, ""
            try:
                rccl_group = dist.new_group(backend="nccl")
                probe = torch.ones(1, dtype=torch.float32, device="cuda")
                dist.all_reduce(probe, group=rccl_group)
                torch.cuda.synchronize()
                ok = 1 if abs(float(probe.item()) - world) < 0.5 else 0
                why = "" if ok else "the probe all-reduce returned a wrong sum"
            except Exception as exc:  # noqa: BLE001 - whatever RCCL raises, the bench goes on through host memory
                ok, why = 0, f"{type(exc).__name__}: {exc}"
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                use_host_reduce = True
                reduce_note = "host memory (gloo): RCCL failed on some rank" + (f" - here: {why[:200]}" if why else "")
                print(f"[bench rank {rank}] RCCL unavailable, reducing through host memory. {why[:300]}", file=sys.stderr)

    import numpy as np
    import raytracer_amd as ra
    from raytracer_amd import multi_gpu as mg
    ctx = ra.Context(local_rank)
    for kv in args.option:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    rig = Rig(ra, ctx, args.workload)
    W, H = rig.W, rig.H
    frame_samples = W * H * rig.spp
    seed0 = 0x853C49E6748FEA9B
    # Everything runs on a stream of torch's own: its handle is non-zero, so the library launches ON it (a NULL stream
    # means "the context's own stream" in the ABI) and the RCCL reduce, which torch orders against the current stream,
    # sees finished frames without any host wait.
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    d_sum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    d_u8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    # the process's FIRST frame (the reference renders one frame per process): nothing primed, nothing cached
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b0, e0, s0 = mg.sample_slice(rank, world, rig.spp)
    rig.dev.render_device(rig.cam, d_sum.data_ptr(), seed=seed0 ^ 0x5555, sample_begin=b0, sample_end=e0, sample_stride=s0, stream=stream)
    torch.cuda.synchronize()
    first_frame_ms = (time.perf_counter() - t0) * 1e3

    def reduce_sum(t):
        # partial frame of every rank -> their sum on rank 0, in place
        if world == 1:
            return
        if use_host_reduce:
            h = t.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=rccl_group)

    def reduce_to_rank0():
        reduce_sum(d_sum)

    def resolve(spp_total):
        if rank == 0:
            ctx.check(ctx.L.rt_resolve_device(ctx.h, d_sum.data_ptr(), W, H, spp_total, 2.2, None, d_u8.data_ptr(), stream))

    def step_strong(i):
        # the config's frame, split: rank r renders sample indices r, r + N, ... of every pixel (same key on all ranks)
        b, e, s = mg.sample_slice(rank, world, rig.spp)
        rig.dev.render_device(rig.cam, d_sum.data_ptr(), seed=(seed0 + i) & 0xFFFFFFFFFFFFFFFF, sample_begin=b, sample_end=e,
                              sample_stride=s, stream=stream)
        reduce_to_rank0()
        resolve(rig.spp)

    def step_weak(i):
        # every rank renders ALL the config's sample indices with its own key: the frame converges with N x spp samples
        seed = (seed0 + 0x9E3779B97F4A7C15 * (rank + 1) + i) & 0xFFFFFFFFFFFFFFFF
        rig.dev.render_device(rig.cam, d_sum.data_ptr(), seed=seed, stream=stream)
        reduce_to_rank0()
        resolve(rig.spp * world)

    def barrier():
        if world > 1:
            torch.cuda.synchronize()  # this rank's launches and reduces are done ...
            dist.barrier()            # ... and so are everybody's (gloo)
        torch.cuda.synchronize()

    def timed(step, n_steps, n_warm):
        for i in range(n_warm):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(n_warm + i)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    elapsed = timed(step_strong, args.steps, args.warmup)

    # per-step hipEvent time of this rank's render kernel (events recorded by the library on the launch stream); measured
    # on an extra replay after the timed region so that reading stats does not serialise the steps
    kernel_ms = []
    for i in range(min(args.steps, 3)):
        step_strong(args.warmup + i)
        torch.cuda.synchronize()
        kernel_ms.append(ctx.stats()["ms_total"])
    stats = ctx.stats()
    k_ms = sorted(kernel_ms)[len(kernel_ms) // 2]
    my_samples = stats["paths"]

    extra = {"first_frame_ms": first_frame_ms,
             "first_frame_note": "wall time of the first rt_render_device call of this process on this rank (cold: code-object "
                                 "load, LDS opt-in, occupancy query, tile order unprimed), scene already uploaded"}
    own = None
    if rank == 0 and not args.lean_run:
        own = measure_own_tree(ctx, rig, stream, d_sum.data_ptr())
    if world == 1 and args.workload in ("c1", "c2", "c4") and not args.lean_run:
        # steady state without the tile order learned from the previous frame of the same view
        ctx.set_option("fused_feedback", 0)
        el_off = timed(step_strong, max(3, min(args.steps, 10)), 1)
        ctx.set_option("fused_feedback", 1)
        extra["ms_per_step_feedback_off"] = el_off / max(3, min(args.steps, 10)) * 1e3
    # Batched frames (rt_render_batch_device): B successive frames of the config — this rank's slice of each — in ONE
    # persistent launch, then B reduces and resolves: the per-frame drain of the deepest paths is paid once per batch
    # instead of once per frame, which is what limits strong scaling of a 9 ms frame.  Secondary field;
    # `value` stays the one-launch-per-frame figure (the reference renders one frame per call).
    if args.workload in ("c1", "c2") and not args.lean_run:
        B = 8
        b_, e_, s_ = mg.sample_slice(rank, world, rig.spp)
        # every frame of a batch must select the same number of sample indices: this rank's slice as it is
        outs = [torch.zeros_like(d_sum) for _ in range(B)]
        ptrs = [o.data_ptr() for o in outs]

        def batch_step(i):
            ra.render_batch_device(rig.dev, rig.cam, ptrs, [(seed0 + 1000 + i * B + k) & 0xFFFFFFFFFFFFFFFF for k in range(B)],
                                   [b_] * B, sample_end=e_, sample_stride=s_, stream=stream)
            for o in outs:
                reduce_sum(o)
                if rank == 0:
                    ctx.check(ctx.L.rt_resolve_device(ctx.h, o.data_ptr(), W, H, rig.spp, 2.2, None, d_u8.data_ptr(), stream))

        nb = max(1, min(args.steps, 16) // 2)
        elb = timed(batch_step, nb, 1)
        extra["batched"] = {"frames_per_launch": B, "launches": nb, "ms_per_frame": elb / (nb * B) * 1e3,
                            "value": frame_samples * nb * B / elb / 1e6, "unit": "Msamples/s",
                            "note": "B frames per rt_render_batch_device launch + B reduces / resolves; frames bit-identical to "
                                    "single launches (tests/test_gpu_batch.py)"}
        del outs
    if world > 1:
        # (1) the N-rank frame is the one-rank frame: rank 0 renders the whole frame itself and compares
        step_strong(12345)
        torch.cuda.synchronize()
        if rank == 0:
            got = d_sum.cpu().numpy()
            whole = torch.zeros_like(d_sum)
            rig.dev.render_device(rig.cam, whole.data_ptr(), seed=(seed0 + 12345) & 0xFFFFFFFFFFFFFFFF, stream=stream)
            torch.cuda.synchronize()
            ref = whole.cpu().numpy()
            err = float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)))
            assert err <= 1e-5, f"the {world}-rank frame differs from the one-rank frame (max rel. error {err:.3g})"
            extra["split_frame_max_rel_error_vs_one_rank"] = err
        # (2) the reduce alone (all ranks enter together)
        barrier()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R):
            reduce_to_rank0()
        barrier()
        extra["reduce_ms"] = (time.perf_counter() - t0) / R * 1e3
        extra["reduce_bytes_per_rank"] = W * H * 12
        # (3) weak scaling, secondary
        wsteps = max(1, min(args.steps, 5))
        wel = timed(step_weak, wsteps, 1)
        extra["weak"] = {"value": frame_samples * world * wsteps / wel / 1e6, "unit": "Msamples/s",
                         "ms_per_step": wel / wsteps * 1e3, "steps": wsteps, "spp_per_gpu": rig.spp,
                         "note": "every rank renders the config's full 100 spp with its own key (per-GPU work fixed)"}

    # (4) the configs BASELINE quotes "at 1/2/4/8 GPUs", split N ways exactly like the headline frame: Cornell smoke at its
    # full 1024 spp and a 256-spp slice of the final scene at 4K (its 10 000 spp are 40 such slices).  Frames long enough
    # that the per-frame drain is small: the scaling north_star's 85 % target speaks of.
    strong = {}
    if args.workload == "c2" and not args.no_strong_workloads:
        for w in ("c4_1024spp", "c5_256spp"):
            r3 = Rig(ra, ctx, w)
            d3 = torch.zeros((r3.H, r3.W, 3), dtype=torch.float32, device="cuda")
            u3 = torch.zeros((r3.H, r3.W, 3), dtype=torch.uint8, device="cuda")

            def step3(i, r3=r3, d3=d3, u3=u3):
                b, e, st = mg.sample_slice(rank, world, r3.spp)
                r3.dev.render_device(r3.cam, d3.data_ptr(), seed=(seed0 + i) & 0xFFFFFFFFFFFFFFFF, sample_begin=b, sample_end=e,
                                     sample_stride=st, stream=stream)
                reduce_sum(d3)
                if rank == 0:
                    ctx.check(ctx.L.rt_resolve_device(ctx.h, d3.data_ptr(), r3.W, r3.H, r3.spp, 2.2, None, u3.data_ptr(), stream))

            n3 = 2
            el3 = timed(step3, n3, 1)
            strong[w] = {"workload": r3.label(), "value": r3.W * r3.H * r3.spp * n3 / el3 / 1e6, "unit": "Msamples/s",
                         "ms_per_step": el3 / n3 * 1e3, "steps": n3, "scaling": "strong",
                         "spp_per_gpu": [mg.samples_in_slice(*mg.sample_slice(r, world, r3.spp)) for r in range(world)],
                         "reduce_bytes_per_rank": r3.W * r3.H * 12 if world > 1 else 0}
            del d3, u3, r3
        extra["strong_workloads"] = strong

    if rank == 0:
        value = frame_samples * args.steps / elapsed / 1e6
        out = {
            "metric": "Msamples/sec (pixels x spp)", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64 geometry + f32 culling/shading",
            "data": f"synthetic (scene builder {rig.scene_name}, fixed seeds)" + (" REHEARSAL, not a measurement" if args.rehearse else ""),
            "config": {"workload": rig.label(), "samples_per_step": frame_samples,
                       "spp_per_gpu": [mg.samples_in_slice(*mg.sample_slice(r, world, rig.spp)) for r in range(world)],
                       "parallelism": (f"sample indices split x{world} (residue classes) + one reduce of "
                                       f"{W * H * 12} B per rank to rank 0 ({reduce_note})") if world > 1 else "single GPU"},
            "roofline": roofline_block(args.workload, rig.wl, my_samples, k_ms, stats, own, frame_samples),
        }
        out["roofline"]["note"] = ("rank 0's launch: %d of the frame's %d samples" % (my_samples, frame_samples)) if world > 1 else \
            "one launch = the whole frame"
        out.update(extra)
        if world > 1:
            out["reduce"] = "host" if use_host_reduce else "rccl"
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N=1 only (the other ranks would idle on it)
            out["cpu_baseline"] = cpu_baseline(rig.scene_name, rig.width_req, rig.depth, rig.texels)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if world == 1 and args.workload == "c2" and not args.no_workloads:
            # every other BASELINE config, each with its own roofline block and a bounded CPU leg (C5 = the north_star's
            # target workload).  CPU samples: C1 in full; 16 / 4 spp of the 1080p frames; 1 spp at 4K.
            out["workloads"] = {}
            for w, csp in (("c1", 10), ("c3a", 16), ("c3b", 4), ("c4", 4), ("c5", 1)):
                r2 = Rig(ra, ctx, w)
                d2 = torch.zeros((r2.H, r2.W, 3), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r2.dev.render_device(r2.cam, d2.data_ptr(), seed=seed0, stream=stream)
                torch.cuda.synchronize()
                first2 = (time.perf_counter() - t0) * 1e3
                r2.dev.render_device(r2.cam, d2.data_ptr(), seed=seed0 + 1, stream=stream)
                torch.cuda.synchronize()
                n2 = 3 if r2.W * r2.H * r2.spp > (1 << 26) else 20
                t0 = time.perf_counter()
                for i in range(n2):
                    r2.dev.render_device(r2.cam, d2.data_ptr(), seed=seed0 + 2 + i, stream=stream)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / n2 * 1e3
                st2 = ctx.stats()
                own2 = measure_own_tree(ctx, r2, stream, d2.data_ptr(), spp_slice=16 if r2.spp >= 256 else None)
                entry = {"workload": r2.label(), "value": r2.W * r2.H * r2.spp / ms / 1e3, "unit": "Msamples/s",
                         "ms_per_step": ms, "steps": n2, "first_frame_ms": first2, "config_spp": r2.config_spp, "spp": r2.spp,
                         "roofline": roofline_block(w, r2.wl, st2["paths"], st2["ms_total"], st2, own2)}
                if not args.no_cpu_baseline:
                    entry["cpu_baseline"] = cpu_baseline(r2.scene_name, r2.width_req, r2.depth, r2.texels, cspp=csp, sampler=r2.sampler)
                    entry["gpu_over_cpu"] = entry["value"] / entry["cpu_baseline"]["value"]
                out["workloads"][w] = entry
                del d2, r2
            if not args.no_c5_full:
                # BASELINE configs[4] at its stated size on ONE GPU: 3840 x 2160 x 10 000 spp = 82 944 M samples, as 10
                # launches of 1000 sample indices each (residue classes mod 10 of the 100 x 100 grid: the checkpointable
                # form, Camera::render_progressive's), partial sums added on the device.
                r5 = Rig(ra, ctx, "c5_full")
                acc5 = torch.zeros((r5.H, r5.W, 3), dtype=torch.float32, device="cuda")
                p5 = torch.zeros_like(acc5)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k_ms5, rays5 = 0.0, 0
                for ps in range(C5_FULL_PASSES):
                    r5.dev.render_device(r5.cam, p5.data_ptr(), seed=seed0, sample_begin=ps, sample_stride=C5_FULL_PASSES, stream=stream)
                    acc5 += p5
                    torch.cuda.synchronize()  # (also: a progress point every ~3 s)
                    st5 = ctx.stats()         # raises if a frame was cut short (watchdog)
                    k_ms5 += st5["ms_total"]
                    rays5 += st5["rays"]
                    assert st5["samples_nan"] == 0 and st5["samples_clamped"] == 0
                el5 = time.perf_counter() - t0
                n5 = r5.W * r5.H * r5.spp
                mean5 = float(acc5.mean().item()) / r5.spp
                # the 16-spp frame of the `c5` entry estimates the same image: means agree to Monte-Carlo noise
                r16 = Rig(ra, ctx, "c5")
                d16 = torch.zeros((r16.H, r16.W, 3), dtype=torch.float32, device="cuda")
                r16.dev.render_device(r16.cam, d16.data_ptr(), seed=seed0 + 77, stream=stream)
                torch.cuda.synchronize()
                mean16 = float(d16.mean().item()) / r16.spp
                assert abs(mean5 - mean16) <= 0.01 * abs(mean5), (mean5, mean16)
                out["workloads"]["c5_full"] = {
                    "workload": r5.label(), "value": n5 / el5 / 1e6, "unit": "Msamples/s", "seconds": el5, "steps": 1,
                    "launches": C5_FULL_PASSES, "config_spp": r5.config_spp, "spp": r5.spp, "samples": n5, "rays": rays5,
                    "kernel_ms_total": k_ms5, "mean_linear_radiance": mean5, "mean_linear_radiance_16spp_frame": mean16,
                    "roofline": roofline_block("c5", r5.wl, n5, k_ms5, {"rays": rays5, "paths": n5}, out["workloads"]["c5"]["roofline"].get("own_tree", {}).get("per_sample"),
                                               frame_samples=r16.W * r16.H * r16.spp)}  # (the PMC profile is that of the 16-spp frame)
                if "cpu_baseline" in out["workloads"]["c5"]:
                    out["workloads"]["c5_full"]["gpu_over_cpu"] = n5 / el5 / 1e6 / out["workloads"]["c5"]["cpu_baseline"]["value"]
                del acc5, p5, d16, r5, r16
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
