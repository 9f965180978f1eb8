#!/usr/bin/env python3
"""bench.py — headline benchmark: Msamples/s of the GPU `Camera::render` replacement on BASELINE.json's
configs[1] (book_cover: bouncing spheres + motion blur + checker ground, 1200x675, 100 spp, depth 50).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full render of the frame on every rank: each rank renders ALL of the config's per-pixel sample
indices (100 spp) for the whole frame with its own Philox key (decorrelated streams per GPU), the partial f32
frames are summed on rank 0 with one RCCL reduce, and rank 0 resolves (mean over N x 100 samples, gamma 2.2,
8-bit).  Scene and camera are resident in HBM before the timed region; outputs stay on the device.

Scaling is therefore WEAK (per-GPU work fixed: `spp_per_gpu` = 100; at N GPUs the frame converges with N x 100
spp); value = samples of all ranks / max-over-ranks time.  (A fixed sample budget is split by residue class of the
sample index instead — `rt_render(..., sample_begin = rank, sample_stride = N)`, raytracer_amd/multi_gpu.py.)

The JSON line also carries
  roofline      FP32-VALU roofline of the render kernels (algorithmic flops/sample from the
                instrumented oracle on the reference-topology BVH x samples per step / hipEvent kernel time)
  cpu_baseline  the CPU oracle (C++ restatement of the reference) timed on this host on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 128 lanes x 2 flop x 2.4 GHz

WORKLOADS = {
    # name: (scene, width, sampler, spp rendered per step, depth, workload fixture)
    "c1": ("book_cover", 400, "random", 10, 50, "workload_c1.json"),
    "c2": ("book_cover", 1200, "uniform", 100, 50, "workload_c2.json"),
    "c3a": ("earth", 1920, "uniform", 256, 50, "workload_c3a.json"),
    "c3b": ("something_blocky", 1920, "uniform", 256, 50, "workload_c3b.json"),
    "c4": ("cornell_smoke", 1920, "uniform", 64, 50, "workload_c4.json"),    # config: 1024 spp; step = 64 spp slice
    "c5": ("final_scene", 3840, "uniform", 16, 50, "workload_c5.json"),      # config: 10 000 spp; step = 16 spp slice
}


def cpu_baseline(scene, width, depth, texels=None):
    """Oracle (kind 'port') on a bounded sample of the same workload (about 10-20 s of wall time on 16 threads):
    the same frame at the config's own 100 spp for C2 (width 1200), 16 spp for the 1920-wide frames, 1 spp at 4K."""
    from oracle_lib import OracleScene
    threads = min(16, os.cpu_count() or 1)  # the GPU box's CPU share is 16
    cspp = 100 if width <= 1200 else (16 if width <= 1920 else 1)
    sc = OracleScene(scene, width, 16.0 / 9.0, "uniform", cspp, depth, texels=texels)
    r = sc.render(threads=threads)
    n = sc.width * sc.height * cspp
    # 1-thread figure on a sub-window (the reference's Arc refcount contention makes it scale sub-linearly)
    win = (0, sc.height // 2 // 32 * 32, sc.width, 32)
    r1 = sc.render(threads=1, window=win)
    n1 = min(32, sc.height - win[1]) * sc.width * cspp
    return {
        "value": n / r["seconds"] / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{scene} {sc.width}x{sc.height}, uniform {cspp} spp, depth {depth}: {n} samples, "
                  f"{r['seconds']:.2f} s",
        "one_thread_value": n1 / r1["seconds"] / 1e6,
        "one_thread_sample": f"rows {win[1]}..{win[1] + 32}, {n1} samples, {r1['seconds']:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-stages", action="store_true", help="extra untimed step with per-stage hipEvents")
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
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import raytracer_amd as ra
    scene_name, width, sampler, spp, depth, fixture = WORKLOADS[args.workload]
    ctx = ra.Context(local_rank)
    # weak scaling: every rank renders `spp` samples per pixel; the frame holds world*spp samples.
    # Uniform sampler needs a square count: use the grid of the config per rank by offsetting the seed per rank
    # (decorrelated Philox streams per GPU) and letting each rank render the config's full grid.
    texels = None
    if scene_name in ("earth", "final_scene"):
        from golden_data import load_earth_rgba
        texels = load_earth_rgba()
    host_scene = ra.Scene(scene_name, width, 16.0 / 9.0, sampler, spp, depth, texels=texels)
    dev_scene = host_scene.upload(ctx)
    cam = host_scene.camera_copy()
    W, H = host_scene.width, host_scene.height
    samples_per_rank = W * H * host_scene.spp
    d_sum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    d_u8 = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    seed0 = 0x853C49E6748FEA9B

    def step(i):
        # rank r draws from its own Philox key => decorrelated streams per GPU (north_star)
        seed = (seed0 + 0x9E3779B97F4A7C15 * (rank + 1) + i) & 0xFFFFFFFFFFFFFFFF
        dev_scene.render_device(cam, d_sum.data_ptr(), seed=seed, stream=stream)
        if world > 1 and args.rehearse:
            h = d_sum.cpu()
            dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
            d_sum.copy_(h)
        elif world > 1:
            dist.reduce(d_sum, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            ctx.check(ctx.L.rt_resolve_device(ctx.h, d_sum.data_ptr(), W, H, host_scene.spp * world, 2.2, None,
                                              d_u8.data_ptr(), stream))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    # per-step hipEvent time of the render kernels (events recorded by the library on the launch stream);
    # measured on an extra replay after the timed region so that reading stats does not serialise the steps
    for i in range(min(args.steps, 3)):
        step(args.warmup + i)
        torch.cuda.synchronize()
        kernel_ms.append(ctx.stats()["ms_total"])
    stats = ctx.stats()
    stage = None
    if args.profile_stages:
        ctx.set_option("profile", 1)
        step(0)
        torch.cuda.synchronize()
        s = ctx.stats()
        stage = {k: s[k] for k in ("ms_generate", "ms_intersect", "ms_shade", "ms_accumulate", "ms_total")}
        ctx.set_option("profile", 0)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_samples = samples_per_rank * world * args.steps
        value = total_samples / elapsed / 1e6
        with open(os.path.join(ROOT, "tests", "golden", fixture)) as f:
            wl = json.load(f)
        fps = wl["flops_per_sample"]
        k_ms = sorted(kernel_ms)[len(kernel_ms) // 2]
        achieved = samples_per_rank * fps / (k_ms * 1e-3) / 1e12
        # HBM bytes per launch from the committed PMC profile of this workload (FETCH_SIZE / WRITE_SIZE passes of
        # scripts/gpu_profile.sh, summarised by scripts/summarize_profile.py; c5: scripts/pmc_c5.sh)
        traffic, pmc_extra = None, None
        traffic_src = "profiles/pmc_latest.json"
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        c5_path = os.path.join(ROOT, "profiles", "r01_c5_v9_pmc.json")
        if args.workload == "c5" and world == 1 and os.path.exists(c5_path):
            with open(c5_path) as f:
                der = json.load(f).get("derived", {})
            if "hbm_write_bytes" in der and "hbm_fetch_bytes_raw" in der:
                traffic = 2 * der["hbm_fetch_bytes_raw"] + der["hbm_write_bytes"]  # FETCH_SIZE x2: upper bound (guide)
                traffic_src = "profiles/r01_c5_v9_pmc.json"
        if args.workload == "c2" and world == 1 and os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f)
            der, cnt = pmc.get("derived", {}), pmc.get("counters", {})
            if "hbm_write_bytes" in der and "hbm_fetch_bytes_x2_upper" in der:
                traffic = der["hbm_fetch_bytes_x2_upper"] + der["hbm_write_bytes"]
            if "SQ_INSTS_VALU" in cnt:  # committed profile of the same command (not measured in this run)
                pmc_extra = {"valu_wave_instructions_per_launch": cnt["SQ_INSTS_VALU"],
                             "valu_active_lane_fraction": der.get("valu_active_lane_fraction"),
                             "l2_hit_rate": der.get("l2_hit_rate"), "source": "profiles/pmc_latest.json"}
        out = {
            "metric": "Msamples/sec (pixels x spp)", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64 geometry + f32 culling/shading",
            "data": f"synthetic (scene builder {scene_name}, fixed seeds)" + (" REHEARSAL, not a measurement" if args.rehearse else ""),
            "config": {"workload": f"{args.workload}: {scene_name} {W}x{H}, {sampler}_sampler({host_scene.spp}), "
                                   f"depth {depth}, {host_scene.desc.n_prims} primitives",
                       "spp_per_gpu": host_scene.spp, "samples_per_step": samples_per_rank * world,
                       "parallelism": f"sample-split x{world} + RCCL reduce" if world > 1 else "single GPU"},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": VALU_FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / VALU_FP32_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 upper bound + WRITE_SIZE, "
                                         + traffic_src + "); the kernel is compute-bound, scene and path "
                                         "state live in LDS / L2 / registers",
                         "kernel": "k_fused (persistent fused path-tracing kernel), per rt_render_device call",
                         "kernel_ms_per_step": k_ms, "algorithmic_flops_per_sample": fps,
                         "rays_per_sample": stats["rays"] / max(stats["paths"], 1)},
        }
        if pmc_extra:
            out["roofline"]["pmc"] = pmc_extra
        if stage:
            out["roofline"]["stage_ms"] = stage
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs at N=1 only (the other ranks would idle on it)
            out["cpu_baseline"] = cpu_baseline(scene_name, width, depth, texels)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
