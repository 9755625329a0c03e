#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on the 2.37 M-triangle scene, 1024^2, 16 spp, 4 bounces.

One "step" = one full pass of the trace path (raygen -> closest/shadow traversal -> shading
-> resolve -> RGB8 tiles) over the frame, scene already resident in HBM.  With N > 1 ranks
(one process per GPU, torch.distributed over RCCL) the frame is split into interleaved
32x32 tiles, every rank traces its tiles against its own copy of the scene, and the
finished RGB8 tiles are all-gathered; total work is fixed, so scaling is "strong".

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints one JSON line on rank 0 (see DESIGN.md for the roofline / cpu_baseline definitions).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
RAY_IO_BYTES = 40      # 32 B ray in + 8 B hit out
NODE_BYTES = 128       # one wide node (two tree levels) = one 128-byte line
PRIM_BYTES = 64


SCENE_TEXT = {
    "scan": "load_basic + 4 synthetic scan-like meshes (BASELINE config 3, stand-in for the Stanford scans)",
    "scan_glass": "config-3 geometry with mirror+glass+Phong compositions (BASELINE config 4)",
    "soup": "random-triangle soup, 4 point lights + ambient (BASELINE config 5)",
    "basic_suzanne": "load_basic + Suzanne.obj + directional light (BASELINE config 2)",
    "random_spheres": "load_random_spheres (BASELINE config 1, what main.rs:90 renders)",
    "basic": "load_basic (scene.rs:11-108)",
    "suzanne": "load_suzanne (scene.rs:188-214)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="scan")
    ap.add_argument("--tris", type=int, default=0, help="triangle count of the synthetic scene (0 = 2,374,525)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out", default=None, help="rank 0 writes the last frame to this PNG")
    ap.add_argument("--cpu-row-stride", type=int, default=1, help="oracle traces every n-th image row (1: the whole frame, ~10 s of CPU work)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo: rehearsal, tiles gathered via host)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.one_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    rt = importlib.import_module("rust-raytracer_amd")
    ctx = rt.Context(local_rank)
    t0 = time.time()
    scene = rt.HostScene(args.scene, args.width, args.tris)
    t_scene = time.time() - t0
    info = ctx.upload(scene)
    part = (args.tile, rank, world)
    tl = importlib.import_module("rust-raytracer_amd.tiles")
    n_pad = tl.padded_tiles_per_rank(args.width, args.width, args.tile, world)
    assert ctx.partition_pixels(part) <= n_pad * args.tile * args.tile
    tiles = torch.zeros(n_pad * args.tile * args.tile * 3, dtype=torch.uint8, device=device)
    frame = [None]

    # the library renders on its own stream; the gather runs on torch's: order the next frame's
    # writes to `tiles` after the previous frame's all-gather has read them (device-side wait)
    lib_stream = torch.cuda.ExternalStream(ctx.stream, device=device) if world > 1 else None

    def step(flags=0):
        if lib_stream is not None:
            lib_stream.wait_stream(torch.cuda.current_stream(device))
        st = ctx.render_device(tiles.data_ptr(), args.spp, args.bounces, seed=1, part=part, flags=flags)
        if world > 1:  # the frame's one exchange step: RGB8 tiles over RCCL, then de-interleave
            frame[0] = tl.gather_image(tiles, args.width, args.width, args.tile, rank, world, dist)
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    counted = step(rt.FLAG_COUNT_VISITS)  # untimed pass with visit counters on
    fence()
    t0 = time.perf_counter()
    stats = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    rays_step = stats[-1].rays
    if world > 1:
        cdev = device if args.backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays_step], dtype=torch.int64, device=cdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_step = int(rr.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    if args.out:
        img = frame[0] if world > 1 else tl.gather_image(tiles, args.width, args.width, args.tile, 0, 1)
        rt.png_write(args.out, img.cpu().numpy())
    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_step * args.steps / elapsed / 1e6
    # Dominant kernel of rank 0 = the persistent path kernel: ONE launch per batch traces every ray of
    # every bounce level (closest-hit and shadow queries), timed with HIP events on the library's
    # stream (rtx_stats.path_kernel_ms) inside the timed region.
    p_ms = sum(s.path_kernel_ms for s in stats)
    p_l = sum(s.path_launches for s in stats)
    r_ms = sum(s.resolve_ms for s in stats) / args.steps
    avg_launch_ms = p_ms / max(p_l, 1)
    launches_per_step = p_l / args.steps
    c_rays = counted.primary_rays + counted.bounce_rays
    s_rays = counted.shadow_rays
    c_bytes = NODE_BYTES * counted.closest_nodes + PRIM_BYTES * counted.closest_prims + RAY_IO_BYTES * c_rays
    s_bytes = NODE_BYTES * counted.shadow_nodes + PRIM_BYTES * counted.shadow_prims + RAY_IO_BYTES * s_rays
    t_bytes, t_rays = c_bytes + s_bytes, c_rays + s_rays
    bytes_per_launch = t_bytes / max(launches_per_step, 1)
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    traffic = None

    def per_class(nbytes, rays, nodes, prims):
        return {"rays_per_step": int(rays), "algorithmic_bytes_per_step": int(nbytes), "bytes_per_ray": round(nbytes / max(rays, 1), 1),
                "nodes_per_ray": round(nodes / max(rays, 1), 3), "prims_per_ray": round(prims / max(rays, 1), 3)}

    roofline = {"bound": "hbm", "kernel": "path_kernel (closest-hit + shadow queries of all bounce levels, one persistent launch per batch)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(avg_launch_ms, 4),
                "launches_per_step": launches_per_step, "resolve_ms_per_step": round(r_ms, 4), "rays_per_step": int(t_rays),
                "bytes_per_ray": round(t_bytes / max(t_rays, 1), 1),
                "per_class": {"closest": per_class(c_bytes, c_rays, counted.closest_nodes, counted.closest_prims),
                              "shadow": per_class(s_bytes, s_rays, counted.shadow_nodes, counted.shadow_prims)}}

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import orc  # the oracle, as the reported CPU baseline only
        threads = min(16, os.cpu_count() or 1)
        o = orc.Scene(scene)
        ys = max(1, args.cpu_row_stride)
        _, cst = o.render(spp=args.spp, bounces=args.bounces, seed=1, threads=threads,
                          region=(0, args.width, 1, 0, args.width, ys))
        cpu = {"value": round(cst.rays / cst.trace_seconds / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
               "sample": (f"every {ys}. row of the same" if ys > 1 else "one whole") + f" {args.width}x{args.width} {args.spp}-spp frame "
                         f"({cst.rays} rays in {cst.trace_seconds:.1f} s; oracle BVH build {cst.build_seconds:.1f} s extra)"}
        o.close()

    scene_text = SCENE_TEXT.get(args.scene, args.scene)
    data_text = "synthetic"
    if args.scene in ("scan", "scan_glass") and args.tris == 0:
        found = [f for f in rt.SCAN_FILES if os.path.exists(os.path.join(rt.MEDIA_DIR, f))]
        if found:
            data_text = "Stanford scans from assets/ (" + ", ".join(found) + "), rest synthetic"
            scene_text += f"; real Stanford scans used in place of their stand-ins: {', '.join(found)}"
    out = {"metric": "Mrays/s at 2.3M-tri 1024^2 16spp", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": data_text,
           "config": {"workload": f"{args.scene}: {scene_text}, {info.n_objects} primitives, "
                                  f"{args.width}x{args.width}, {args.spp} spp, {args.bounces} bounces",
                      "tile": args.tile, "parallelism": f"image tiles x{world}, scene replicated",
                      "rays_per_step": int(rays_step), "primary_samples_per_step": args.width * args.width * args.spp,
                      "bvh": {"internal_nodes": info.n_internal, "leaves": info.n_leaves, "depth": info.depth,
                              "build_s": round(info.build_seconds, 3), "flatten_s": round(info.flatten_seconds, 3),
                              "upload_s": round(info.upload_seconds, 3), "scene_gen_s": round(t_scene, 3),
                              "device_MB": round(info.device_bytes / 2**20, 1)}},
           "mprimary_samples_per_s": round(args.width * args.width * args.spp * args.steps / elapsed / 1e6, 2),
           "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
