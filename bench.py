#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on the 2.37 M-triangle scene, 1024^2, 16 spp, 4 bounces.

One "step" = one full pass of the trace path (raygen -> the persistent path kernel: closest-hit and
shadow queries of every bounce level -> resolve -> RGB8 tiles) over the frame, scene already
resident in HBM.  Consecutive steps alternate between the context's two render slots, so one frame's
drain (a few long dependent rays) runs under the next frame's bulk; every step is a complete frame.
With N > 1 ranks (one process per GPU, started by torch.distributed.run) the frame is split into interleaved
32x32 tiles, every rank traces its tiles against its own copy of the scene, and the finished RGB8
tiles are all-gathered over RCCL by the library's own communicator; total work is fixed, so scaling is
"strong".  The ranks meet over a plain TCP socket (rust-raytracer_amd/rendezvous.py): this process never
imports torch, so the library runs on the RCCL and HIP runtime it is linked against and tested with
(the line names them: config.gather).  If the communicator cannot be set up the run FAILS (exit 2);
--allow-host-gather lets it go on with the tiles gathered through the host, and the line says so.

    python bench.py --gpus 1 --steps 20 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints one JSON line on rank 0 (see DESIGN.md section 4 for the roofline / cpu_baseline definitions).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8 TB/s spec; random whole-line gathers from a 151 MB table (L2 + Infinity
# Cache) 7.4-7.9 TB/s.  The path kernel's reads are such gathers (one 128-byte node or 64-byte
# primitive record per lane and step); which row bounds them depends on the scene's size.
HBM_PEAK_GBS = 8000.0
MALL_GATHER_PEAK_GBS = 7650.0
MALL_BYTES = 256 << 20
RAY_IO_BYTES = 40      # 32 B ray in + 8 B hit out
NODE_BYTES = 128       # one wide node (two tree levels) = one 128-byte line
PRIM_BYTES = 64
README_BUILD_S = 27.0  # reference README.md:7, 4-core i5-6600K (other hardware)


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="scan")
    ap.add_argument("--tris", type=int, default=0, help="triangle count of the synthetic scene (0 = the config's own)")
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--slots", type=int, default=0, help="render slots the steps alternate between (0: one at N = 1, where the kernel times are then those of launches that have the GPU to themselves, two at N = 2..3, four from N = 4 on)")
    ap.add_argument("--also-two-slots", action="store_true", help="N = 1: after the timed region, time the same steps alternating between two slots and report them as `two_slots`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-visits", action="store_true", help="skip the untimed pass of the verbatim bvh.rs traversal")
    ap.add_argument("--out", default=None, help="rank 0 writes the last frame to this PNG")
    ap.add_argument("--cpu-row-stride", type=int, default=1, help="oracle traces every n-th image row (1: the whole frame, ~10 s of CPU work)")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "torch"],
                    help="N > 1: rccl = the library's own communicator (rtx_comm_init / rtx_gather_image; the ranks meet over a TCP "
                         "socket, torch is not imported), torch = torch.distributed all_gather_into_tensor (round-1 path, for comparison)")
    ap.add_argument("--backend", default="nccl", help="--gather torch: torch.distributed backend (gloo: rehearsal, tiles gathered via host)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--force-comm", action="store_true", help="N = 1: run the N-rank code path (communicator of one rank, rtx_gather_image) all the same")
    ap.add_argument("--allow-host-gather", action="store_true",
                    help="N > 1: if the RCCL communicator cannot be set up, gather the tiles through the host instead of failing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device:
        local_rank = 0
    torch = dist = device = rdzv = None
    multi = world > 1 or args.force_comm
    lib_gather = multi and args.gather == "rccl"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if multi and lib_gather:
        # host-side rendezvous only (RCCL id, barriers, two reductions): no torch in this process
        rdzv = importlib.import_module("rust-raytracer_amd.rendezvous").Rendezvous(rank, world)
    elif multi:
        import torch
        import torch.distributed as dist
        # torch first: its bundled HIP runtime has to initialise before the library's (/opt/rocm) does
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    rt = importlib.import_module("rust-raytracer_amd")
    tl = importlib.import_module("rust-raytracer_amd.tiles") if multi and not lib_gather else None
    ctx = rt.Context(local_rank)
    host_gather = False
    comm_text = None
    if lib_gather:
        # The library's communicator.  Should RCCL refuse on any rank (rendezvous, a trial gather) the run fails: a curve
        # measured through the host must not pass for an RCCL curve.  --allow-host-gather: every rank gathers through the host.
        ok, why = 1, ""
        try:
            cid = rdzv.broadcast(rt.comm_unique_id().hex() if rank == 0 else None)
            ctx.comm_init(bytes.fromhex(cid), rank, world)
            comm_text = rt.comm_info()
        except Exception as e:  # noqa: BLE001
            ok, why = 0, repr(e)
        oks = rdzv.allgather([ok, why])
        if not all(o[0] for o in oks):
            why = "; ".join(f"rank {r}: {o[1]}" for r, o in enumerate(oks) if not o[0])
            if not args.allow_host_gather:
                print(f"[bench rank {rank}] RCCL communicator unavailable ({why}); --allow-host-gather would gather the tiles through the host",
                      file=sys.stderr, flush=True)
                rdzv.close()
                sys.exit(2)
            print(f"[bench rank {rank}] RCCL communicator unavailable ({why}): tiles gathered through the host", file=sys.stderr, flush=True)
            lib_gather, host_gather = False, True
    t0 = time.time()
    scene = rt.HostScene(args.scene, args.width, args.tris)
    t_scene = time.time() - t0
    info = ctx.upload(scene)
    part = (args.tile, rank, world)
    # frames in flight: small shares of the frame want more of them (their fixed tail and launch chain hide under the others' bulk)
    n_slots = max(1, min(args.slots or (1 if world == 1 else 2 if world < 4 else 4), ctx.slots))
    # --also-two-slots: at N = 1 the timed region uses one slot and a second, separately reported region alternates between two
    n_alloc = max(n_slots, min(2, ctx.slots)) if (world == 1 and args.also_two_slots and not multi) else n_slots
    npix = ctx.partition_pixels(part)
    frame = [None]
    if lib_gather:
        nbytes = ctx.gather_bytes_per_rank(args.tile)
        tiles = [rt.DeviceBuffer(nbytes) for _ in range(n_slots)]
        ptrs = [t.ptr for t in tiles]
        all_tiles = [rt.DeviceBuffer(nbytes * world) for _ in range(n_slots)]
        images = [rt.DeviceBuffer(args.width * args.width * 3) for _ in range(n_slots)]
    elif host_gather:
        nbytes = ctx.gather_bytes_per_rank(args.tile)
        tiles = [rt.DeviceBuffer(nbytes) for _ in range(n_slots)]
        ptrs = [t.ptr for t in tiles]
        host_all = [None] * world
    elif multi:
        n_pad = tl.padded_tiles_per_rank(args.width, args.width, args.tile, world)
        assert npix <= n_pad * args.tile * args.tile
        tiles = [torch.zeros(n_pad * args.tile * args.tile * 3, dtype=torch.uint8, device=device) for _ in range(n_slots)]
        ptrs = [t.data_ptr() for t in tiles]
        # the library renders on its slots' streams, the gather runs on torch's: a slot's next frame may
        # overwrite its tile buffer only after the gather of its previous frame has read it
        lib_streams = [torch.cuda.ExternalStream(ctx.slot_stream(k), device=device) for k in range(n_slots)]
    else:
        tiles = [rt.DeviceBuffer(npix * 3) for _ in range(n_alloc)]
        ptrs = [t.ptr for t in tiles]

    issued = [0]
    stats = []

    def step(flags=0, keep=True):
        slot = issued[0] % n_slots
        if issued[0] >= n_slots:
            st = ctx.slot_stats(slot)                 # the frame this slot rendered n_slots steps ago (waits for it only)
            if keep:
                stats.append(st)
        if multi and not lib_gather and not host_gather:
            lib_streams[slot].wait_stream(torch.cuda.current_stream(device))
        ctx.render_device(ptrs[slot], args.spp, args.bounces, seed=1, part=part, flags=flags, stats=False, slot=slot)
        if lib_gather:  # the frame's one exchange step, on the slot's stream: RCCL all-gather of the RGB8 tiles, de-interleave kernel
            ctx.gather_image(ptrs[slot], all_tiles[slot].ptr, images[slot].ptr, slot=slot, tile=args.tile)
        elif host_gather:  # --allow-host-gather: D2H, all-gather over the rendezvous sockets (the image is assembled from host_all on demand)
            ctx.slot_synchronize(slot)
            host_all[:] = rdzv.allgather_bytes(tiles[slot].to_host().tobytes())
        elif multi:  # the same through torch.distributed
            torch.cuda.current_stream(device).wait_stream(lib_streams[slot])
            frame[0] = tl.gather_image(tiles[slot], args.width, args.width, args.tile, rank, world, dist)
        issued[0] += 1

    def drain(keep=True):
        for k in range(max(issued[0] - n_slots, 0), issued[0]):
            st = ctx.slot_stats(k % n_slots)
            if keep:
                stats.append(st)
        issued[0] = 0

    def fence():
        ctx.synchronize()
        if lib_gather or host_gather:
            rdzv.barrier()
            ctx.synchronize()
        elif multi:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(keep=False)
    drain(keep=False)
    # untimed passes with the visit counters on: the shipped traversal, and bvh.rs:359-395 verbatim for orientation
    ctx.render_device(ptrs[0], args.spp, args.bounces, seed=1, part=part, flags=rt.FLAG_COUNT_VISITS, stats=False)
    counted = ctx.slot_stats(0)
    ref_counted = None
    if world == 1 and not args.no_reference_visits:
        ctx.render_device(ptrs[0], args.spp, args.bounces, seed=1, part=part,
                          flags=rt.FLAG_COUNT_VISITS | rt.FLAG_REFERENCE_TRAVERSAL, stats=False)
        ref_counted = ctx.slot_stats(0)
    fence()
    stats.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    assert len(stats) == args.steps, (len(stats), args.steps)
    rays_step = stats[-1].rays
    overlapped = None
    if world == 1 and n_slots == 1 and n_alloc > 1:
        # the same K steps once more, alternating between two slots (the next frame's launch fills the
        # tail of this one's): reported beside `value`, whose kernel times are those of launches alone on the GPU
        timed_stats = list(stats)
        n_slots = n_alloc
        for _ in range(2):
            step(keep=False)
        drain(keep=False)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(keep=False)
        drain(keep=False)
        fence()
        e2 = time.perf_counter() - t0
        overlapped = {"slots": n_alloc, "ms_per_step": round(e2 / args.steps * 1e3, 3), "value": round(rays_step * args.steps / e2 / 1e6, 2), "unit": "Mrays/s"}
        n_slots = 1
        stats[:] = timed_stats
    one_frame = None
    if multi and n_slots > 1:
        # the same K steps with ONE frame in flight (the time of a frame, not the throughput of a pipeline of frames):
        # reported beside `value` so that the scaling curve can be read either way
        timed_stats = list(stats)
        pipelined_slots, n_slots = n_slots, 1
        step(keep=False)
        drain(keep=False)
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(keep=False)
        drain(keep=False)
        fence()
        one_frame = time.perf_counter() - t0
        n_slots = pipelined_slots
        stats[:] = timed_stats
    if lib_gather or host_gather:
        both = rdzv.allgather([elapsed, rays_step, one_frame])
        elapsed = max(b[0] for b in both)
        rays_step = sum(b[1] for b in both)
        if one_frame is not None:
            one_frame = max(b[2] for b in both)
    elif multi:
        cdev = device if args.backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([elapsed, one_frame or 0.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt[0].item())
        one_frame = float(tt[1].item()) if one_frame is not None else None
        rr = torch.tensor([rays_step], dtype=torch.int64, device=cdev)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        rays_step = int(rr.item())
    if rank != 0:
        if rdzv:
            rdzv.close()
        if dist:
            dist.destroy_process_group()
        return

    if args.out:
        if lib_gather:
            img = images[(args.steps - 1) % n_slots].to_host().reshape(args.width, args.width, 3)
        elif host_gather:
            import numpy as np
            img = np.zeros((args.width, args.width, 3), np.uint8)
            for r, blob in enumerate(host_all):   # rank r's packed tiles: [local tile][row][col][rgb]
                t = np.frombuffer(blob, np.uint8).reshape(-1, args.tile, args.tile, 3)
                tx = (args.width + args.tile - 1) // args.tile
                for lt in range(t.shape[0]):
                    gt = lt * world + r
                    if gt >= tx * tx:
                        break
                    y0, x0 = (gt // tx) * args.tile, (gt % tx) * args.tile
                    h, w = min(args.tile, args.width - y0), min(args.tile, args.width - x0)
                    img[y0:y0 + h, x0:x0 + w] = t[lt, :h, :w]
        elif multi:
            img = frame[0].cpu().numpy()
        else:
            img = rt.untile(tiles[(args.steps - 1) % n_slots].to_host(), args.width, args.width, args.tile)
        rt.png_write(args.out, img)
    ms_per_step = elapsed / args.steps * 1e3
    mrays = rays_step * args.steps / elapsed / 1e6
    # Dominant kernel of rank 0 = the persistent path kernel: ONE launch per batch traces every ray of
    # every bounce level (closest-hit and shadow queries), timed with HIP events on the slot's stream
    # (rtx_stats.path_kernel_ms) inside the timed region.  With two slots the launches of consecutive
    # steps overlap, so a launch's own duration is longer than the step time it accounts for.
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
    delivery = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    # the bound: scenes that fit the Infinity Cache are bounded by its gather rate, larger ones by HBM
    in_mall = info.device_bytes <= MALL_BYTES
    peak = MALL_GATHER_PEAK_GBS if in_mall else HBM_PEAK_GBS
    # measured fabric traffic of the kernel (rocprofv3 --pmc FETCH_SIZE, doubled as the guide prescribes), from
    # the committed PMC summary -- valid only for the kernel sources it was measured on and for this workload
    traffic, pmc_note, valu = None, "no PMC summary for this workload", None
    ppath = os.path.join(ROOT, "profiles", "pmc.json")
    wkey = f"{args.scene}:{args.tris}:{args.width}:{args.spp}:{args.bounces}:{world}"
    if os.path.exists(ppath):
        pj = json.load(open(ppath))
        w = pj.get("workloads", {}).get(wkey)
        if w and w.get("kernel_hash") == rt.kernel_source_hash():
            traffic = int(w["path_fetch_bytes_per_launch"])
            valu = w.get("valu")
            pmc_note = f"profiles/pmc.json ({w.get('source', '')})"
        elif w:
            pmc_note = "profiles/pmc.json is stale: kernel sources changed since it was measured"
    if traffic is not None:
        achieved = traffic / (avg_launch_ms * 1e-3) / 1e9
        basis = "measured fabric reads behind L2 (PMC) per launch / launch time"
    else:
        achieved = None
        basis = ("no valid PMC traffic for this kernel and workload: achieved / frac are not stated (record_delivery_GBps is the algorithmic rate -- "
                 "records come from LDS, L1 and L2 too, it can exceed the peak)")
    # what binds: the kernel issues VALU instructions for >= 90 % of its cycles (PMC: SQ_INSTS_VALU x 4 cycles / SIMD / kernel cycles) while
    # the fabric carries a fraction of its peak -> "valu"; else the memory level the scene lives in
    valu_share = (valu or {}).get("issue_cycles_over_kernel_cycles")
    bound = "valu" if (valu_share or 0.0) >= 0.9 else ("mall" if in_mall else "hbm")

    def per_class(nbytes, rays, nodes, prims):
        return {"rays_per_step": int(rays), "algorithmic_bytes_per_step": int(nbytes), "bytes_per_ray": round(nbytes / max(rays, 1), 1),
                "nodes_per_ray": round(nodes / max(rays, 1), 3), "prims_per_ray": round(prims / max(rays, 1), 3)}

    # The memory side (fabric reads behind L2 against the level the scene lives in) and, when the PMC summary is valid, the
    # VALU side: lane-instructions = VALU instructions x the lanes that executed them, against the rate at which this chip
    # retires unpacked f32 vector instructions, MEASURED (tools/valu_ubench.hip, profiles/r03_valu_ubench.txt): with 5-8 waves
    # per SIMD a SIMD takes 3.9-4.4 cycles per wave64 v_fma_f32 / v_min_f32 / v_max3_f32 / v_cndmask_b32 -- 16 lanes per
    # cycle, 40 T lane-instr/s over 1,024 SIMDs at the clock it held -- 2.4-2.8 cycles per v_mul_f32 / v_sub_f32 (two waves
    # issue together) and 4.3-4.6 per v_pk_*_f32 (two operations per lane: the 157.3 TFLOP/s of the data sheet, 78.6 T
    # lane-FMA/s, is v_pk_fma_f32).  The path kernel is comparisons, minima, maxima, selects and divisions: the 4-cycle
    # class is its ceiling; packing its 48 slab subtractions / multiplications per node was measured (+1.6 %).
    memory = {"achieved": round(achieved, 2) if achieved is not None else None, "peak": peak, "unit": "GB/s",
              "frac": round(achieved / peak, 5) if achieved is not None else None,
              "peak_is": ("random 128-byte gathers from L2 + Infinity Cache (scene %.0f MB <= 256 MB)" % (info.device_bytes / 2**20)) if in_mall
                         else "HBM3E (scene %.0f MB > Infinity Cache)" % (info.device_bytes / 2**20),
              "frac_basis": basis}
    VALU_PEAK_T = 40.2          # measured: v_fma_f32, 8 waves per SIMD, 1.632 ns per wave-instruction per SIMD
    VALU_PEAK_PACKED_T = 78.6   # data sheet: 157.3 TFLOP/s of packed f32 FMA
    if valu:
        li = valu["insts_per_launch"] * 64.0 * valu["active_lane_fraction"]
        rate = li / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0
        valu = dict(valu, lane_instr_per_ray=round(li / max(t_rays / max(launches_per_step, 1), 1), 1),
                    lane_instr_rate_T_per_s=round(rate, 2), lane_instr_peak_T_per_s=VALU_PEAK_T, lane_instr_peak_packed_T_per_s=VALU_PEAK_PACKED_T,
                    note="issue_cycles_over_kernel_cycles prices a wave64 VALU instruction at 4 cycles of its SIMD -- what tools/valu_ubench.hip "
                         "measures for v_fma / v_min / v_max3 / v_cndmask with 5-8 waves per SIMD (profiles/r03_valu_ubench.txt; subtractions and "
                         "multiplications retire in 2.4-2.8): 0.9+ means the SIMDs' vector issue is taken, at active_lane_fraction of the lanes; "
                         "lane_instr_peak_T_per_s is that measured unpacked rate, lane_instr_peak_packed_T_per_s the data sheet's packed-FMA rate")
    if bound == "valu":
        head = {"achieved": valu["lane_instr_rate_T_per_s"], "peak": VALU_PEAK_T, "unit": "T lane-instr/s",
                "frac": round(valu["lane_instr_rate_T_per_s"] / VALU_PEAK_T, 5)}
    else:
        head = {k: memory[k] for k in ("achieved", "peak", "unit", "frac")}
    roofline = {"bound": bound, "kernel": "path_kernel (closest-hit + shadow queries of all bounce levels, one persistent launch per batch)",
                **head, "traffic": traffic, "memory": memory, "pmc": pmc_note,
                "record_delivery_GBps": round(delivery, 2),   # algorithmic bytes / time: records come from LDS, L1, L2 too
                "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(avg_launch_ms, 4),
                "launches_per_step": launches_per_step, "slots": n_slots, "resolve_ms_per_step": round(r_ms, 4),
                "rays_per_step": int(t_rays), "bytes_per_ray": round(t_bytes / max(t_rays, 1), 1), "valu": valu,
                "max_records_one_ray": int(counted.max_ray_records),
                "per_class": {"closest": per_class(c_bytes, c_rays, counted.closest_nodes, counted.closest_prims),
                              "shadow": per_class(s_bytes, s_rays, counted.shadow_nodes, counted.shadow_prims)}}
    if ref_counted is not None:  # the reference's own traversal (both children, no pruning) on the same rays, 64-byte binary nodes
        roofline["reference_traversal_visits"] = {
            "closest": {"nodes_per_ray": round(ref_counted.closest_nodes / max(c_rays, 1), 2), "prims_per_ray": round(ref_counted.closest_prims / max(c_rays, 1), 2)},
            "shadow": {"nodes_per_ray": round(ref_counted.shadow_nodes / max(s_rays, 1), 2), "prims_per_ray": round(ref_counted.shadow_prims / max(s_rays, 1), 2)},
            "path_kernel_ms": round(ref_counted.path_kernel_ms, 2)}

    cpu = None
    build = {"hip_backend_s": round(info.build_seconds, 3), "reference_README_s": README_BUILD_S}
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
        build["oracle_s"] = round(cst.build_seconds, 3)
        # the reference's own default: -t 4 (main.rs:45; README.md:7 quotes a 4-core run) on a quarter of the rows
        ys4 = ys * 4
        _, c4 = o.render(spp=args.spp, bounces=args.bounces, seed=1, threads=4, region=(0, args.width, 1, 0, args.width, ys4))
        cpu["threads_4"] = {"value": round(c4.rays / c4.trace_seconds / 1e6, 3), "unit": "Mrays/s", "cores": 4,
                            "sample": f"every {ys4}. row ({c4.rays} rays in {c4.trace_seconds:.1f} s)",
                            "reference_README": "39 s for this frame incl. 27 s BVH build on a 4-core i5-6600K (README.md:7): about 6 Mrays/s"}
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
                      "gather": None if not multi else (f"rtx_gather_image (library RCCL communicator): {comm_text}" if lib_gather else
                                                        "FALLBACK (--allow-host-gather): tiles through the host over the rendezvous sockets (RCCL communicator unavailable)" if host_gather else
                                                        f"torch.distributed {args.backend}"),
                      "frames_in_flight": n_slots,
                      "one_frame_in_flight": None if one_frame is None else
                          {"ms_per_step": round(one_frame / args.steps * 1e3, 3), "value": round(rays_step * args.steps / one_frame / 1e6, 2), "unit": "Mrays/s",
                           "what": "the same K steps with ONE frame in flight: the time of a frame; `value` is the throughput of frames_in_flight frames in rotation"},
                      "rays_per_step": int(rays_step), "primary_samples_per_step": args.width * args.width * args.spp,
                      "bvh": {"internal_nodes": info.n_internal, "leaves": info.n_leaves, "depth": info.depth,
                              "build_s": round(info.build_seconds, 3), "flatten_s": round(info.flatten_seconds, 3),
                              "upload_s": round(info.upload_seconds, 3), "scene_gen_s": round(t_scene, 3),
                              "device_MB": round(info.device_bytes / 2**20, 1), "build_comparison": build}},
           "mprimary_samples_per_s": round(args.width * args.width * args.spp * args.steps / elapsed / 1e6, 2),
           "two_slots": overlapped, "roofline": roofline, "cpu_baseline": cpu}
    print(json.dumps(out), flush=True)
    if rdzv:
        rdzv.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
