#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

A step is one frame: every pixel's camera ray, BVH closest hit, shadow rays, shading and
the 3-bounce reflection chain, with the scene already resident in HBM.  For N > 1 the
frame is cut into interleaved screen tiles (tile k -> rank k mod N), each rank renders
its tiles on its own GPU, the packed tile buffers are all-gathered with RCCL and
re-assembled into the row-major frame: total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0 (metric = Mrays/s over primary + reflection + shadow rays).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
L2_PEAK_GBS = 34500.0   # aggregate of the eight XCDs' L2s
CLOCK_GHZ = 2.4


def profiled_counters(workload_key, kernel_build):
    """Counters per launch of the render kernel from the committed PMC passes (profiles/traffic.json, written by
    tools/pmc_summary.py from `rocprofv3 --pmc` runs of the same frame): HBM bytes = FETCH_SIZE doubled, as the gfx950
    note on that counter prescribes, + WRITE_SIZE (both reported in KiB); wave-level VALU instructions = SQ_INSTS_VALU;
    lanes active per VALU instruction = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU.  An entry counts only if it was
    taken from the kernels this process runs (`kernel_build` = the hash of their sources in rtx_version()).
    Returns (counters or None, reason or None)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[workload_key]
    except Exception:
        return None, "no committed counter pass (profiles/traffic.json) for this workload"
    if t.get("kernel_build") != kernel_build:
        return None, (f"the committed counter pass of this workload was taken from kernel build {t.get('kernel_build')}, "
                      f"this library is build {kernel_build}")
    c = {"bytes": int((2 * t["FETCH_SIZE_KiB"] + t["WRITE_SIZE_KiB"]) * 1024), "source": t["source"],
         "valu_instructions": t.get("SQ_INSTS_VALU"), "kernel_build": t["kernel_build"]}
    if t.get("SQ_THREAD_CYCLES_VALU") and t.get("SQ_ACTIVE_INST_VALU"):
        c["lanes_per_valu_instruction"] = t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"]
    c["l1_accesses"] = t.get("TCP_TOTAL_CACHE_ACCESSES_sum")
    c["l2_requests"], c["l2_hits"] = t.get("TCC_REQ_sum"), t.get("TCC_HIT_sum")
    return c, None


def cpu_baseline(scene_name, n_tris, W, H, depth, obj_path):
    """Times the CPU path on the host cores: oracle/_ref/ref_bvh (the reference's own headers,
    compiled in the build container) when present, else the oracle restatement."""
    # A one-GPU box gives a job 16 host cores, whatever `nproc` says about the machine (many times as many): the
    # baseline runs on that share, as the reference runs on all of ITS cores (8 threads, src/main.cpp:35)
    nproc = os.cpu_count() or 1
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = nproc
    cores = max(1, min(share, int(os.environ.get("RTX_CPU_BASELINE_THREADS", "16"))))
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_bvh")
    cli = os.path.join(ROOT, "oracle", "oracle_cli")
    scene_root = os.path.dirname(os.path.dirname(obj_path))
    linked = None
    if os.path.basename(os.path.dirname(obj_path)) != "scenes" or os.path.basename(obj_path) != scene_name + ".obj":
        # the reference's factories open scenes/<name>.obj under the current directory (scene.hpp:262): give them that view
        import tempfile
        linked = tempfile.mkdtemp(prefix="rtx_bench_scene_")
        os.makedirs(os.path.join(linked, "scenes"))
        os.symlink(os.path.abspath(obj_path), os.path.join(linked, "scenes", scene_name + ".obj"))
        scene_root = linked
    try:
        if os.path.exists(ref):
            cmd, kind = [ref, scene_name, scene_root, str(W), str(H), str(depth), "-", str(cores), "3"], "reference"
        elif os.path.exists(cli):
            cmd, kind = [cli, scene_name, obj_path, str(W), str(H), str(depth), "-", str(cores), "1", "1", "3"], "port"
        else:
            return None
        def frames(command):
            out = subprocess.run(command, capture_output=True, text=True, timeout=900).stdout
            return [float(l.split()[1]) for l in out.splitlines() if l.startswith("frame_seconds")]

        secs = frames(cmd)
        # the reference times three frames in a row (src/main.cpp:42-61); the median one counts here
        res = {"kind": kind, "cores": cores, "nproc": nproc, "affinity": share, "seconds": sorted(secs)[len(secs) // 2], "all": secs,
               "frames": len(secs), "cmd": os.path.basename(cmd[0])}
        # the same program built for this host's vector units (-march=x86-64-v3, contraction off: same pixels)
        v3 = ref + "_v3"
        try:
            flags = open("/proc/cpuinfo").read()
            if kind == "reference" and os.path.exists(v3) and " avx2" in flags and " fma" in flags:
                s3 = frames([v3] + cmd[1:])
                if s3:
                    res["v3_seconds"] = sorted(s3)[len(s3) // 2]
        except Exception:
            pass
        return res
    except Exception as e:  # a missing baseline must not take the GPU number down with it
        return {"error": repr(e)}
    finally:
        if linked:
            import shutil
            shutil.rmtree(linked, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="dragon")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--mode", choices=["reference", "pruned"], default="pruned")
    ap.add_argument("--tile", type=int, default=32)
    ap.add_argument("--inflight", type=int, default=None,
                    help="frames in flight: step i runs on stream i %% INFLIGHT, so the end of a frame (lanes running dry) "
                         "overlaps the start of the next one; 1 = every frame waits for the previous one; 4 = one stream "
                         "per hardware queue of the process (the default); from 8 ranks up the default is 8 and the process asks for 8 "
                         "hardware queues: a rank's share of a frame is short then, so more of them overlap")
    ap.add_argument("--family", choices=["blob", "knot"], default="knot",
                    help="stand-in mesh family of the headline: knot = a knotted tube (concave, self-occluding, self-reflecting) in a "
                         "box that fills the view: 6.4 rays per pixel, the ray statistics of the reference's one real scene "
                         "(teapot: 6.1); blob = a bumpy star-shaped surface in a box around z = 0: 2.8 rays per pixel, most "
                         "reflected rays leave the scene.  The other family is measured in the "
                         "same run and reported as `other_family`")
    ap.add_argument("--no-other-family", action="store_true", help="skip the second stand-in family")
    ap.add_argument("--obj", default=None, help="real .obj for the scene (default: the seeded stand-in mesh)")
    ap.add_argument("--tolerant", action="store_true",
                    help="read --obj with the general loader (vt/vn index groups, negative indices, comments) instead of the "
                         "reference's grammar; the last face is read twice, as the reference's own pipeline (tools/denormalize.py, "
                         "then parseOBJ) reads it, so that triangle ids and materials match the reference's render of the mesh")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo stages the tile gather through host memory: a rehearsal of the N > 1 path on a box with fewer "
                         "GPUs than ranks (every rank then renders on --device); never a benchmark")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default LOCAL_RANK)")
    ap.add_argument("--cpu-sample-div", type=int, default=1, help="CPU baseline renders (W/div) x (H/div)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand without a launcher: become the launcher (a child process; nothing here has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 2000), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    if args.inflight is None:
        args.inflight = 8 if int(os.environ.get("WORLD_SIZE", "1")) >= 8 else 4       # measured: tools/shard_time.py
    if args.inflight > 4 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # Read by the HIP runtime when it starts; 4 is its default.  Frames in flight overlap only while their streams sit on
        # different hardware queues, and a process's live streams share the queues in turn: with RCCL's own streams in the
        # process, four queues no longer keep four frames apart (eight queues with four frames measure the same as four).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    # Libraries (gloo, RCCL, the HIP runtime) print notices on file descriptor 1; the one JSON line must be all that
    # reaches standard output, so everything else is sent to standard error and the line is written to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import ray_tracer_amd as rt

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the render path)")
    ordinal = local_rank if args.device is None else args.device
    torch.cuda.set_device(ordinal)
    dev = torch.device("cuda", ordinal)
    if world > 1:
        if args.backend == "nccl":
            # the gather's kernels compete with persistent render launches that fill every CU: a high-priority
            # stream lets them take the first slots that come free
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:
                opts = None
            dist.init_process_group("nccl", device_id=dev, pg_options=opts)
        else:
            dist.init_process_group("gloo")

    def all_reduce_sum(t):
        if args.backend == "nccl":
            dist.all_reduce(t)
        else:
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)

    W, H = args.width, args.height
    mode = rt.TRAVERSAL_PRUNED if args.mode == "pruned" else rt.TRAVERSAL_REFERENCE
    tmp = os.environ.get("TMPDIR", "/tmp")
    F = max(1, min(8, args.inflight))
    packed = world > 1
    kernel_build = rt.kernel_build()
    props = torch.cuda.get_device_properties(dev)
    simds = 4 * int(props.multi_processor_count)
    from ray_tracer_amd.distributed import ShardedRenderer

    # N > 1: every rank must see the same world through RCCL itself, whatever the launcher's environment says
    ranks_seen = None
    if world > 1:
        one = torch.ones(1, dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    def measure(family, steps, warmup, headline):
        """One stand-in family (or the real mesh): scene into HBM, census, the timed loop with F frames in flight, a frame
        at a time, and for the headline the other traversal mode.  Returns what the JSON line is made of."""
        obj_dir = os.path.join(tmp, f"rtx_bench_{os.getpid()}_{family}", "scenes")
        obj_path = os.path.join(obj_dir, f"{args.scene}.obj")
        t0 = time.time()
        if args.obj:
            host = rt.HostScene.load_obj(args.obj, args.scene, no_bvh=True, tolerant=args.tolerant, last_face_twice=args.tolerant)
            obj_path = args.obj
        else:
            want_obj = headline and rank == 0 and not args.no_cpu_baseline and world == 1
            if want_obj:
                os.makedirs(obj_dir, exist_ok=True)
            host = rt.HostScene.standin(args.scene, seed=0, obj_path_out=obj_path if want_obj else None, no_bvh=True, family=family)
        load_s = time.time() - t0
        t0 = time.time()
        scene = rt.Scene(host, device=ordinal)        # the BVH is built and flattened on the GPU
        create_s = time.time() - t0

        # the product's multi-GPU frame loop: render own tiles -> all-gather -> re-assembly, F frames in flight
        renderer = ShardedRenderer(scene, W, H, rank, world, dev, tile=args.tile, inflight=F, depth=args.depth, spp=args.spp,
                                   traversal=mode)
        params = renderer.params
        frames, buf = renderer.frames, renderer.locals[0]
        stream = renderer.streams[0].cuda_stream

        def census(traversal):
            """Untimed pre-pass with the counting kernel: rays, work and scheduling of this rank's tiles."""
            st = torch.zeros(16, dtype=torch.int64, device=dev)
            p = rt.make_params(W, H, depth=args.depth, spp=args.spp, traversal=traversal, tile=(args.tile, args.tile),
                               tile_rank=rank, tile_count=world, layout=params.layout)
            scene.render_device(p, rgb_u8=buf.data_ptr(), stats=st.data_ptr(), stream=stream)
            if world > 1:
                all_reduce_sum(st)
            torch.cuda.synchronize()
            v = st.cpu().numpy()
            names = ["primary_rays", "reflection_rays", "shadow_rays", "shadow_occluded", "closest_hits", "node_visits",
                     "box_hits", "leaf_visits", "triangle_tests"]
            out = {n: int(v[i]) for i, n in enumerate(names)}
            # rounds a wave ran of each step kind and the lanes that took part: the scheduler's lane utilisation
            sched = {k: (round(int(v[12 + i]) / (64.0 * int(v[9 + i])), 4) if int(v[9 + i]) else None)
                     for i, k in enumerate(("node_pair_steps", "triangle_steps", "shading_steps"))}
            sched["all_steps"] = round(int(v[12:15].sum()) / (64.0 * max(1, int(v[9:12].sum()))), 4)
            return out, sched

        cen_ref, _ = census(rt.TRAVERSAL_REFERENCE)     # canonical (reference traversal) work, whole frame
        cen_own, sched_own = census(mode)               # what the timed kernel itself visits, and how full its rounds are
        # The frame's rays as the reference traces them (the unit of `value` and of the CPU baseline).  The pruned walk does not
        # ask a shadow query whose light cannot add anything to the surface (it faces away and no specular term applies:
        # scene.hpp:165-180), so it traces fewer rays for the same pixels; that count is reported beside it.
        rays = cen_ref["primary_rays"] + cen_ref["reflection_rays"] + cen_ref["shadow_rays"]
        rays_traced = cen_own["primary_rays"] + cen_own["reflection_rays"] + cen_own["shadow_rays"]

        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        # device-side start (as its complement) and end of every timed launch (100 MHz ticks), left by the kernel itself in
        # a zeroed pair
        ticks = torch.zeros(max(1, steps) * 2, dtype=torch.int64, device=dev)
        for k in range(warmup):
            renderer.submit()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            renderer.submit(events=ev[i], ticks=ticks[2 * i:].data_ptr())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        event_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, steps)
        tk = ticks.cpu().numpy().astype("uint64").reshape(-1, 2)
        kernel_ms = float(((tk[:, 1] - ~tk[:, 0]).astype("float64") * 1e-5).mean())     # [0] holds the complement of the start
        frame_crc = None
        if rank == 0:
            import zlib
            frame_crc = "%08x" % zlib.crc32(frames[(warmup + steps - 1) % F].cpu().numpy().tobytes())

        # one frame at a time on one stream, each waited for: the latency of a frame (the headline is throughput)
        n1 = max(3, min(10, steps))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(n1):
            renderer.submit()
            renderer.synchronize()
        alone_ms = (time.perf_counter() - t1) * 1e3 / n1

        m = dict(family=family, host=host, scene=scene, renderer=renderer, obj_path=obj_path, obj_dir=obj_dir, load_s=load_s,
                 create_s=create_s, cen_ref=cen_ref, cen_own=cen_own, sched=sched_own, rays=rays, rays_traced=rays_traced, elapsed=elapsed, steps=steps,
                 ms_per_step=elapsed * 1e3 / steps, event_ms=event_ms, kernel_ms=kernel_ms, crc=frame_crc, alone_ms=alone_ms)

        # several frames per launch (rtx_render_device_batch): the overlap of frames in flight without their streams
        if world == 1:
            n_batch = 4
            big = torch.zeros(n_batch * H * W * 3, dtype=torch.uint8, device=dev)
            scene.render_device_batch(params, n_batch, rgb_u8=big.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            nb = max(2, min(10, steps // n_batch))
            t2 = time.perf_counter()
            for _ in range(nb):
                scene.render_device_batch(params, n_batch, rgb_u8=big.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            m["batch_ms_per_frame"] = (time.perf_counter() - t2) * 1e3 / (nb * n_batch)
            m["batch_frames"] = n_batch
            del big

        if headline:
            # a second, short measurement of the other traversal mode (reported, never the headline)
            other_mode = rt.TRAVERSAL_REFERENCE if mode == rt.TRAVERSAL_PRUNED else rt.TRAVERSAL_PRUNED
            p2 = rt.make_params(W, H, depth=args.depth, spp=args.spp, traversal=other_mode, tile=(args.tile, args.tile),
                                tile_rank=rank, tile_count=world, layout=params.layout)
            n2 = max(2, min(5, steps))
            scene.render_device(p2, rgb_u8=buf.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n2):
                scene.render_device(p2, rgb_u8=buf.data_ptr(), stream=stream)
            e1.record()
            torch.cuda.synchronize()
            m["other_mode"] = other_mode
            m["other_kernel_ms"] = e0.elapsed_time(e1) / n2
        return m

    def workload_text(m):
        return (f"{args.scene} ({m['host'].n_triangles} triangles, {'real' if args.obj else 'stand-in, ' + m['family'] + ' family'} mesh) "
                f"{W}x{H}, depth {args.depth}, {args.spp} spp, traversal={args.mode}, {F} frame(s) in flight")

    def roofline_of(m):
        """The roofs of one measurement.  What binds this kernel is VALU issue: FP64 throughout, one wave64 instruction
        occupies a 16-lane SIMD for 4 cycles, so a launch's wave-level VALU instructions (committed counter pass of this very
        frame and this very kernel build) x 4 cycles over all SIMDs is the least time its instruction stream needs.
        achieved / peak are in lane-operations per second (peak = SIMDs x 16 lanes x clock; the 78.6 TFLOP/s FP64 vector
        figure is this peak with every slot an FMA)."""
        scene_key = args.scene if m["family"] == "blob" else f"{args.scene}_{m['family']}"
        counters, why_not = (profiled_counters(f"{scene_key}_{W}x{H}_{args.mode}", kernel_build)
                             if world == 1 and not args.obj and args.spp == 1 else (None, "counter passes exist for the single-GPU stand-in frames at one sample per pixel"))
        ms = m["ms_per_step"]
        peak_laneops = simds * 16 * CLOCK_GHZ * 1e9
        cen_ref, cen_own = m["cen_ref"], m["cen_own"]
        alg_bytes_frame = cen_ref["node_visits"] * 64 + cen_ref["triangle_tests"] * 96       # the REFERENCE traversal's bytes
        own_bytes = cen_own["node_visits"] * 64 + cen_own["triangle_tests"] * 112            # what the timed kernel itself requests
        launch_alg_bytes = alg_bytes_frame / world         # this rank's launch covers 1/world of the tiles
        if counters and counters.get("valu_instructions"):
            valu = float(counters["valu_instructions"])
            achieved = valu * 64 / (ms * 1e-3)
            lanes = counters.get("lanes_per_valu_instruction")
            r = {
                "bound": "valu_issue",
                "achieved": round(achieved / 1e12, 3),
                "peak": round(peak_laneops / 1e12, 3),
                "unit": "Tlaneop/s",
                "frac": round(achieved / peak_laneops, 4),
                # issue slots are not useful work: lanes switched on per VALU instruction (of 64), and the fraction of
                # the peak that did arithmetic for a live ray
                "lane_utilisation": None if not lanes else round(lanes / 64.0, 4),
                "frac_useful_lanes": None if not lanes else round(achieved / peak_laneops * lanes / 64.0, 4),
                "wave_valu_instructions_per_launch": int(valu),
                "min_ms_per_launch": round(valu * 4 / (simds * CLOCK_GHZ * 1e9) * 1e3, 3),
                "counter_source": counters["source"],
                "counter_build": counters["kernel_build"],
            }
        else:
            r = {"bound": "valu_issue", "achieved": None, "peak": round(peak_laneops / 1e12, 3), "unit": "Tlaneop/s", "frac": None,
                 "lane_utilisation": None, "reason": why_not}
        r.update({
            "kernel_build": kernel_build,
            # lanes that took part per round of the wave scheduler, counted by the kernel itself in the census pass
            "lane_utilisation_by_step_kind": m["sched"],
            # HBM side: bytes that crossed the fabric per launch (counter pass) over the time per frame
            "traffic": counters["bytes"] if counters else None,
            "traffic_reason": None if counters else why_not,
            "hbm": None if not counters else {
                "achieved": round(counters["bytes"] / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(counters["bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
            # vector L1 (one per CU): tag accesses of a launch (counter pass) per clock and CU over the time per frame; every
            # 16-byte load of a lane that does not share its line with a neighbour is one access, and the kernel lives on such loads
            "l1": None if not (counters and counters.get("l1_accesses")) else {
                "accesses_per_launch": int(counters["l1_accesses"]),
                "per_clock_per_cu": round(counters["l1_accesses"] / (ms * 1e-3 * CLOCK_GHZ * 1e9) / max(1, simds // 4), 4),
                "l2_hit_rate": None if not counters.get("l2_requests") else round(counters["l2_hits"] / counters["l2_requests"], 4)},
            # cache side: the bytes the timed kernel asks for (its own census: 128 B per node pair, 112 B per triangle;
            # the top of the tree is served from LDS, the rest by L1 / L2 / Infinity Cache) against the L2s' aggregate peak
            "l2": {"own_bytes_per_launch": int(own_bytes / world), "achieved": round(own_bytes / world / (ms * 1e-3) / 1e9, 1),
                   "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(own_bytes / world / (ms * 1e-3) / 1e9 / L2_PEAK_GBS, 4)},
            "kernel": "render_kernel",
            "kernel_ms": round(m["kernel_ms"], 4),
            "kernel_ms_by_stream_events": round(m["event_ms"], 4),
            "launches_in_flight": F,
            # the canonical work unit: bytes of the REFERENCE traversal (node visits x 64 B + triangle tests x 96 B).  The
            # pruned walk returns the same pixels with fewer visits, so these are not a fraction of any roof: they say how
            # much reference work a second of this kernel stands for
            "canonical_bytes_per_ray": round(alg_bytes_frame / max(1, m["rays"]), 1),
            "canonical_bytes_per_launch": int(launch_alg_bytes),
            "canonical_GBps_per_launch": round(launch_alg_bytes / (m["kernel_ms"] * 1e-3) / 1e9, 1),
            "canonical_GBps_aggregate": round(launch_alg_bytes / (ms * 1e-3) / 1e9, 1),
            "note": "frac = wave-level VALU instructions of one launch x 64 lanes / time per frame / (SIMDs x 16 lanes x 2.4 GHz): "
                    "<= 1 by construction, and an occupancy of issue slots, not useful work: frac_useful_lanes scales it by the "
                    "lanes switched on per instruction. kernel_ms is the average duration of a launch in the timed region (first "
                    "block start to last block end on the device clock, what a kernel trace reports); with launches_in_flight > 1 "
                    "that many launches share the chip, so every rate here divides by the time per frame (ms_per_step), not by kernel_ms.",
        })
        return r, launch_alg_bytes

    head = measure(args.family, args.steps, args.warmup, True)
    other = None
    if world == 1 and not args.obj and not args.no_other_family:
        # the scene handle of the headline stays alive for the CPU baseline's ray count; a dragon is 250 MB of 288 GB
        other = measure("blob" if args.family == "knot" else "knot", max(8, min(args.steps, 40)), args.warmup, False)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = head["ms_per_step"]
    rays = head["rays"]
    mrays = rays / (ms_per_step * 1e-3) / 1e6
    roofline, launch_alg_bytes = roofline_of(head)
    host, scene = head["host"], head["scene"]
    result = {
        "metric": "Mrays/s (primary+reflection+shadow)",
        "value": round(mrays, 2),
        "unit": "Mrays/s",
        # `value` counts the frame's rays as the reference traces them (what the CPU baseline is measured in); the timed
        # kernel skips shadow queries that cannot change the pixel, so it traces fewer: the same frames per second in its own rays
        "value_in_rays_the_timed_kernel_traced": round(head["rays_traced"] / (ms_per_step * 1e-3) / 1e6, 2),
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        # a frame rendered alone and waited for (ms_per_step is throughput: F frames in flight share the chip)
        "frame_ms_latency": round(head["alone_ms"], 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "real mesh" if args.obj else "synthetic",
        "config": {
            "workload": workload_text(head),
            "parallelism": f"screen tiles {args.tile}x{args.tile} interleaved over {world} GPU(s)"
                           + ((", RCCL all-gather of packed tiles" if args.backend == "nccl" else ", gloo all-gather through host memory (rehearsal)") if packed else ""),
            "rays_per_frame": rays,
            "rays_per_pixel": round(rays / (W * H), 3),
            "rays_traced_per_frame_by_the_timed_kernel": head["rays_traced"],
            "bvh_nodes": scene.bvh_nodes,
            "bvh_depth": scene.bvh_depth,
            "scene_hbm_bytes": scene.device_bytes,
            "mesh_load_seconds": round(head["load_s"], 3),
            "scene_create_seconds": round(head["create_s"], 4),
            "bvh_build_kernels_ms": round(scene.build_seconds * 1e3, 3),
            "frame_crc32": head["crc"],
            "frames_in_flight": F,
            "ms_per_frame_one_at_a_time": round(head["alone_ms"], 4),
            "ms_per_frame_four_frames_per_launch": None if "batch_ms_per_frame" not in head else round(head["batch_ms_per_frame"], 4),
            "rccl_ranks_seen": ranks_seen,
        },
        "roofline": roofline,
        "other_mode": {
            "traversal": "reference" if head["other_mode"] == rt.TRAVERSAL_REFERENCE else "pruned",
            "kernel_ms": round(head["other_kernel_ms"], 4),
            "mrays_per_s_kernel_only": round(rays / world / (head["other_kernel_ms"] * 1e-3) / 1e6, 2),
            "canonical_GBps": round(launch_alg_bytes / (head["other_kernel_ms"] * 1e-3) / 1e9, 1),
        },
        "census_reference_traversal": head["cen_ref"],
        "census_timed_traversal": head["cen_own"],
    }
    if other:
        o_roof, _ = roofline_of(other)
        result["other_family"] = {
            "workload": workload_text(other),
            "value": round(other["rays"] / (other["ms_per_step"] * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "steps": other["steps"],
            "ms_per_step": round(other["ms_per_step"], 4),
            "frame_ms_latency": round(other["alone_ms"], 4),
            "ms_per_frame_four_frames_per_launch": None if "batch_ms_per_frame" not in other else round(other["batch_ms_per_frame"], 4),
            "rays_per_frame": other["rays"],
            "rays_traced_per_frame_by_the_timed_kernel": other["rays_traced"],
            "rays_per_pixel": round(other["rays"] / (W * H), 3),
            "frame_crc32": other["crc"],
            "roofline": {k: o_roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "lane_utilisation", "frac_useful_lanes",
                                                    "counter_source", "counter_build", "reason", "lane_utilisation_by_step_kind",
                                                    "traffic", "kernel_ms")},
        }
    if world == 1 and not args.no_cpu_baseline:
        div = max(1, args.cpu_sample_div)
        cw, ch = (W // div) // 8 * 8, (H // div) // 2 * 2
        cb = cpu_baseline(args.scene, host.n_triangles, cw, ch, args.depth, head["obj_path"])
        if cb and "seconds" in cb:
            # rays of the sample: same scene and camera at (W/div)x(H/div); rays/pixel measured on the GPU census
            ps = rt.make_params(cw, ch, depth=args.depth, traversal=rt.TRAVERSAL_REFERENCE)
            st = torch.zeros(16, dtype=torch.int64, device=dev)
            sbuf = torch.zeros(cw * ch * 3, dtype=torch.uint8, device=dev)
            scene.render_device(ps, rgb_u8=sbuf.data_ptr(), stats=st.data_ptr(), stream=head["renderer"].streams[0].cuda_stream)
            torch.cuda.synchronize()
            v = st.cpu().numpy()
            srays = int(v[0] + v[1] + v[2])
            result["cpu_baseline"] = {
                "value": round(srays / cb["seconds"] / 1e6, 3),
                "unit": "Mrays/s",
                "cores": cb["cores"],
                "nproc": cb["nproc"],
                "cores_note": f"os.cpu_count() = {cb['nproc']}, affinity mask = {cb['affinity']} CPUs; a one-GPU job's share of the "
                              f"box is 16 cores, so the baseline runs {cb['cores']} threads (RTX_CPU_BASELINE_THREADS overrides)",
                "kind": cb["kind"],
                "sample": f"same scene ({head['family']} family) and camera at {cw}x{ch} ({srays} rays), median of {cb['frames']} frames "
                          f"({', '.join('%.3f' % x for x in cb['all'])} s) of {cb['cmd']} with {cb['cores']} threads",
            }
            if "v3_seconds" in cb:
                result["cpu_baseline"]["built_for_avx2_fma"] = {
                    "value": round(srays / cb["v3_seconds"] / 1e6, 3), "unit": "Mrays/s",
                    "flags": "-O3 -march=x86-64-v3 -ffp-contract=off (the reference's Makefile uses plain -O3)"}
        else:
            result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": repr(cb)}
    for m in (head, other):
        if m and not args.obj and os.path.isdir(os.path.dirname(m["obj_dir"])):
            import shutil
            shutil.rmtree(os.path.dirname(m["obj_dir"]), ignore_errors=True)      # the stand-in .obj written for the CPU baseline
    sys.stdout.flush()
    os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
