#!/usr/bin/env python3
"""bench.py -- primary Mrays/s and frame ms of the ray-casting hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode frames|bands] [--sequence]
                    [--config c3_3840x2160_n1024] [--variant tiled|literal] [--no-cpu]

A step is one pass of the hot path (rc_render -> the draw kernel) over one frame of the
workload, inputs (scene arrays, sine tables) already resident in HBM.  Default workload:
3840x2160, 1024 spheres, camera (0,0,20) -- the configuration the headline metric is quoted on.

Multi-GPU (one rank per GPU; started by torch.distributed.run, or -- when `--gpus N` is given
to a plain `python bench.py` -- bench.py starts the N ranks itself before it touches the GPU):
  --mode frames (default)  whole frames are sharded over ranks, no data-path collective;
                           per-GPU work is fixed as N grows -> "scaling": "weak".
  --mode bands             ONE frame per step is split into row bands, every rank renders its
                           band and the bands are gathered on rank 0 over RCCL ("strong").

One JSON line is printed by rank 0.  `roofline` prices the dominant kernel (the draw kernel)
against the FP32 vector-ALU peak with the floating-point work the kernel EXECUTES per
ray-sphere test (its separable filter: (1 + 1/P) FMA + one max input = 3.25 FLOP at 8 pixels
per lane), so `frac` <= 1 by construction; counter-derived figures (VALU busy, HBM bytes per
launch) come from profiles/counters.json, written by tools/profile.sh + tools/pmc_summary.py.
`cpu_baseline` times the CPU oracle (a port of the reference's per-pixel function) on this
box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cudaraytracing_amd import abi, configs, sharding  # noqa: E402
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3     # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
SURVEY_FLOP_PER_TEST = 7            # b' = v.d (3 mul + 2 add), disc' = b'b' - c (1 mul + 1 sub): the hoisted
                                    # form of kernel.cu:53-58; reported as an equivalent rate, never as `frac`
REF_FLOP_PER_TEST = 21              # kernel.cu:53-58 as written (what the CPU baseline executes)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3_3840x2160_n1024", choices=sorted(configs.CONFIGS))
    ap.add_argument("--mode", default="frames", choices=["frames", "bands"])
    ap.add_argument("--sequence", action="store_true", help="animate the camera along the 300-frame path")
    ap.add_argument("--subbands", type=int, default=2, help="bands mode: pieces per band, rendered and sent one after the other")
    ap.add_argument("--gather", default="float", choices=["float", "rgba8"],
                    help="bands mode: what is gathered on rank 0 -- the float RGBA bands (16 B/pixel) or the fused "
                         "RGBA8 bands (4 B/pixel, a quarter of the xGMI traffic)")
    ap.add_argument("--band-layout", default="contiguous", choices=["contiguous", "interleaved"],
                    help="bands mode: contiguous = rank r renders rows [r H/N, (r+1) H/N) in `subbands` pieces; interleaved = the "
                         "frame is cut into 2N bands and rank r renders band r and band N+r (an outer band with an "
                         "inner one: evens out the hit-heavy middle of the picture)")
    ap.add_argument("--gather-impl", default="torch", choices=["torch", "c"],
                    help="bands mode: torch = torch.distributed batch_isend_irecv on device tensors; c = librc_gather.so "
                         "(grouped ncclSend/ncclRecv issued from C on its own stream; include/rc_gather.h)")
    ap.add_argument("--variant", default="tiled", choices=["tiled", "tiled2", "tiled4", "tiled8", "culled", "literal"])
    ap.add_argument("--shading", default="ref", choices=["ref", "ext"],
                    help="ref: the reference's math; ext: + plane, point light, Phong, hard shadows (unpinned by the reference)")
    ap.add_argument("--readback", default="none", choices=["none", "rgba8", "float"],
                    help="also time K frames end to end through the double-buffered read-back pipeline "
                         "(reported as end_to_end, never as value)")
    ap.add_argument("--graph", action="store_true",
                    help="also capture the K steps into one HIP graph and time its replay (reported as graph_replay, never as value)")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the headline measurement: no culled_variant / display_conversion legs (for profiling runs)")
    ap.add_argument("--tuning", type=int, default=0, help="rc_opts.tuning bits (RC_TUNE_*): launcher overrides for A/B runs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-bands-check", action="store_true",
                    help="with more than one rank in frames mode: skip the untimed bands_check leg (one 7680x4320 x 4096 frame "
                         "through both row-band gathers, checked against the whole frame)")
    ap.add_argument("--bands-check-timeout", type=float, default=240.0,
                    help="seconds after which a stuck bands_check leg is abandoned: the headline line is printed without it")
    ap.add_argument("--no-clock-ramp", action="store_true",
                    help="do not extend the untimed warm-up until the GPU has reached its steady clock")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads of the cpu_baseline leg (0 = every hardware thread this process may run on)")
    return ap.parse_args()


def cpu_share():
    """(hardware threads this process may be scheduled on, CPU-time quota in cores or None)."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    return allowed, quota


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: start the N ranks as a
    child `python -m torch.distributed.run ... bench.py <same arguments>` and relay its one JSON line.
    Called before anything in this process has touched the GPU (a process that has must never be
    replaced, and nothing here is: the launcher is a child, this process only waits for it)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # RCCL shares device buffers between the ranks' processes through dmabuf handles; the legacy IPC mode this
    # switches off is not supported by the host driver here (hipIpcGetMemHandle fails with "invalid argument")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    return proc.returncode


def cpu_baseline(cfg, threads, shading):
    """Oracle (port of kernel.cu:21-112) on the host cores, bounded sample of the same frame.

    `threads` = 0: every hardware thread this process may run on.  Where the box hands a job a
    smaller CPU share than it has hardware threads (OMP_NUM_THREADS, a cgroup quota), the sample
    is timed at that share as well and the faster of the two is `value`; both are in the result."""
    from oracle import binding as ob     # the checker, used here only as the timed CPU baseline
    w, h, n = cfg["width"], cfg["height"], cfg["n"]
    pos, rad, col = ob.generate_scene(n, cfg["seed"])
    cam = ob.camera_data(cfg["position"], cfg["rotation"], w, h)
    ext = None
    if shading == "ext":
        ext = ob.Ext(**{k: (ob.Vec3(*v) if isinstance(v, tuple) else v) for k, v in configs.DEFAULT_EXT.items()})
    allowed, quota = cpu_share()
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or None
    if threads > 0:
        counts = [threads]
    else:
        counts = [min(allowed, 512)]
        if share and share < counts[0]:
            counts.append(share)

    def sample(t):
        # ~6.4 ns per test per core: pick rows so that one pass is ~20 core-seconds of work
        rows = int(min(h, max(t, 20.0 / (6.4e-9 * w * max(n, 1)))))
        rows = min(h, max(t, rows - rows % t))
        y0 = (h - rows) // 2
        ob.render(pos, rad, col, cam, w, min(rows, 4 * t), y_offset=y0, full_height=h, threads=t, ext=ext)  # warm
        reps, dt = 0, 0.0
        t0 = time.perf_counter()
        while dt < 2.0 and reps < 200:          # small frames: repeat until the sample is >= 2 s
            ob.render(pos, rad, col, cam, w, rows, y_offset=y0, full_height=h, threads=t, ext=ext)
            reps += 1
            dt = time.perf_counter() - t0
        dt /= reps
        return {"threads": t, "value": w * rows / dt / 1e6,
                "sample": f"{reps} x {rows} centre rows of one {w}x{h} frame, {n} spheres "
                          f"({w * rows * n / 1e9:.2f} G tests per pass), {dt:.3f} s per pass",
                "ns_per_test_per_thread": dt * t / (w * rows * max(n, 1)) * 1e9}

    runs = [sample(t) for t in counts]
    best = max(runs, key=lambda r: r["value"])
    return {
        "value": best["value"], "unit": "Mrays/s", "cores": best["threads"], "kind": "port", "sample": best["sample"],
        "ns_per_test_per_core": best["ns_per_test_per_thread"],
        "box_threads": allowed, "box_cpu_count": os.cpu_count(), "cpu_quota_cores": quota,
        "job_cpu_share": share, "runs": runs,
    }


def pipelined_readback(lib, scene, base_opts, w, h, steps, warmup, what):
    """K frames rendered AND read back to pinned host memory: two device surfaces, a render
    stream and a copy stream, the read-back of frame f overlapping the render of frame f+1
    (the replacement of the reference's sync + full-frame copy, RayTracingKernel.cpp:18-27)."""
    bpp = 4 if what == "rgba8" else 16
    render, copy = C.c_void_p(), C.c_void_p()
    abi.check(lib.rc_stream_create(C.byref(render)))
    abi.check(lib.rc_stream_create(C.byref(copy)))
    surf, dev8, host, ev_r, ev_c = [], [], [], [], []
    for _ in range(2):
        surf.append(DeviceBuffer(w * 16 * h))
        dev8.append(DeviceBuffer(w * 4 * h) if what == "rgba8" else None)
        hp = C.c_void_p()
        abi.check(lib.rc_host_alloc(C.byref(hp), w * bpp * h))
        host.append(hp)
        for lst in (ev_r, ev_c):
            e = C.c_void_p()
            abi.check(lib.rc_event_create(C.byref(e)))
            lst.append(e)
    opts = abi.RcOpts.from_buffer_copy(bytes(base_opts))
    opts.ext = base_opts.ext
    opts.stream = render

    def submit(i):
        b = i & 1
        if i >= 2:
            abi.check(lib.rc_stream_wait_event(render, ev_c[b]))
        screen = abi.RcScreen(surf[b].ptr.value, w, h, w * 16)
        if what == "rgba8":
            opts.rgba8, opts.rgba8_pitch, opts.rgba8_flip = dev8[b].ptr.value, w * 4, 1
        abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)), "rc_render")
        abi.check(lib.rc_event_record(ev_r[b], render))
        abi.check(lib.rc_stream_wait_event(copy, ev_r[b]))
        src = dev8[b].ptr if what == "rgba8" else surf[b].ptr
        abi.check(lib.rc_download_2d_async(host[b], w * bpp, src, w * bpp, w * bpp, h, copy), "download")
        abi.check(lib.rc_event_record(ev_c[b], copy))

    for i in range(max(warmup, 2)):
        submit(i)
    abi.check(lib.rc_sync(None))
    t0 = time.perf_counter()
    for i in range(steps):
        submit(i + 2)
    abi.check(lib.rc_sync(None))
    dt = time.perf_counter() - t0
    for b in range(2):
        surf[b].free()
        if dev8[b] is not None:
            dev8[b].free()
        lib.rc_host_free(host[b])
        lib.rc_event_destroy(ev_r[b])
        lib.rc_event_destroy(ev_c[b])
    lib.rc_stream_destroy(render)
    lib.rc_stream_destroy(copy)
    return {"readback": what, "value": w * h * steps / dt / 1e6, "unit": "Mrays/s", "ms_per_frame": dt / steps * 1e3,
            "pcie_gbs": w * h * bpp * steps / dt / 1e9, "bytes_per_frame": w * h * bpp}


def make_scene(cfg):
    """Scene arrays of a workload resident in HBM (SceneManager.cpp:17-24) -> (RcScene, buffers to keep alive)."""
    n = cfg["n"]
    pos, rad, col = generate_scene(n, cfg["seed"])
    bufs = []
    for a in (pos, rad, col):
        b = DeviceBuffer(max(a.nbytes, 4))
        b.upload(a)
        bufs.append(b)
    scene = abi.RcScene()
    scene.sphere_num = n
    scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
    cam = Camera(cfg["width"], cfg["height"])
    cam.position, cam.rotation = cfg["position"], cfg["rotation"]
    scene.cam = cam.get_data()
    return scene, bufs


def bands_check(lib, torch, dist, world, rank, comm_device, stream, reps=3):
    """With more than one rank, outside the timed headline: ONE frame of 7680x4320 x 4096 spheres cut into row
    bands (two pieces per rank), rendered by all ranks and assembled on rank 0 -- once through
    torch.distributed (sharding.BandGather) and once through librc_gather.so -- each checked bit for bit
    against the frame rank 0 renders whole, with its time and every rank's render time.  The first run on a
    multi-GPU node thereby exercises both gather paths whatever mode the headline used."""
    cfg = configs.CONFIGS["c4_7680x4320_n4096"]
    w, h = cfg["width"], cfg["height"]
    parts = 2
    scene, keep = make_scene(cfg)
    g = sharding.BandGather(h, world, rank, parts=parts, layout="contiguous")
    band = torch.empty((g.rows, w, 4), dtype=torch.float32, device="cuda")
    full = torch.empty((h, w, 4), dtype=torch.float32, device="cuda") if rank == 0 else None
    base = abi.RcOpts(full_height=h, stream=stream)

    def render_piece(k):
        sb, se = g.pieces[k]
        if se <= sb:
            return
        piece = abi.RcScreen(band[g.offsets[k]:].data_ptr(), w, se - sb, w * 16)
        o = abi.RcOpts.from_buffer_copy(bytes(base))
        o.y_offset = sb
        abi.check(lib.rc_render(C.byref(piece), C.byref(scene), C.byref(o)), "rc_render (bands_check)")

    def fence():
        dist.barrier()
        torch.cuda.synchronize()

    out = {"workload": "c4_7680x4320_n4096", "pieces_per_rank": parts, "band_layout": "contiguous",
           "rows_of_this_run": [list(sharding.rank_pieces(h, world, r, parts)) for r in range(world)]}
    # every rank's render alone (no exchange)
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    abi.check(lib.rc_event_create(C.byref(ev0)))
    abi.check(lib.rc_event_create(C.byref(ev1)))
    for k in range(parts):
        render_piece(k)
    fence()
    abi.check(lib.rc_event_record(ev0, stream))
    for _ in range(reps):
        for k in range(parts):
            render_piece(k)
    abi.check(lib.rc_event_record(ev1, stream))
    fence()
    ms = C.c_float()
    abi.check(lib.rc_event_elapsed_ms(ev0, ev1, C.byref(ms)))
    out["per_rank_render_ms"] = sharding.gather_floats(ms.value / reps, comm_device)

    whole = None
    if rank == 0:
        whole = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        ws = abi.RcScreen(whole.data_ptr(), w, h, w * 16)
        abi.check(lib.rc_render(C.byref(ws), C.byref(scene), C.byref(base)), "rc_render (whole)")
        torch.cuda.synchronize()

    def torch_frame():
        if comm_device == "cuda":
            g.post_receives(full)
            for k in range(parts):
                render_piece(k)
                g.piece_rendered(k, band, full)
            g.wait()
        else:                                   # rehearsal backend: the same exchange staged through host memory
            for k in range(parts):
                render_piece(k)
            torch.cuda.current_stream().synchronize()
            host_band = band.cpu()
            host_full = torch.empty(full.shape, dtype=full.dtype) if rank == 0 else None
            g.post_receives(host_full)
            for k in range(parts):
                g.piece_rendered(k, host_band, host_full)
            g.wait()
            if rank == 0:
                full.copy_(host_full)

    def run(frame_fn):
        if rank == 0:
            full.zero_()
        frame_fn()                               # warm-up, and the frame that is checked
        fence()
        ok = bool(torch.equal(full, whole)) if rank == 0 else None
        t0 = time.perf_counter()
        for _ in range(reps):
            frame_fn()
        fence()
        dt = sharding.max_over_ranks(time.perf_counter() - t0, comm_device) / reps
        return {"frame_ms": dt * 1e3, "gather_check": ok, "value": w * h / dt / 1e6, "unit": "Mrays/s"}

    out["torch"] = run(torch_frame)
    if comm_device == "cuda":
        from cudaraytracing_amd import gather as rcg
        ident = [rcg.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        cg = rcg.Gather(world, rank, ident[0])
        full_ptr = full.data_ptr() if full is not None else None

        def c_frame():
            cg.begin(parts, full_ptr, w * 16, h, stream)
            for k in range(parts):
                render_piece(k)
                cg.piece(k, parts, band.data_ptr(), full_ptr, w * 16, h, stream)
            cg.wait(stream)

        out["c"] = run(c_frame)
        cg.destroy()
    else:
        out["c"] = {"skipped": "librc_gather.so needs RCCL; this run used the %s rehearsal backend" % comm_device}
    lib.rc_event_destroy(ev0)
    lib.rc_event_destroy(ev1)
    for b in keep:
        b.free()
    return out


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))       # nothing in this process has touched the GPU yet
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # stdout carries exactly one line, the JSON result: anything libraries print there (RCCL
    # writes a version banner to stdout when it initialises) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    if os.environ.get("RC_BENCH_DRY_RUN") == "1":
        # Launch rehearsal for a machine without a GPU (tests only; renders nothing, measures nothing):
        # rendezvous, barrier, the MAX reduction of the timed region, one line from rank 0.
        dist.init_process_group("gloo")
        dist.barrier()
        worst = sharding.max_over_ranks(float(rank), "cpu")
        if rank == 0:
            os.write(result_fd, (json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                                             "max_rank": worst, "value": None}) + "\n").encode())
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal switches for a box with fewer GPUs than ranks (never set by the driver):
    # RC_BENCH_SHARE_GPU=1 puts every rank on device 0, RC_BENCH_BACKEND=gloo replaces RCCL.
    backend = os.environ.get("RC_BENCH_BACKEND", "nccl")
    if os.environ.get("RC_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    lib = abi.load()
    abi.check(lib.rc_set_device(local_rank), "rc_set_device")
    # a process group also for one rank when launched through torch.distributed.run, and always in
    # bands mode (exercises the RCCL set-up, barrier, MAX reduction and the band assembly on
    # device tensors on a one-GPU box)
    from_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    use_dist = world > 1 or from_launcher or args.mode == "bands"
    if use_dist:
        kw = {}
        if not from_launcher:                    # one rank, started by hand: rendezvous with itself
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                kw = dict(init_method=f"tcp://127.0.0.1:{sk.getsockname()[1]}", rank=0, world_size=1)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(backend, **kw)
    comm_device = "cuda" if backend == "nccl" else "cpu"

    cfg = configs.CONFIGS[args.config]
    w, h, n = cfg["width"], cfg["height"], cfg["n"]
    variant = {"tiled": abi.VARIANT_TILED, "tiled2": abi.VARIANT_TILED2, "tiled4": abi.VARIANT_TILED4, "tiled8": abi.VARIANT_TILED8, "culled": abi.VARIANT_CULLED, "literal": abi.VARIANT_LITERAL}[args.variant]

    # scene resident in HBM before the timed region (SceneManager.cpp:17-24)
    pos, rad, col = generate_scene(n, cfg["seed"])
    bufs = []
    for a in (pos, rad, col):
        b = DeviceBuffer(max(a.nbytes, 4))
        b.upload(a)
        bufs.append(b)
    scene = abi.RcScene()
    scene.sphere_num = n
    scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
    cam = Camera(w, h)
    cam.position, cam.rotation = cfg["position"], cfg["rotation"]
    scene.cam = cam.get_data()

    banded = args.mode == "bands"      # also with one rank: the rows are still rendered piece by piece
                                       # with y_offset and assembled into `full` by the same code
    gatherer = sharding.BandGather(h, world, rank, parts=args.subbands, layout=args.band_layout) if banded else None
    if banded and args.gather_impl == "c" and args.band_layout != "contiguous":
        raise SystemExit("--gather-impl c assembles contiguous bands only")
    y0 = gatherer.pieces[0][0] if banded else 0
    rows = gatherer.rows if banded else h
    # the frame lives in a torch tensor so that the band gather can use torch.distributed; in bands mode
    # it holds this rank's pieces one after the other
    frame = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
    screen = abi.RcScreen(frame.data_ptr(), w, rows, w * 16)
    stream = torch.cuda.current_stream().cuda_stream
    opts = abi.RcOpts(y_offset=y0, full_height=h, variant=variant, stream=stream, tuning=args.tuning)
    rc_ext = None
    if args.shading == "ext":
        rc_ext = abi.RcExt(**{k: (abi.RcVec3(*v) if isinstance(v, tuple) else v) for k, v in configs.DEFAULT_EXT.items()})
        opts.mode = abi.MODE_EXT
        opts.ext = C.pointer(rc_ext)
    full = None
    gather8 = banded and args.gather == "rgba8"
    frame8 = torch.empty((rows, w), dtype=torch.int32, device="cuda") if gather8 else None
    if gather8:           # the draw kernel writes the display surface too; that one travels
        opts.rgba8, opts.rgba8_pitch, opts.rgba8_flip = frame8.data_ptr(), w * 4, 0
    if banded and rank == 0:
        full = (torch.empty((h, w), dtype=torch.int32, device="cuda") if gather8
                else torch.empty((h, w, 4), dtype=torch.float32, device="cuda"))
    sent = frame8 if gather8 else frame          # what the band gather moves

    cgather = None
    if banded and args.gather_impl == "c":
        if comm_device != "cuda":
            raise SystemExit("--gather-impl c needs RCCL (it cannot run on the gloo rehearsal backend)")
        from cudaraytracing_amd import gather as rcg
        ident = [rcg.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)          # carries the 128-byte identifier to every rank
        cgather = rcg.Gather(world, rank, ident[0])

    def render_piece(k):
        """Piece k of this rank's rows into its place in `frame` (rc_opts.y_offset = its first frame row)."""
        sb, se = gatherer.pieces[k]
        if se <= sb:
            return
        o = gatherer.offsets[k]
        piece = abi.RcScreen(frame[o:].data_ptr(), w, se - sb, w * 16)
        popts = abi.RcOpts.from_buffer_copy(bytes(opts))
        popts.ext = opts.ext
        popts.y_offset = sb
        if gather8:
            popts.rgba8 = frame8[o:].data_ptr()
        abi.check(lib.rc_render(C.byref(piece), C.byref(scene), C.byref(popts)), "rc_render")

    def step(i):
        if args.sequence:
            f = (i * (world if args.mode == "frames" else 1) + (rank if args.mode == "frames" else 0)) % configs.SEQUENCE_FRAMES
            cam.position, cam.rotation = configs.sequence_camera(f)
            scene.cam = cam.get_data()                     # SceneManager::UpdateCamera
            if rc_ext is not None:
                rc_ext.light_pos = abi.RcVec3(*configs.sequence_light(f))
        if gatherer is None:
            abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)), "rc_render")
        elif cgather is not None:
            # the pieces handed to librc_gather.so: each transfer runs on the library's stream behind an
            # event on ours, so piece k travels while piece k + 1 renders.  Every rank makes the same calls;
            # a band shorter than `subbands` rows has empty pieces.
            row_bytes = w * (4 if gather8 else 16)
            cgather.begin(args.subbands, full.data_ptr() if full is not None else None, row_bytes, h, stream)
            for k in range(args.subbands):
                if k < len(gatherer.pieces):
                    render_piece(k)
                cgather.piece(k, args.subbands, sent.data_ptr(), full.data_ptr() if full is not None else None,
                              row_bytes, h, stream)
            cgather.wait(stream)
        elif comm_device == "cuda":
            # render piece by piece; each piece is sent while the next one renders
            gatherer.post_receives(full)
            for k in range(len(gatherer.pieces)):
                render_piece(k)
                gatherer.piece_rendered(k, sent, full)
            gatherer.wait()
        else:                                   # rehearsal backend: the same exchange staged through host memory
            for k in range(len(gatherer.pieces)):
                render_piece(k)
            torch.cuda.current_stream().synchronize()
            host_band = sent.cpu()
            host_full = torch.empty(full.shape, dtype=full.dtype) if rank == 0 else None
            gatherer.post_receives(host_full)
            for k in range(len(gatherer.pieces)):
                gatherer.piece_rendered(k, host_band, host_full)
            gatherer.wait()
            if rank == 0:
                full.copy_(host_full)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    # Clock ramp: an idle MI355X needs ~30 ms of continuous work to reach its steady clock
    # (tools/ramp_probe.py: the first 80 frames of this workload run up to 35 % slower), so the
    # untimed warm-up is extended until the GPU has been busy for ~0.1 s.  The count is agreed
    # over ranks (bands mode needs every rank to run the same number of steps).
    ramp_steps = 0
    if not args.no_clock_ramp:
        t_p = time.perf_counter()
        for i in range(8):
            step(args.warmup + i)
        torch.cuda.current_stream().synchronize()
        est = max((time.perf_counter() - t_p) / 8, 1e-6)
        ramp_steps = 8 + int(sharding.max_over_ranks(float(min(4096, int(0.1 / est) + 1)), comm_device))
        for i in range(8, ramp_steps):
            step(args.warmup + i)
        fence()
    first = args.warmup + ramp_steps
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    abi.check(lib.rc_event_create(C.byref(ev0)))
    abi.check(lib.rc_event_create(C.byref(ev1)))
    t0 = time.perf_counter()
    abi.check(lib.rc_event_record(ev0, stream))
    for i in range(args.steps):
        step(first + i)
    abi.check(lib.rc_event_record(ev1, stream))
    fence()
    dt = time.perf_counter() - t0
    ev_ms = C.c_float()
    abi.check(lib.rc_event_elapsed_ms(ev0, ev1, C.byref(ev_ms)))

    dt_max = sharding.max_over_ranks(dt, comm_device)
    # per-GPU time of the K steps on each rank's own stream (load balance of bands / frames)
    per_rank_ms = sharding.gather_floats(ev_ms.value / args.steps, comm_device) if use_dist else None

    # Per-frame distribution (median, p10, p90): 100 more frames, each between
    # its own pair of HIP events, outside the timed region.
    frame_stats = None
    if rank == 0 and args.mode == "frames":
        evs = []
        for i in range(101):
            e = C.c_void_p()
            abi.check(lib.rc_event_create(C.byref(e)))
            evs.append(e)
        abi.check(lib.rc_event_record(evs[0], stream))
        for i in range(100):
            step(first + args.steps + i)
            abi.check(lib.rc_event_record(evs[i + 1], stream))
        torch.cuda.synchronize()
        per = []
        for i in range(100):
            ms = C.c_float()
            abi.check(lib.rc_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms)))
            per.append(ms.value)
        for e in evs:
            lib.rc_event_destroy(e)
        per.sort()
        frame_stats = {"frames": 100, "median_ms": per[50], "p10_ms": per[10], "p90_ms": per[90], "min_ms": per[0], "max_ms": per[-1]}
    if use_dist:
        dist.barrier()

    # Bands mode self-check (outside the timed region): one more gathered frame on rank 0 against the
    # same frame rendered whole by rank 0 alone, bit for bit.
    gather_check = None
    if banded:
        step(first + args.steps)
        fence()
        if rank == 0:
            whole = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
            wopts = abi.RcOpts.from_buffer_copy(bytes(opts))
            wopts.ext = opts.ext
            wopts.y_offset, wopts.full_height = 0, h
            whole8 = torch.empty((h, w), dtype=torch.int32, device="cuda") if gather8 else None
            if gather8:
                wopts.rgba8 = whole8.data_ptr()
            wscreen = abi.RcScreen(whole.data_ptr(), w, h, w * 16)
            abi.check(lib.rc_render(C.byref(wscreen), C.byref(scene), C.byref(wopts)), "rc_render (check)")
            torch.cuda.synchronize()
            gather_check = bool(torch.equal(full, whole8 if gather8 else whole))
        fence()

    # Secondary measurement (N=1, default variant only): the opt-in tile-culling variant on the same
    # workload.  It is NOT `value`: it no longer performs one test per (pixel, sphere) pair.
    culled = None
    if world == 1 and args.variant == "tiled" and args.mode == "frames" and not args.no_extras:
        copts = abi.RcOpts.from_buffer_copy(bytes(opts))
        copts.ext = opts.ext
        copts.variant = abi.VARIANT_CULLED
        # records built once per frame in a caller workspace (16 bytes per sphere); below ~1000
        # spheres the extra launch costs more than every workgroup building its own
        use_ws = n >= 1024
        culled_counters = None
        try:
            with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
                culled_counters = json.load(f).get("%s|%s|culled" % (args.config, args.shading))
        except (OSError, ValueError):
            pass
        if use_ws:       # records + per-bin sphere lists
            wsp = torch.empty((lib.rc_workspace_bytes_frame(n, w, rows) + 3) // 4, dtype=torch.float32, device="cuda")
            copts.workspace, copts.workspace_bytes = wsp.data_ptr(), wsp.numel() * 4
        for i in range(max(args.warmup, 3)):
            abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(copts)), "rc_render")
        fence()
        abi.check(lib.rc_event_record(ev0, stream))
        for i in range(args.steps):
            abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(copts)), "rc_render")
        abi.check(lib.rc_event_record(ev1, stream))
        fence()
        cms = C.c_float()
        abi.check(lib.rc_event_elapsed_ms(ev0, ev1, C.byref(cms)))
        cms = cms.value / args.steps
        culled = {"variant": "culled (RC_VARIANT_CULLED, opt-in): interval tests per workgroup tile, then per wave tile, in place of the per-pixel loop; "
                             + ("records and per-bin sphere lists built once per frame in rc_opts.workspace; " if use_ws else "") +
                             "bit-identical surfaces, not one test per (pixel, sphere) pair",
                  "value": w * h / (cms * 1e-3) / 1e6, "unit": "Mrays/s", "frame_ms": cms,
                  # no roofline block: the variant is instruction-bound (VALU busy in profiles/counters.json under
                  # "<config>|<shading>|culled"), not HBM-bound, and it does not run one test per (pixel, sphere) pair,
                  # so neither of the line's rooflines prices it; the write rate is given for scale only
                  "hbm_write_gbs": 16.0 * w * h / (cms * 1e-3) / 1e9,
                  "valu_busy": (culled_counters or {}).get("valu_busy"), "counters_source": (culled_counters or {}).get("source")}

    # Secondary measurement (N=1): the display step after the path, float RGBA -> R8G8B8A8
    # (DXManager.cpp:134,299,328), standalone (rc_to_rgba8: 16 B read + 4 B written per pixel,
    # HBM-bound) and fused into the draw kernel as a second surface (+4 B written per pixel).
    display = None
    if world == 1 and args.variant == "tiled" and args.mode == "frames" and not args.sequence and not args.no_extras:
        out8 = torch.empty((rows, w), dtype=torch.int32, device="cuda")
        def timed(fn):
            for i in range(max(args.warmup, 3)):
                fn()
            fence()
            abi.check(lib.rc_event_record(ev0, stream))
            for i in range(args.steps):
                fn()
            abi.check(lib.rc_event_record(ev1, stream))
            fence()
            ms = C.c_float()
            abi.check(lib.rc_event_elapsed_ms(ev0, ev1, C.byref(ms)))
            return ms.value / args.steps
        conv_ms = timed(lambda: abi.check(lib.rc_to_rgba8(C.byref(screen), out8.data_ptr(), w * 4, 1, stream), "rc_to_rgba8"))
        fopts = abi.RcOpts.from_buffer_copy(bytes(opts))
        fopts.ext = opts.ext
        fopts.rgba8, fopts.rgba8_pitch, fopts.rgba8_flip = out8.data_ptr(), w * 4, 1
        fused_ms = timed(lambda: abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(fopts)), "rc_render"))
        gbs = 20.0 * w * rows / (conv_ms * 1e-3) / 1e9
        display = {"kernel": "rc_rgba8_kernel", "standalone_ms": conv_ms, "bytes_per_pixel": 20,
                   "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
                   "note": ("back-to-back launches on one frame: the %dx%d float surface (%.0f MB) %s the 256 MiB Infinity Cache, "
                            "so the achieved rate is %s" % (
                                w, rows, 16.0 * w * rows / 1e6,
                                "fits" if 20 * w * rows <= 256 * 2 ** 20 else "does not fit",
                                "a cache-assisted figure, not raw HBM" if 20 * w * rows <= 256 * 2 ** 20 else "an HBM figure")),
                   "fused_frame_ms": fused_ms, "fused_extra_ms": fused_ms - ev_ms.value / args.steps}

    # Secondary measurement (--graph, N=1): the same K steps captured into ONE HIP graph and replayed
    # with a single submission -- what a launch-bound sequence of small frames gains.
    graph_replay = None
    if args.graph and world == 1 and args.mode == "frames":
        gstream = C.c_void_p()
        abi.check(lib.rc_stream_create(C.byref(gstream)))
        gopts = abi.RcOpts.from_buffer_copy(bytes(opts))
        gopts.ext = opts.ext
        gopts.stream = gstream
        abi.check(lib.rc_graph_begin(gstream), "rc_graph_begin")
        for i in range(args.steps):
            if args.sequence:
                cam.position, cam.rotation = configs.sequence_camera(i % configs.SEQUENCE_FRAMES)
                scene.cam = cam.get_data()
            abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(gopts)), "rc_render (capture)")
        gexec = C.c_void_p()
        abi.check(lib.rc_graph_end(gstream, C.byref(gexec)), "rc_graph_end")
        g0, g1 = C.c_void_p(), C.c_void_p()
        abi.check(lib.rc_event_create(C.byref(g0)))
        abi.check(lib.rc_event_create(C.byref(g1)))
        reps = 5
        for i in range(2):
            abi.check(lib.rc_graph_launch(gexec, gstream))
        abi.check(lib.rc_sync(gstream))
        t_g = time.perf_counter()
        abi.check(lib.rc_event_record(g0, gstream))
        for i in range(reps):
            abi.check(lib.rc_graph_launch(gexec, gstream))
        abi.check(lib.rc_event_record(g1, gstream))
        abi.check(lib.rc_sync(gstream))
        t_g = time.perf_counter() - t_g
        gms = C.c_float()
        abi.check(lib.rc_event_elapsed_ms(g0, g1, C.byref(gms)))
        graph_replay = {"frames_per_graph": args.steps, "replays": reps,
                        "frame_ms": gms.value / (reps * args.steps), "wall_ms_per_frame": t_g * 1e3 / (reps * args.steps),
                        "value": w * h * reps * args.steps / t_g / 1e6, "unit": "Mrays/s"}
        lib.rc_graph_destroy(gexec)
        lib.rc_event_destroy(g0)
        lib.rc_event_destroy(g1)
        lib.rc_stream_destroy(gstream)

    end_to_end = None
    if args.readback != "none" and args.mode == "frames":
        end_to_end = pipelined_readback(lib, scene, opts, w, h, args.steps, args.warmup, args.readback)

    frames_total = args.steps * (world if args.mode == "frames" else 1)
    value = w * h * frames_total / dt_max / 1e6
    kernel_ms = ev_ms.value / args.steps           # HIP events on the launch stream, this rank
    tests = w * rows * n
    shadow_tests = 0                 # algorithmic: one shadow ray per hit pixel, tested against every sphere
    shadow_tests_executed = None     # what the shadow loop actually ran (instrumented build, tools/count_tests.py)
    executed_source = None
    if args.shading == "ext":
        hit_pixels = int((frame[..., :3] != 0).any(dim=-1).sum().item())   # counted on the rendered frame
        shadow_tests = hit_pixels * n
        try:
            with open(os.path.join(ROOT, "profiles", "executed_tests.json")) as f:
                ex = json.load(f).get("%s|ext" % args.config)
            if ex and args.variant == "tiled" and rows == h and not args.sequence and args.tuning == 0:
                shadow_tests_executed = int(ex["shadow_tests_executed"])
                executed_source = "profiles/executed_tests.json"
        except (OSError, ValueError, KeyError):
            pass
    # Floating-point work the kernel EXECUTES per test.  Tiled kernel, primary rays: the separable
    # filter T = fma(a_pixel, beta, r), r = fma(b_row, gamma, alpha) shared by the lane's P pixels,
    # then one input of the running maximum: (1 + 1/P) FMA (2 FLOP each) + 1.  Shadow rays (EXT,
    # brute-force pass): h.e as mul + 2 FMA (2 FMA in the dominant-axis form), then one max input: 6 (5) FLOP, counted for hit pixels only
    # (lanes of a wave with a hit that hold no hit pixel run the loop too and are not counted).
    # The exact re-tests of flagged pairs, ray set-up, shading and stores are not counted either.
    # Literal kernel: kernel.cu:53-58 as written, 21 FLOP to the discriminant.
    cu_count = C.c_int()
    abi.check(lib.rc_device_info(local_rank, C.byref(cu_count), None, None))
    # (bands mode launches one kernel per piece: the launcher sees a piece's rows and the frame's height)
    launch_rows = (gatherer.pieces[0][1] - gatherer.pieces[0][0]) if banded else rows
    ppl = configs.pixels_per_lane(w, launch_rows, n, args.shading == "ext", args.variant == "culled", cu_count.value,
                                  full_height=h)
    ppl = {"tiled2": 2, "tiled4": 4, "tiled8": 8 if args.shading == "ref" else 4}.get(args.variant, ppl)
    axis = False
    if args.variant == "literal":
        executed_flop, executed_shadow_flop = float(REF_FLOP_PER_TEST), float(REF_FLOP_PER_TEST)
    else:
        # shadow filter: the dominant-axis form (2 FMA + max input) from 1024 spheres or when forced, else mul + 2 FMA + max
        axis = (args.tuning & abi.TUNE_SHADOW_AXIS_ON) or (n >= 1024 and not (args.tuning & abi.TUNE_SHADOW_AXIS_OFF))
        executed_flop, executed_shadow_flop = 2.0 * (1.0 + 1.0 / ppl) + 1.0, (5.0 if axis else 6.0)
    secs = kernel_ms * 1e-3
    # EXT: the shadow loop is priced with the tests it EXECUTED where the instrumented build has counted them for
    # this workload (whole waves of every block run: idle lanes and padding included, early exits excluded);
    # otherwise with the algorithmic count, which overstates it when the any-hit exit leaves the loop early
    priced_shadow_tests = shadow_tests_executed if shadow_tests_executed is not None else shadow_tests
    executed_tflops = (executed_flop * tests + executed_shadow_flop * priced_shadow_tests) / secs / 1e12
    # counter-derived figures of this exact workload / shading / variant, if tools/profile.sh has been
    # run for it (separate rocprofv3 --pmc passes; summary committed under profiles/)
    counters = None
    try:
        with open(os.path.join(ROOT, "profiles", "counters.json")) as f:
            counters = json.load(f).get("%s|%s|%s" % (args.config, args.shading, args.variant))
    except (OSError, ValueError):
        pass
    if counters is not None and (rows != h or args.sequence):
        counters = None                                         # measured on the whole static frame only
    traffic = (counters["hbm_write_bytes"] + counters["hbm_read_bytes"]) if counters else None

    out = None
    if rank == 0:
        cu, khz = C.c_int(), C.c_int()
        abi.check(lib.rc_device_info(local_rank, C.byref(cu), C.byref(khz), None))
        name = C.create_string_buffer(256)
        abi.check(lib.rc_device_name(local_rank, name, 256))
        out = {
            "metric": "primary Mrays/sec at %dx%d, %d spheres" % (w, h, n),
            "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "warmup_clock_ramp_steps": ramp_steps,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak" if args.mode == "frames" else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.config, "width": w, "height": h, "spheres": n,
                       "mode": "REF" if args.shading == "ref" else "EXT (plane + point light + Phong + hard shadows; unpinned by the reference)",
                       "variant": args.variant, "tuning": args.tuning, "sharding": args.mode, "gathered": (args.gather if banded else None), "gather_impl": (args.gather_impl if banded else None),
                       "band_layout": (args.band_layout if banded else None), "backend": backend if world > 1 else None,
                       "camera": "sequence" if args.sequence else "static",
                       "device": name.value.decode(), "compute_units": cu.value, "clock_mhz": khz.value / 1e3},
            "frame_ms": kernel_ms, "frame_ms_stats": frame_stats, "per_rank_step_ms": per_rank_ms, "gather_check": gather_check,
            "roofline": {
                # The contract knows two rooflines, hbm | mfma.  This kernel is compute-bound, so it goes in the
                # compute ("mfma") slot, although it issues NO MFMA: what bounds it is the FP32 vector ALU
                # (VALU) issue rate.  `peak` is the FP32 vector peak (= the dense f32 matrix peak, 157.3 TFLOP/s).
                "bound": "mfma", "bound_detail": "compute-bound: the contract's compute slot is named 'mfma', but this kernel "
                                                 "issues NO MFMA instruction; it is bound by FP32 vector-ALU (VALU) issue, and "
                                                 "`peak` is the f32 vector peak (= the dense f32 matrix peak, 157.3 TFLOP/s at 2.4 GHz)",
                "achieved": executed_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": executed_tflops / FP32_VECTOR_PEAK_TFLOPS,
                "frac_basis": "FLOP the kernel executes in its inner loops (executed_flop_per_test x tests_per_launch"
                              " [+ executed_flop_per_shadow_test x shadow_tests_per_launch]) / kernel_ms / peak",
                "traffic": traffic, "algorithmic_bytes": 16 * w * rows,
                "traffic_note": "HBM bytes per launch (TCC_EA0_WRREQ x 64 B + TCC_EA0_RDREQ x 64 B x 2) from separate "
                                "rocprofv3 --pmc passes of this workload, profiles/counters.json; algorithmic bytes = 16*W*H",
                "valu_busy": counters.get("valu_busy") if counters else None,
                "lds_bank_conflict_cycles": counters.get("lds_bank_conflict") if counters else None,
                "effective_clock_ghz": counters.get("effective_clock_ghz") if counters else None,
                "counters_source": counters.get("source") if counters else None,
                "pixels_per_lane": None if args.variant == "literal" else ppl,
                "executed_flop_per_test": executed_flop, "executed_flop_per_shadow_test": executed_shadow_flop if shadow_tests else None,
                # template arguments: pixels per lane, EXT, CULL, BINS, SHFAST (dominant-axis shadow filter)
                "kernel": ("rc_draw_tiled<%d, %s, false, false, %s>" % (ppl, "true" if args.shading == "ext" else "false",
                                                                        "true" if (args.shading == "ext" and axis) else "false"))
                          if args.variant.startswith("tiled") else args.variant,
                "kernel_ms": kernel_ms, "tests_per_launch": tests, "shadow_tests_per_launch": shadow_tests,
                "shadow_tests_executed_per_launch": shadow_tests_executed, "shadow_tests_executed_source": executed_source,
                "shadow_tests_note": (("frac prices the shadow loop with the tests it executed (counted by the instrumented build)"
                                       if shadow_tests_executed is not None else
                                       "frac prices the shadow loop with hit pixels x spheres, the algorithmic count: an upper bound "
                                       "of the executed work where the any-hit exit leaves the loop early (no executed count on file "
                                       "for this workload)") if shadow_tests else None),
                "tests_per_second": tests / secs,
                # equivalent rates for comparisons, NOT roofline fractions: the same tests priced at the hoisted
                # 7-FLOP form of kernel.cu:53-58 and at the 21 FLOP the reference (and the CPU baseline) execute
                "survey_equiv_tflops_7flop": (SURVEY_FLOP_PER_TEST * tests + 16 * shadow_tests) / secs / 1e12,
                "ref_equiv_tflops_21flop": REF_FLOP_PER_TEST * (tests + shadow_tests) / secs / 1e12,
                "hbm_write_gbs": 16.0 * w * rows / secs / 1e9, "hbm_peak_gbs": HBM_PEAK_GBS,
            },
        }
        if culled is not None:
            out["culled_variant"] = culled
        if display is not None:
            out["display_conversion"] = display
        if graph_replay is not None:
            out["graph_replay"] = graph_replay
        if end_to_end is not None:
            out["end_to_end"] = end_to_end
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_threads, args.shading)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None

    def emit():
        if rank == 0:
            sys.stdout.flush()
            os.write(result_fd, (json.dumps(out) + "\n").encode())

    # More than one rank, default (frames) mode: the headline above exchanged nothing on the data path, so both
    # row-band gathers are exercised here, untimed by the headline.  A watchdog abandons the leg if a rank is stuck
    # in it (the line is then printed without it and every rank leaves at once).
    if world > 1 and args.mode == "frames" and not args.no_bands_check and not args.no_extras:
        import threading

        def abandon():
            if rank == 0:
                out["bands_check"] = {"error": "abandoned after %.0f s" % args.bands_check_timeout}
                emit()
            os._exit(0)

        dog = threading.Timer(args.bands_check_timeout, abandon)
        dog.daemon = True
        dog.start()
        try:
            res = bands_check(lib, torch, dist, world, rank, comm_device, stream)
        except Exception as e:                   # noqa: BLE001  (reported in the line; the headline stands)
            res = {"error": repr(e)}
        dog.cancel()
        if rank == 0:
            out["bands_check"] = res
    emit()

    if world > 1:
        # The line is out; a rank stuck in communicator teardown (a peer that died in the leg above) must not keep
        # the launcher waiting.
        import threading
        late = threading.Timer(60.0, lambda: os._exit(0))
        late.daemon = True
        late.start()
    if cgather is not None:
        cgather.destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
