#!/usr/bin/env python3
"""bench.py -- FFB6D hot-path benchmark on MI355X (driver contract: see the task statement).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            (no launcher: re-executes itself under torch.distributed.run, N ranks)
    python bench.py --config 4 | --config 5  (BASELINE configs: N=24576 fp32 | bs=16 bf16 mixed precision)

A *step* is one pass of the hot path over one batch of synthetic RGB-D frames that is already
resident in HBM: the on-device index pyramid (the 22 exact-KNN searches per frame that the
reference runs on the CPU in its DataLoader, linemod_dataset.py:318-353) followed by
FFB6D.forward (ffb6d.py:203-337) in fp32, eval mode -- by default the forward builds the pyramid itself (one batch of
searches) on a third HIP stream under the colour stem (--overlap-pyramid 0: pyramid first, then the forward).
Workload = BASELINE.json configs[1]:
bs=8, N=12288 points, 480x640, 4 encoder + 3 decoder fusion layers, one MI355X.  With N>1
ranks every rank runs its own batch of 8 (weak scaling, no data-path collective: batch items
are independent in the forward pass -- SURVEY.md section 8e); value = total frames / max-over-ranks time.

The JSON line also carries
  roofline      achieved algorithmic TFLOP/s (MFMA-bound launches) or GB/s (HBM-bound ones) of the dominant
                hand-written kernel, measured live with HIP events on its launch stream during the timed steps
                (which kernel that is, and hot_path_ops for all of them, come from untimed fully bracketed steps
                between warm-up and the timed region);
  cpu_baseline  the CPU oracle path (reference nanoflann from oracle/_ref when present, else the
                C restatement, + the plain-torch forward of oracle/forward_ref.py) timed on this
                host's cores on a bounded sample (a few single frames).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TFLOPS = 157.3     # fp32 peak, vector FMA = f32-input MFMA (MI355X_MICROARCH.md)
BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline figure is 2:1 sparse)
METRIC = "RGB-D frames/sec fwd (480x640, N=12288, bs=8)"      # BASELINE.json; --config 4 reports N=24576 in its line


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default: 8)")
    ap.add_argument("--n-points", type=int, default=None, help="points per frame (default: by --config)")
    ap.add_argument("--n-classes", type=int, default=22)
    ap.add_argument("--index-dtype", choices=["int64", "int32"], default="int64")
    ap.add_argument("--roofline-op", default="auto", help="op whose launches are event-timed in the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=5)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--trace-all", action="store_true", help="extra untimed pass: per-op table to stderr")
    ap.add_argument("--cudnn-benchmark", type=int, default=1,
                    help="torch.backends.cudnn.benchmark (MIOpen find mode for the colour branch's dense convs)")
    ap.add_argument("--miopen-db", choices=["pinned", "fresh"], default="pinned",
                    help="pinned (default): MIOpen reads the committed find-db of ONE search (ffb6d_amd/miopen_pin.py) instead of timing "
                         "its near-tied solvers again on every machine -- the same convolution kernels on every run; fresh: MIOpen's own "
                         "search on an empty user database (what rounds 1-4 measured: the step moves by +-0.4-0.8 ms with the draw)")
    ap.add_argument("--mode", choices=["infer", "train", "e2e"], default="infer",
                    help="e2e: sensor -> pose as one pipeline (ffb6d_amd/pipeline.py: input assembly from depth + rgb, forward with the "
                         "pyramid inside, pose solver), serial and overlapped schedules in one line.  "
                         "infer (default, the BASELINE metric): pyramid + forward, eval, no_grad.  train: BASELINE "
                         "config 3 shape -- pyramid + forward + backward + Adam step in train() mode, wrapped in "
                         "DistributedDataParallel (RCCL gradient all-reduce) when launched with more than one rank")
    ap.add_argument("--objective", choices=["reference", "proxy"], default="reference",
                    help="train mode: reference = the reference's objective (2 * focal(gamma=2) on the segmentation + masked L1 on the "
                         "keypoint and centre offsets, train_lm.py:245-259, ffb6d_amd/loss.py) on seeded synthetic labels / target "
                         "offsets; proxy = mean of squares of the three outputs (what the round-3 records were taken with)")
    ap.add_argument("--local-bn", action="store_true",
                    help="train mode, >1 rank: per-rank BatchNorm statistics, gradients are the only collective (north_star's "
                         "wording); default = torch.nn.SyncBatchNorm like the reference's apex SyncBN (train_lm.py:592)")
    ap.add_argument("--bn-variants", type=int, default=1,
                    help="train mode, >1 rank, SyncBatchNorm run: 1 (default) = also time the same steps with per-rank BatchNorm statistics "
                         "and report both in the line (train_extra.bn_variants); 0 = skip the second variant")
    ap.add_argument("--channels-last", type=int, default=1, help="train mode: keep the colour branch in channels_last (NHWC) memory format")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL, one GPU per rank) for real runs; gloo lets several ranks share one GPU "
                         "to exercise the multi-process path on a single-GPU box")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2),
                    help="2: point branch on a second HIP stream under the colour branch (inference); "
                         "1: everything on one stream")
    ap.add_argument("--overlap-pyramid", type=int, default=1,
                    help="1 (default, inference with --streams 2): the forward builds the index pyramid itself on a third HIP "
                         "stream (forward_pm.StreamedPyramid: the 22 searches as a K = 16 and a K = 1 batch, ~1 ms, under the colour stem); "
                         "0: whole pyramid first, then the forward")
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5),
                    help="BASELINE.json workload: 2 = bs=8, N=12288, 22 classes, fp32 (the headline metric, default); "
                         "4 = YCB-shaped bs=8, N=24576, 22 classes, fp32; 5 = bs=16, N=12288, bf16 mixed precision "
                         "(bfloat16 activations / weights, fp32 accumulation).  Explicit --batch / --n-points override it")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default=None, help="default: by --config")
    ap.add_argument("--mark-region", action="store_true",
                    help="launch a marker kernel (check_range_kernel) right before and after the timed steps "
                         "so scripts/rocpd_stats.py --between can cut the warm-up out of a rocprofv3 trace")
    ap.add_argument("--path", choices=["fused", "dropin"], default="fused",
                    help="dropin (inference, one GPU): the operator-level drop-in -- the reference's dataflow in the reference's channel-major "
                         "layout through stock conv / BatchNorm modules (ffb6d_amd/dropin.py, pinned to the reference's end_points on CPU) "
                         "with patch.patch_classes applied, i.e. what patch_reference makes an unmodified reference model execute; "
                         "fused (default): the package's own point-major inference path")
    ap.add_argument("--e2e-objects", type=int, default=5, help="--mode e2e: objects per frame of the synthetic vote field")
    ap.add_argument("--fit-spread", type=int, default=None, choices=(0, 1, 2),
                    help="--mode e2e: the pose solver's chip-wide first rounds (pose.set_fit_spread): 0 never, 2 always, 1 (default) = "
                         "calls of few sets, and never under the overlapped schedule")
    ap.add_argument("--form", action="append", default=[], metavar="NAME=0|1",
                    help="A/B runs: set a boolean form attribute of ffb6d_amd.forward_pm (HEADS_SHARE_FIRST=0 ...) before the model is "
                         "built; the line's config.forms records what ran")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 16 if args.config == 5 else 8
    if args.precision is None:
        args.precision = "bf16" if args.config == 5 else "fp32"
    if args.n_points is None:
        args.n_points = 24576 if args.config == 4 else 12288
    return args


# hot_path_ops key -> kernel instantiation as rocprofv3 lists it (csrc/mlp_pm.hip: tile id from ffb6d_mlp_pm_tile)
MLP_KERNEL_NAMES = {"mlp_pm<128x128>": "mlp_pm_kernel<2, 2, 2, 2, false>", "mlp_pm<64x256>": "mlp_pm_kernel<2, 2, 1, 4, false>",
                    "mlp_pm<32x256>": "mlp_pm_kernel<1, 2, 1, 4, false>", "mlp_pm<64x64>": "mlp_pm_kernel<1, 1, 2, 2, false>",
                    "mlp_pm<64x32,ksplit>": "mlp_pm_kernel<2, 1, 2, 2, true>",
                    "mlp_pm<stream>": "mlp_pm_stream_kernel<T, TM, NS, LSM, TWO>", "mlp_pm<lds128x128>": "mlp_pm_lds_kernel<T>",
                    "mlp_pm<seq128x128>": "mlp_pm_seq_kernel<TWO>", "mlp_pm<big256x256>": "mlp_pm_big_kernel",
                    "att_pool_pm": "att_pool_pm_kernel<TM, TN>", "lfa_pm": "lfa_pm_kernel<T, D, MODE, P>"}
PM_TILES = {1: "128x128", 2: "64x256", 3: "32x256", 4: "64x64", 5: "64x32,ksplit", 6: "stream", 7: "lds128x128", 8: "seq128x128", 9: "big256x256"}


def gemm_flops(name, rec_tag, batch):
    """algorithmic flops of one traced GEMM launch (tags: see ffb6d_amd/ops.py / ops_pm.py)"""
    if name.startswith("mlp_pm"):
        return 2.0 * rec_tag[0] * rec_tag[1] * rec_tag[2]                  # (K, Cout, rows of all frames, tile)
    if name.startswith("att_pool_pm"):
        return 2.0 * rec_tag[0] * rec_tag[0] * 16 * rec_tag[1] * batch     # (d, N): score GEMM d x d over 16 N pairs
    if name.startswith("lfa_pm"):
        return float(rec_tag[4])                                           # (mode, d, N, dtype, flops): ops_pm.lfa_half
    return 0.0


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and args.dist_backend == "nccl":
        raise SystemExit(f"--gpus {args.gpus} requested but only {have} GPU(s) are visible")
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")).returncode)


def state_dict(n_classes):
    from ffb6d_amd import synth
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as fh:
        shapes = json.load(fh)
    return synth.synth_state_dict_from_shapes(shapes, seed=0, n_classes=n_classes)


def cpu_baseline(args, sd):
    """Oracle path on the host CPU: single frames (bs=1), all cores for the forward; the KNN
    pyramid is single-threaded per frame by construction in the reference (knn_.cxx:108)."""
    from ffb6d_amd import synth
    from oracle import forward_ref
    from oracle import knn as oknn
    from oracle import pyramid as opyr
    from oracle import ref_harness

    use_ref = os.path.exists(ref_harness.REF_KNN_SO)
    if use_ref:
        def search(s, q, k):
            return ref_harness.ref_knn_batch(s, q, k, omp=True).astype(np.int32)
    else:
        search = oknn.knn_search
    # more threads than ~32 make the CPU path slower (oversubscribed small convolutions, OpenMP
    # team start-up inside the per-frame KNN calls): measured 108 s/frame at 256 threads on the
    # 256-thread bench host vs ~1 s/frame at 8 threads on the build container
    cores = min(os.cpu_count() or 1, args.cpu_threads)
    torch.set_num_threads(cores)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cores)
    except Exception:  # pragma: no cover
        pass
    t_knn, t_fwd = [], []
    t_start = time.perf_counter()
    for i in range(args.cpu_frames + 1):        # first frame = warm-up
        if i > 1 and time.perf_counter() - t_start > 45.0:
            break
        frames = synth.make_batch(2, 1, n_points=args.n_points) if i == 0 else \
            {k: v[i % args.batch: i % args.batch + 1] for k, v in cpu_baseline.frames.items()}
        t0 = time.perf_counter()
        pyr = opyr.build_batch(frames, search)
        t1 = time.perf_counter()
        inputs = {"rgb": torch.from_numpy(frames["rgb"].astype(np.float32)),
                  "cld_rgb_nrm": torch.from_numpy(frames["cld_rgb_nrm"]),
                  "choose": torch.from_numpy(frames["choose"].astype(np.int64))}
        for k, v in pyr.items():
            inputs[k] = torch.from_numpy(v.astype(np.int64) if v.dtype == np.int32 else v)
        with torch.no_grad():
            forward_ref.ffb6d_forward(sd, inputs)
        t2 = time.perf_counter()
        if i > 0:
            t_knn.append(t1 - t0)
            t_fwd.append(t2 - t1)
    knn_s, fwd_s = float(np.median(t_knn)), float(np.median(t_fwd))
    return {
        "value": 1.0 / (knn_s + fwd_s), "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": f"{len(t_fwd)} single 480x640 frames (bs=1, N={args.n_points}) after 1 warm-up; "
                  f"KNN pyramid {'reference nanoflann (oracle/_ref), 1 thread' if use_ref else 'C oracle brute force, all cores'} "
                  f"{knn_s * 1e3:.0f} ms/frame + plain-torch fp32 forward (oracle/forward_ref.py, {cores} threads) "
                  f"{fwd_s * 1e3:.0f} ms/frame",
        "knn_ms_per_frame": knn_s * 1e3, "forward_ms_per_frame": fwd_s * 1e3,
        "forward_only_fps": 1.0 / fwd_s,
    }


def run_e2e(args, dev, net, frames):
    """--mode e2e: K batches through ffb6d_amd.pipeline.SensorToPose, serial and overlapped; one JSON line."""
    from ffb6d_amd import inputs, pipeline, pose, synth
    B, N = args.batch, args.n_points
    if args.fit_spread is not None:
        pose.set_fit_spread(args.fit_spread)              # (1: the overlapped schedule still switches it off, pipeline.py)
    rgb = torch.from_numpy(frames["rgb"]).to(dev)                                         # uint8 [B,3,H,W]
    depth = torch.from_numpy(np.ascontiguousarray(frames["dpt_xyz"][:, 2])).to(dev)       # metres, zeros where invalid
    sensor = {"rgb": rgb, "depth": depth}
    n_obj = args.e2e_objects
    cases = [synth.make_pose_case(900 + b, n_pts=N, n_obj=n_obj, mesh_seed=9) for b in range(B)]
    stack = lambda key: torch.from_numpy(np.stack([c[key] for c in cases])).to(dev)       # noqa: E731
    votes = (stack("pcld"), stack("mask"), stack("ctr_of"), stack("kp_of"))
    rng = np.random.RandomState(3)
    n_cls = args.n_classes
    variants = {
        # the judge-specified workload: 5 objects per frame.  Random-init weights segment nothing, so the solver is fed a synthetic
        # 5-object vote field of the same shapes; the stage still starts from the forward's completion event
        "synthetic_votes": dict(pose_inputs=lambda inp, out: votes, mesh_kps=cases[0]["mesh_kps"], mesh_ctr=cases[0]["mesh_ctr"],
                                r_lst=cases[0]["r_lst"]),
        # the network's own (noise) votes: argmax of the segmentation, every class that occurs in a frame is solved
        "network_votes": dict(pose_inputs=None, mesh_kps=((rng.rand(n_cls, 8, 3) - 0.5) * 0.2).astype(np.float32),
                              mesh_ctr=((rng.rand(n_cls, 3) - 0.5) * 0.02).astype(np.float32),
                              r_lst=(0.08 + 0.05 * rng.rand(n_cls - 1)).astype(np.float32)),
    }
    ev = lambda: torch.cuda.Event(enable_timing=True)                                     # noqa: E731
    out = {}
    K = args.steps
    for name, kw in variants.items():
        pipe = pipeline.SensorToPose(net, synth.LINEMOD_K, N, kw["mesh_kps"], kw["mesh_ctr"], r_lst=kw["r_lst"],
                                     pose_inputs=kw["pose_inputs"], seed=7)
        batches = [sensor] * K
        res = {}
        for mode in ("serial", "overlapped"):
            pipe.run([sensor] * max(2, args.warmup), overlap=mode == "overlapped")
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            poses = pipe.run(batches, overlap=mode == "overlapped")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[mode] = {"ms_per_batch": 1e3 * dt / K, "frames_per_s": B * K / dt}
            res[mode + "_objects_per_frame"] = float(np.mean([len(f[0]) for p in poses for f in p]))
        # the three stages alone (HIP events / wall clock for the pose stage, which reads its results back)
        inp = pipe.assemble(sensor, 0)
        o = pipe.forward(inp)
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        for _ in range(10):
            pipe.assemble(sensor, 0)
        b.record()
        torch.cuda.synchronize()
        res["stage_ms"] = {"inputs": a.elapsed_time(b) / 10}
        a, b = ev(), ev()
        a.record()
        for _ in range(5):
            pipe.forward(inp)
        b.record()
        torch.cuda.synchronize()
        res["stage_ms"]["forward"] = a.elapsed_time(b) / 5
        t0 = time.perf_counter()
        for _ in range(5):
            pipe.solve(inp, o)
        torch.cuda.synchronize()
        res["stage_ms"]["pose"] = 1e3 * (time.perf_counter() - t0) / 5
        out[name] = res
    main = out["synthetic_votes"]
    line = {
        "metric": f"RGB-D frames/sec sensor->pose (480x640, N={N}, bs={B}, {n_obj} objects/frame)", "value": main["overlapped"]["frames_per_s"],
        "unit": "frames/s", "n_gpus": 1, "steps": K, "warmup": args.warmup, "ms_per_step": main["overlapped"]["ms_per_batch"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16",
        "data": "synthetic",
        "config": {"workload": f"sensor -> pose: depth + rgb resident in HBM -> normals, cloud, point sampling (inputs.py) -> FFB6D.forward incl. "
                               f"the 22-call KNN index pyramid -> pose solver (mean-shift votes + least-squares fit) for {n_obj} objects per "
                               f"frame; bs={B}, N={N}, 480x640, {args.precision}; serial = the three stages of a batch one after the other, "
                               f"overlapped = input assembly of batch i+1 and pose solver of batch i on side streams under the forward of "
                               f"batch i+1 (ffb6d_amd/pipeline.py)"},
        "e2e": out,
        "e2e_note": "synthetic_votes is the measured workload (5 objects per frame, ~1700 clustered votes per set).  network_votes feeds the "
                    "random-init network's own outputs to the solver: one class for all 12288 points and offsets of ~80 m, i.e. one "
                    "scattered 12288-vote set per frame whose points never meet -- max_iter + 1 rounds of 12288^2 pairs "
                    "(profiles/r06_pose_netvotes_probe.txt); kept as the solver's worst case, not a pose workload",
        "reference_published": "57 ms forward + 18 ms pose = 75 ms per FRAME on the reference's GPU (README.md:305-330); other hardware, "
                               "not a baseline for vs_baseline",
        "roofline": None,
    }
    print(json.dumps(line), flush=True)


def run_dropin(args, dev, net, cld, dpt_xyz, fixed, idt):
    """--path dropin: pyramid + the reference-shaped forward with the channel-major HIP operators patched in; one JSON line with the
    per-operator table (algorithmic bytes by the SURVEY 8d rule / HIP-event time) and the same forward with plain-torch operators."""
    from ffb6d_amd import _lib, dropin, patch, pyramid
    stand = dropin.FFB6D(net)
    ev = lambda: torch.cuda.Event(enable_timing=True)                                     # noqa: E731

    def step():
        inputs = pyramid.build_index_pyramid(cld, dpt_xyz, index_dtype=idt)
        inputs.update(fixed)
        with torch.no_grad():
            return stand(inputs)

    def timed(n):
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        for _ in range(n):
            step()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    plain_ms = timed(max(2, args.steps // 2))                 # unpatched: index expansion + torch.gather + permutes, softmax / mul / sum
    undo = patch.patch_classes(dropin.FFB6D, dropin.Building_block, dropin.Att_pooling)
    try:
        ms = timed(args.steps)
        tr = _lib.Tracer(None)
        _lib.TRACER = tr
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        _lib.TRACER = None
    finally:
        undo()
    table = {}
    for name, r in tr.summary().items():
        table[name] = {"launches_per_step": r["launches"] / 3, "ms_per_step": r["total_ms"] / 3, "algorithmic_GBps": r["gbps"],
                       "frac_of_hbm_peak": r["gbps"] / HBM_PEAK_GBS}
    by_shape = {f"{name}{tag}": {"launches_per_step": r["launches"] / 3, "avg_us": r["avg_us"], "algorithmic_GBps": r["gbps"],
                                 "frac_of_hbm_peak": r["gbps"] / HBM_PEAK_GBS}
                for (name, tag), r in sorted(tr.summary(by_tag=True).items(), key=lambda kv: -kv[1]["total_ms"])
                if name in ("random_sample", "nearest_interpolation", "gather_neighbour", "relative_pos_encoding", "att_pool")}
    gathers = [k for k in ("random_sample", "nearest_interpolation", "gather_neighbour", "relative_pos_encoding", "att_pool") if k in table]
    dom = max(gathers, key=lambda k: table[k]["ms_per_step"])
    line = {"metric": METRIC, "value": args.batch / (ms * 1e-3), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"operator-level drop-in: on-device 22-call KNN index pyramid + the reference's forward dataflow in the "
                                   f"reference's channel-major layout (stock conv / BatchNorm / activation modules, NCHW) with the five "
                                   f"neighbour operators of ffb6d_amd.ops patched in (patch.patch_classes on ffb6d_amd/dropin.py); "
                                   f"bs={args.batch}, N={args.n_points}, 480x640, fp32, one stream",
                       "path": "dropin"},
            "same_forward_with_plain_torch_operators": {"ms_per_step": plain_ms, "frames_per_s": args.batch / (plain_ms * 1e-3)},
            "roofline": {"bound": "hbm", "achieved": table[dom]["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": table[dom]["frac_of_hbm_peak"], "traffic": None, "kernel": dom,
                         "note": "the drop-in operator with the largest share of the step; every operator: hot_path_ops / hot_path_ops_by_shape"},
            "hot_path_ops": table, "hot_path_ops_by_shape": by_shape}
    print(json.dumps(line), flush=True)


def main():
    args = parse()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)            # never returns: a multi-GPU request is never silently run on one rank
    miopen_db = "MIOpen's own search (fresh user find-db)"
    if args.miopen_db == "pinned":
        from ffb6d_amd import miopen_pin
        # before the first convolution: a private copy per process; the fp32 forward-only run also pins the solver
        miopen_db = miopen_pin.use(rank=local_rank, only_solver=miopen_pin.FWD_SOLVER if (args.mode == "infer" and args.precision == "fp32" and args.path == "fused") else None)
    if world_env > 1:
        # one MIOpen find-db / kernel cache per rank: N ranks tuning the same convolutions at the same
        # time would otherwise contend for the lock of one sqlite user database
        os.environ.setdefault("MIOPEN_USER_DB_PATH", f"/tmp/ffb6d_miopen_db_rank{local_rank}")
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", f"/tmp/ffb6d_miopen_cache_rank{local_rank}")
        for d in (os.environ["MIOPEN_USER_DB_PATH"], os.environ["MIOPEN_CUSTOM_CACHE_DIR"]):
            os.makedirs(d, exist_ok=True)
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the product path has no CPU fallback)")
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.backends.cudnn.benchmark = bool(args.cudnn_benchmark)

    from ffb6d_amd import _lib, distributed, forward_pm, model, ops, pyramid, synth
    for item in args.form:
        name, _, val = item.partition("=")
        if not isinstance(getattr(forward_pm, name, None), bool) or val not in ("0", "1"):
            raise SystemExit(f"--form {item}: not a boolean form attribute of ffb6d_amd.forward_pm")
        setattr(forward_pm, name, val == "1")
    group = distributed.init_from_env(backend=args.dist_backend, device=dev)   # nccl == RCCL on ROCm
    rank, world = group.rank, group.world
    _lib.load()
    idt = torch.int64 if args.index_dtype == "int64" else torch.int32

    sd = state_dict(args.n_classes)
    net = model.FFB6D(n_classes=args.n_classes, n_pts=args.n_points)
    net.load_state_dict(sd)
    net = net.to(dev)
    train = args.mode == "train"
    opt = None
    if train:
        net.train()
        if args.channels_last:      # NHWC convolutions forward and backward (MIOpen's fastest path on gfx950), no layout transposes.
            # (Converting the colour branch only was measured slower: 92.9 against 98.0 frames/s in bf16 -- the fusion layers then
            # hand NCHW maps back to it.)
            if args.channels_last == 2:     # A/B: pixel-map modules only (colour branch + the p2r fusion convolutions); point-branch
                for name, mod in net.named_children():     # modules keep NCHW = channel-major [B,C,N,1], what the neighbour ops read
                    if name.startswith("cnn_") or name.endswith("p2r_fuse_layers"):
                        mod.to(memory_format=torch.channels_last)
            else:
                net = net.to(memory_format=torch.channels_last)
        ddp = distributed.wrap_ddp(net, dev, sync_bn=False if args.local_bn else None) if world > 1 else net     # RCCL all-reduce of 33.85 M fp32 grads
        opt = torch.optim.Adam(net.parameters(), lr=1e-5)              # train_lm.py:596
    else:
        net.eval()

    # per-rank batch, resident in HBM before the timed region: rank r holds samples
    # [r*batch, (r+1)*batch) of the config-2 synthetic stream (seeds 1000*2 + sample)
    frames = distributed.shard_frames(args.config, args.batch, rank, None, n_points=args.n_points)
    cpu_baseline.frames = frames
    rgb = torch.from_numpy(frames["rgb"]).to(dev).float()
    if train and args.channels_last:
        rgb = rgb.contiguous(memory_format=torch.channels_last)
    cld_rgb_nrm = torch.from_numpy(frames["cld_rgb_nrm"]).to(dev)
    choose = torch.from_numpy(frames["choose"]).to(dev).long()
    cld = torch.from_numpy(frames["cld"]).to(dev)
    dpt_xyz = torch.from_numpy(frames["dpt_xyz"]).to(dev)

    targets = None
    if train and args.objective == "reference":
        from ffb6d_amd import loss as ffb6d_loss
        tg = [synth.make_targets(synth.frame_seed(args.config, rank * args.batch + s), frames["cld"][s], n_classes=args.n_classes)
              for s in range(args.batch)]
        targets = tuple(torch.from_numpy(np.stack([t[k] for t in tg])).to(dev) for k in ("labels", "kp_targ_ofst", "ctr_targ_ofst"))
        targets = (targets[0].long(),) + targets[1:]

    if args.path == "dropin":
        if world > 1 or train:
            raise SystemExit("--path dropin is a one-GPU inference record")
        run_dropin(args, dev, net, cld, dpt_xyz, {"rgb": rgb, "cld_rgb_nrm": cld_rgb_nrm, "choose": choose}, idt)
        group.close()
        return
    if args.mode == "e2e":
        if world > 1:
            raise SystemExit("--mode e2e is a one-GPU pipeline record")
        net.two_streams, net.precision, net.index_dtype = bool(args.streams == 2), args.precision, idt
        run_e2e(args, dev, net, frames)
        group.close()
        return

    ev = lambda: torch.cuda.Event(enable_timing=True)
    phase = {"pyramid": [], "forward": []}

    # streams: the point branch of the forward runs on a second HIP stream under the colour branch's
    # convolutions (forward_pm.forward); with --overlap-pyramid the index pyramid is
    # enqueued on that stream too (nothing in the colour stem needs an index)
    overlap = bool(args.streams == 2) and not train
    net.two_streams = overlap
    net.precision = args.precision
    net.index_dtype = idt
    side = True if (overlap and args.overlap_pyramid) else None       # pyramid streamed inside the forward

    def step(record=False):
        e0, e1, e2 = (ev(), ev(), ev()) if record else (None, None, None)
        if record:
            e0.record()
        if side is not None:
            inputs = {"dpt_xyz": dpt_xyz}
            if record:
                e1.record()
        else:
            inputs = pyramid.build_index_pyramid(cld, dpt_xyz, index_dtype=idt)
            if record:
                e1.record()
        inputs.update(rgb=rgb, cld_rgb_nrm=cld_rgb_nrm, choose=choose)
        if train:
            opt.zero_grad(set_to_none=True)
            # --precision bf16: mixed precision as the reference trains (apex amp, train_lm.py:600) -- torch.autocast runs
            # the convolutions / 1x1 layers in bfloat16 with fp32 master weights; the neighbour operators (ops_cl), the decoder's
            # LogSoftmax and the up-samplings work on rows in that activation dtype with fp32 arithmetic inside and round their outputs
            # to it
            with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.precision == "bf16"):
                out = ddp(inputs)
                if targets is not None:        # the reference's objective (train_lm.py:245-259) on the synthetic targets
                    loss = ffb6d_loss.training_loss(out, *targets)[0]
                else:                          # --objective proxy: touches all three heads, no labels needed
                    loss = sum((v.float() ** 2).mean() for v in out.values())
            loss.backward()
            opt.step()
        else:
            out = net(inputs)
        if record:
            e2.record()
            phase["pyramid"].append((e0, e1))
            phase["forward"].append((e0 if side is not None else e1, e2))
        return out

    # op name of a traced launch as hot_path_ops lists it: the shared-MLP launches are split by the kernel instantiation
    # rocprofv3 lists them under -- mlp_pm by tile / kernel form -- and also by the side of the ridge the layer is on (157.3 TFLOP/s / 8 TB/s =
    # 19.7 flop per byte in fp32: arithmetic intensity of a row = 2 K Cout flop over esz (K + Cout) bytes)
    def split_name(name, tag):
        if name == "mlp_pm":
            k, cout = tag[0], tag[1]
            esz, peak = (2.0, BF16_PEAK_TFLOPS) if args.precision == "bf16" else (4.0, VALU_PEAK_TFLOPS)
            ridge_side = "mfma" if 2.0 * k * cout / (esz * (k + cout)) >= peak * 1e3 / HBM_PEAK_GBS else "hbm"
            return "mlp_pm<%s,%s>" % (PM_TILES.get(tag[3], "?"), ridge_side)
        return name

    def split_mlp(tr):
        for rec in tr.records.pop("mlp_pm", []):
            tr.records.setdefault(split_name("mlp_pm", rec[3]), []).append(rec)

    with torch.no_grad():
        marker = torch.zeros(4, dtype=torch.int32, device=dev)
        for _ in range(args.warmup):            # the W untimed warm-up steps (MIOpen find, allocator, caches)
            step()
        torch.cuda.synchronize()

        # Untimed measurement passes between warm-up and the timed region (every rank runs them, to stay in step):
        #  1. N_FULL steps in the benchmarked stream configuration with EVERY hand-written op bracketed by HIP events on
        #     its launch stream -> hot_path_ops (with several streams an interval also counts the time a launch waits
        #     behind the other streams' kernels);
        #  2. the same steps on ONE stream: clean per-kernel durations -> which kernel is the dominant one, and
        #     roofline.isolated.
        # The timed region then brackets only the launches of that dominant kernel (two event records per launch are
        # measurement overhead: ~180 launches per step cost ~3 % of a step when all of them are bracketed).
        N_FULL = 3
        full = _lib.Tracer(None if args.roofline_op == "auto" else [args.roofline_op])
        _lib.TRACER = full
        for _ in range(N_FULL):
            step()
        torch.cuda.synchronize()
        _lib.TRACER = None
        serial = None
        pyramid_on_side = side is not None
        if overlap:
            net.two_streams, keep_side, side = False, side, None
            serial = _lib.Tracer(None if args.roofline_op == "auto" else [args.roofline_op])
            step()
            torch.cuda.synchronize()
            _lib.TRACER = serial
            for _ in range(N_FULL):
                step()
            torch.cuda.synchronize()
            _lib.TRACER = None
            net.two_streams, side = True, keep_side
        pyr_alone_ms = None
        if pyramid_on_side:       # the pyramid by itself (it has no interval of its own inside a streamed step)
            pe = [ev(), ev()]
            pyramid.build_index_pyramid(cld, dpt_xyz, index_dtype=idt)
            pe[0].record()
            for _ in range(5):
                pyramid.build_index_pyramid(cld, dpt_xyz, index_dtype=idt)
            pe[1].record()
            torch.cuda.synchronize()
            pyr_alone_ms = pe[0].elapsed_time(pe[1]) / 5

        split_mlp(full)
        if serial is not None:
            split_mlp(serial)
        pick_from = (serial if serial is not None else full).summary()
        cand = {k: v for k, v in pick_from.items() if not k.startswith("knn") and v["launches"] and k in full.records}
        roof_op = args.roofline_op if args.roofline_op != "auto" else \
            (max(cand, key=lambda k: cand[k]["total_ms"]) if cand else None)
        tracer = _lib.Tracer(pred=lambda name, tag: roof_op is not None and split_name(name, tag) == roof_op)

        def one_step(timed):
            if timed and _lib.TRACER is None:
                if args.mark_region:
                    ops.check_index_range(marker, 1)
                _lib.TRACER = tracer
            step(record=timed)

        elapsed = distributed.timed_steps(one_step, 0, args.steps, group, sync=torch.cuda.synchronize)
        _lib.TRACER = None
        # >1 rank, training: what does the step cost in collectives, and what does SyncBatchNorm cost?  One counted step, then the
        # same steps on a second copy of the model with per-rank BatchNorm statistics (gradients the only collective) -- both variants
        # in one invocation, so that one run on the 8-GPU node prices SyncBatchNorm.
        train_extra = None
        if train and world > 1:
            with distributed.count_collectives() as cc:
                step()
                torch.cuda.synchronize()
            train_extra = {"python_side_collectives_per_step": cc.counts, "ddp_gradient_buckets": distributed.ddp_bucket_count(ddp),
                           "sync_batchnorm_layers": sum(isinstance(m, torch.nn.SyncBatchNorm) for m in ddp.modules())}
            if train_extra["sync_batchnorm_layers"] and args.bn_variants:
                net2 = model.FFB6D(n_classes=args.n_classes, n_pts=args.n_points)
                net2.load_state_dict(sd)
                net2 = net2.to(dev).train()
                if args.channels_last:
                    net2 = net2.to(memory_format=torch.channels_last)
                ddp2 = distributed.wrap_ddp(net2, dev, sync_bn=False)
                opt2 = torch.optim.Adam(net2.parameters(), lr=1e-5)

                def step2(_timed=False):
                    inputs = pyramid.build_index_pyramid(cld, dpt_xyz, index_dtype=idt)
                    inputs.update(rgb=rgb, cld_rgb_nrm=cld_rgb_nrm, choose=choose)
                    opt2.zero_grad(set_to_none=True)
                    with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.precision == "bf16"):
                        out = ddp2(inputs)
                        loss = ffb6d_loss.training_loss(out, *targets)[0] if targets is not None else sum((v.float() ** 2).mean() for v in out.values())
                    loss.backward()
                    opt2.step()
                el2 = distributed.timed_steps(step2, args.warmup, args.steps, group, sync=torch.cuda.synchronize)
                train_extra["bn_variants"] = {
                    "sync_batchnorm": {"ms_per_step": 1e3 * elapsed / args.steps, "frames_per_s": args.batch * world * args.steps / elapsed},
                    "local_batchnorm": {"ms_per_step": 1e3 * el2 / args.steps, "frames_per_s": args.batch * world * args.steps / el2}}
        if args.mark_region:
            ops.check_index_range(marker, 1)

        if args.trace_all and rank == 0:
            tall = _lib.Tracer(None)
            _lib.TRACER = tall
            for _ in range(max(2, args.steps // 2)):
                step()
            torch.cuda.synchronize()
            _lib.TRACER = None
            print("%-24s %9s %10s %10s %10s" % ("op", "launches", "total ms", "avg us", "alg GB/s"), file=sys.stderr)
            for name, r in sorted(tall.summary().items(), key=lambda kv: -kv[1]["total_ms"]):
                print("%-24s %9d %10.3f %10.1f %10.1f" % (name, r["launches"], r["total_ms"], r["avg_us"], r["gbps"]),
                      file=sys.stderr)
            print("--- by shape ---", file=sys.stderr)
            for (name, tag), r in sorted(tall.summary(by_tag=True).items(), key=lambda kv: -kv[1]["total_ms"])[:40]:
                print("%-24s %-22s %5d %9.3f %9.1f %9.1f" % (name, str(tag), r["launches"], r["total_ms"], r["avg_us"],
                                                            r["gbps"]), file=sys.stderr)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.batch * world * args.steps / elapsed
        pyr_ms = float(np.mean([a.elapsed_time(b) for a, b in phase["pyramid"]]))
        fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in phase["forward"]]))
        def is_gemm(name):
            return name.startswith(("mlp_pm", "att_pool_pm", "lfa_pm"))

        def mfma_bound(name):
            return is_gemm(name) and not name.endswith(",hbm>")

        def roofline_of(tr, op):
            summ = tr.summary().get(op)
            if not summ or not summ["launches"]:
                return None
            sec = summ["total_ms"] * 1e-3
            if mfma_bound(op):
                flops = sum(gemm_flops(op, t, args.batch) for _, _, _, t in tr.records[op])
                ach = flops / sec / 1e12
                peak = BF16_PEAK_TFLOPS if args.precision == "bf16" and op.startswith(("mlp_pm", "att_pool_pm", "lfa_pm")) else VALU_PEAK_TFLOPS
                return {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                        "kernel": MLP_KERNEL_NAMES.get(op.replace(",mfma>", ">"), op) +
                        (" (bf16 MFMA 32x32x16)" if peak == BF16_PEAK_TFLOPS else " (fp32 MFMA 32x32x2)"),
                        "launches": summ["launches"], "avg_launch_us": summ["avg_us"], "flops": flops, "bytes": summ["bytes"]}
            return {"bound": "hbm", "achieved": summ["gbps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": summ["gbps"] / HBM_PEAK_GBS, "kernel": MLP_KERNEL_NAMES.get(op.replace(",hbm>", ">"), op),
                    "launches": summ["launches"],
                    "avg_launch_us": summ["avg_us"], "flops": 0.0, "bytes": summ["bytes"]}

        split_mlp(tracer)
        summary = full.summary()         # every op: the N_FULL fully bracketed steps before the timed region
        roofline = None
        r = roofline_of(tracer, roof_op) if roof_op else None
        if r:
            traffic, traffic_set = None, None
            pmc_file = os.path.join(ROOT, "profiles", "pmc_traffic.json" if args.config == 2 else "pmc_traffic_config%d.json" % args.config)
            # HBM bytes per launch measured offline with rocprofv3 --pmc on the same workload (config 2: fp32; config 5: bf16).  The table holds
            # the mean over ALL launches of the kernel instantiation in a step -- both sides of the ridge -- so `traffic_set` gives the
            # algorithmic bytes per launch over that same set of launches (the figure to set `traffic` beside), not this group's only
            if os.path.exists(pmc_file) and args.precision == ("bf16" if args.config == 5 else "fp32") and args.batch == (16 if args.config == 5 else 8):
                with open(pmc_file) as fh:
                    table = json.load(fh)
                    key = roof_op.replace(",mfma>", ">").replace(",hbm>", ">")
                    traffic = table.get(key, table.get(key.split("<")[0], {})).get("hbm_bytes_per_launch")
                    same = [v for n, v in summary.items() if n.replace(",mfma>", ">").replace(",hbm>", ">") == key]
                    nl = sum(v["launches"] for v in same)
                    if traffic is not None and nl:
                        traffic_set = {"kernel": key, "launches_per_step": nl / N_FULL,
                                       "algorithmic_bytes_per_launch": sum(v["bytes"] for v in same) / nl}
            roofline = {"bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                        "frac": r["frac"], "traffic": traffic, "traffic_set": traffic_set, "kernel": r["kernel"],
                        "launches_per_step": r["launches"] / args.steps, "avg_launch_us": r["avg_launch_us"],
                        "algorithmic_bytes_per_step": r["bytes"] / args.steps}
            if r["bound"] == "mfma":
                roofline["algorithmic_flops_per_step"] = r["flops"] / args.steps
            else:
                roofline["algorithmic_bytes_per_launch"] = r["bytes"] / r["launches"]
        if roofline and r["bound"] == "mfma" and roof_op.startswith("mlp_pm"):
            # the same launches split by role: GEMMs of the north-star path (p2r / r2p fusion, decoder, heads) against those of
            # the colour decoder (folded PSPUpsample z GEMMs, PSP bottleneck), which SURVEY section 2 lists as CNN
            recs = tracer.records.get(roof_op, [])
            parts = {}
            for st, en, _, tag in recs:
                role = "north_star_path" if (len(tag) < 6 or tag[5] == "path") else "colour_decoder"
                a = parts.setdefault(role, [0.0, 0.0, 0])
                a[0] += gemm_flops(roof_op, tag, args.batch); a[1] += st.elapsed_time(en) * 1e-3; a[2] += 1
            roofline["by_role"] = {k: {"launches_per_step": v[2] / args.steps, "achieved": v[0] / v[1] / 1e12 if v[1] > 0 else 0.0,
                                       "frac": (v[0] / v[1] / 1e12 / r["peak"]) if v[1] > 0 else 0.0, "ms_per_step": 1e3 * v[1] / args.steps,
                                       "flops_per_step": v[0] / args.steps} for k, v in parts.items()}
        if roofline and serial is not None:
            iso = roofline_of(serial, roof_op)
            if iso:
                roofline["isolated"] = {"achieved": iso["achieved"], "frac": iso["frac"], "avg_launch_us": iso["avg_launch_us"],
                                        "note": "same kernel, same steps on one stream after the timed region: "
                                                "no kernel of the other stream shares the CUs"}
        ops_table = {k: {"launches_per_step": v["launches"] / N_FULL, "ms_per_step": v["total_ms"] / N_FULL,
                         "algorithmic_GBps": v["gbps"]} for k, v in summary.items()}
        for k in ops_table:
            if is_gemm(k):      # GEMM-shaped ops: MFMA rate next to the byte rate
                fl = sum(gemm_flops(k, t, args.batch) for _, _, _, t in full.records[k])
                ops_table[k]["algorithmic_TFLOPs"] = fl / (summary[k]["total_ms"] * 1e-3) / 1e12
            if k == "lfa_pm":   # fused kernel: also the bytes of the unfused reference ops it replaces over its time
                rb = sum(t[5] for _, _, _, t in full.records[k] if len(t) > 5)
                ops_table[k]["reference_equivalent_GBps"] = rb / (summary[k]["total_ms"] * 1e-3) / 1e9
        # the same table from the one-stream pass: durations no kernel of another stream shares the CUs with (the numbers a
        # per-kernel roofline fraction should be read from; hot_path_ops above is the benchmarked, overlapped schedule)
        ops_one_stream = None
        if serial is not None:
            one = serial.summary()
            ops_one_stream = {k: {"launches_per_step": v["launches"] / N_FULL, "ms_per_step": v["total_ms"] / N_FULL,
                                  "algorithmic_GBps": v["gbps"], "frac_of_hbm_peak": v["gbps"] / HBM_PEAK_GBS}
                              for k, v in one.items() if v["launches"] and v["total_ms"] > 0}
            for k in ops_one_stream:
                if is_gemm(k):
                    fl = sum(gemm_flops(k, t, args.batch) for _, _, _, t in serial.records[k])
                    ops_one_stream[k]["algorithmic_TFLOPs"] = fl / (one[k]["total_ms"] * 1e-3) / 1e12
                if k == "lfa_pm":
                    rb = sum(t[5] for _, _, _, t in serial.records[k] if len(t) > 5)
                    ops_one_stream[k]["reference_equivalent_GBps"] = rb / (one[k]["total_ms"] * 1e-3) / 1e9
        if "knn" in summary:
            # exact KNN is VALU/latency bound, not HBM bound: brute-force-equivalent pairs/s, and the pairs
            # the pruned search really evaluated (device counter, one extra untimed pyramid) against the fp32 VALU roof
            recs = full.records["knn"]

            def triples(tag):       # one (S, Q, K) per search; a batched call (search_many) carries a tuple of them
                return list(tag) if tag and isinstance(tag[0], tuple) else [tag]
            pairs = sum(t[0] * t[1] for _, _, _, tag in recs for t in triples(tag)) * args.batch
            sec = summary["knn"]["total_ms"] * 1e-3
            ops_table["knn"]["bruteforce_equivalent_Gpairs_per_s"] = pairs / sec / 1e9
            lib = _lib.load()
            ctr = torch.zeros(1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            _lib.check(lib.ffb6d_knn_set_pair_counter(ctr.data_ptr()), "ffb6d_knn_set_pair_counter")
            pyramid.build_index_pyramid(cld, dpt_xyz, index_dtype=idt)
            torch.cuda.synchronize()
            _lib.check(lib.ffb6d_knn_set_pair_counter(None), "ffb6d_knn_set_pair_counter")
            scanned = sum(t[0] * t[1] for _, _, _, tag in recs[:len(recs) // N_FULL] for t in triples(tag)
                          if not lib.ffb6d_knn_uses_pruning(args.batch, t[0], t[1], t[2])) * args.batch
            evaluated = int(ctr.item()) + scanned
            per_step_s = sec / N_FULL
            ops_table["knn"].update({
                "evaluated_Mpairs_per_step": evaluated / 1e6,
                "evaluated_fraction_of_bruteforce": evaluated / (pairs / N_FULL),
                "evaluated_Gpairs_per_s": evaluated / per_step_s / 1e9,
                # 8 flop per pair (3 sub, 3 mul, 2 add) against the fp32 vector peak
                "valu_frac_of_fp32_peak": evaluated * 8.0 / per_step_s / (VALU_PEAK_TFLOPS * 1e12)})
        line = {
            "metric": (METRIC.replace("12288", str(args.n_points)).replace("bs=8", "bs=%d" % args.batch)) if not train else
                      f"RGB-D frames/sec train step (fwd+bwd+Adam, 480x640, N={args.n_points}, bs={args.batch}/GPU)",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": ("FFB6D forward" if not train else "FFB6D training step (forward + backward + Adam, "
                                    "train-mode BatchNorm with %s statistics, DDP gradient all-reduce when n_gpus > 1)" %
                                    ("synchronised (SyncBatchNorm)" if not args.local_bn and world > 1 and args.dist_backend == "nccl" else "per-rank")) +
                                   " incl. on-device 22-call KNN index pyramid; "
                                   f"bs={args.batch}/GPU, N={args.n_points} pts, 480x640 RGB-D, "
                                   f"{args.n_classes} classes, " + ("bf16 activations/weights with fp32 accumulation" if args.precision == "bf16" else "fp32") +
                                   f", {'train' if train else 'eval'}" +
                                   ((", objective: 2 * focal(gamma=2) + masked L1 offsets (train_lm.py:245-259) on synthetic targets"
                                     if args.objective == "reference" else ", objective: mean of squared outputs (proxy)") if train else ""),
                       "baseline_config": args.config, "global_batch": args.batch * world, "n_points": args.n_points,
                       "index_dtype": args.index_dtype, "layout": "pm",
                       # the forms a record was taken with (module attributes since round 4: no environment switches)
                       "forms": {"lfa_fused": forward_pm.LFA_FUSED, "posenc_fused": forward_pm.POSENC_FUSED, "stem_fused": forward_pm.STEM_FUSED,
                                 "last_stage_at_chosen": forward_pm.LAST_STAGE_AT_CHOSEN, "heads_share_first": forward_pm.HEADS_SHARE_FIRST, "heads_align_last": forward_pm.HEADS_ALIGN_LAST,
                                 "heads_on_both_streams": forward_pm.HEADS_ON_BOTH_STREAMS, "heads_chain_fused": forward_pm.HEADS_CHAIN_FUSED,
                                 "gemm_seq_form": forward_pm.GEMM_SEQ_FORM, "gemm_big_form": forward_pm.GEMM_BIG_FORM, "gemm_seq_lin": forward_pm.GEMM_SEQ_LIN, "gemm_seq_lin_one": forward_pm.GEMM_SEQ_LIN_ONE,
                                 "miopen": ("find mode (cudnn.benchmark), " if args.cudnn_benchmark else "immediate mode, ") + miopen_db,
                                 "upconv_fold": forward_pm.UPCONV_FOLD if isinstance(forward_pm.UPCONV_FOLD, str) else
                                 (None if forward_pm.UPCONV_FOLD is None else sorted(forward_pm.UPCONV_FOLD)),
                                 "upconv_fold_bf16": forward_pm.UPCONV_FOLD_BF16,
                                 "psp_train_fold": model.PyramidPooling.fold_in_training,
                                 "final_rows_log_softmax": model.FinalHead.rows_log_softmax, "upsample_rows": ops.UPSAMPLE_ROWS,
                                 "pyramid_one_event": forward_pm.PYRAMID_ONE_EVENT},
                       "parallelism": f"dp{world} (independent batches, one process per GPU, "
                                      f"{args.dist_backend if world > 1 else 'no'} process group)"},
            "breakdown_ms": ({"step": fwd_ms, "knn_pyramid_alone": pyr_alone_ms,
                              "note": "the forward builds the index pyramid itself (the 22 searches as two batches: K = 16, then K = 1) on a third HIP stream, "
                                      "under the network (forward_pm.StreamedPyramid); knn_pyramid_alone = the same 22 "
                                      "searches run by themselves after the timed region"}
                             if pyramid_on_side else {"knn_pyramid": pyr_ms, "forward": fwd_ms}),
            "streams": (3 if pyramid_on_side else 2) if overlap else 1,
            **({} if pyramid_on_side else {"forward_only_fps": args.batch * world / (fwd_ms * 1e-3)}),
            "roofline": roofline,
            # hot-path-only time, i.e. without the MIOpen convolutions of the colour branch = sum of
            # the HIP-event durations of every hand-written launch of a step (with two streams these intervals
            # overlap the convolutions and each other, so the sum is GPU time, not a share of ms_per_step)
            "hot_path_only": {"kernel_ms_per_step": sum(v["ms_per_step"] for v in ops_table.values()),
                              "launches_per_step": sum(v["launches_per_step"] for v in ops_table.values())},
            "hot_path_ops": ops_table,
            "hot_path_ops_one_stream": ops_one_stream,
            "hot_path_ops_source": f"{N_FULL} untimed steps between warm-up and the timed region with every hand-written launch "
                                   "bracketed by HIP events; the timed region brackets only the roofline kernel's launches",
        }
        if train_extra is not None:
            line["train_extra"] = train_extra
        if not args.no_cpu_baseline and world == 1 and not train:
            line["cpu_baseline"] = cpu_baseline(args, sd)
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)

    group.close()


if __name__ == "__main__":
    main()
