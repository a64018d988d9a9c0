"""One-process A/B of the opt-in forms (folded up-convolution, fused position encoding) on the GPU box: stage-level parity
against the plain-torch restatement with the test-suite's own bars, step time with each form switched on, per-op event
times.  Results are written to gpurun_out/final_shot.json after every stage (a cut-off call still leaves what was measured).

    python tests/ab_opt_in_forms.py [--skip-bf16]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out", "final_shot.json")
RES = {"stages": []}
T0 = time.perf_counter()


def save(stage):
    RES["stages"].append(stage)
    RES["elapsed_s"] = time.perf_counter() - T0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT + ".tmp", "w") as fh:
        json.dump(RES, fh, indent=1)
    os.replace(OUT + ".tmp", OUT)
    print("[final_shot] %-28s %.1f s" % (stage, RES["elapsed_s"]), flush=True)


def errs(got, want):
    got, want = got.double(), want.double()
    scale = float(want.abs().max())
    diff = (got - want).abs()
    ok = bool((diff <= 1e-4 * want.abs() + 1e-5 * scale).all())          # the elementwise bar of tests/test_forward_gpu.py
    return {"max_over_range": float(diff.max()) / scale, "mean_over_range": float(diff.mean()) / scale, "elementwise_bar": ok}


def main():
    from ffb6d_amd import _lib, distributed, forward_pm, pyramid, synth
    from oracle import forward_ref          # checker only (this file is test tooling and lives under tests/)
    import test_forward_gpu as T
    dev = torch.device("cuda", 0)
    RES["device"] = torch.cuda.get_device_name(0)
    _lib.load()
    variants = [("base", frozenset(), False), ("fold1024", frozenset([1024]), False), ("fold1024_256", frozenset([1024, 256]), False),
                ("fold_all", None, False), ("fold_all+posenc", None, True), ("posenc", frozenset(), True)]

    def set_variant(fold, pos):
        forward_pm.UPCONV_FOLD, forward_pm.POSENC_FUSED = fold, pos

    # ---- 1. stage-level parity, fp32, bs=2, N=12288, 480x640 (the configuration of test_every_fusion_stage_matches_plain_torch)
    frames = synth.make_batch(1, 2, n_points=12288, height=480, width=640)
    net = T.build(22, 12288, dev)
    inputs = pyramid.frames_to_device(frames, dev)
    ref_taps = {}
    with torch.no_grad():
        ref = forward_ref.ffb6d_forward(dict(net.state_dict()), inputs, taps=ref_taps)
    RES["parity_fp32_bs2"] = {}
    for name, fold, pos in [variants[0], variants[3], variants[4]]:
        set_variant(fold, pos)
        taps = {}
        with torch.no_grad():
            ep = net(inputs, taps=taps)
        torch.cuda.synchronize()
        rec = {k: errs(ep[k], ref[k]) for k in ref}
        rec.update({k: errs(taps[k], ref_taps[k]) for k in sorted(ref_taps)})
        rec["worst"] = max(v["max_over_range"] for v in rec.values())
        rec["all_elementwise"] = all(v["elementwise_bar"] for v in rec.values() if isinstance(v, dict))
        RES["parity_fp32_bs2"][name] = rec
        save("parity fp32 " + name)
    del net, inputs, ref, ref_taps

    # ---- 1b. the z GEMMs of the folded up-convolution: csrc/mlp_pm.hip against the library GEMM torch.mm dispatches to
    from ffb6d_amd import ops_pm
    RES["z_gemm"] = {}
    for rows, k, cout in ((38400, 1024, 2304), (153600, 256, 576), (614400, 64, 576)):
        x = torch.randn(rows, k, device=dev)
        w = torch.randn(cout, k, device=dev) / k ** 0.5
        out = torch.empty(rows, cout, device=dev)
        wt = w.t().contiguous()
        rec = {}
        for name, fn in (("mlp_pm", lambda: ops_pm.mlp(x, w, out=out)), ("torch_mm", lambda: torch.mm(x, wt, out=out)),
                         ("torch_linear", lambda: torch.nn.functional.linear(x, w))):
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / 5
            rec[name] = {"us": us, "TFLOPs": 2.0 * rows * k * cout / (us * 1e-6) / 1e12, "GBps": 4.0 * (rows * (k + cout)) / (us * 1e-6) / 1e9}
        rec["tile"] = int(_lib.load().ffb6d_mlp_pm_choice(rows, cout, k, 0, 0, 0, 0))
        RES["z_gemm"]["%dx%d->%d" % (rows, k, cout)] = rec
        del x, w, out, wt
    save("z gemm a/b")

    # ---- 2. step time, config 2 (bs=8, N=12288, fp32), streamed pyramid, like bench.py
    torch.backends.cudnn.benchmark = True

    def bench_setup(precision, batch):
        net = T.build(22, 12288, dev)
        net.two_streams, net.precision, net.index_dtype = True, precision, torch.int64
        fr = distributed.shard_frames(2 if precision == "fp32" else 5, batch, 0, None, n_points=12288)
        base = {"rgb": torch.from_numpy(fr["rgb"]).to(dev).float(), "cld_rgb_nrm": torch.from_numpy(fr["cld_rgb_nrm"]).to(dev),
                "choose": torch.from_numpy(fr["choose"]).to(dev).long(), "dpt_xyz": torch.from_numpy(fr["dpt_xyz"]).to(dev)}
        return net, base

    def time_steps(net, base, n):
        with torch.no_grad():
            for _ in range(2):
                out = net(dict(base))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = net(dict(base))
            torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n, {k: v.clone() for k, v in out.items()}

    def op_table(net, base):
        """event-bracketed launches of two steps on ONE stream: clean per-op durations"""
        net.two_streams = False
        tr = _lib.Tracer(None)
        with torch.no_grad():
            net(dict(base))
            torch.cuda.synchronize()
            _lib.TRACER = tr
            for _ in range(2):
                net(dict(base))
            torch.cuda.synchronize()
            _lib.TRACER = None
        net.two_streams = True
        rows = {}
        for (name, tag), r in tr.summary(by_tag=True).items():
            if name in ("upconv_combine_pm", "bilinear_resize_pm", "posenc_mlp_pm", "relative_pos_encoding_pm") or \
                    (name == "mlp_pm" and (tag[1] in (2304, 576) or tag[0] == 16)):
                rows["%s %s" % (name, tag)] = {"launches": r["launches"], "avg_us": r["avg_us"], "alg_GBps": r["gbps"],
                                               "TFLOPs": (2.0 * tag[0] * tag[1] * tag[2] / (r["avg_us"] * 1e-6) / 1e12) if name == "mlp_pm" else None}
        rows["_all_ops_ms_per_step"] = sum(r["total_ms"] for r in tr.summary().values()) / 2
        return rows

    for precision, batch, names in (("fp32", 8, [v[0] for v in variants]), ("bf16", 16, ["base", "fold_all", "fold_all+posenc"])):
        if precision == "bf16" and "--skip-bf16" in sys.argv:
            continue
        net, base = bench_setup(precision, batch)
        key = "step_%s_bs%d" % (precision, batch)
        RES[key] = {}
        outs0 = None
        if precision == "bf16":                   # fp32 answer of the same frames: the yardstick of the bf16 bars
            set_variant(frozenset(), False)
            net.precision = "fp32"
            torch.backends.cudnn.benchmark = False       # no MIOpen search for a yardstick that is run three times
            _, want32 = time_steps(net, base, 1)
            torch.backends.cudnn.benchmark = True
            net.precision = "bf16"
        for name, fold, pos in variants:
            if name not in names:
                continue
            set_variant(fold, pos)
            ms, outs = time_steps(net, base, 6)
            rec = {"ms_per_step": ms, "frames_per_s": batch / (ms * 1e-3)}
            if outs0 is None:
                outs0 = outs
            else:
                rec["vs_base"] = {k: errs(outs[k], outs0[k]) for k in outs}
            if precision == "bf16":
                rec["vs_fp32"] = {k: errs(outs[k], want32[k]) for k in outs}
            RES[key][name] = rec
            save("%s %s %.2f ms" % (key, name, ms))
        for name, fold, pos in (variants[0], variants[4]):
            set_variant(fold, pos)
            RES[key]["ops_one_stream_" + name] = op_table(net, base)
            save("%s op table %s" % (key, name))
        del net, base


if __name__ == "__main__":
    main()
