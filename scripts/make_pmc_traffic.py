#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the two rocprofv3 --pmc summaries of scripts/profile_round.sh
(gpurun_out/<tag>_pmc_FETCH_SIZE.txt / _WRITE_SIZE.txt): HBM bytes per launch of every hand-written op, keyed like
bench.py's hot_path_ops.  Counter unit = 1024 B.  MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half the bytes of a wide
(16 B per lane) coalesced read on gfx950 -> x2 for the float4-streaming kernels (all point-major kernels), x1 for the
4-byte gather kernels of the channel-major path and the KNN kernels (uncalibrated width, taken at face value).

    python scripts/make_pmc_traffic.py r02
"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
KEYS = [("mlp_pm_kernel<2, 2, 1, 4, false>", "mlp_pm<64x256>", 2.0), ("mlp_pm_kernel<2, 2, 2, 2, false>", "mlp_pm<128x128>", 2.0),
        ("mlp_pm_kernel<1, 2, 1, 4, false>", "mlp_pm<32x256>", 2.0), ("mlp_pm_kernel<1, 1, 2, 2, false>", "mlp_pm<64x64>", 2.0),
        ("mlp_pm_kernel<2, 1, 2, 2, true>", "mlp_pm<64x32,ksplit>", 2.0), ("mlp_pm_lds_kernel", "mlp_pm<lds128x128>", 2.0),
        ("mlp_pm_seq_kernel", "mlp_pm<seq128x128>", 2.0), ("mlp_pm_big_kernel", "mlp_pm<big256x256>", 2.0), ("mlp_chain3_kernel", "mlp_chain3_pm", 2.0),
        ("mlp_pm_stream_kernel", "mlp_pm<stream>", 2.0), ("att_pool_pm_kernel", "att_pool_pm", 2.0),
        ("affine_act_pm_kernel", "affine_act_pm", 2.0), ("bilinear_pm_kernel", "bilinear_resize_pm", 2.0),
        ("upsampled_patch_rows_pm_kernel", "upsampled_patch_rows_pm", 2.0),
        ("random_sample_pm_kernel", "random_sample_pm", 2.0), ("rel_pos_enc_pm_kernel", "relative_pos_encoding_pm", 1.0),
        ("psp_rowsum_pm_kernel", "psp_pool_pm", 2.0), ("psp_binsum_pm_kernel", "psp_pool_pm", 2.0),
        ("psp_prior_sum_pm_kernel", "psp_prior_sum_pm", 2.0), ("knn_row16", "knn", 1.0), ("knn_pruned", "knn", 1.0),
        ("knn_scan", "knn", 1.0), ("lfa_pm_kernel", "lfa_pm", 2.0), ("upconv_combine", "upconv_combine_pm", 2.0),
        ("affine_relu_maxpool_pm_kernel", "affine_relu_maxpool_pm", 2.0), ("posenc_mlp_pm_kernel", "posenc_mlp_pm", 1.0)]


def read(path):
    rows = []
    for line in open(path):
        m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+(\d+)\s+([\d.]+)\s*$", line.rstrip())
        if m:
            rows.append((m.group(1), int(m.group(3)), float(m.group(4))))
    return rows


fetch = read(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_pmc_FETCH_SIZE.txt"))
write = read(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_pmc_WRITE_SIZE.txt"))
out = {"_comment": f"HBM bytes per launch from rocprofv3 --pmc (separate FETCH_SIZE and WRITE_SIZE passes over `python bench.py "
                   f"--steps 2 --warmup 2 --no-cpu-baseline --cudnn-benchmark 0` (+ the workload flags the tag names); profiles/{tag}_rocprofv3_pmc_*.txt). Counter unit "
                   "= 1024 B. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE under-reports wide coalesced 16 B/lane reads by 2x on "
                   "gfx950: fetch_correction 2.0 is applied to the float4-streaming point-major kernels, 1.0 elsewhere. Values are "
                   "means over all launches of the op in one forward (shapes differ per layer); with the x2 correction they are an "
                   "UPPER bound wherever part of the reads are narrower or L2-resident re-reads."}
acc = {}
for rows, what in ((fetch, "f"), (write, "w")):
    for name, n, mean in rows:
        for pat, key, corr in KEYS:
            if pat in name:
                a = acc.setdefault(key, {"f": 0.0, "w": 0.0, "nf": 0, "nw": 0, "corr": corr, "parts": 0})
                a[what] += n * mean
                a["n" + what] = max(a["n" + what], n) if key == "psp_pool_pm" else a["n" + what] + n
                break
for key, a in acc.items():
    nf, nw = max(a["nf"], 1), max(a["nw"], 1)
    fk, wk = a["f"] / nf, a["w"] / nw
    out[key] = {"fetch_kib": round(fk, 1), "write_kib": round(wk, 1), "fetch_correction": a["corr"],
                "hbm_bytes_per_launch": int((a["corr"] * fk + wk) * 1024),
                "hbm_bytes_per_launch_uncorrected": int((fk + wk) * 1024)}
# tag r06 -> profiles/pmc_traffic.json (the default workload); tag r06_config5 -> profiles/pmc_traffic_config5.json
suffix = tag.split("_", 1)[1] if "_" in tag else ""
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic%s.json" % ("_" + suffix if suffix else "")), "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items() if k != "_comment"}, indent=1))
