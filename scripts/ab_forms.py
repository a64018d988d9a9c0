"""In-process A/B of forward_pm's boolean form attributes on the default workload (bs = 8, N = 12288, fp32, three streams; --precision bf16
--batch 16 = BASELINE configuration 5):
one model and one set of MIOpen solver choices, the forms toggled between blocks of steps.  `bench.py --form` runs each setting in
its own process, where MIOpen's search alone moves the step by +-0.8 ms (round 4; since round 5 the bench pins MIOpen, ffb6d_amd/miopen_pin.py) -- more than the forms compared here.

    python scripts/ab_forms.py HEADS_SHARE_FIRST,HEADS_ALIGN_LAST HEADS_ON_BOTH_STREAMS
compares: every listed attribute off / the first group on / the first two groups on / ...; prints one JSON line."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                                  # noqa: E402
from ffb6d_amd import distributed, forward_pm, model          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("groups", nargs="+", help="comma-separated attribute names; group i is switched on from setting i on")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    groups = [g.split(",") for g in args.groups]
    for g in groups:
        for name in g:
            if not isinstance(getattr(forward_pm, name, None), bool):
                raise SystemExit(f"{name}: not a boolean form attribute of ffb6d_amd.forward_pm")
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    net = model.FFB6D(n_classes=22, n_pts=12288)
    net.load_state_dict(bench.state_dict(22))
    net = net.to(dev).eval()
    net.two_streams, net.precision, net.index_dtype = True, args.precision, torch.int64
    frames = distributed.shard_frames(2, args.batch, 0, None, n_points=12288)
    fixed = {"rgb": torch.from_numpy(frames["rgb"]).to(dev).float(), "cld_rgb_nrm": torch.from_numpy(frames["cld_rgb_nrm"]).to(dev),
             "choose": torch.from_numpy(frames["choose"]).to(dev).long()}
    dpt_xyz = torch.from_numpy(frames["dpt_xyz"]).to(dev)

    def step():
        with torch.no_grad():
            return net(dict(fixed, dpt_xyz=dpt_xyz))

    def setting(i):                                           # groups[:i] on, the rest off
        for j, g in enumerate(groups):
            for name in g:
                setattr(forward_pm, name, j < i)

    n_set = len(groups) + 1
    for i in range(n_set):                                    # warm every setting: weight caches, MIOpen's choices
        setting(i)
        for _ in range(5):
            step()
    torch.cuda.synchronize()
    ms = [[] for _ in range(n_set)]
    for r in range(args.rounds):
        order = range(n_set) if r % 2 == 0 else reversed(range(n_set))
        for i in order:
            setting(i)
            step()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.steps):
                step()
            b.record()
            torch.cuda.synchronize()
            ms[i].append(a.elapsed_time(b) / args.steps)
    out = {"workload": f"bs={args.batch}, N=12288, {args.precision}, three streams, one process", "steps_per_block": args.steps, "rounds": args.rounds, "settings": []}
    for i in range(n_set):
        on = [n for g in groups[:i] for n in g]
        out["settings"].append({"on": on, "ms_per_step": [round(x, 4) for x in ms[i]], "mean_ms": round(sum(ms[i]) / len(ms[i]), 4),
                                "min_ms": round(min(ms[i]), 4)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
