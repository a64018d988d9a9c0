#!/usr/bin/env python3
"""Micro-benchmark of the KNN kernel on the 22 call shapes of one FFB6D index pyramid
(B frames batched), HIP-event timed.  GPU box only.

    python scripts/bench_knn.py [--batch 8] [--n-points 12288] [--iters 5]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ffb6d_amd import pyramid, synth
from ffb6d_amd.nearest_neighbors import knn_batch_device, PreparedPoints, knn_prepared, uses_pruning


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--n-points", type=int, default=12288)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    frames = synth.make_batch(2, a.batch, n_points=a.n_points)
    cld = torch.from_numpy(frames["cld"]).to(dev)
    dpt = torch.from_numpy(frames["dpt_xyz"]).to(dev)
    grids = {s: pyramid.strided_grid(dpt, s) for s in (2, 4, 8)}
    calls, cur = [], cld
    lv = [cld]
    for i in range(4):
        sub = cur[:, : cur.shape[1] // 4].contiguous()
        g = grids[pyramid.RGB_DS_SR[i]]
        calls += [(f"self16_L{i}", cur, cur, 16), (f"up1_L{i}", sub, cur, 1),
                  (f"r2p_ds{i}", g, sub, 16), (f"p2r_ds{i}", sub, g, 1)]
        cur = sub
        lv.append(sub)
    for i in range(3):
        g = grids[pyramid.RGB_UP_SR[i]]
        pts = lv[3 - i]
        calls += [(f"r2p_up{i}", g, pts, 16), (f"p2r_up{i}", pts, g, 1)]
    tot = 0.0
    totpairs = 0
    print("%-12s %7s %7s %3s %10s %10s" % ("call", "S", "Q", "K", "us", "Gpairs/s"))
    prep = {}
    for name, s, q, k in calls:
        pruned = uses_pruning(a.batch, s.shape[1], q.shape[1], k)
        if pruned:
            for t in (s, q):
                if id(t) not in prep:
                    prep[id(t)] = PreparedPoints(t)
            run = lambda: knn_prepared(prep[id(s)], prep[id(q)], k, dtype=torch.int32)
            name += "*"
        else:
            run = lambda: knn_batch_device(s, q, k, dtype=torch.int32)
        for _ in range(2):
            run()
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        us = float(np.median(ts))
        pairs = a.batch * s.shape[1] * q.shape[1]
        tot += us; totpairs += pairs
        print("%-12s %7d %7d %3d %10.1f %10.1f" % (name, s.shape[1], q.shape[1], k, us, pairs / us / 1e3))
    print("TOTAL %.1f us for %d frames  (%.1f Gpairs/s brute-force equivalent); * = Morton-pruned search" % (tot, a.batch, totpairs / tot / 1e3))
    sets = [cld, lv[1], lv[2], lv[3], lv[4], grids[2], grids[4], grids[8]]
    tp = 0.0
    for t in sets:
        PreparedPoints(t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); PreparedPoints(t); e1.record(); torch.cuda.synchronize()
        print("prepare S=%6d: %8.1f us" % (t.shape[1], e0.elapsed_time(e1) * 1e3)); tp += e0.elapsed_time(e1) * 1e3
    print("prepare total %.1f us" % tp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pyramid.build_index_pyramid(cld, dpt); torch.cuda.synchronize()
    e0.record(); pyramid.build_index_pyramid(cld, dpt); e1.record(); torch.cuda.synchronize()
    print("build_index_pyramid end-to-end: %.1f us" % (e0.elapsed_time(e1) * 1e3))


if __name__ == "__main__":
    main()
