"""Sensor -> pose as ONE pipeline on one GPU: what the reference does per frame between the camera and the pose
(demo.py:154-182; train_lm.py:371-420 -> utils/pvn3d_eval_utils_kpls.py:448-500; README.md:305-330 publishes 57 ms forward +
18 ms pose = 75 ms per frame):

    stage A  depth (+ rgb) resident in HBM  ->  normals, cloud, valid-pixel sampling, input assembly       (ffb6d_amd.inputs)
    stage B  FFB6D.forward with the index pyramid built inside (22 exact-KNN searches per frame)             (ffb6d_amd.forward_pm)
    stage C  argmax of the segmentation, mean-shift voting per object / keypoint, least-squares fit         (ffb6d_amd.pose)

`run(batches, overlap=False)` runs A, B, C of a batch one after the other.  `overlap=True` is the pipelined schedule: stage A of
batch i + 1 (a side stream) and stage C of batch i (another side stream; its host-side polling and read-backs too) run under stage B
of batch i + 1 -- the stages of ONE batch still follow each other through events, so every batch's results are the bits the serial
schedule produces (tests/test_pipeline_gpu.py).  No collective, no host geometry; the only read-backs are the poses.
"""
import numpy as np
import torch

from . import inputs as _inputs
from . import pose as _pose


class SensorToPose:
    """net: model.FFB6D in eval mode on a GPU (net.two_streams / precision / index_dtype as the caller set them).
    K: camera intrinsics [3,3]; mesh_kps [n_cls,n_kps,3], mesh_ctr [n_cls,3], r_lst: the objects' model keypoints / radii
    (pvn3d_eval_utils_kpls.py:149-152); classes: None = every class present in a frame's mask (YCB flow), or a list of class ids.
    normals: "depth" = estimate them from the depth image (inputs.depth_normal: the LINE-MOD estimator behind the reference's
    normalSpeed call, linemod_dataset.py:252-254) or "given" = the sensor batch carries them.
    pose_inputs: optional callable (inp, out) -> (pcld [B,N,3], mask [B,N], ctr_of [B,1,N,3], kp_of [B,n_kps,N,3]) replacing the
    network's own votes (benchmarks with random weights feed a synthetic 5-object vote field; the stage still waits for the forward)."""

    def __init__(self, net, K, n_points, mesh_kps, mesh_ctr, r_lst=None, classes=None, normals="depth", cam_scale=1.0,
                 pose_inputs=None, seed=0):
        self.net, self.K, self.n_points = net, np.asarray(K, np.float64), int(n_points)
        self.mesh_kps, self.mesh_ctr, self.r_lst, self.classes = mesh_kps, mesh_ctr, r_lst, classes
        self.normals, self.cam_scale, self.pose_inputs, self.seed = normals, float(cam_scale), pose_inputs, int(seed)
        self.dev = next(net.parameters()).device
        self._streams = None

    # ---- the three stages, each on the current stream -------------------------------------------------------
    def assemble(self, batch, index):
        """batch: dict rgb [B,3,H,W] (uint8 or float), depth [B,H,W] float32 in metres * cam_scale (zeros = invalid), optionally
        normals [B,3,H,W] -> the forward's input dict without the index pyramid (the forward builds it) + `cld` [B,N,3]"""
        rgb, depth = batch["rgb"], batch["depth"]
        if self.normals == "depth":
            nrm = _inputs.depth_normal(depth * (1000.0 / self.cam_scale), self.K[0, 0], self.K[1, 1], 5, 2000, 20, False)
        else:
            nrm = batch["normals"]
        dpt_xyz = _inputs.depth_to_cloud(depth, self.K, self.cam_scale)
        pts = _inputs.sample_points(depth / self.cam_scale if self.cam_scale != 1.0 else depth, self.n_points, dpt_xyz, rgb, nrm,
                                    seed=self.seed + index)
        return {"rgb": rgb.float(), "cld_rgb_nrm": pts["cld_rgb_nrm"], "choose": pts["choose"], "dpt_xyz": dpt_xyz,
                "cld": pts["cld"], "n_valid": pts["n_valid"]}

    def forward(self, inp):
        with torch.no_grad():
            return self.net({k: inp[k] for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")})

    def solve(self, inp, out):
        """-> list over frames of (class ids, poses [n,3,4] float64, keypoints [n,n_kps+1,3]) (pose.solve_poses)"""
        if self.pose_inputs is not None:
            pcld, mask, ctr_of, kp_of = self.pose_inputs(inp, out)
        else:
            pcld, mask = inp["cld"], out["pred_rgbd_segs"].argmax(dim=1)            # train_lm.py:385 / demo.py:160
            ctr_of, kp_of = out["pred_ctr_ofs"], out["pred_kp_ofs"]
        return _pose.solve_poses(pcld, mask, ctr_of, kp_of, self.mesh_kps, self.mesh_ctr, r_lst=self.r_lst, classes=self.classes)

    # ---- schedules ------------------------------------------------------------------------------------------------
    def _side(self):
        if self._streams is None:
            with torch.cuda.device(self.dev):
                self._streams = (torch.cuda.Stream(), torch.cuda.Stream())
        return self._streams

    def run(self, batches, overlap=True, keep_outputs=False):
        """batches: sequence of sensor batches (dicts of GPU tensors).  Returns a list with one entry per batch: the poses
        (and, with keep_outputs, (poses, input dict, end points))."""
        results = []
        pack = (lambda p, i, o: (p, i, o)) if keep_outputs else (lambda p, i, o: p)
        if not overlap:
            for n, b in enumerate(batches):
                inp = self.assemble(b, n)
                out = self.forward(inp)
                results.append(pack(self.solve(inp, out), inp, out))
            return results
        # the solver runs UNDER the next forward here: its one-workgroup fits on 40 CUs cost the forward less than chip-wide first rounds
        # (overlapped period 25.3 ms without them, 25.8 ms with; serial 30.3 / 29.2 ms -- profiles/r06_e2e_spread_ab.txt)
        spread = _pose.set_fit_spread(0) if _pose.FIT_SPREAD == 1 else None
        try:
            return self._run_overlapped(batches, results, pack)
        finally:
            if spread is not None:
                _pose.set_fit_spread(spread)

    def _run_overlapped(self, batches, results, pack):
        main = torch.cuda.current_stream(self.dev)
        s_in, s_pose = self._side()

        s_in.wait_stream(main)                          # the sensor batches were produced on `main` (before this call)

        def assemble_on_side(n):
            with torch.cuda.stream(s_in):
                inp = self.assemble(batches[n], n)
                ev = torch.cuda.Event()
                ev.record(s_in)
            return inp, ev

        def solve_on_side(inp, out, ev):
            s_pose.wait_event(ev)
            with torch.cuda.stream(s_pose):
                for t in list(out.values()) + [inp["cld"]]:
                    t.record_stream(s_pose)
                return self.solve(inp, out)

        nxt = assemble_on_side(0)
        pending = None
        for n in range(len(batches)):
            inp, ev_in = nxt
            main.wait_event(ev_in)
            for t in inp.values():
                t.record_stream(main)
            out = self.forward(inp)                     # enqueued; the GPU is now busy for a forward's length
            ev_f = torch.cuda.Event()
            ev_f.record(main)
            if n + 1 < len(batches):
                nxt = assemble_on_side(n + 1)           # batch n + 1's inputs under batch n's forward
            if pending is not None:                     # batch n - 1's poses under batch n's forward (host polling included)
                results.append(pack(solve_on_side(*pending), pending[0], pending[1]))
            pending = (inp, out, ev_f)
        results.append(pack(solve_on_side(*pending), pending[0], pending[1]))
        main.wait_stream(s_pose)
        main.wait_stream(s_in)
        return results
