"""PointNet++ backbone: 4 set-abstraction + 2 feature-propagation layers
(reference models/backbone_module.py:21-139; layer hyper-parameters :38-75 are hard-coded there
and here).  Fills the `sa*_`, `fp2_` and `seed_` keys of `end_points`.
"""
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for _p in (_ROOT, os.path.join(_ROOT, "pointnet2")):
    if _p not in sys.path:
        sys.path.append(_p)

from pointnet2_modules import PointnetSAModuleVotes, PointnetFPModule  # noqa: E402
import pointnet2_utils  # noqa: E402

GROUP_AHEAD = True          # prefetched sampling chains also make the stages' centres / ball queries / row plans (see _launch_plan)

# (name, npoint, radius, nsample)
_SA_GEOMETRY = (("sa1", 2048, 0.2, 64), ("sa2", 1024, 0.4, 32), ("sa3", 512, 0.8, 16),
                ("sa4", 256, 1.2, 16))


class Pointnet2Backbone(nn.Module):
    """input (B, N, 3 + input_feature_dim) -> 1024 seeds with 288-channel features."""

    def __init__(self, input_feature_dim=0, width=2, depth=2):
        super().__init__()
        self.depth = depth
        self.width = width
        hidden = [128 * width] * depth
        in_ch = [input_feature_dim, 128 * width, 256 * width, 256 * width]
        mid = [[64 * width] * depth, hidden, hidden, hidden]
        out_ch = [128 * width, 256 * width, 256 * width, 256 * width]
        for (name, npoint, radius, nsample), ci, cm, co in zip(_SA_GEOMETRY, in_ch, mid, out_ch):
            setattr(self, name, PointnetSAModuleVotes(npoint=npoint, radius=radius, nsample=nsample,
                                                      mlp=[ci] + cm + [co], use_xyz=True,
                                                      normalize_xyz=True))
        for name_ in ("sa1", "sa2", "sa3", "sa4"):
            getattr(self, name_).omnipq_stage = name_      # label of the stage in per-stage timings (sa_fused.run)
        self.fp1 = PointnetFPModule(mlp=[256 * width + 256 * width, 256 * width, 256 * width])
        self.fp2 = PointnetFPModule(mlp=[256 * width + 256 * width, 256 * width, 288])

    # ---- sampling plan -----------------------------------------------------------------------
    # The four FPS passes of the backbone (and the one of FPSModule on the seeds) depend on the input
    # coordinates only, never on features: 40000 -> 2048 -> 1024 -> 512 -> 256 is a chain of
    # latency-bound kernels (thousands of dependent argmax rounds on a handful of CUs).  On a GPU
    # they run on a side stream: the sa1 pass can be started one batch ahead (`prefetch`), the
    # later ones overlap with the MLP of the previous layer.  Results are identical to calling
    # furthest_point_sample inside each SA layer (reference pointnet2_modules.py:233-236).
    _plan = None
    _side = None
    _extra = None
    # optional extra level of the plan: {"sa2": npoint} = also sample npoint of the sa2 centres (the model's
    # FPSModule on the seeds, pq_transformer.py:211-212, is such a coordinate-only sampling)
    plan_extra = None

    def __getstate__(self):
        # streams, events and the plan's index buffers are per-instance run-time state: copies and pickles of
        # the module (copy.deepcopy for an EMA teacher, torch.save of the whole model) start without them
        state = self.__dict__.copy()
        for k in ("_plan", "_side", "_plan_bufs", "_extra", "_pending"):
            state.pop(k, None)
        return state

    def _side_stream(self, device):
        if self._side is None or self._side.device != device:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    @staticmethod
    def _key(xyz_src):
        return (xyz_src.data_ptr(), xyz_src._version, tuple(xyz_src.shape))

    def _plan_buffers(self, pointcloud):
        """Persistent index buffers of the plan (one set per batch shape), so that a plan started inside a
        captured hipGraph lands at the same addresses on every replay."""
        key = (pointcloud.shape[0], str(pointcloud.device))
        bufs = self.__dict__.setdefault("_plan_bufs", {})
        if key not in bufs:
            bufs[key] = [torch.zeros((pointcloud.shape[0], getattr(self, n).npoint), device=pointcloud.device,
                                     dtype=torch.int32) for n in ("sa1", "sa2", "sa3", "sa4")]
        return bufs[key]

    # Grouping ahead of the stages (round 5).  The centres, the ball-query indices and the row plan of a stage depend on
    # coordinates only, like the sampling itself: a chain started by prefetch() appends them for all four stages (gather_xyz,
    # ball query, the plan's three small kernels: ~0.25 ms and ~14 launches that otherwise sit on the main stream between the
    # stages' GEMMs) and leaves them in ONE flat persistent buffer; forward() takes it with one copy and hands every stage its
    # views on the index tensor (`inds.omnipq_group`).  Same kernels, same inputs: identical results.  Switch: GROUP_AHEAD.

    def _group_layout(self, B, n_points):
        """-> (total int32 words, [(name, M, S, n_src, off_xyz, off_idx, off_plan | None)])"""
        import sa_fused
        r4 = lambda n: (n + 3) // 4 * 4
        off, levels, n_src = 0, [], n_points
        for name in ("sa1", "sa2", "sa3", "sa4"):
            mod = getattr(self, name)
            M, S = mod.npoint, mod.nsample
            P = B * M * S
            o_inds, off = off, off + r4(B * M)
            o_xyz, off = off, off + r4(B * M * 3)
            o_idx, off = off, off + r4(P)
            o_plan = None
            if sa_fused.plan_static_ok(S, P):
                o_plan, off = off, off + sum(sa_fused.plan_words(B, M, P))
            o_csr = None
            if name != "sa1":           # stages with features: the CSR "which grouped positions read point k" of their backward
                o_csr, off = off, off + r4(B * (n_src + 1)) + r4(P)
            levels.append((name, M, S, n_src, o_xyz, o_idx, o_plan, o_csr, o_inds))
            n_src = M
        # the two feature-propagation modules (fp1: sa3 <- sa4, fp2: sa2 <- sa3): 3-NN weights / indices and the CSR of their
        # backward, per (unknown level, known level)
        fps = []
        for u, k in ((2, 3), (1, 2)):
            n, m = levels[u][1], levels[k][1]
            o_w, off = off, off + r4(B * n * 3)
            o_i, off = off, off + r4(B * n * 3)
            o_o, off = off, off + r4(B * (m + 1))
            o_r, off = off, off + r4(B * n * 3)
            fps.append((u, k, n, m, o_w, o_i, o_o, o_r))
        # the sampled indices themselves (copied in at the end of the chain) and the extra level's
        extra = None
        if self.plan_extra:
            (ename, en), = list(self.plan_extra.items())[:1]
            extra = (ename, en, off)
            off += r4(B * en)
        return off, levels, fps, extra

    def _group_views(self, flat, B, n_points, layout=None):
        """flat int32 -> ([(centres (B,M,3) f32, idx (B,M,S) i32, plan state (flat i32) | None, (offsets, order) | None)] per stage,
        [(unknown level, known level, weight (B,n,3) f32, idx (B,n,3) i32, offsets (B,m+1), order (B,3n))] per FP module)"""
        import sa_fused
        # layout: the one the buffer was FILLED with (kept in the plan: a switch flipped between a chain's launch and the
        # forward that consumes it must not move the pieces)
        total, levels, fp_layout, extra_layout = layout if layout is not None else self._group_layout(B, n_points)
        out = []
        r4 = lambda n: (n + 3) // 4 * 4
        for name, M, S, n_src, o_xyz, o_idx, o_plan, o_csr, o_inds in levels:
            P = B * M * S
            cen = flat[o_xyz:o_xyz + B * M * 3].view(torch.float32).view(B, M, 3)
            idx = flat[o_idx:o_idx + P].view(B, M, S)
            st = None if o_plan is None else flat[o_plan:o_plan + sum(sa_fused.plan_words(B, M, P))]
            csr = None
            if o_csr is not None:
                o2 = o_csr + r4(B * (n_src + 1))
                csr = (flat[o_csr:o_csr + B * (n_src + 1)].view(B, n_src + 1), flat[o2:o2 + P].view(B, M * S))
            out.append((cen, idx, st, csr, flat[o_inds:o_inds + B * M].view(B, M)))
        fp_views = []
        for u, k, n, m, o_w, o_i, o_o, o_r in fp_layout:
            fp_views.append((u, k, flat[o_w:o_w + B * n * 3].view(torch.float32).view(B, n, 3),
                                   flat[o_i:o_i + B * n * 3].view(B, n, 3), flat[o_o:o_o + B * (m + 1)].view(B, m + 1),
                                   flat[o_r:o_r + B * n * 3].view(B, n * 3)))
        extra = None if extra_layout is None else \
            (extra_layout[0], flat[extra_layout[2]:extra_layout[2] + B * extra_layout[1]].view(B, extra_layout[1]))
        return out, fp_views, extra

    def _launch_plan(self, pointcloud, trusted=False, small=False, group=False):
        """FPS chain for `pointcloud` on the side stream -> {"key", "inds": [4 x (B,npoint) int32],
        "events": [4 x Event]}; the caller's stream has to wait on events[i] before using inds[i].
        group: also the stages' centres / ball queries / row plans (GROUP_AHEAD) -> plan["group"] = (flat buffer, event)."""
        main = torch.cuda.current_stream(pointcloud.device)
        side = self._side_stream(pointcloud.device)
        side.wait_stream(main)
        bufs = self._plan_buffers(pointcloud)
        plan = {"key": self._key(pointcloud), "inds": [], "events": [], "src": pointcloud, "trusted": trusted,
                "extra": None, "group": None}
        ext = pointnet2_utils._ext
        group = bool(group) and GROUP_AHEAD and hasattr(ext, "set_timing_sink")
        gflat, gviews, gfp, gsrc = None, None, None, []
        if group:
            B0, n0 = pointcloud.shape[0], pointcloud.shape[1]
            gkey = ("group", B0, n0, str(pointcloud.device))
            store = self.__dict__.setdefault("_plan_bufs", {})
            glayout = self._group_layout(B0, n0)
            if gkey not in store or store[gkey].numel() < glayout[0]:
                store[gkey] = torch.zeros((glayout[0],), device=pointcloud.device, dtype=torch.int32)
            gflat = store[gkey]
            gviews, gfp, gextra = self._group_views(gflat, B0, n0, glayout)
            if not torch.cuda.is_current_stream_capturing():
                gflat.record_stream(side)
        if not torch.cuda.is_current_stream_capturing():
            # The cloud and the persistent index buffers come from the caller's stream's pool but are read / written
            # by the sampling stream: if either dies while a plan is still running (a model dropped right after a
            # prefetch -- seen in the test-suite as FPS indices landing in the next test's fresh tensor), the
            # allocator must not hand the block out before the sampling stream is done with it.
            pointcloud.record_stream(side)
            for buf in bufs:
                buf.record_stream(side)
        with torch.cuda.stream(side), torch.no_grad():
            xyz = pointcloud[..., 0:3].contiguous()
            for li, name in enumerate(("sa1", "sa2", "sa3", "sa4")):
                npoint = getattr(self, name).npoint
                if hasattr(ext, "set_timing_sink"):          # the product binding: write in place
                    inds = ext.furthest_point_sampling(xyz, npoint, out=bufs[li], small_footprint=small)
                else:
                    inds = ext.furthest_point_sampling(xyz, npoint)
                ev = torch.cuda.Event()
                ev.record(side)
                plan["inds"].append(inds)
                plan["events"].append(ev)
                if group:
                    src = xyz.contiguous()
                    xyz = ext.gather_xyz(src, inds, out=gviews[li][0])
                    gsrc.append(src)
                elif name != "sa4":
                    if hasattr(ext, "gather_xyz"):
                        xyz = ext.gather_xyz(xyz.contiguous(), inds)
                    else:
                        xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds) \
                            .transpose(1, 2).contiguous()
                if self.plan_extra and name in self.plan_extra:
                    extra = self.__dict__.setdefault("_plan_bufs", {}).setdefault(
                        ("extra", name, pointcloud.shape[0], str(pointcloud.device)),
                        torch.zeros((pointcloud.shape[0], self.plan_extra[name]), device=pointcloud.device,
                                    dtype=torch.int32))
                    if not torch.cuda.is_current_stream_capturing():
                        extra.record_stream(side)
                    if hasattr(ext, "set_timing_sink"):
                        e_inds = ext.furthest_point_sampling(xyz, self.plan_extra[name], out=extra)
                    else:
                        e_inds = ext.furthest_point_sampling(xyz, self.plan_extra[name])
                    e_ev = torch.cuda.Event()
                    e_ev.record(side)
                    plan["extra"] = (name, e_inds, e_ev)
            if group:
                # behind every sampling level (their events fire first): ball query and row plan per stage
                import sa_fused
                planned = []
                for li, name in enumerate(("sa1", "sa2", "sa3", "sa4")):
                    mod = getattr(self, name)
                    cen, idx, st, csr, ginds = gviews[li]
                    ginds.copy_(plan["inds"][li])
                    with sa_fused._tagged("@sa", name):
                        ext.ball_query(cen, gsrc[li], mod.radius, mod.nsample, out=idx)
                        rp = sa_fused.make_row_plan(idx, idx.numel(), into=st) if st is not None else None
                        if csr is not None:
                            # in the row space the stage will run in: compact under a plan (a planned stage = training)
                            sa_fused.build_csr_ahead(idx, gsrc[li].shape[1], rp if mod.training else None, csr[0], csr[1])
                    planned.append(rp is not None and mod.training)
                for u, k, w, i3, offs, order in gfp:
                    ext.three_nn_weights(gviews[u][0], gviews[k][0], out=(w, i3))
                    sa_fused.build_csr_ahead(i3, gviews[k][0].shape[1], None, offs, order)
                if gextra is not None and plan["extra"] is not None and plan["extra"][0] == gextra[0]:
                    gextra[1].copy_(plan["extra"][1])
                g_ev = torch.cuda.Event()
                g_ev.record(side)
                plan["group"] = (gflat, g_ev, planned, glayout, sa_fused.PLAN_GROUP)
        return plan

    def prefetch(self, pointcloud, trusted=False, at_next_forward=False, footprint=None):
        """Start the sampling plan of a FUTURE batch now (e.g. while the current batch is in backward).
        A later forward() on the very same tensor picks the result up; any other input recomputes.
        trusted=True: the next forward() takes the plan whatever tensor it is given (the caller vouches
        that the contents match -- used when a captured graph feeds forward() from a static buffer).
        at_next_forward=True: do not start now but inside the NEXT forward(), right after it has taken (and copied) the plan
        it runs on -- the earliest point at which the plan's persistent index buffers may be overwritten.  The sampling
        chain (7 ms of dependent argmax rounds for 8 x 40 000 points) then has the whole step to hide under, forward
        included, instead of the backward pass only.
        footprint: "small" runs the 40 000-point level on 3 instead of 5 compute units per scene with ~40 % longer rounds
        (_ext.furthest_point_sampling(small_footprint=True): same indices) -- right when the chain ends well before the
        step it hides under, which is the default for at_next_forward; "fast" otherwise (e.g. two chains per step)."""
        if not pointcloud.is_cuda:
            return
        small = (footprint or ("small" if at_next_forward else "fast")) == "small"
        if at_next_forward:
            self._pending = (pointcloud, trusted, small, True)
        else:
            self._plan = self._launch_plan(pointcloud, trusted, small, True)

    def forget_plan(self):
        """Drop the HOST-side record of a plan in flight / pending (not the device work).  A captured step replays its
        sampling chains from the graph; the plan record its capture run (or an up-front `prefetch` before a replay) left behind
        is marked `trusted` and would hand a LATER eager forward() -- an evaluation pass between two training steps -- the
        indices of whatever batch the static buffers held last.  train_step.CapturedStep calls this after the capture and
        after every replay."""
        self._plan = None
        self._pending = None
        self._extra = None

    def join(self, device=None):
        """Make the current stream wait for everything queued on the sampling stream."""
        if self._side is not None:
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)

    def take_extra(self, name):
        """Indices of the plan's extra level `name` for the batch just run through forward(), or None."""
        extra, self._extra = self._extra, None
        if extra is None or extra[0] != name:
            return None
        if extra[2] is None:
            return extra[1]                 # forward() copied it out of the plan's buffer already
        torch.cuda.current_stream(extra[1].device).wait_event(extra[2])
        return extra[1].clone()             # the plan's buffer is reused by the next plan

    def _take_plan(self, pointcloud):
        if not pointcloud.is_cuda or os.environ.get("OMNIPQ_SAMPLING_PLAN", "1") == "0":
            return None
        plan = self._plan
        self._plan = None
        if plan is None or not (plan["trusted"] or plan["key"] == self._key(pointcloud)):
            plan = self._launch_plan(pointcloud)
        return plan

    def _break_up_pc(self, pc):
        xyz = pc[..., 0:3].contiguous()
        features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
        return xyz, features

    def forward(self, pointcloud, end_points=None):
        if not end_points:
            end_points = {}
        xyz, features = self._break_up_pc(pointcloud)
        plan = self._take_plan(pointcloud)

        self._extra = None
        if plan is not None and plan["extra"] is not None:
            self._extra = plan["extra"]
        taken = [None] * 4
        groups = [None] * 4
        if plan is not None and plan.get("group") is not None:
            # made ahead of the stages (GROUP_AHEAD): one copy of the chain's flat buffer, views per stage
            gflat, g_ev, planned, glayout, gs_then = plan["group"]
            torch.cuda.current_stream(pointcloud.device).wait_event(g_ev)
            gv, gfp, gex = self._group_views(gflat.clone(), pointcloud.shape[0], pointcloud.shape[1], glayout)
            import sa_fused
            if gs_then != sa_fused.PLAN_GROUP:        # plans (and the CSRs in their row space) of another group size: not these
                groups = [(g[0], g[1], None, None, False) for g in gv]
            else:
                groups = [g[:4] + (pl,) for g, pl in zip(gv, planned)]
            # the indices travel in the same copy: nothing else of this plan is read below
            taken = [g[4] for g in gv]
            if gex is not None and plan["extra"] is not None and plan["extra"][0] == gex[0]:
                self._extra = (gex[0], gex[1], None)
                plan = dict(plan, extra=None)
            plan = dict(plan, inds=None)
            for u, k, w, i3, offs, order in gfp:                         # the FP modules find theirs on the unknown centres
                groups[u][0].omnipq_nn = (groups[k][0], w, i3, (offs, order))
        pending, self._pending = getattr(self, "_pending", None), None
        if pending is not None:
            if plan is not None:
                # the next plan starts below and reuses this plan's buffers: copy everything it holds NOW
                cur = torch.cuda.current_stream(pointcloud.device)
                for li in range(4):
                    if plan["inds"] is None:
                        break                          # (they came with the group's copy)
                    cur.wait_event(plan["events"][li])
                    taken[li] = plan["inds"][li].clone()
                if plan["extra"] is not None:
                    name, e_inds, e_ev = plan["extra"]
                    cur.wait_event(e_ev)
                    self._extra = (name, e_inds.clone(), None)
                plan = None
            self._plan = self._launch_plan(*pending)
        for li, name in enumerate(("sa1", "sa2", "sa3", "sa4")):
            inds = taken[li]
            if plan is not None and plan["inds"] is not None:
                torch.cuda.current_stream(pointcloud.device).wait_event(plan["events"][li])
                inds = plan["inds"][li].clone()        # the plan's buffers are reused by the next plan
            if groups[li] is not None and inds is not None:
                inds.omnipq_group = groups[li]
            xyz, features, inds = getattr(self, name)(xyz, features, inds)
            if name in ("sa1", "sa2"):          # the reference records inds for these two only
                end_points[name + "_inds"] = inds
            end_points[name + "_xyz"] = xyz
            end_points[name + "_features"] = features

        features = self.fp1(end_points["sa3_xyz"], end_points["sa4_xyz"],
                            end_points["sa3_features"], end_points["sa4_features"])
        features = self.fp2(end_points["sa2_xyz"], end_points["sa3_xyz"],
                            end_points["sa2_features"], features)
        end_points["fp2_features"] = features
        end_points["fp2_xyz"] = end_points["sa2_xyz"]
        num_seed = end_points["fp2_xyz"].shape[1]
        # seeds are the first num_seed FPS picks of sa1, i.e. indices into the input cloud
        end_points["fp2_inds"] = end_points["sa1_inds"][:, 0:num_seed]
        end_points["seed_inds"] = end_points["fp2_inds"]
        end_points["seed_xyz"] = end_points["fp2_xyz"]
        end_points["seed_features"] = end_points["fp2_features"]
        return end_points
