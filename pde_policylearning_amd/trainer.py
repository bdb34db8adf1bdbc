"""Training-step counterpart of run_pde_observers.py:167-239 and the data-parallel
gradient exchange the reference lacks.

One process per GPU; parameters are replicated; the global batch is split on dim 0; the
loss is a SUM over samples (LpLoss(size_average=False), run_pde_observers.py:138), so the
matching collective is ONE all-reduce(SUM) of a flat gradient bucket with no averaging.
`torch.distributed` backend "nccl" is RCCL on ROCm (xGMI on MI355X nodes); "gloo" is used
by the CPU tests.
"""
import weakref

import torch
import torch.distributed as dist


class LpLoss(object):
    """Relative / absolute Lp loss with the reference semantics (libs/utilities3.py:295-337)."""

    def __init__(self, d=2, p=2, size_average=True, reduction=True):
        assert d > 0 and p > 0
        self.d, self.p, self.reduction, self.size_average = d, p, reduction, size_average

    def abs(self, x, y):
        n = x.size()[0]
        h = 1.0 / (x.size()[1] - 1.0)
        norms = (h ** (self.d / self.p)) * torch.norm(x.view(n, -1) - y.view(n, -1), self.p, 1)
        if self.reduction:
            return torch.mean(norms) if self.size_average else torch.sum(norms)
        return norms

    def rel(self, x, y):
        n = x.size()[0]
        diff = torch.norm(x.reshape(n, -1) - y.reshape(n, -1), self.p, 1)
        yn = torch.norm(y.reshape(n, -1), self.p, 1)
        if self.reduction:
            return torch.mean(diff / yn) if self.size_average else torch.sum(diff / yn)
        return diff / yn

    def __call__(self, x, y):
        return self.rel(x, y)


class FusedLpLoss(LpLoss):
    """LpLoss whose relative form (the one the training loops call) runs in the engine: decode of
    prediction and target, both norms and the gradient in two streaming passes over the fields
    (include/fnoengine.h fno_lploss_rel_*).  `decoder` = MeanStdDecoder or None.  GPU only."""

    def __init__(self, d=2, p=2, size_average=True, reduction=True, decoder=None):
        super().__init__(d, p, size_average, reduction)
        assert p == 2 and reduction, "the engine implements the reduced L2 form used by the training loops"
        self.decoder = decoder

    def rel(self, x, y):
        from . import functional as F
        dec = self.decoder
        mean = dec.mean.to(x.device) if dec is not None else None
        std = dec.std.to(x.device) if dec is not None else None
        return F.lp_loss_rel(x, y, mean, std, dec.eps if dec is not None else 0.0, self.size_average)


class MeanStdDecoder(object):
    """x * (std + eps) + mean  (NormalizerGivenMeanStd.cuda_decode, libs/utilities3.py:115-129),
    with the statistics placed on the model's device once."""

    def __init__(self, mean, std, eps=1e-5, device=None):
        self.mean = torch.as_tensor(mean, dtype=torch.float32, device=device)
        self.std = torch.as_tensor(std, dtype=torch.float32, device=device)
        self.eps = eps

    def decode(self, x):
        return x * (self.std + self.eps) + self.mean


class FullFieldObjective(object):
    """Loss of the FullFieldNSDataset branch of the observer loop (run_pde_observers.py:207-231):
    data term = LpLoss over the decoded target planes, physics term = pde_loss_weight * sum_b pde_loss(U_b, V_b, V_b with the
    predicted planes written in, W_b) on the channel-flow RHS kernels (libs/envs/control_env.ChannelFlowRHS).

    __call__(pred_raw, batch): pred_raw (B, P, X, Z, T) model output; batch = (v_field (B, T, P, X, Z) normalised targets,
    U (B, T, Nx, Ny+1, Nz), V (B, T, Nx, Ny, Nz), W).  The reference squeezes T (only T = 1 runs there); here T folds into
    the batch of the physics term."""

    def __init__(self, decoder, plane_indexs, env=None, pde_loss_weight=0.0, data_loss=None):
        self.decoder, self.plane_indexs = decoder, list(plane_indexs)
        self.env, self.weight = env, float(pde_loss_weight)
        self.data_loss = data_loss if data_loss is not None else FusedLpLoss(size_average=False)
        if self.weight > 0 and env is None:
            raise ValueError("pde_loss_weight > 0 needs the channel grid (ChannelFlowRHS)")
        self.last_terms = None
        self._plane_idx = None

    def __call__(self, pred_raw, batch):
        v_field, U, V, W = batch
        B = pred_raw.shape[0]
        pred = self.decoder.decode(pred_raw.permute(0, 4, 1, 2, 3))           # 'bpxzt -> btpxz', then x * (std + eps) + mean
        target = self.decoder.decode(v_field)
        loss = self.data_loss(pred.reshape(B, -1), target.reshape(B, -1))
        self.last_terms = (loss.detach(), None)
        if self.weight > 0:
            full = V.clone()
            # (the plane indices as a DEVICE tensor built once: a Python list would be turned into a host tensor and copied
            # over on every call - not allowed while a hipGraph is being captured)
            idx = self._plane_idx
            if idx is None or idx.device != full.device:
                idx = self._plane_idx = torch.tensor([i % full.shape[3] for i in self.plane_indexs], device=full.device)
            full.index_copy_(3, idx, pred.permute(0, 1, 3, 2, 4).to(full.dtype))                  # (B, T, X, P, Z)
            pde = self.env.pde_loss(U.flatten(0, 1), V.flatten(0, 1), full.flatten(0, 1), W.flatten(0, 1))
            self.last_terms = (self.last_terms[0], pde.detach())
            loss = loss + self.weight * pde
        return loss


class MultiStepLR(object):
    """torch.optim.lr_scheduler.MultiStepLR for FusedAdam (train_pino.py:201-203): lr = base * gamma ** #(milestones <= epoch)."""

    def __init__(self, optimizer, milestones, gamma=0.1, last_epoch=0):
        self.optimizer, self.milestones, self.gamma = optimizer, sorted(milestones), gamma
        self.base_lr, self.last_epoch = optimizer.lr, last_epoch
        self._apply()

    def _apply(self):
        self.optimizer.lr = self.base_lr * self.gamma ** sum(1 for m in self.milestones if m <= self.last_epoch)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [self.optimizer.lr]

    def state_dict(self):
        import collections
        return dict(milestones=collections.Counter(self.milestones), gamma=self.gamma, base_lrs=[self.base_lr],
                    last_epoch=self.last_epoch, _step_count=self.last_epoch + 1, _last_lr=[self.optimizer.lr])

    def load_state_dict(self, sd):
        """This class's own state or torch.optim.lr_scheduler.MultiStepLR's (milestones as a Counter, base_lrs list)."""
        ms = sd["milestones"]
        self.milestones = sorted(ms.elements()) if hasattr(ms, "elements") else sorted(ms)
        self.gamma, self.last_epoch = sd["gamma"], sd["last_epoch"]
        self.base_lr = sd["base_lr"] if "base_lr" in sd else sd["base_lrs"][0]
        self._apply()


class PinoObjective(object):
    """Loss of the PINO fine-tuning step (train_pino.py:86-106): xy_weight * LpLoss(out, u) + f_weight * loss_f +
    ic_weight * loss_ic, the last two from the Navier-Stokes vorticity residual of the model output (engine kernels,
    libs.envs.diff_control_env.Channelflow_PINO_loss).  The reference evaluates the model twice when both the data and the
    PDE term are on (same input, same weights: identical outputs); here it is evaluated once.

    __call__(out, (u, a_in, re)): out (B, S, S, T, 1), u (B, S, S, T), a_in (B, S, S, T, 4) whose last channel at t = 0 is
    the initial condition, re (B,)."""

    def __init__(self, forcing, t_duration, ic_weight, f_weight, xy_weight, scale=1.0):
        self.forcing, self.t_duration = forcing, t_duration
        self.ic_weight, self.f_weight, self.xy_weight, self.scale = ic_weight, f_weight, xy_weight, scale
        self.lploss = FusedLpLoss(size_average=True)
        self.last_terms = {}

    def __call__(self, out, batch):
        from .libs.envs.diff_control_env import Channelflow_PINO_loss
        u, a_in, re = batch
        loss = 0.0
        self.last_terms = {}
        if self.xy_weight > 0:
            data_loss = self.lploss(out.reshape(u.shape), u)
            self.last_terms["data"] = data_loss.detach()
            loss = loss + self.xy_weight * data_loss
        if self.f_weight != 0.0:
            loss_ic, loss_f = Channelflow_PINO_loss(out, a_in[:, :, :, 0, -1], self.forcing, 1.0 / re.float(), self.t_duration)
            self.last_terms["IC"], self.last_terms["PDE"] = loss_ic.detach(), loss_f.detach()
            loss = loss + self.f_weight * loss_f + self.ic_weight * loss_ic
        if not torch.is_tensor(loss):
            raise ValueError("PinoObjective: every loss weight is zero")
        return loss * self.scale if self.scale != 1.0 else loss


class FlatGradBucket(object):
    """All parameter gradients live in ONE contiguous buffer (p.grad are views into it), so
    the data-parallel exchange is a single all-reduce(SUM) - sized for xGMI: one 9.6 MB message
    for FNO2d(12,12,64) instead of ~30 small ones."""

    def __init__(self, params, process_group=None, direct_module=None, zero_all=False):
        """direct_module: a model whose engine calls may WRITE their gradients into the bucket (no autograd
        accumulation kernels, no zeroing); valid when every such parameter is used by exactly one engine call per
        step, as in the reference training steps.  A fused engine FNO covers all of its parameters; modules exposing
        `direct_grad_params()` (the PINO observers' spectral convolutions: > 99 % of their parameter bytes) cover those,
        and the remaining parameters are laid out at the tail of the bucket, which is the only part zero() clears.
        zero_all: clear the whole bucket anyway - for models that fall back to autograd accumulation when a parameter
        is used more than once in a step (RNO2d over several time steps: functional.single_use)."""
        self.direct_module = direct_module
        self.params = [p for p in params if p.requires_grad]
        self.user_order = list(self.params)       # the order an optimizer built on the same iterable would number them in
        self.group = process_group
        direct_ids = set()
        if direct_module is not None:
            for m in direct_module.modules():
                if hasattr(m, "fused_supported"):
                    # a new bucket for the model supersedes an overlapped one (for_fno) that was built on it before: the fused
                    # module must not keep handing the old bucket to the engine (an extra async all-reduce of its late
                    # segment per step, never waited for: ADVICE r05).  for_fno() installs itself again behind this constructor.
                    old = getattr(m, "_grad_overlap", None)
                    if old is not None:
                        old._drop_inflight()
                        m._grad_overlap = None
                    m._direct_grads = True
                    direct_ids.update(id(p) for p in m.parameters())
                elif hasattr(m, "direct_grad_params"):
                    m._direct_grads = True
                    direct_ids.update(id(p) for p in m.direct_grad_params())
            if any(id(p) not in direct_ids for p in self.params):
                self.params = [p for p in self.params if id(p) in direct_ids] + [p for p in self.params if id(p) not in direct_ids]
        n = sum(self._nfloat(p) for p in self.params)
        self._zero_from = sum(self._nfloat(p) for p in self.params if id(p) in direct_ids) if direct_module is not None else 0
        if zero_all:
            self._zero_from = 0
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, v in zip(self.params, self.views(self.flat)):
            p.grad = v

    @staticmethod
    def _nfloat(p):
        """fp32 slots of a parameter: complex64 parameters (libs/models/pino_models/basics.py:74-77) take two."""
        return p.numel() * (2 if p.is_complex() else 1)

    def views(self, flat):
        """Per-parameter views of a flat fp32 buffer laid out like the bucket (complex parameters as
        complex views of interleaved pairs, which is also how torch.optim.Adam treats them)."""
        out, off = [], 0
        for p in self.params:
            k = self._nfloat(p)
            seg = flat[off:off + k]
            if p.is_contiguous():
                out.append(torch.view_as_complex(seg.view(*p.shape, 2)) if p.is_complex() else seg.view_as(p))
            else:
                # a dense parameter in another memory order (plane-major dialect-C weights, functional.plane_major): the
                # same order inside the bucket, so that the engine, the optimizer and the exchange see one layout
                base = torch.view_as_complex(seg.view(-1, 2)) if p.is_complex() else seg
                expect = 1
                for size, stride in sorted(zip(p.shape, p.stride()), key=lambda t: t[1]):
                    if size > 1 and stride != expect:
                        raise RuntimeError("FlatGradBucket: parameters must be dense (a permutation of a contiguous tensor)")
                    expect *= size
                out.append(base.as_strided(p.shape, p.stride()))
            off += k
        return out

    def zero(self):
        if self._zero_from < self.flat.numel():
            self.flat[self._zero_from:].zero_()

    def check_views(self, skip=()):
        """autograd accumulates in place into an existing .grad; re-attach if something replaced it (`skip`: ids of
        parameters whose bucket region an asynchronous all-reduce is still reducing)."""
        for p, v in zip(self.params, self.views(self.flat)):
            if id(p) in skip:
                continue
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    def all_reduce(self):
        """SUM over ranks, no division (sum-reduced loss).  With an overlapped bucket (for_fno) the late
        layers' part is already in flight: wait for it and exchange the rest."""
        self._check_live_extents()
        if not self._collective_needed():
            self.check_views()
            self._inflight = None
            for sg in getattr(self, "_segments", None) or []:
                sg["work"], sg["pending"] = None, set(sg["ids"])
            return
        ev0 = self._event()
        if getattr(self, "_segments", None) is not None:
            # gradients of segments already on the wire must not be touched: re-attach views only where nothing is in flight
            self.check_views(skip={i for sg in self._segments if sg["work"] is not None for i in sg["ids"]})
            self.wire_bytes = self.wire_bytes if any(sg["work"] is not None for sg in self._segments) else 0
            self._finish_segments()
            self.wire_bytes_last, self.wire_bytes = self.wire_bytes, 0
        else:
            self.wire_bytes_last = 4 * self.flat.numel()
            work, self._inflight = getattr(self, "_inflight", None), None
            if work is not None:
                late = {id(p) for p in self.params[:self._late_count]}
                self.check_views(skip=late)
                dist.all_reduce(self.flat[self._late_numel:], op=dist.ReduceOp.SUM, group=self.group)
                work.wait()
            else:
                self.check_views()
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if ev0 is not None:
            first, self._ev_first = getattr(self, "_ev_first", None) or ev0, None
            self.exchange_events.append((first, ev0, self._event()))

    # ---- timing of the exchange (bench.py, N > 1): HIP events on the compute stream ---------------------------------
    def time_exchange(self, on=True):
        """Record, per step, events at the first collective's launch, at the start of the post-backward wait and at its end
        (exchange_ms() turns them into total / exposed milliseconds)."""
        self.exchange_events = [] if on else None
        self._ev_first = None

    def _event(self):
        if getattr(self, "exchange_events", None) is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _mark_first_launch(self):
        if getattr(self, "exchange_events", None) is not None and getattr(self, "_ev_first", None) is None:
            self._ev_first = self._event()

    def exchange_ms(self):
        """(total, exposed) milliseconds per step, averaged over the recorded steps: total = first collective enqueued ->
        every collective complete on the compute stream; exposed = the part after the backward pass's last kernel (what
        the overlap did not hide).  Call after a device synchronise."""
        ev = getattr(self, "exchange_events", None)
        if not ev:
            return None
        tot = sum(a.elapsed_time(c) for a, _, c in ev) / len(ev)
        exp = sum(b.elapsed_time(c) for _, b, c in ev) / len(ev)
        return tot, exp

    def _check_live_extents(self):
        """The live last-dim extents of dialect-C weights were recorded once (enable_dp_exchange); a later batch with a
        longer last dimension puts gradient into slices that are never packed, and the ranks would drift apart silently."""
        for m, k in getattr(self, "_live_modules", ()):
            now = m.__dict__.get("_live_last")
            if now is not None and now > k:
                raise RuntimeError(f"FlatGradBucket: {type(m).__name__} now has {now} live last-dim modes, the gradient "
                                   f"exchange was planned for {k} (longer last dimension than the sample batch): call "
                                   "enable_dp_exchange again with a sample of the new shape")

    # ---- segmented exchange: any model, segments go on the wire as their gradients complete ---------------------------
    def enable_segmented_exchange(self, min_bytes=16 << 20, live_last=None):
        """Cut the bucket into segments of consecutive parameters (>= min_bytes each, big parameters alone) and start an
        ASYNC all-reduce of a segment as soon as the backward pass has produced all of its gradients: autograd's
        post-accumulate hooks for ordinary parameters, functional.DIRECT_WRITE_HOOKS for gradients the engine writes in
        place.  all_reduce() then waits for the segments in flight and exchanges what is left.  Same sums as the single
        exchange: every element is reduced exactly once.

        live_last: {parameter: k} for dialect-C spectral weights (.., m3) of which only the last-dim slice [..., :k] can
        receive a gradient (libs/models/pino_models/basics.py:119-139 zero-pads the spectrum to modes3: with Nz/2+1 <
        modes3 the rest of the weight never sees data).  Their dead slices are exactly zero on every rank, so only the live
        slice is packed and exchanged (PINObserverFullField at T = 1: 1/12 of 906 MB)."""
        from . import functional as F
        live_last = live_last or {}
        self._live = {id(p): int(k) for p, k in live_last.items() if int(k) < p.shape[-1]}
        segs, cur, cur_bytes, off = [], None, 0, 0
        for p in self.params:
            n = self._nfloat(p)
            sliced = id(p) in self._live
            if cur is None or cur_bytes >= min_bytes or sliced != cur["sliced"]:
                cur = dict(start=off, end=off, params=[], sliced=sliced)
                segs.append(cur)
                cur_bytes = 0
            cur["params"].append(p)
            cur["end"] = off + n
            cur_bytes += 4 * n
            off += n
        for sg in segs:
            sg["ids"] = {id(p) for p in sg["params"]}
            sg["pending"], sg["work"], sg["pack"] = set(sg["ids"]), None, None
            if sg["sliced"]:
                nlive = sum(self._nfloat(p) // p.shape[-1] * self._live[id(p)] for p in sg["params"])
                sg["pack"] = torch.zeros(nlive, dtype=torch.float32, device=self.flat.device)
        self._segments = segs
        self._seg_of = {i: sg for sg in segs for i in sg["ids"]}
        self._by_ptr = {}
        self.wire_bytes = 0
        self.close()
        import weakref
        me = weakref.ref(self)

        def arrived(p):              # the parameters must not keep the bucket (and its buffers) alive
            b = me()
            if b is not None:
                b._arrived(p)
        for p, v in zip(self.params, self.views(self.flat)):
            self._by_ptr[(torch.view_as_real(v) if v.is_complex() else v).data_ptr()] = p
            self._hook_handles.append(p.register_post_accumulate_grad_hook(arrived))
        self._direct_ref = weakref.WeakMethod(self._direct_written)
        F.DIRECT_WRITE_HOOKS.append(self._direct_ref)
        return self

    def _drop_inflight(self):
        """wait for an asynchronous all-reduce of the late segment that nobody collected (for_fno)"""
        work, self._inflight = getattr(self, "_inflight", None), None
        if work is not None:
            work.wait()

    def close(self):
        """Detach from the parameters' gradient hooks, from functional.DIRECT_WRITE_HOOKS and - an overlapped bucket
        (for_fno) - from the fused module it installed itself on (a bucket that is dropped without close() is released
        as well: the registrations hold weak references only)."""
        from . import functional as F
        self._drop_inflight()
        fno = getattr(self, "_fno_ref", None)
        fno = fno() if fno is not None else None
        if fno is not None and getattr(fno, "_grad_overlap", None) is self:
            fno._grad_overlap = None
        for h in getattr(self, "_hook_handles", []):
            h.remove()
        self._hook_handles = []
        ref = getattr(self, "_direct_ref", None)
        if ref is not None and ref in F.DIRECT_WRITE_HOOKS:
            F.DIRECT_WRITE_HOOKS.remove(ref)
        self._direct_ref = None

    def planned_wire_bytes(self):
        """bytes one rank puts on the wire per step: the whole bucket, or with a segmented exchange the segments' buffers
        (live slices only for the dialect-C weights)"""
        segs = getattr(self, "_segments", None)
        if segs is None:
            return 4 * self.flat.numel()
        return sum(4 * (sg["pack"].numel() if sg["sliced"] else sg["end"] - sg["start"]) for sg in segs)

    def _direct_written(self, tensors):
        for t in tensors:
            p = self._by_ptr.get(t.data_ptr())
            if p is not None:
                self._arrived(p)

    def _arrived(self, p):
        from . import functional as F
        sg = self._seg_of.get(id(p))
        if sg is None or sg["work"] is not None:
            return
        if not getattr(self, "single_use_step", F.LAST_FORWARD_SINGLE_USE[0]):
            return      # parameters used several times per step (RNO2d over T > 1 steps) accumulate several times: the
                        # first arrival is not the last - all_reduce() exchanges every segment after the backward pass
        sg["pending"].discard(id(p))
        if not sg["pending"] and self._collective_needed():
            self._launch(sg, async_op=True)

    def _live_views(self, sg):
        """(live slice of the gradient, its place in the pack buffer) for every parameter of a sliced segment"""
        out, off = [], 0
        gviews = {id(p): v for p, v in zip(self.params, self.views(self.flat))}
        for p in sg["params"]:
            g = gviews[id(p)]
            g = torch.view_as_real(g) if g.is_complex() else g.unsqueeze(-1)
            k = self._live[id(p)]
            live = g[..., :k, :]
            out.append((live, sg["pack"][off:off + live.numel()].view(live.shape)))
            off += live.numel()
        return out

    def _launch(self, sg, async_op):
        self._mark_first_launch()
        if sg["sliced"]:
            for live, slot in self._live_views(sg):
                slot.copy_(live)
            buf = sg["pack"]
        else:
            buf = self.flat[sg["start"]:sg["end"]]
        self.wire_bytes += 4 * buf.numel()
        sg["work"] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op) or True

    def _finish_segments(self):
        for sg in self._segments:
            if sg["work"] is None:
                self._launch(sg, async_op=False)
        for sg in self._segments:
            if sg["work"] is not True and sg["work"] is not None:
                sg["work"].wait()
            if sg["sliced"]:
                for live, slot in self._live_views(sg):
                    live.copy_(slot)
            sg["work"], sg["pending"] = None, set(sg["ids"])

    # ---- overlap of the gradient exchange with the backward pass (engine FNO only) -------------------------
    @classmethod
    def for_fno(cls, model, split_layer=1, process_group=None):
        """Bucket laid out for comm / compute overlap: [projection | blocks L-1 .. split_layer | rest].  The engine
        differentiates the late layers first (fno_model_backward_part) and calls late_gradients_ready(), which starts
        an ASYNC all-reduce of the first segment; the blocks below `split_layer`, the lifting and the shared bias
        tensor follow in all_reduce().  Same sums as the single exchange (each gradient element is reduced once)."""
        fno = next(m for m in model.modules() if hasattr(m, "fused_supported"))
        L = fno.n_layers
        nc = 2 ** (fno.n_dim - 1)
        blocks = fno.fno_blocks
        late = list(fno.projection.parameters())
        for l in range(L - 1, split_layer - 1, -1):
            late += [blocks.fno_skips[l].weight] + [blocks.convs.weight[nc * l + c].tensor for c in range(nc)]
        late_ids = {id(p) for p in late}
        rest = [p for p in model.parameters() if id(p) not in late_ids and p.requires_grad]
        bucket = cls([p for p in late if p.requires_grad] + rest, process_group=process_group, direct_module=model)
        bucket._late_numel = sum(cls._nfloat(p) for p in late if p.requires_grad)
        bucket._late_count = sum(1 for p in late if p.requires_grad)
        bucket._inflight = None
        bucket.split_layer = split_layer
        bucket._fno_ref = weakref.ref(fno)
        fno._grad_overlap = bucket
        return bucket

    def _collective_needed(self):
        """More than one rank (or `force_collective`, used by the single-rank GPU test of the async path)."""
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or getattr(self, "force_collective", False)

    def late_gradients_ready(self):
        if self._collective_needed():
            self._mark_first_launch()
            self._inflight = dist.all_reduce(self.flat[:self._late_numel], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True)


def enable_dp_exchange(bucket, model, sample_inputs=None, min_bytes=16 << 20):
    """Segmented, overlapped gradient exchange for any model (the fused FNO has its own: FlatGradBucket.for_fno), with the
    dead last-dim slices of dialect-C weights left off the wire.  The live extents depend on the input's last dimension, so
    one forward pass under no_grad on `sample_inputs` (any batch size) records them first."""
    if sample_inputs is not None:
        was = model.training
        model.eval()
        with torch.no_grad():
            model(*sample_inputs)
        model.train(was)
    bucket._live_modules = [(m, int(m.__dict__["_live_last"])) for m in model.modules()
                            if m.__dict__.get("_live_last") is not None and hasattr(m, "modes3")]
    return bucket.enable_segmented_exchange(min_bytes=min_bytes, live_last=live_last_of(model))


def live_last_of(model):
    """{weight: live last-dim extent} of the dialect-C 3-D spectral convolutions in `model` after a forward pass (the extent
    depends on the input's last dimension): the argument of FlatGradBucket.enable_segmented_exchange(live_last=...)."""
    out = {}
    for m in model.modules():
        k = m.__dict__.get("_live_last")
        if k is not None and hasattr(m, "modes3"):
            for w in (m.weights1, m.weights2, m.weights3, m.weights4):
                out[w] = int(k)
    return out


class _DeadSliceGuard(object):
    """What FusedAdam leaves on a model whose dead weight slices it skips: called by a dialect-C spectral convolution before
    it reads more last-dim slices than the plan holds live, and (before_state_dict) by the model's state_dict().  Holds weak
    references only; pickles / deep-copies as a detached no-op (torch.save(model) of run_pde_observers.py:307-315 keeps
    working - call optimizer.sync_dead_slices() first, as train_observer does)."""

    def __init__(self, opt=None, model=None):
        import weakref
        self._opt = weakref.ref(opt) if opt is not None else (lambda: None)
        self._model = weakref.ref(model) if model is not None else (lambda: None)

    def __call__(self, module, k3):
        opt, root = self._opt(), self._model()
        if opt is not None and root is not None and opt._runs is not None and k3 > module.__dict__.get("_dead_slice_k", k3):
            module._live_last = k3
            opt.skip_dead_slices(root)

    def before_state_dict(self, *args, **kwargs):
        opt = self._opt()
        if opt is not None:
            opt.sync_dead_slices()

    def __reduce__(self):
        return (_DeadSliceGuard, ())


class FusedAdam(object):
    """torch.optim.Adam semantics (run_pde_observers.py:134: lr, weight_decay, default betas / eps) as ONE
    kernel over a flat parameter bucket.  Parameters are re-pointed at views of one contiguous buffer
    laid out exactly like the FlatGradBucket, so the step reads the all-reduced gradient bucket directly.

    Dead last-dim slices (skip_dead_slices): a dialect-C 3-D spectral weight (.., modes3) only ever sees data in
    [..., :k], k = min(Nz/2+1, modes3) (libs/models/pino_models/basics.py:119-139; PINObserverFullField at T = 1: 1/12 of
    906 MB).  The gradient of the rest is exactly zero, so Adam's effect on it is a recurrence on (p, m, v) alone
    (g = weight_decay * p): those elements are not stepped - the step touches the live slices (moments kept compact) and the
    rest of the bucket - and are REPLAYED through the skipped steps in one kernel, bit-identical to stepping them every
    time, whenever somebody needs them: state_dict() / load_state_dict() here, the model's state_dict() (pre-hook), a
    forward pass with a longer last dimension (the modules call back before they read the weights), sync_dead_slices().
    Reading a weight's `.data` directly in between shows the dead slice as of the last replay - hence opt-in
    (skip_dead_slices=True: train_observer's full-field loop, bench.py).  The live extents are taken from the bucket's
    direct_module at the first step (after a forward pass has recorded them)."""

    def __init__(self, bucket, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False,
                 skip_dead_slices=False):
        """capturable: keep the step count on the device (fno_adam_step_dev) so that a captured hipGraph of
        the training step (GraphedTrainStep) can be replayed.
        skip_dead_slices: see the class docstring - opt-in, because code that reads `parameter.data` directly between steps
        (instead of state_dict()) would see the dead slices as of the last replay."""
        self.bucket = bucket
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.capturable = capturable
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=bucket.flat.device) if capturable else None
        self.scratch = torch.zeros(2, dtype=torch.float32, device=bucket.flat.device) if capturable else None
        self.flat_param = torch.empty_like(bucket.flat)
        with torch.no_grad():
            for p, v in zip(bucket.params, bucket.views(self.flat_param)):
                v.copy_(p.data)
                p.data = v
        self.exp_avg = torch.zeros_like(bucket.flat)
        self.exp_avg_sq = torch.zeros_like(bucket.flat)
        self.step_count = 0
        self._skip_dead = bool(skip_dead_slices)
        self._runs = None            # lazy layout: [("dense", off, n, coff) | ("rows", off, rows, row_len, live_len, coff)]
        self._dead = {}              # rows-run index -> [dead exp_avg, dead exp_avg_sq] (compact) or None (all zero)
        self._dead_step = 0          # the step count the dead slices (parameters and moments) are current at
        self._hp_log = []            # [(first step, lr, beta1, beta2, eps, weight_decay)]: what the skipped steps used
        self._guarded, self._sd_hook = [], None

    # ---- dead slices -----------------------------------------------------------------------------------------------
    def _hyper(self):
        return (float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay))

    def _current_step(self):
        return int(self.step_dev.item()) if self.capturable else self.step_count

    def skip_dead_slices(self, model=None, live_last=None):
        """Plan the step around the dead last-dim slices of `model`'s dialect-C weights (default: the bucket's
        direct_module; live extents as recorded by its last forward pass, or `live_last` = {parameter: k}).  Returns
        whether anything is skipped.  Called by step() on its own the first time; call it again after a change."""
        model = model if model is not None else self.bucket.direct_module
        if live_last is None:
            live_last = {}
            for m in (model.modules() if model is not None else ()):
                k = m.__dict__.get("_live_last")
                if k is not None and hasattr(m, "modes3"):
                    k = max(int(k), int(m.__dict__.get("_dead_slice_k", 0)))      # a plan never shrinks
                    m._live_last = k
                    live_last.update({w: k for w in (m.weights1, m.weights2, m.weights3, m.weights4)})
        live = {id(p): int(k) for p, k in live_last.items() if p.is_complex() and 0 < int(k) < p.shape[-1]}
        self.sync_dead_slices()
        full = self._full_moments() if self._runs is not None else (self.exp_avg, self.exp_avg_sq)
        runs, off, coff, ok = [], 0, 0, bool(live)
        for p in self.bucket.params:
            n = self.bucket._nfloat(p)
            k = live.get(id(p))
            if k is not None:
                from . import functional as F
                if F.plane_major(p):         # last dim outermost in memory: the live planes are a prefix of the tensor
                    rows, row_len, live_len = 1, n, n // p.shape[-1] * k
                elif p.is_contiguous():      # last dim innermost: the live part is the head of every row of 2 * modes3 floats
                    row_len, live_len = 2 * p.shape[-1], 2 * k
                    rows = n // row_len
                else:
                    ok = False
                    rows, row_len, live_len = 1, n, n
                ok = ok and off % 2 == 0
                last = runs[-1] if runs else None
                if last and last[0] == "rows" and last[3:5] == (row_len, live_len):
                    runs[-1] = ("rows", last[1], last[2] + rows, row_len, live_len, last[5])
                else:
                    coff += -coff % 4
                    runs.append(("rows", off, rows, row_len, live_len, coff))
                coff += rows * live_len
            else:
                last = runs[-1] if runs else None
                if last and last[0] == "dense":
                    runs[-1] = ("dense", last[1], last[2] + n, last[3])
                else:
                    coff += -coff % 4
                    ok = ok and off % 4 == 0
                    runs.append(("dense", off, n, coff))
                coff += n
            off += n
        self._release_guards()
        # the moment buffers are about to be replaced: a hipGraph captured on the old ones (GraphedTrainStep) must not replay
        self.plan_generation = getattr(self, "plan_generation", 0) + 1
        if not ok:
            if self._runs is not None:
                self._runs, self._dead = None, {}
                self.exp_avg, self.exp_avg_sq = full
            return False
        self._runs, self._dead = runs, {}
        self.exp_avg = torch.zeros(coff, dtype=torch.float32, device=self.flat_param.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self._scatter_full(*full)
        self._dead_step = self._current_step()
        self._hp_log = []
        if model is not None:
            import weakref
            guard = _DeadSliceGuard(self, model)
            for m in model.modules():
                if m.__dict__.get("_live_last") is not None and hasattr(m, "modes3"):
                    m._dead_slice_k = int(m._live_last)
                    m._dead_slice_guard = guard
                    self._guarded.append(weakref.ref(m))
            if hasattr(model, "register_state_dict_pre_hook"):
                self._sd_hook = model.register_state_dict_pre_hook(guard.before_state_dict)
        return True

    def _release_guards(self):
        for r in self._guarded:
            m = r()
            if m is not None:
                m.__dict__.pop("_dead_slice_guard", None)
                m.__dict__.pop("_dead_slice_k", None)
        self._guarded = []
        if self._sd_hook is not None:
            self._sd_hook.remove()
            self._sd_hook = None

    def close(self):
        """Bring the dead slices up to date and drop the hooks on the model."""
        self.sync_dead_slices()
        self._release_guards()

    def sync_dead_slices(self):
        """Replay the steps the dead slices have skipped (parameters and moments), so that every parameter holds the value
        torch.optim.Adam would have left in it.  Cheap when nothing is pending."""
        if self._runs is None:
            return
        from . import functional as F
        t0, t = self._dead_step, self._current_step()
        if t <= t0:
            return
        log = self._hp_log or [(t0 + 1,) + self._hyper()]
        for i, seg in enumerate(log):
            first = max(seg[0], t0 + 1)
            last = (log[i + 1][0] - 1) if i + 1 < len(log) else t
            if last < first:
                continue
            lr, b1, b2, eps, wd = seg[1:]
            scal = None
            for ri, run in enumerate(self._runs):
                if run[0] != "rows":
                    continue
                dead = self._dead.get(ri)
                if wd == 0.0 and dead is None:
                    continue      # zero gradient, zero moments: Adam leaves the element where it is
                if scal is None:
                    scal = F.adam_replay_scalars(first, last - first + 1, lr, (b1, b2), self.flat_param.device,
                                                 on_device=self.capturable)
                _, off, rows, row_len, live_len, _ = run
                zero = dead is None
                if zero:
                    dead = [torch.empty(rows * (row_len - live_len), dtype=torch.float32, device=self.flat_param.device)
                            for _ in range(2)]
                    self._dead[ri] = dead
                F.adam_replay_dead(rows, row_len, live_len, self.flat_param[off:off + rows * row_len], dead[0], dead[1], zero,
                                   scal, (b1, b2), eps, wd)
        self._dead_step = t
        self._hp_log = []

    def _full_moments(self):
        """(exp_avg, exp_avg_sq) in the bucket's full layout (new tensors) from the compact live + dead representation."""
        out = []
        for which, comp in ((0, self.exp_avg), (1, self.exp_avg_sq)):
            full = torch.zeros_like(self.flat_param)
            for ri, run in enumerate(self._runs):
                if run[0] == "dense":
                    _, off, n, coff = run
                    full[off:off + n] = comp[coff:coff + n]
                else:
                    _, off, rows, row_len, live_len, coff = run
                    v = full[off:off + rows * row_len].view(rows, row_len)
                    v[:, :live_len] = comp[coff:coff + rows * live_len].view(rows, live_len)
                    if self._dead.get(ri) is not None:
                        v[:, live_len:] = self._dead[ri][which].view(rows, row_len - live_len)
            out.append(full)
        return tuple(out)

    def _scatter_full(self, m_full, v_full):
        """Take moments given in the bucket's full layout into the current representation."""
        if self._runs is None:
            self.exp_avg.copy_(m_full); self.exp_avg_sq.copy_(v_full)
            return
        self._dead = {}
        for ri, run in enumerate(self._runs):
            if run[0] == "dense":
                _, off, n, coff = run
                self.exp_avg[coff:coff + n] = m_full[off:off + n]
                self.exp_avg_sq[coff:coff + n] = v_full[off:off + n]
            else:
                _, off, rows, row_len, live_len, coff = run
                mv = m_full[off:off + rows * row_len].view(rows, row_len)
                vv = v_full[off:off + rows * row_len].view(rows, row_len)
                self.exp_avg[coff:coff + rows * live_len].view(rows, live_len).copy_(mv[:, :live_len])
                self.exp_avg_sq[coff:coff + rows * live_len].view(rows, live_len).copy_(vv[:, :live_len])
                if bool(mv[:, live_len:].any()) or bool(vv[:, live_len:].any()):
                    self._dead[ri] = [mv[:, live_len:].contiguous().view(-1), vv[:, live_len:].contiguous().view(-1)]

    # ---- torch.optim surface ----------------------------------------------------------------------------------------
    def zero_grad(self, set_to_none=False):
        self.bucket.zero()

    def step(self):
        from . import functional as F
        self.bucket.check_views()
        if self._skip_dead and self._runs is None and self.bucket.direct_module is not None:
            self._skip_dead = False      # one attempt; skip_dead_slices() can be called again by hand
            self.skip_dead_slices()
        self.step_count += 1            # host mirror; under graph replay the device counter is authoritative
        if self._runs is None:
            F.adam_step(self.flat_param, self.bucket.flat, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                        self.betas, self.eps, self.weight_decay, step_counter=self.step_dev, scratch=self.scratch)
            return
        hp = self._hyper()
        if not self._hp_log:
            self._hp_log.append((self.step_count,) + hp)
        elif self._hp_log[-1][1:] != hp:
            if self.capturable:
                # the host's step count is only a mirror there (graph replays do not advance it): close the stretch that
                # ran on the old hyperparameters at the DEVICE's count, then start a new one
                self.step_count -= 1
                self.sync_dead_slices()
                self.step_count += 1
                self._hp_log = [(self._dead_step + 1,) + hp]
            else:
                self._hp_log.append((self.step_count,) + hp)
        F.adam_step_runs(self._runs, self.flat_param, self.bucket.flat, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                         self.betas, self.eps, self.weight_decay, step_counter=self.step_dev, scratch=self.scratch)

    def state_dict(self):
        """torch.optim.Adam's layout ({'state': {i: step / exp_avg / exp_avg_sq}, 'param_groups': [...]}, parameters numbered
        in the order the bucket was given them), so that train_pino's {'model', 'optim', 'scheduler'} checkpoints
        (libs/pino_utils/utils.py:178-194) load into torch.optim.Adam and back."""
        if self.capturable:
            self.step_count = int(self.step_dev.item())
        self.sync_dead_slices()
        m_full, v_full = self._full_moments() if self._runs is not None else (self.exp_avg, self.exp_avg_sq)
        pos = {id(p): i for i, p in enumerate(self.bucket.user_order)}
        state = {}
        for p, m, v in zip(self.bucket.params, self.bucket.views(m_full), self.bucket.views(v_full)):
            if self.step_count > 0:
                state[pos[id(p)]] = dict(step=torch.tensor(float(self.step_count)), exp_avg=m.detach().clone(),
                                         exp_avg_sq=v.detach().clone())
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay, amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(self.bucket.user_order))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        """Accepts torch.optim.Adam's state_dict (a reference checkpoint's 'optim' entry) or this class's own."""
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("FusedAdam.load_state_dict: expected torch.optim.Adam's {'state', 'param_groups'} layout")
        group = sd["param_groups"][0]
        order = self.bucket.user_order
        if len(group["params"]) != len(order):
            raise ValueError(f"optimizer state holds {len(group['params'])} parameters, the bucket {len(order)}")
        self.sync_dead_slices()          # (the parameters stay: bring their dead slices to the step being replaced)
        self.lr, self.betas, self.eps = group["lr"], tuple(group["betas"]), group["eps"]
        self.weight_decay = group["weight_decay"]
        m_full, v_full = torch.zeros_like(self.flat_param), torch.zeros_like(self.flat_param)
        mv = {id(p): (m, v) for p, m, v in zip(self.bucket.params, self.bucket.views(m_full), self.bucket.views(v_full))}
        steps = set()
        with torch.no_grad():
            for i, p in enumerate(order):
                st = sd["state"].get(group["params"][i], sd["state"].get(i))
                m, v = mv[id(p)]
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"optimizer state of parameter {i}: shape {tuple(st['exp_avg'].shape)} != {tuple(p.shape)}")
                m.copy_(st["exp_avg"].to(m.device)); v.copy_(st["exp_avg_sq"].to(v.device))
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError(f"optimizer state with different step counts per parameter ({sorted(steps)}): one flat Adam step "
                             "cannot represent it")
        self._scatter_full(m_full, v_full)
        self.step_count = steps.pop() if steps else 0
        if self.capturable:
            self.step_dev.fill_(self.step_count)
        self._dead_step, self._hp_log = self.step_count, []


def dense_view(t):
    """A contiguous view of the same storage for a tensor that is a dense PERMUTATION of its memory (the plane-major
    dialect-C weights of libs/models/pino_models/basics.py: last dim outermost): RCCL / NCCL collectives refuse
    non-contiguous tensors ("Tensors must be contiguous"), gloo does not."""
    if t.is_contiguous():
        return t
    order = sorted(range(t.dim()), key=lambda d: -t.stride(d))
    v = t.permute(order)
    if not v.is_contiguous():
        raise RuntimeError(f"broadcast_parameters: parameter of shape {tuple(t.shape)}, strides {t.stride()} is not a dense "
                           "permutation of its storage")
    return v


def broadcast_parameters(module, src=0, group=None, force=False):
    """Identical replicas on every rank (rank `src` wins).  `force`: also with one rank (exercises the backend's argument
    checks on a single GPU)."""
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(dense_view(t.data), src=src, group=group)


def shard_batch(t, rank, world):
    """Even split of the global batch on dim 0 (weak or strong scaling decided by the caller)."""
    n = t.shape[0]
    assert n % world == 0, f"global batch {n} not divisible by world size {world}"
    per = n // world
    return t[rank * per:(rank + 1) * per]


_UNIT_GRADS = {}


def _unit_gradient(loss):
    """The implicit gradient of `loss.backward()` (ones_like of a scalar loss), kept per device and dtype; None (= torch's
    own default) for anything that is not a scalar."""
    if loss.dim() != 0:
        return None
    key = (loss.device, loss.dtype)
    g = _UNIT_GRADS.get(key)
    if g is None:
        if loss.is_cuda and torch.cuda.is_current_stream_capturing():
            # first use inside a graph capture: the fill would only be RECORDED, and the cached tensor - in the graph's private
            # pool - would hold uninitialised memory for any eager step before the first replay: torch's own default
            return None
        g = _UNIT_GRADS[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    return g


def train_step(model_fn, bucket, optimizer, inputs, target, loss_fn, decoder=None):
    """zero_grad -> forward -> decode -> loss -> backward -> all-reduce -> optimizer step
    (run_pde_observers.py:185-193).  Returns the local loss tensor (no host sync)."""
    loss = local_gradients(model_fn, bucket, inputs, target, loss_fn, decoder)
    bucket.all_reduce()
    if optimizer is not None:
        optimizer.step()
    return loss


def local_gradients(model_fn, bucket, inputs, target, loss_fn, decoder=None):
    """The rank-local part of train_step: zero_grad -> forward -> decode -> loss -> backward.  Leaves this rank's gradients
    in the bucket and returns the local loss tensor; no collective (with a plain bucket), no optimizer step."""
    from . import functional as F
    bucket.zero()
    pred = model_fn(*inputs)
    # whether this forward used every parameter once (read by the bucket's early-launch decision during THIS backward pass;
    # another model's forward in between must not change it)
    bucket.single_use_step = F.LAST_FORWARD_SINGLE_USE[0]
    if isinstance(loss_fn, FusedLpLoss):
        # decode, both norms and the gradient happen inside the engine's loss kernels
        if decoder is not None:
            loss_fn.decoder = decoder
        loss = loss_fn(pred.reshape(target.shape), target)
    else:
        if decoder is not None:
            pred = decoder.decode(pred.reshape(target.shape))
            tgt = decoder.decode(target)
        else:
            tgt = target
        loss = loss_fn(pred, tgt)
    # `loss.backward()` with the unit gradient it would build handed in: torch fills a fresh ones_like(loss) per call (one
    # 4.7 us fill kernel per step on the measured stack); the same values, no launch
    loss.backward(_unit_gradient(loss))
    return loss.detach()


class GraphedTrainStep(object):
    """The whole training step (zero_grad -> forward -> loss -> backward -> [all-reduce] -> Adam) captured ONCE
    into a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~50 launches of a small configuration
    (BASELINE config 1) become one graph launch.  Everything the step enqueues goes through the engine's C ABI
    on the capture stream; no host-side scalar changes between replays (FusedAdam(capturable=True) keeps its step
    count on the device).  Warm-up steps run before capture on a side stream and their effect on the optimizer /
    parameters is rolled back, so the first replay is step 1.

    Data parallel (round 6): when the bucket's exchange is a real collective (more than one rank, or `force_collective`) the
    step is captured as TWO graphs - the rank-local part (zero_grad .. backward) and the optimizer step - with the
    all-reduce of the flat bucket issued eagerly between their replays: the collective stays outside the capture (RCCL
    inside a hipGraph ties the graph to one communicator state), the launch-bound part of the step is still two graph
    launches.  Needs a plain bucket: the overlapped / segmented exchanges start collectives from inside the backward pass."""

    def __init__(self, model_fn, bucket, optimizer, inputs, target, loss_fn, decoder=None, warmup=2):
        assert isinstance(optimizer, FusedAdam) and optimizer.capturable, "GraphedTrainStep needs FusedAdam(capturable=True)"
        self.inputs = [t.clone() for t in inputs]
        # (objectives with several reference tensors take a tuple: trainer.FullFieldObjective, PinoObjective)
        self.target = tuple(t.clone() for t in target) if isinstance(target, (tuple, list)) else target.clone()
        opt = optimizer
        if opt._skip_dead and opt._runs is None and bucket.direct_module is not None:
            # the optimizer plans its dead-slice layout at its first step: do it now (one forward pass records the live
            # extents), so that the snapshot below and the captured launches see the final buffers
            with torch.no_grad():
                model_fn(*self.inputs)
            opt._skip_dead = False
            opt.skip_dead_slices()
        snap = [t.clone() for t in (opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_dev, bucket.flat)]
        host_step = opt.step_count
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                train_step(model_fn, bucket, opt, self.inputs, self.target, loss_fn, decoder)
        torch.cuda.current_stream().wait_stream(side)
        for dst, src in zip((opt.flat_param, opt.exp_avg, opt.exp_avg_sq, opt.step_dev, bucket.flat), snap):
            dst.copy_(src)
        opt.step_count = host_step
        self._bucket = bucket
        self._dist = bool(bucket._collective_needed())
        self.graph = torch.cuda.CUDAGraph()
        if self._dist:
            if getattr(bucket, "_late_numel", None) is not None or getattr(bucket, "_segments", None) is not None:
                raise RuntimeError("GraphedTrainStep with a data-parallel exchange needs a plain FlatGradBucket (one all-reduce "
                                   "between the two captured parts): the overlapped (for_fno) and segmented (enable_dp_exchange) "
                                   "buckets start collectives from inside the backward pass")
            with torch.cuda.graph(self.graph):
                self.loss = local_gradients(model_fn, bucket, self.inputs, self.target, loss_fn, decoder)
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                opt.step()
        else:
            with torch.cuda.graph(self.graph):
                self.loss = train_step(model_fn, bucket, opt, self.inputs, self.target, loss_fn, decoder)
        opt.step_count = host_step            # capture enqueued nothing
        # lr / betas / eps / weight decay are launch arguments of the captured Adam kernel: a scheduler that changes them
        # afterwards would be ignored silently on replay
        self._opt = opt
        self._hyper = (opt.lr, tuple(opt.betas), opt.eps, opt.weight_decay)
        self._plan_generation = getattr(opt, "plan_generation", 0)

    def __call__(self, inputs=None, target=None):
        if getattr(self._opt, "plan_generation", 0) != self._plan_generation:
            raise RuntimeError("GraphedTrainStep: the optimizer re-planned its dead weight slices after capture (a forward pass "
                               "with a longer last dimension than the plan held live); the captured launches address the old "
                               "moment buffers - build a new GraphedTrainStep")
        if (self._opt.lr, tuple(self._opt.betas), self._opt.eps, self._opt.weight_decay) != self._hyper:
            raise RuntimeError("GraphedTrainStep: the optimizer's lr / betas / eps / weight_decay changed after capture "
                               f"({self._hyper} -> {(self._opt.lr, tuple(self._opt.betas), self._opt.eps, self._opt.weight_decay)}); "
                               "they are baked into the captured launches - build a new GraphedTrainStep")
        if inputs is not None:
            for dst, src in zip(self.inputs, inputs):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        if target is not None:
            pairs = zip(self.target, target) if isinstance(self.target, tuple) else ((self.target, target),)
            for dst, src in pairs:
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        self.graph.replay()
        if self._dist:
            self._bucket.all_reduce()          # eager, on the replay stream: between the two captured parts
            self.graph_opt.replay()
        return self.loss


class DevicePrefetcher(object):
    """Input staging for the training loop: batches of host tensors (e.g. a torch DataLoader over libs.pde_data_loader
    datasets) are copied into PINNED buffers and sent to the GPU on a separate copy stream one batch ahead of the
    consumer, cast to float32 on the device - the reference does a synchronous `.cuda().float()` per step
    (run_pde_observers.py:173).  Iterating yields tuples of device tensors that are safe to use on the current stream."""

    def __init__(self, loader, device, depth=2):
        self.loader, self.device, self.depth = loader, torch.device(device), max(1, depth)
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        items = batch if isinstance(batch, (tuple, list)) else (batch,)
        out = []
        with torch.cuda.stream(self.stream):
            for t in items:
                t = torch.as_tensor(t)
                pinned = t if t.is_pinned() else t.contiguous().pin_memory()
                out.append(pinned.to(self.device, non_blocking=True).float())
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return tuple(out), ev

    def __iter__(self):
        import collections
        q = collections.deque()
        it = iter(self.loader)
        try:
            while len(q) < self.depth:
                q.append(self._stage(next(it)))
        except StopIteration:
            it = None
        while q:
            tensors, ev = q.popleft()
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in tensors:
                t.record_stream(torch.cuda.current_stream(self.device))
            if it is not None:
                try:
                    q.append(self._stage(next(it)))
                except StopIteration:
                    it = None
            yield tensors
