"""torch.autograd bridges onto the fnoengine C ABI.

PyTorch is plumbing here: it owns device memory (inputs, outputs, workspace, the
forward->backward stash) and the stream; all arithmetic happens in the HIP library.
"""
import ctypes as C

import weakref

import torch

from . import _lib

_spec_plans = {}
_model_plans = {}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"fnoengine: `{name}` must live on the GPU (got {t.device}); "
                           "the engine has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"fnoengine: `{name}` must be float32 (got {t.dtype})")


def _bytes(n, device):
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------
# standalone spectral convolution
# ----------------------------------------------------------------------------
def spec_plan(ndim, cin, cout, dims, modes, weight_last_extent, norm, device, input_gelu=False, weight_planes=False):
    key = (ndim, cin, cout, tuple(dims), tuple(modes), weight_last_extent, norm, device.index, bool(input_gelu),
           bool(weight_planes))
    plan = _spec_plans.get(key)
    if plan is None:
        d = _lib.FnoSpecDesc()
        d.ndim, d.Cin, d.Cout = ndim, cin, cout
        for i in range(ndim):
            d.dims[i], d.modes[i] = int(dims[i]), int(modes[i])
        d.weight_last_extent = int(weight_last_extent)
        d.norm = _lib.NORM_CODES[norm]
        d.input_gelu = 1 if input_gelu else 0
        d.weight_planes = 1 if weight_planes else 0
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().fno_spec_plan_create(C.byref(d), C.byref(h)), "spec_plan_create")
        plan = h
        _spec_plans[key] = plan
    return plan


# Listeners for gradients the engine writes straight into the caller's storage (direct_grads): autograd never sees those
# tensors, so a data-parallel trainer that starts a gradient exchange as soon as a segment is complete
# (trainer.FlatGradBucket.enable_segmented_exchange) learns about them here.  Called with the list of written tensors.
DIRECT_WRITE_HOOKS = []


# Direct gradient writes are valid only when a parameter is used by exactly ONE engine call per step.  Models whose
# parameters may be reused (RNO2d over several time steps / predicted steps) run their forward under single_use(flag): with
# the flag off, `direct_grads` requests inside are ignored and autograd accumulates as usual (the bucket is zeroed in full
# for such models: FlatGradBucket(direct_module=..., zero_all=True)).
_SINGLE_USE = [True]
LAST_FORWARD_SINGLE_USE = [True]       # what the most recent declaring forward said (read by the gradient buckets: a parameter
                                       # used several times "arrives" several times, so segments must not leave early)


class single_use(object):
    def __init__(self, ok):
        self.ok = bool(ok)

    def __enter__(self):
        _SINGLE_USE.append(self.ok)
        LAST_FORWARD_SINGLE_USE[0] = self.ok

    def __exit__(self, *exc):
        _SINGLE_USE.pop()


def plane_major(w):
    """True when a corner weight (complex (Cin, Cout, m.., wl) or its real view (.., wl, 2)) is stored PLANE-MAJOR: the last
    mode dim outermost in memory, everything else contiguous behind it (include/fnoengine.h, weight_planes) - the layout
    libs.models.pino_models.basics.SpectralConv3d gives its parameters, so that the live last-dim slices are one
    contiguous prefix.  A tensor that is also contiguous in the ordinary sense (wl = 1) counts as ordinary."""
    r = torch.view_as_real(w) if w.is_complex() else w
    if r.dim() < 4 or r.is_contiguous() or r.stride(-1) != 1:
        return False
    plane = r[..., 0, :]
    return plane.is_contiguous() and r.stride(-2) == plane.numel()


def to_plane_major(w):
    """The same values with the last dim outermost in memory (shape unchanged)."""
    nd = w.dim()
    return w.permute(nd - 1, *range(nd - 1)).contiguous().permute(*range(1, nd), 0)


def _weights_ready(ws):
    """(tensors the engine can read in place, weight_planes flag): all plane-major -> as they are; otherwise contiguous"""
    if ws and all(plane_major(t) for t in ws):
        return list(ws), True
    return [t.contiguous() for t in ws], False


def _check_corner_weights(spec_ws, cin, cout, modes, planes, what):
    """The C ABI takes raw pointers: a corner weight of the wrong extent is an out-of-bounds read on the device, not an
    error.  Contiguous corner weights (real view) must be (cin, cout, *modes, 2) - what the reference's einsum would
    have refused otherwise; plane-major ones carry their own (checked) extents."""
    if planes:
        return
    want = (int(cin), int(cout)) + tuple(int(m) for m in modes) + (2,)
    for i, t in enumerate(spec_ws):
        if tuple(t.shape) != want:
            raise RuntimeError(f"fnoengine {what}: spectral weight {i} has shape {tuple(t.shape)}, the plan's kept modes "
                               f"{tuple(modes)} need {want} (real view of a ({cin}, {cout}, {', '.join(str(int(m)) for m in modes)}) complex corner)")


def _same_layout(g, w):
    return g is not None and g.shape == w.shape and g.stride() == w.stride() and (g.is_contiguous() or plane_major(g))


def _fresh_grads(ws, planes):
    """gradient tensors laid out like the weights; plane-major: zeros (the engine writes the live planes only)"""
    return [torch.zeros_like(t) if planes else torch.empty_like(t) for t in ws]


def _direct_views(direct_grads, spec_ws, last_dim=None):
    """the weights' own .grad storage as real views, or None when direct writes are off / not possible.
    `last_dim` = the data's last extent: with plane-major weights the engine writes only the live planes
    [0, min(last_dim / 2 + 1, modes3)) of a gradient and nobody else clears a direct-write region (the bucket's zero()
    skips it), so when the live extent SHRINKS against the previous direct write the planes in between are cleared here -
    they would otherwise keep the last step's gradient (a batch with a shorter last dim behind a longer one)."""
    if not (direct_grads and _SINGLE_USE[-1] and torch.is_grad_enabled()
            and all(_same_layout(t.grad, t) for t in spec_ws)):
        return None
    if last_dim is not None:
        for t in spec_ws:
            if t.is_complex() and plane_major(t):
                k = min(int(last_dim) // 2 + 1, t.shape[-1])
                prev = t.__dict__.get("_fno_direct_k", 0)
                if k < prev:
                    t.grad[..., k:prev].zero_()
                t._fno_direct_k = k
    return [torch.view_as_real(t.grad) if t.grad.is_complex() else t.grad for t in spec_ws]


def _notify_direct(tensors):
    """DIRECT_WRITE_HOOKS holds weak references to bound methods (a bucket that went away must not be kept alive, nor
    polled): dead entries are dropped here."""
    dead = []
    for ref in DIRECT_WRITE_HOOKS:
        h = ref() if isinstance(ref, weakref.WeakMethod) else ref
        if h is None:
            dead.append(ref)
        else:
            h(tensors)
    for ref in dead:
        DIRECT_WRITE_HOOKS.remove(ref)


class _SpectralConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, modes, norm, weight_last_extent, direct, *weights):
        _require_cuda(x, "x")
        x = x.contiguous()
        ws_list, planes = _weights_ready(weights)            # real views (.., 2)
        for w in ws_list:
            _require_cuda(w, "weight")
        B, cin = x.shape[0], x.shape[1]
        dims = tuple(x.shape[2:])
        ndim = len(dims)
        cout = ws_list[0].shape[1]
        # (raw pointers from here on: extents the plan will assume are checked against the tensors first)
        if len(ws_list) != 2 ** (ndim - 1) or len(modes) != ndim:
            raise RuntimeError(f"fnoengine spectral_conv: {len(ws_list)} corner weights / {len(modes)} mode counts for {ndim}-d data")
        wl = int(weight_last_extent) if weight_last_extent else int(modes[-1])
        want = (int(cin), int(cout)) + tuple(int(m) for m in modes[:-1]) + (wl, 2)
        for i, w in enumerate(ws_list):
            if tuple(w.shape) != want:
                raise RuntimeError(f"fnoengine spectral_conv: corner weight {i} has shape {tuple(w.shape)}, kept modes {tuple(modes)} "
                                   f"(last extent {wl}) need {want}")
        if bias is not None and bias.numel() != cout:
            raise RuntimeError(f"fnoengine spectral_conv: bias has {bias.numel()} elements, {cout} output channels")
        L = _lib.lib()
        plan = spec_plan(ndim, cin, cout, dims, modes, weight_last_extent, norm, x.device, weight_planes=planes)
        ctx.planes = planes
        y = torch.empty((B, cout) + dims, dtype=torch.float32, device=x.device)
        xhat = _bytes(L.fno_spec_xhat_bytes(plan, B), x.device)
        nws = L.fno_spec_workspace_bytes(plan, B)
        ws = _bytes(nws, x.device)
        wp = (C.c_void_p * 4)(*[w.data_ptr() for w in ws_list] + [0] * (4 - len(ws_list)))
        b = bias.contiguous() if bias is not None else None
        with torch.cuda.device(x.device):
            _lib.check(L.fno_spec_forward(plan, B, _ptr(x), wp, _ptr(b), _ptr(y), _ptr(xhat), _ptr(ws), nws,
                                          _stream()), "spec_forward")
        ctx.plan, ctx.B, ctx.has_bias = plan, B, bias is not None
        ctx.x_shape = x.shape
        ctx.direct = direct
        ctx.save_for_backward(xhat, *ws_list)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, *ws_list = ctx.saved_tensors
        dy = dy.contiguous()
        L = _lib.lib()
        need_dx = ctx.needs_input_grad[0]
        need_db = ctx.has_bias and ctx.needs_input_grad[1]
        need_dw = any(ctx.needs_input_grad[6:])
        dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dy.device) if need_dx else None
        direct = ctx.direct if need_dw else None
        dws = (direct if direct is not None else _fresh_grads(ws_list, ctx.planes)) if need_dw else None
        db = torch.empty(dy.shape[1], dtype=torch.float32, device=dy.device) if need_db else None
        nws = L.fno_spec_workspace_bytes(ctx.plan, ctx.B)
        ws = _bytes(nws, dy.device)
        wp = (C.c_void_p * 4)(*[w.data_ptr() for w in ws_list] + [0] * (4 - len(ws_list)))
        dwp = (C.c_void_p * 4)(*[w.data_ptr() for w in dws] + [0] * (4 - len(dws))) if need_dw else None
        with torch.cuda.device(dy.device):
            _lib.check(L.fno_spec_backward(ctx.plan, ctx.B, _ptr(dy), _ptr(xhat), wp, _ptr(dx),
                                           dwp, _ptr(db), _ptr(ws), nws, _stream()), "spec_backward")
        if direct is not None:                     # written in place into the caller's gradient storage
            _notify_direct(direct)
            return (dx, db, None, None, None, None) + (None,) * len(ws_list)
        return (dx, db, None, None, None, None) + (tuple(dws) if need_dw else (None,) * len(ws_list))


def spectral_conv(x, weights, bias, modes, norm="backward", weight_last_extent=None, direct_grads=False):
    """y = irfftn(pad(W_c . rfftn(x)[corner_c]), s=x.shape[2:]) (+ bias[None, :, None..]).

    weights: corner tensors in canonical order, real (Cin, Cout, m.., 2) or complex.
    modes:   kept extent per corner along each dim.
    direct_grads: the backward WRITES dL/dW into the weights' existing contiguous `.grad` storage (e.g. views of a
    trainer.FlatGradBucket) instead of returning it to autograd: no accumulation kernel, no zeroing needed.  Only valid when
    each weight feeds exactly one spectral_conv call per step.
    """
    ws = [torch.view_as_real(w) if w.is_complex() else w for w in weights]
    wle = int(weight_last_extent) if weight_last_extent is not None else int(ws[0].shape[-2])
    b = bias.reshape(-1) if bias is not None else None
    direct = _direct_views(direct_grads, weights, last_dim=x.shape[-1])
    return _SpectralConvFn.apply(x, b, tuple(int(m) for m in modes), norm, wle, direct, *ws)


# ----------------------------------------------------------------------------
# fused FNO model
# ----------------------------------------------------------------------------
def model_plan(ndim, cin, c, cout, hidden_proj, n_layers, dims, modes, norm, gelu_mask, device, weight_planes=False):
    key = (ndim, cin, c, cout, hidden_proj, n_layers, tuple(dims), tuple(modes), norm, gelu_mask, device.index) \
        + ((True,) if weight_planes else ())
    plan = _model_plans.get(key)
    if plan is None:
        d = _lib.FnoModelDesc()
        d.ndim, d.Cin, d.C, d.Cout = ndim, cin, c, cout
        d.hidden_proj, d.n_layers = hidden_proj, n_layers
        for i in range(ndim):
            d.dims[i], d.modes[i] = int(dims[i]), int(modes[i])
        d.norm = _lib.NORM_CODES[norm]
        d.gelu_mask = gelu_mask
        d.weight_planes = 1 if weight_planes else 0
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().fno_model_plan_create(C.byref(d), C.byref(h)), "model_plan_create")
        plan = h
        _model_plans[key] = plan
    return plan


def model_plan_available(ndim, cin, c, cout, hidden_proj, n_layers, dims, modes, norm, gelu_mask, device):
    """True when fno_model_plan_create accepts the configuration (result cached; an unsupported shape is not an error
    for callers that have an unfused path)."""
    key = ("avail", ndim, cin, c, cout, hidden_proj, n_layers, tuple(dims), tuple(modes), norm, gelu_mask, device.index)
    ok = _model_plans.get(key)
    if ok is None:
        try:
            model_plan(ndim, cin, c, cout, hidden_proj, n_layers, dims, modes, norm, gelu_mask, device)
            ok = True
        except RuntimeError:
            ok = False
        _model_plans[key] = ok
    return ok


def _fill_params(struct, n_layers, ncorner, lift_w, lift_b, skip_ws, spec_ws, spec_bias, w1, b1, w2, b2):
    struct.lift_w, struct.lift_b = lift_w.data_ptr(), lift_b.data_ptr()
    for l in range(n_layers):
        struct.skip_w[l] = skip_ws[l].data_ptr()
        for c in range(ncorner):
            struct.spec_w[l][c] = spec_ws[l * ncorner + c].data_ptr()
    struct.spec_bias = spec_bias.data_ptr() if spec_bias is not None else 0
    struct.proj_w1, struct.proj_b1 = w1.data_ptr(), b1.data_ptr()
    struct.proj_w2, struct.proj_b2 = w2.data_ptr(), b2.data_ptr()


class _FNOModelFn(torch.autograd.Function):
    """Whole FNO forward/backward in the HIP engine.  Tensor arguments, in order:
    x, lift_w, lift_b, spec_bias (or None), w1, b1, w2, b2, skip_w[0..L), spec_w[0..L*ncorner)."""

    @staticmethod
    def forward(ctx, cfg, x, lift_w, lift_b, spec_bias, w1, b1, w2, b2, *rest):
        n_layers, modes, norm, gelu_mask, direct, overlap = cfg
        ctx.direct = direct
        ctx.overlap = overlap
        _require_cuda(x, "x")
        x = x.contiguous()
        dims = tuple(x.shape[2:])
        ndim = len(dims)
        ncorner = 2 ** (ndim - 1)
        skip_ws = [t.contiguous() for t in rest[:n_layers]]
        spec_ws = [t.contiguous() for t in rest[n_layers:]]
        assert len(spec_ws) == n_layers * ncorner
        tensors = [lift_w, lift_b, w1, b1, w2, b2] + skip_ws + spec_ws + ([spec_bias] if spec_bias is not None else [])
        for t in tensors:
            _require_cuda(t, "parameter")
        lift_w, lift_b, w1, b1, w2, b2 = [t.contiguous() for t in (lift_w, lift_b, w1, b1, w2, b2)]
        sb = spec_bias.contiguous() if spec_bias is not None else None
        B, cin = x.shape[0], x.shape[1]
        c, cout, hid = lift_w.shape[0], w2.shape[0], w1.shape[0]
        _check_corner_weights(spec_ws, c, c, modes, False, "fno_model")
        for t, want, nm in ((lift_w, c * cin, "lifting weight"), (lift_b, c, "lifting bias"), (w1, hid * c, "projection W1"), (b1, hid, "projection b1"),
                            (w2, cout * hid, "projection W2"), (b2, cout, "projection b2")):
            if t.numel() != want:
                raise RuntimeError(f"fnoengine fno_model: {nm} has {t.numel()} elements, the model's widths need {want}")
        for l, t in enumerate(skip_ws):
            if t.numel() != c * c:
                raise RuntimeError(f"fnoengine fno_model: skip weight {l} has {t.numel()} elements, need {c * c}")
        L = _lib.lib()
        plan = model_plan(ndim, cin, c, cout, hid, n_layers, dims, modes, norm, gelu_mask, x.device)
        prm = _lib.FnoModelParams()
        _fill_params(prm, n_layers, ncorner, lift_w, lift_b, skip_ws, spec_ws, sb, w1, b1, w2, b2)
        y = torch.empty((B, cout) + dims, dtype=torch.float32, device=x.device)
        saved = _bytes(L.fno_model_saved_bytes(plan, B), x.device)
        nws = L.fno_model_workspace_bytes(plan, B)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.fno_model_forward(plan, B, C.byref(prm), _ptr(x), _ptr(y), _ptr(saved), _ptr(ws), nws,
                                           _stream()), "model_forward")
        ctx.plan, ctx.B, ctx.n_layers, ctx.ncorner = plan, B, n_layers, ncorner
        ctx.has_sb = sb is not None
        ctx.save_for_backward(x, saved, lift_w, lift_b, w1, b1, w2, b2, *skip_ws, *spec_ws,
                              *([sb] if sb is not None else []))
        return y

    @staticmethod
    def backward(ctx, dy):
        sv = ctx.saved_tensors
        x, saved, lift_w, lift_b, w1, b1, w2, b2 = sv[:8]
        nl, nc = ctx.n_layers, ctx.ncorner
        skip_ws = list(sv[8:8 + nl])
        spec_ws = list(sv[8 + nl:8 + nl + nl * nc])
        sb = sv[8 + nl + nl * nc] if ctx.has_sb else None
        dy = dy.contiguous()
        L = _lib.lib()
        prm = _lib.FnoModelParams()
        _fill_params(prm, nl, nc, lift_w, lift_b, skip_ws, spec_ws, sb, w1, b1, w2, b2)
        if ctx.direct is not None:
            # the engine WRITES gradients: hand it the parameters' own (pre-allocated, flat-bucket)
            # .grad storage and return None so autograd launches no accumulation kernels
            dg = ctx.direct
            g = [dg[k] for k in ("lift_w", "lift_b", "w1", "b1", "w2", "b2")]
            g_skip, g_spec, g_sb = dg["skip"], dg["spec"], dg["spec_bias"]
        else:
            g = [torch.empty_like(t) for t in (lift_w, lift_b, w1, b1, w2, b2)]
            g_skip = [torch.empty_like(t) for t in skip_ws]
            g_spec = [torch.empty_like(t) for t in spec_ws]
            g_sb = torch.empty_like(sb) if sb is not None else None
        grd = _lib.FnoModelGrads()
        _fill_params(grd, nl, nc, g[0], g[1], g_skip, g_spec, g_sb, g[2], g[3], g[4], g[5])
        nws = L.fno_model_workspace_bytes(ctx.plan, ctx.B)
        ws = _bytes(nws, dy.device)
        ov = ctx.overlap
        # dL/dx through the lifting layer (run_control.py:186-224 differentiates the observer down to its input field)
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(dy.device):
            if dx is not None:
                _lib.check(L.fno_model_backward_dx(ctx.plan, ctx.B, C.byref(prm), _ptr(x), _ptr(dy), _ptr(saved),
                                                   C.byref(grd), _ptr(dx), _ptr(ws), nws, _stream()), "model_backward_dx")
            elif ov is not None and ctx.direct is not None and 0 < ov.split_layer < nl:
                # late layers first; their finished gradients go on the wire while the early layers are differentiated
                k = ov.split_layer
                _lib.check(L.fno_model_backward_part(ctx.plan, ctx.B, C.byref(prm), _ptr(x), _ptr(dy), _ptr(saved),
                                                     C.byref(grd), None, _ptr(ws), nws, _stream(), nl - 1, k),
                           "model_backward_part")
                ov.late_gradients_ready()
                _lib.check(L.fno_model_backward_part(ctx.plan, ctx.B, C.byref(prm), _ptr(x), _ptr(dy), _ptr(saved),
                                                     C.byref(grd), None, _ptr(ws), nws, _stream(), k - 1, 0),
                           "model_backward_part")
            else:
                _lib.check(L.fno_model_backward(ctx.plan, ctx.B, C.byref(prm), _ptr(x), _ptr(dy), _ptr(saved),
                                                C.byref(grd), _ptr(ws), nws, _stream()), "model_backward")
        if ctx.direct is not None:
            return (None, dx) + (None,) * (7 + len(skip_ws) + len(spec_ws))
        return (None, dx, g[0], g[1], g_sb, g[2], g[3], g[4], g[5]) + tuple(g_skip) + tuple(g_spec)


def fno_model(x, lift_w, lift_b, skip_ws, spec_ws, spec_bias, w1, b1, w2, b2, modes, norm="forward",
              gelu_mask=None, direct_grads=False, overlap=None):
    """Fused neuralop.models.FNO forward (default configuration).  `modes` = kept per
    corner per dim (n_modes // 2); `spec_ws` real-view corner weights, layer-major."""
    n_layers = len(skip_ws)
    if gelu_mask is None:
        gelu_mask = 0
        for l in range(n_layers):
            if l < n_layers - l:                 # fno_block.py:149
                gelu_mask |= 1 << l
    direct = None
    if direct_grads:
        params = [lift_w, lift_b, w1, b1, w2, b2] + list(skip_ws) + list(spec_ws) + ([spec_bias] if spec_bias is not None else [])
        if all(p.grad is not None and p.grad.is_contiguous() for p in params):
            direct = dict(lift_w=lift_w.grad, lift_b=lift_b.grad, w1=w1.grad, b1=b1.grad, w2=w2.grad, b2=b2.grad,
                          skip=[p.grad for p in skip_ws], spec=[p.grad for p in spec_ws],
                          spec_bias=spec_bias.grad if spec_bias is not None else None)
    cfg = (n_layers, tuple(int(m) for m in modes), norm, int(gelu_mask), direct, overlap if direct is not None else None)
    return _FNOModelFn.apply(cfg, x, lift_w, lift_b, spec_bias, w1, b1, w2, b2, *skip_ws, *spec_ws)


# ----------------------------------------------------------------------------
# training-step tail: decode + relative-L2 loss, Adam on a flat bucket
# ----------------------------------------------------------------------------
class _LpLossRelFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, mean, std, eps, size_average):
        _require_cuda(pred, "pred")
        _require_cuda(target, "target")
        L = _lib.lib()
        B = pred.shape[0]
        pred_c, tgt_c = pred.contiguous(), target.contiguous()
        n = pred_c.numel() // B
        if tgt_c.numel() != pred_c.numel():
            raise RuntimeError(f"fnoengine lp_loss_rel: pred {tuple(pred.shape)} vs target {tuple(target.shape)}")
        stat_len = 1
        for s in (mean, std):
            if s is not None:
                _require_cuda(s, "mean/std")
                stat_len = s.numel()
        if mean is not None and std is not None and mean.numel() != std.numel():
            raise RuntimeError("fnoengine lp_loss_rel: mean and std must have the same number of elements")
        mean_c = mean.contiguous() if mean is not None else None
        std_c = std.contiguous() if std is not None else None
        nws = L.fno_lploss_workspace_bytes(B)
        ws = _bytes(nws, pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        _lib.check(L.fno_lploss_rel_forward(B, n, _ptr(pred_c), _ptr(tgt_c), _ptr(mean_c), _ptr(std_c), stat_len,
                                            float(eps), int(bool(size_average)), _ptr(loss), _ptr(ws), nws, _stream()),
                   "lploss_rel_forward")
        ctx.save_for_backward(pred_c, tgt_c, std_c if std_c is not None else pred_c.new_empty(0), ws)
        ctx.meta = (B, n, stat_len, float(eps), std_c is not None, nws, pred.shape)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        pred_c, tgt_c, std_c, ws = ctx.saved_tensors
        B, n, stat_len, eps, has_std, nws, shape = ctx.meta
        L = _lib.lib()
        dpred = torch.empty_like(pred_c)
        g = gloss.contiguous().to(torch.float32)
        _lib.check(L.fno_lploss_rel_backward(B, n, _ptr(pred_c), _ptr(tgt_c), _ptr(std_c if has_std else None), stat_len,
                                             eps, _ptr(g), _ptr(dpred), _ptr(ws), nws, _stream()), "lploss_rel_backward")
        return dpred.view(shape), None, None, None, None, None


def lp_loss_rel(pred, target, mean=None, std=None, eps=1e-5, size_average=False):
    """LpLoss(d=2, p=2).rel of the DECODED fields in two streaming passes (libs/utilities3.py:115-129,
    323-334; run_pde_observers.py:188-192).  mean / std: None (no decode), scalars or per-element planes
    broadcast over the batch."""
    return _LpLossRelFn.apply(pred, target, mean, std, eps, size_average)


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
              step_counter=None, scratch=None):
    """One torch.optim.Adam update of a flat fp32 bucket, in place, one kernel.  With `step_counter`
    (int32 device tensor) the step count lives on the device (fno_adam_step_dev: graph-replayable)."""
    for t, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _require_cuda(t, name)
        if not t.is_contiguous() or t.numel() != param.numel():
            raise RuntimeError(f"fnoengine adam_step: `{name}` must be contiguous with {param.numel()} elements")
    if step_counter is not None:
        _lib.check(_lib.lib().fno_adam_step_dev(param.numel(), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                                float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                float(weight_decay), _ptr(step_counter), _ptr(scratch), _stream()),
                   "adam_step_dev")
        return
    _lib.check(_lib.lib().fno_adam_step(param.numel(), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                        float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
                                        int(step), _stream()), "adam_step")


def adam_step_runs(runs, param, grad, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay, step_counter=None,
                   scratch=None):
    """One Adam update of a bucket planned around dead last-dim slices (trainer.FusedAdam.skip_dead_slices): `runs` lists
    ("dense", offset, n, compact offset) ranges and ("rows", offset, rows, row_len, live_len, compact offset) blocks of
    `param` / `grad` (full layout); exp_avg / exp_avg_sq are compact.  One kernel per run, the dead part of a block untouched."""
    L = _lib.lib()
    for t, name in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _require_cuda(t, name)
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise RuntimeError(f"fnoengine adam_step_runs: `{name}` must be contiguous float32")
    if grad.numel() != param.numel() or exp_avg.numel() != exp_avg_sq.numel():
        raise RuntimeError("fnoengine adam_step_runs: param / grad and exp_avg / exp_avg_sq must pair up")
    dyn = None
    if step_counter is not None:
        _lib.check(L.fno_adam_prep_dev(_ptr(step_counter), _ptr(scratch), float(lr), float(betas[0]), float(betas[1]),
                                       _stream()), "adam_prep_dev")
        dyn = _ptr(scratch)
    hp = (float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step))
    P, G, M, V = param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr()
    for run in runs:
        if run[0] == "dense":
            _, off, n, coff = run
            if off + n > param.numel() or coff + n > exp_avg.numel():
                raise RuntimeError("fnoengine adam_step_runs: run outside the buffers")
            _lib.check(L.fno_adam_step_range(n, P + 4 * off, G + 4 * off, M + 4 * coff, V + 4 * coff, *hp, dyn, _stream()),
                       "adam_step_range")
        else:
            _, off, rows, row_len, live_len, coff = run
            if off + rows * row_len > param.numel() or coff + rows * live_len > exp_avg.numel():
                raise RuntimeError("fnoengine adam_step_runs: block outside the buffers")
            _lib.check(L.fno_adam_step_live(rows, row_len, live_len, P + 4 * off, G + 4 * off, M + 4 * coff, V + 4 * coff, *hp,
                                            dyn, _stream()), "adam_step_live")


def adam_replay_scalars(step_from, nsteps, lr, betas, device, on_device=False):
    """(2 * nsteps,) device tensor: {lr / (1 - beta1^t), sqrt(1 - beta2^t)} for t = step_from .. step_from + nsteps - 1,
    derived as the stepping kernels' callers derive them: on the host in double (fno_adam_step), or - on_device - by the
    device arithmetic of the graph-replayable path (fno_adam_step_dev)."""
    L = _lib.lib()
    if on_device:
        scal = torch.empty(2 * nsteps, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _lib.check(L.fno_adam_replay_prep(_ptr(scal), int(step_from), int(nsteps), float(lr), float(betas[0]),
                                              float(betas[1]), _stream()), "adam_replay_prep")
        return scal
    import ctypes
    host = torch.empty(2 * nsteps, dtype=torch.float32)
    base = host.data_ptr()
    for j in range(nsteps):
        L.fno_adam_scalars(float(lr), float(betas[0]), float(betas[1]), int(step_from + j), ctypes.c_void_p(base + 8 * j))
    return host.to(device)


def adam_replay_dead(rows, row_len, live_len, param_block, dead_m, dead_v, moments_zero, scal, betas, eps, weight_decay):
    """Take the dead part of a row-sliced block (rows x row_len floats of `param_block`, dead = [live_len, row_len) of each
    row) through the Adam steps described by `scal` (adam_replay_scalars) with a zero gradient; dead_m / dead_v: compact
    dead moments, read unless moments_zero, always written."""
    for t, name in ((param_block, "param"), (dead_m, "dead exp_avg"), (dead_v, "dead exp_avg_sq"), (scal, "scalars")):
        _require_cuda(t, name)
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise RuntimeError(f"fnoengine adam_replay_dead: `{name}` must be contiguous float32")
    nd = rows * (row_len - live_len)
    if param_block.numel() != rows * row_len or dead_m.numel() != nd or dead_v.numel() != nd or scal.numel() % 2:
        raise RuntimeError("fnoengine adam_replay_dead: buffer sizes do not match the block")
    with torch.cuda.device(param_block.device):
        _lib.check(_lib.lib().fno_adam_replay_dead(rows, row_len, live_len, _ptr(param_block), _ptr(dead_m), _ptr(dead_v),
                                                   1 if moments_zero else 0, _ptr(scal), scal.numel() // 2, float(betas[0]),
                                                   float(betas[1]), float(eps), float(weight_decay), _stream()),
                   "adam_replay_dead")


# ----------------------------------------------------------------------------
# fused block stack: y = B_{L-1}(...B_0(x)),  B_l(u) = [gelu](specconv_l(u) + conv1x1_l(u) + bias_l)
# ----------------------------------------------------------------------------
class _FNOBlocksFn(torch.autograd.Function):
    """Tensor arguments: x, bias (L, C) or None, skip_w[0..L), spec_w[0..L*ncorner)."""

    @staticmethod
    def forward(ctx, cfg, x, bias, *rest):
        n_layers, modes, norm, gelu_mask, direct = cfg[:5]
        tail = cfg[5] if len(cfg) > 5 else None      # (relu_out, drop_p, seed tensor or None): fno_model_*_tail
        ctx.direct = direct
        _require_cuda(x, "x")
        x = x.contiguous()
        dims = tuple(x.shape[2:])
        ndim = len(dims)
        ncorner = 2 ** (ndim - 1)
        skip_ws = [t.contiguous() for t in rest[:n_layers]]
        spec_ws, planes = _weights_ready(rest[n_layers:])
        ctx.planes = planes
        assert len(spec_ws) == n_layers * ncorner
        for t in skip_ws + spec_ws + ([bias] if bias is not None else []):
            _require_cuda(t, "parameter")
        sb = bias.contiguous() if bias is not None else None
        B, c = x.shape[0], x.shape[1]
        _check_corner_weights(spec_ws, c, c, modes, planes, "fno_blocks")
        L = _lib.lib()
        plan = model_plan(ndim, 0, c, 0, 0, n_layers, dims, modes, norm, gelu_mask, x.device, weight_planes=planes)
        prm = _lib.FnoModelParams()
        for l in range(n_layers):
            prm.skip_w[l] = skip_ws[l].data_ptr()
            for k in range(ncorner):
                prm.spec_w[l][k] = spec_ws[l * ncorner + k].data_ptr()
        prm.spec_bias = sb.data_ptr() if sb is not None else 0
        y = torch.empty_like(x)
        saved = _bytes(L.fno_model_saved_bytes(plan, B), x.device)
        nws = L.fno_model_workspace_bytes(plan, B)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            if tail is None:
                _lib.check(L.fno_model_forward(plan, B, C.byref(prm), _ptr(x), _ptr(y), _ptr(saved), _ptr(ws), nws,
                                               _stream()), "blocks_forward")
            else:
                relu_out, drop_p, seed = tail
                t = _lib.FnoBlockTail(int(relu_out), float(drop_p), _ptr(seed), None)
                _lib.check(L.fno_model_forward_tail(plan, B, C.byref(prm), _ptr(x), _ptr(y), _ptr(saved), _ptr(ws), nws,
                                                    _stream(), C.byref(t)), "blocks_forward_tail")
        ctx.plan, ctx.B, ctx.n_layers, ctx.ncorner, ctx.has_sb = plan, B, n_layers, ncorner, sb is not None
        ctx.tail = None if tail is None else (bool(tail[0]), float(tail[1]))
        extra = []
        if tail is not None:
            extra = [y if tail[0] else x.new_empty(0), tail[2] if tail[2] is not None else x.new_empty(0)]
        ctx.save_for_backward(x, saved, *skip_ws, *spec_ws, *([sb] if sb is not None else []), *extra)
        return y

    @staticmethod
    def backward(ctx, dy):
        sv = ctx.saved_tensors
        x, saved = sv[:2]
        nl, nc = ctx.n_layers, ctx.ncorner
        skip_ws = list(sv[2:2 + nl])
        spec_ws = list(sv[2 + nl:2 + nl + nl * nc])
        sb = sv[2 + nl + nl * nc] if ctx.has_sb else None
        dy = dy.contiguous()
        L = _lib.lib()
        prm, grd = _lib.FnoModelParams(), _lib.FnoModelGrads()
        g_skip = [torch.empty_like(t) for t in skip_ws]
        g_spec = ctx.direct if ctx.direct is not None else _fresh_grads(spec_ws, ctx.planes)   # direct: the weights' own .grad storage
        g_sb = torch.empty_like(sb) if sb is not None else None
        for l in range(nl):
            prm.skip_w[l], grd.skip_w[l] = skip_ws[l].data_ptr(), g_skip[l].data_ptr()
            for k in range(nc):
                prm.spec_w[l][k], grd.spec_w[l][k] = spec_ws[l * nc + k].data_ptr(), g_spec[l * nc + k].data_ptr()
        prm.spec_bias = sb.data_ptr() if sb is not None else 0
        grd.spec_bias = g_sb.data_ptr() if g_sb is not None else 0
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        nws = L.fno_model_workspace_bytes(ctx.plan, ctx.B)
        ws = _bytes(nws, dy.device)
        with torch.cuda.device(dy.device):
            if ctx.tail is None:
                _lib.check(L.fno_model_backward_dx(ctx.plan, ctx.B, C.byref(prm), _ptr(x), _ptr(dy), _ptr(saved),
                                                   C.byref(grd), _ptr(dx), _ptr(ws), nws, _stream()), "blocks_backward")
            else:
                y_out, seed = sv[-2], sv[-1]
                t = _lib.FnoBlockTail(int(ctx.tail[0]), ctx.tail[1], _ptr(seed) if seed.numel() else None,
                                      _ptr(y_out) if y_out.numel() else None)
                _lib.check(L.fno_model_backward_tail(ctx.plan, ctx.B, C.byref(prm), _ptr(x), _ptr(dy), _ptr(saved),
                                                     C.byref(grd), _ptr(dx), _ptr(ws), nws, _stream(), C.byref(t)),
                           "blocks_backward_tail")
        if ctx.direct is not None:
            _notify_direct(ctx.direct)
        return (None, dx, g_sb) + tuple(g_skip) + ((None,) * len(g_spec) if ctx.direct is not None else tuple(g_spec))


def blocks_supported(x, n_layers=1, modes=None, norm="backward", gelu_mask=0):
    """Shapes the fused block kernels cover (fno_model_plan_create): 32 / 64 channels, last dim a
    multiple of 32 (<= 256), planes that tile by 128 (256) pixels; with `modes` the engine itself is asked
    (tile + twiddle tables must fit LDS)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() in (4, 5)):
        return False
    c, w = x.shape[1], x.shape[-1]
    pw = 1
    for s in x.shape[2:]:
        pw *= s
    npx = 256 if w > 128 else 128
    tiled = w % 32 == 0 and w <= 256 and npx % w == 0 and pw % npx == 0
    # "loose rows" (any other last dim in 32..320, e.g. the PINO observers' padded time axis 73, or 96 / 160):
    # 128-pixel tiles of the flattened plane, spectral rows gathered per tile (split-precision GEMM mode only)
    loose = (not tiled) and 32 <= w <= 320 and pw % 128 == 0 and _lib.lib().fno_get_gemm_mode() == 1
    if not (c in (32, 64) and (tiled or loose) and n_layers <= _lib.FNO_MAX_LAYERS):
        return False
    if modes is None:
        return True
    return model_plan_available(x.dim() - 2, 0, c, 0, 0, n_layers, tuple(x.shape[2:]), tuple(int(m) for m in modes), norm,
                                int(gelu_mask), x.device)


def block_tail_supported(x, modes, norm):
    """Shapes fno_block_tail covers: what one fused block covers on whole rows (32 / 64 channels, rows of 32 / 64 / 128
    floats tiling 128-pixel tiles), split-precision GEMM mode."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _lib.lib().fno_get_gemm_mode() == 1):
        return False
    w, pw = x.shape[-1], x.shape[-1] * x.shape[-2]
    if not (x.shape[1] in (32, 64) and w in (32, 64, 128) and pw % 128 == 0):
        return False
    return blocks_supported(x, 1, modes, norm)


def draw_dropout_seed(device):
    """Two 32-bit words for the engine's counter-based dropout, drawn on the device by torch's generator: follows
    torch.manual_seed, and is safe under hipGraph capture (the generator's offset advances per replay)."""
    return torch.randint(-2 ** 31, 2 ** 31 - 1, (2,), device=device, dtype=torch.int32)


def dropout_scale(n, drop_p, seed, device):
    """The 0 / 1/(1-p) field the kernels regenerate from `seed` for a tensor of n elements (tests, oracles)."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().fno_dropout_scale(n, float(drop_p), _ptr(seed), _ptr(out), _stream()), "dropout_scale")
    return out


def fno_block_tail(x, skip_w, spec_ws, bias, modes, norm, relu_out=True, drop_p=0.0, seed=None, direct_grads=False):
    """One fused Fourier layer with the tail of the RNO regressor's layers (rno.py:92-106, channels-first):
    y = relu(specconv(drop(x)) + skip_w x + bias).  `seed`: draw_dropout_seed() (required when drop_p > 0).  The ReLU, its
    derivative, the dropout mask (regenerated in the backward) and the accumulation of the two branches' input gradients
    all happen inside the engine kernels (fno_model_forward_tail / fno_model_backward_tail)."""
    sw = [torch.view_as_real(t) if t.is_complex() else t for t in spec_ws]
    direct = _direct_views(direct_grads, spec_ws)
    if drop_p > 0 and seed is None:
        raise ValueError("fno_block_tail: drop_p > 0 needs a seed (draw_dropout_seed)")
    cfg = (1, tuple(int(m) for m in modes), norm, 0, direct, (bool(relu_out), float(drop_p), seed if drop_p > 0 else None))
    return _FNOBlocksFn.apply(cfg, x, bias, skip_w, *sw)


def fno_blocks(x, skip_ws, spec_ws, bias, modes, norm, gelu_mask=0, direct_grads=False):
    """Stack of fused Fourier layers (include/fnoengine.h, block stacks): per layer one spectral
    convolution (corner weights `spec_ws`, layer-major, real view (C, C, m.., 2)), one 1x1 convolution
    (`skip_ws[l]`, (C, C) or (C, C, 1..)) and one bias row of `bias` (L, C); GELU after layer l iff bit l
    of `gelu_mask`.  Returns (B, C, ...); differentiable w.r.t. x and every parameter."""
    sw = [torch.view_as_real(t) if t.is_complex() else t for t in spec_ws]
    direct = _direct_views(direct_grads, spec_ws, last_dim=x.shape[-1])       # backward WRITES dL/dW of the spectral weights into their existing .grad storage
    cfg = (len(skip_ws), tuple(int(m) for m in modes), norm, int(gelu_mask), direct)
    return _FNOBlocksFn.apply(cfg, x, bias, *skip_ws, *sw)


# ----------------------------------------------------------------------------
# fan-out of Fourier layers over one input (RNO cell: f1, f3, f5, f7 on x; f2, f4, f8 on h)
# ----------------------------------------------------------------------------
FANOUT_MAX = 4


class _FourierFanoutFn(torch.autograd.Function):
    """Tensor arguments: x, then skip_w[n], bias[n], spec_w[n * ncorner] (member-major).  Returns n tensors."""

    @staticmethod
    def forward(ctx, cfg, x, *rest):
        n, modes, norm = cfg[:3]
        ctx.direct = cfg[3] if len(cfg) > 3 else None
        _require_cuda(x, "x")
        x = x.contiguous()
        dims = tuple(x.shape[2:])
        ndim = len(dims)
        nc = 2 ** (ndim - 1)
        skip_ws = [t.contiguous() for t in rest[:n]]
        biases = [t.contiguous() for t in rest[n:2 * n]]
        spec_ws, planes = _weights_ready(rest[2 * n:])
        ctx.planes = planes
        assert len(spec_ws) == n * nc and n <= FANOUT_MAX
        for t in skip_ws + biases + spec_ws:
            _require_cuda(t, "parameter")
        B, c = x.shape[0], x.shape[1]
        _check_corner_weights(spec_ws, c, c, modes, planes, "fourier_fanout")
        L = _lib.lib()
        plan = model_plan(ndim, 0, c, 0, 0, FANOUT_MAX, dims, modes, norm, 0, x.device, weight_planes=planes)
        prm = _lib.FnoModelParams()
        for j in range(n):
            prm.skip_w[j] = skip_ws[j].data_ptr()
            for k in range(nc):
                prm.spec_w[j][k] = spec_ws[j * nc + k].data_ptr()
        ys = [torch.empty_like(x) for _ in range(n)]
        bptr = (C.c_void_p * n)(*[b.data_ptr() for b in biases])
        yptr = (C.c_void_p * n)(*[y.data_ptr() for y in ys])
        saved = _bytes(L.fno_fanout_saved_bytes(plan, B, n), x.device)
        nws = L.fno_fanout_workspace_bytes(plan, B, n)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.fno_fanout_forward(plan, B, n, C.byref(prm), bptr, _ptr(x), yptr, _ptr(saved), _ptr(ws), nws,
                                            _stream()), "fanout_forward")
        ctx.plan, ctx.B, ctx.n, ctx.nc = plan, B, n, nc
        ctx.save_for_backward(x, saved, *skip_ws, *spec_ws)
        ctx.bias_shapes = [b.shape for b in rest[n:2 * n]]
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        sv = ctx.saved_tensors
        x, saved = sv[:2]
        n, nc = ctx.n, ctx.nc
        skip_ws, spec_ws = list(sv[2:2 + n]), list(sv[2 + n:])
        dys = [torch.zeros_like(x) if d is None else d.contiguous() for d in dys]
        L = _lib.lib()
        prm, grd = _lib.FnoModelParams(), _lib.FnoModelGrads()
        g_skip = [torch.empty_like(t) for t in skip_ws]
        g_spec = ctx.direct if ctx.direct is not None else _fresh_grads(spec_ws, ctx.planes)     # direct: the weights' own .grad storage
        g_bias = [torch.empty(x.shape[1], dtype=torch.float32, device=x.device) for _ in range(n)]
        for j in range(n):
            prm.skip_w[j], grd.skip_w[j] = skip_ws[j].data_ptr(), g_skip[j].data_ptr()
            for k in range(nc):
                prm.spec_w[j][k], grd.spec_w[j][k] = spec_ws[j * nc + k].data_ptr(), g_spec[j * nc + k].data_ptr()
        dyptr = (C.c_void_p * n)(*[d.data_ptr() for d in dys])
        dbptr = (C.c_void_p * n)(*[b.data_ptr() for b in g_bias])
        dx = torch.empty_like(x)
        nws = L.fno_fanout_workspace_bytes(ctx.plan, ctx.B, n)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.fno_fanout_backward(ctx.plan, ctx.B, n, C.byref(prm), _ptr(x), dyptr, _ptr(saved), C.byref(grd), dbptr,
                                             _ptr(dx), _ptr(ws), nws, _stream()), "fanout_backward")
        g_bias = [g.view(sh) for g, sh in zip(g_bias, ctx.bias_shapes)]
        if ctx.direct is not None:
            _notify_direct(ctx.direct)
        return ((None, dx if ctx.needs_input_grad[1] else None) + tuple(g_skip) + tuple(g_bias)
                + ((None,) * len(g_spec) if ctx.direct is not None else tuple(g_spec)))


def fanout_supported(x, n, modes, norm):
    if not (1 <= n <= FANOUT_MAX and blocks_supported(x)):
        return False
    return model_plan_available(x.dim() - 2, 0, x.shape[1], 0, 0, FANOUT_MAX, tuple(x.shape[2:]), tuple(int(m) for m in modes),
                                norm, 0, x.device)


def fourier_fanout(x, skip_ws, biases, spec_ws, modes, norm, direct_grads=False):
    """[SpecConv_j(x) + conv1x1(x; skip_ws[j]) + biases[j] for j < n]: n <= 4 Fourier layers (rno.py:215-228) on ONE input,
    whose forward transforms run once and whose input gradients are summed inside the backward kernels
    (include/fnoengine.h, fno_fanout_*).  spec_ws is member-major: member j's corner weights at [j * ncorner, (j+1) * ncorner)."""
    n = len(skip_ws)
    cfg = (n, tuple(int(m) for m in modes), norm, _direct_views(direct_grads, spec_ws))
    return _FourierFanoutFn.apply(cfg, x, *skip_ws, *biases, *spec_ws)


# ----------------------------------------------------------------------------
# PINO residual loss (spectral Navier-Stokes vorticity residual + initial condition)
# ----------------------------------------------------------------------------
class _PinoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, u0, forcing, visc, t_interval):
        for t, name in ((u, "u"), (u0, "u0"), (forcing, "forcing"), (visc, "visc")):
            _require_cuda(t, name)
        B, n, n2, nt = u.shape
        if n != n2:
            raise RuntimeError(f"fnoengine pino_loss: square grids only (got {n} x {n2})")
        u_c, u0_c = u.contiguous(), u0.reshape(B, n, n).contiguous()
        f_c = forcing.reshape(n, n).contiguous()
        v_c = visc.reshape(B).contiguous()
        L = _lib.lib()
        nws = L.fno_pino_loss_workspace_bytes(B, n, nt)
        ws = _bytes(nws, u.device)
        losses = torch.empty(2, dtype=torch.float32, device=u.device)
        with torch.cuda.device(u.device):
            _lib.check(L.fno_pino_loss_forward(B, n, nt, _ptr(u_c), _ptr(u0_c), _ptr(f_c), _ptr(v_c), float(t_interval),
                                               _ptr(losses[0:1]), _ptr(losses[1:2]), _ptr(ws), nws, _stream()),
                       "pino_loss_forward")
        ctx.save_for_backward(u_c, u0_c, f_c, v_c, ws)
        ctx.meta = (B, n, nt, float(t_interval), nws, u.shape)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_ic, g_f):
        u_c, u0_c, f_c, v_c, ws = ctx.saved_tensors
        B, n, nt, t_interval, nws, shape = ctx.meta
        L = _lib.lib()
        du = torch.empty_like(u_c)
        gi = g_ic.contiguous().to(torch.float32).reshape(1)
        gf = g_f.contiguous().to(torch.float32).reshape(1)
        with torch.cuda.device(du.device):
            _lib.check(L.fno_pino_loss_backward(B, n, nt, _ptr(u_c), _ptr(u0_c), _ptr(f_c), _ptr(v_c), t_interval,
                                                _ptr(gi), _ptr(gf), _ptr(du), _ptr(ws), nws, _stream()),
                       "pino_loss_backward")
        return du.view(shape), None, None, None, None


def pino_loss(u, u0, forcing, visc, t_interval=1.0):
    """(loss_ic, loss_f) of Channelflow_PINO_loss / PINO_loss3d (libs/envs/diff_control_env.py:44-60):
    u (B, n, n, nt) model output, u0 (B, n, n), forcing (n, n) or (1, n, n, 1), visc (B,) = 1 / Re.
    Differentiable w.r.t. u.  n in {32, 64, 128} (one workgroup per plane, in-LDS FFTs) or 256 (row / column / row
    slab passes through HBM)."""
    return _PinoLossFn.apply(u, u0, forcing, visc, t_interval)


# ----------------------------------------------------------------------------
# channel-flow RHS and the physics-informed loss (libs/envs/control_env.py:429-530, 627-633)
# ----------------------------------------------------------------------------
class ChannelGrid:
    """The staggered channel grid the kernels need: sizes, uniform spacings dx, dz, viscosity nu and the wall-normal
    metrics y (Ny faces), ym (Ny-1 centres), yg (Ny+1 ghost-extended centres).  Packs the reciprocal spacings once on the
    host (fno_chanflow_pack_metrics) and keeps one device copy per GPU."""

    def __init__(self, Nx, Nz, dx, dz, y, ym, yg, nu):
        import numpy as np
        y, ym, yg = (np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1)) for a in (y, ym, yg))
        self.Nx, self.Ny, self.Nz = int(Nx), int(y.shape[0]), int(Nz)
        if ym.shape[0] != self.Ny - 1 or yg.shape[0] != self.Ny + 1:
            raise ValueError(f"channel grid: y has {self.Ny} faces, so ym needs {self.Ny - 1} and yg {self.Ny + 1} entries "
                             f"(got {ym.shape[0]}, {yg.shape[0]})")
        self.dx, self.dz, self.nu = float(dx), float(dz), float(nu)
        self.y, self.ym, self.yg = y, ym, yg
        self._packed = None
        self._dev = {}

    def desc(self):
        return _lib.FnoChanflowGrid(self.Nx, self.Ny, self.Nz, self.dx, self.dz, self.nu)

    def metrics(self, device):
        import numpy as np
        if self._packed is None:
            packed = np.zeros(3 * (self.Ny + 2), dtype=np.float64)
            dp = C.POINTER(C.c_double)
            _lib.check(_lib.lib().fno_chanflow_pack_metrics(self.Ny, self.y.ctypes.data_as(dp), self.ym.ctypes.data_as(dp),
                                                            self.yg.ctypes.data_as(dp), packed.ctypes.data_as(dp)),
                       "chanflow_pack_metrics")
            self._packed = packed
        if device not in self._dev:
            self._dev[device] = torch.from_numpy(self._packed).to(device)
        return self._dev[device]

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_dev"] = {}
        return st

    def _check_fields(self, U, V, W, who):
        B = U.shape[0]
        su, sv = (B, self.Nx, self.Ny + 1, self.Nz), (B, self.Nx, self.Ny, self.Nz)
        if tuple(U.shape) != su or tuple(W.shape) != su or tuple(V.shape) != sv:
            raise RuntimeError(f"fnoengine {who}: expected U, W {su} and V {sv}, got {tuple(U.shape)}, {tuple(W.shape)}, "
                               f"{tuple(V.shape)}")
        for t, n in ((U, "U"), (V, "V"), (W, "W")):
            if not t.is_cuda:
                raise RuntimeError(f"fnoengine {who}: `{n}` must live on the GPU (got {t.device}); the engine has no CPU path")


def chanflow_rhs(grid, U, V, W, dPdx):
    """Fu, Fv, Fw = NSControlEnvMatlab.compute_rhs_py(U, V, W, dPdx) (libs/envs/control_env.py:429-530) for a batch of
    fields: U, W (B, Nx, Ny+1, Nz), V (B, Nx, Ny, Nz), fp32 or fp64; dPdx a float or a (B,) tensor.  Not differentiable
    (the reference uses it under autograd only through pde_loss -> chanflow_pde_loss)."""
    grid._check_fields(U, V, W, "chanflow_rhs")
    if U.dtype not in (torch.float32, torch.float64) or V.dtype != U.dtype or W.dtype != U.dtype:
        raise RuntimeError("fnoengine chanflow_rhs: U, V, W must share one dtype, float32 or float64")
    U, V, W = U.contiguous(), V.contiguous(), W.contiguous()
    B = U.shape[0]
    dp, dflt = None, 0.0
    if torch.is_tensor(dPdx) and dPdx.numel() > 1:
        dp = dPdx.to(device=U.device, dtype=U.dtype).reshape(B).contiguous()
    else:
        dflt = float(dPdx)
    Fu, Fv, Fw = torch.empty_like(U), torch.empty_like(V), torch.empty_like(W)
    g = grid.desc()
    with torch.cuda.device(U.device):
        _lib.check(_lib.lib().fno_chanflow_rhs(C.byref(g), B, 0 if U.dtype == torch.float32 else 1, _ptr(grid.metrics(U.device)),
                                               _ptr(U), _ptr(V), _ptr(W), _ptr(dp), dflt, _ptr(Fu), _ptr(Fv), _ptr(Fw),
                                               _stream()), "chanflow_rhs")
    return Fu, Fv, Fw


class _ChanflowPdeLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grid, U, Vgt, V, W):
        B = U.shape[0]
        U, Vgt, V, W = U.contiguous(), Vgt.contiguous(), V.contiguous(), W.contiguous()
        L = _lib.lib()
        g = grid.desc()
        nws = L.fno_chanflow_pde_loss_workspace_bytes(C.byref(g), B)
        ws = _bytes(nws, U.device)
        loss = torch.empty(1, dtype=torch.float32, device=U.device)
        m = grid.metrics(U.device)
        with torch.cuda.device(U.device):
            _lib.check(L.fno_chanflow_pde_loss_forward(C.byref(g), B, _ptr(m), _ptr(U), _ptr(Vgt), _ptr(V), _ptr(W), _ptr(loss),
                                                       _ptr(ws), nws, _stream()), "chanflow_pde_loss_forward")
        ctx.save_for_backward(U, Vgt, V, W, ws, m)
        ctx.meta = (grid, B, nws)
        return loss[0]

    @staticmethod
    def backward(ctx, gl):
        U, Vgt, V, W, ws, m = ctx.saved_tensors
        grid, B, nws = ctx.meta
        g = grid.desc()
        dV = torch.empty_like(V)
        glc = gl.contiguous().to(torch.float32).reshape(1)
        with torch.cuda.device(V.device):
            _lib.check(_lib.lib().fno_chanflow_pde_loss_backward(C.byref(g), B, _ptr(m), _ptr(U), _ptr(Vgt), _ptr(V), _ptr(W),
                                                                 _ptr(glc), _ptr(dV), _ptr(ws), nws, _stream()),
                       "chanflow_pde_loss_backward")
        return None, None, None, dV, None


def chanflow_pde_loss(grid, U, Vgt, V, W):
    """sum_b ||Fu(U,Vgt,W) - Fu(U,V,W)|| + ||Fv ..|| + ||Fw ..||: NSControlEnvMatlab.pde_loss (libs/envs/control_env.py:627-633)
    summed over the batch as the training loop does (run_pde_observers.py:226-230).  fp32 fields on the GPU;
    differentiable w.r.t. V (the predicted wall-normal velocity); the pressure gradient cancels and is not an argument."""
    grid._check_fields(U, V, W, "chanflow_pde_loss")
    if tuple(Vgt.shape) != tuple(V.shape):
        raise RuntimeError(f"fnoengine chanflow_pde_loss: Vgt {tuple(Vgt.shape)} must match V {tuple(V.shape)}")
    for t, n in ((U, "U"), (Vgt, "Vgt"), (V, "V"), (W, "W")):
        _require_cuda(t, n)
    return _ChanflowPdeLossFn.apply(grid, U, Vgt, V, W)


# ----------------------------------------------------------------------------
# RNO cell gates (neuralop/models/rno.py:254-260)
# ----------------------------------------------------------------------------
def gates_supported(*tensors):
    t0 = tensors[0]
    return all(t.is_cuda and t.dtype == torch.float32 and t.shape == t0.shape for t in tensors) and t0.numel() % 4 == 0


class _RnoResetGateFn(torch.autograd.Function):
    """rh = sigmoid(a3 + a4 + b2) * h."""

    @staticmethod
    def forward(ctx, a3, a4, b2, h):
        if not (a3.shape == a4.shape == h.shape):
            raise RuntimeError(f"fnoengine rno_reset_gate: operands of shapes {tuple(a3.shape)}, {tuple(a4.shape)}, {tuple(h.shape)}")
        a3, a4, h = a3.contiguous(), a4.contiguous(), h.contiguous()
        r, rh = torch.empty_like(h), torch.empty_like(h)
        _lib.check(_lib.lib().fno_rno_reset_gate_forward(h.numel(), _ptr(a3), _ptr(a4), _ptr(b2), _ptr(h), _ptr(r), _ptr(rh),
                                                         _stream()), "rno_reset_gate_forward")
        ctx.save_for_backward(r, h)
        return rh

    @staticmethod
    def backward(ctx, d_rh):
        r, h = ctx.saved_tensors
        L = _lib.lib()
        d_rh = d_rh.contiguous()
        ds, dh = torch.empty_like(h), torch.empty_like(h)
        part = torch.empty(L.fno_rno_gate_partials(), dtype=torch.float64, device=h.device)
        _lib.check(L.fno_rno_reset_gate_backward(h.numel(), _ptr(d_rh), _ptr(r), _ptr(h), _ptr(ds), _ptr(dh), _ptr(part),
                                                 _stream()), "rno_reset_gate_backward")
        return ds, ds, part.sum().float().reshape(()), dh


class _RnoOutputGateFn(torch.autograd.Function):
    """h_new = (1 - sigmoid(a1 + a2 + b1)) * h + sigmoid(a7 + a8 + b4) * selu(a5 + a6 + b3)."""

    @staticmethod
    def forward(ctx, a1, a2, b1, a7, a8, b4, a5, a6, b3, h):
        if any(t.shape != h.shape for t in (a1, a2, a7, a8, a5, a6)):
            raise RuntimeError(f"fnoengine rno_output_gate: operands must all have the state's shape {tuple(h.shape)}")
        a1, a2, a7, a8, a5, a6, h = [t.contiguous() for t in (a1, a2, a7, a8, a5, a6, h)]
        z, z2, s3, hn = (torch.empty_like(h) for _ in range(4))
        _lib.check(_lib.lib().fno_rno_output_gate_forward(h.numel(), _ptr(a1), _ptr(a2), _ptr(b1), _ptr(a7), _ptr(a8), _ptr(b4),
                                                          _ptr(a5), _ptr(a6), _ptr(b3), _ptr(h), _ptr(z), _ptr(z2), _ptr(s3),
                                                          _ptr(hn), _stream()), "rno_output_gate_forward")
        ctx.save_for_backward(z, z2, s3, h)
        return hn

    @staticmethod
    def backward(ctx, g):
        z, z2, s3, h = ctx.saved_tensors
        L = _lib.lib()
        g = g.contiguous()
        d1, d7, d3, dh = (torch.empty_like(h) for _ in range(4))
        P = L.fno_rno_gate_partials()
        part = torch.empty(3, P, dtype=torch.float64, device=h.device)
        _lib.check(L.fno_rno_output_gate_backward(h.numel(), _ptr(g), _ptr(z), _ptr(z2), _ptr(s3), _ptr(h), _ptr(d1), _ptr(d7),
                                                  _ptr(d3), _ptr(dh), _ptr(part), _stream()), "rno_output_gate_backward")
        db = part.sum(dim=1).float()
        return d1, d1, db[0].reshape(()), d7, d7, db[1].reshape(()), d3, d3, db[2].reshape(()), dh


def rno_reset_gate(a3, a4, b2, h):
    return _RnoResetGateFn.apply(a3, a4, b2, h)


def rno_output_gate(a1, a2, b1, a7, a8, b4, a5, a6, b3, h):
    return _RnoOutputGateFn.apply(a1, a2, b1, a7, a8, b4, a5, a6, b3, h)


# ----------------------------------------------------------------------------
# pointwise channel mix + bias + residual add (the Conv1d(k=1) beside a spectral convolution)
# ----------------------------------------------------------------------------
def pointwise_supported(x):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3):
        return False
    pw = 1
    for s in x.shape[2:]:
        pw *= s
    return x.shape[1] in (32, 64) and pw % 128 == 0


class _PointwiseAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, addend):
        _require_cuda(x, "x")
        x = x.contiguous()
        B, Cc = x.shape[0], x.shape[1]
        pw = x.numel() // (B * Cc)
        w2 = w.reshape(Cc, Cc).contiguous()
        if addend is not None and addend.shape != x.shape:
            raise RuntimeError(f"fnoengine pointwise_conv_add: addend of shape {tuple(addend.shape)} for x of shape {tuple(x.shape)}")
        if bias is not None and bias.numel() != Cc:
            raise RuntimeError(f"fnoengine pointwise_conv_add: bias of {bias.numel()} elements for {Cc} channels")
        add_c = addend.contiguous() if addend is not None else None
        b_c = bias.contiguous() if bias is not None else None
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().fno_pointwise_forward(B, Cc, pw, _ptr(x), _ptr(w2), _ptr(b_c), _ptr(add_c), 0, _ptr(y),
                                                        _stream()), "pointwise_forward")
        ctx.save_for_backward(x, w2)
        ctx.meta = (B, Cc, pw, w.shape, bias is not None, addend is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        B, Cc, pw, wshape, has_b, has_add = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w2)
        db = torch.empty(Cc, dtype=torch.float32, device=x.device) if has_b else None
        nws = L.fno_pointwise_workspace_bytes(Cc)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.fno_pointwise_backward(B, Cc, pw, _ptr(x), _ptr(w2), _ptr(dy), None, 0, _ptr(dx), _ptr(dw), _ptr(db),
                                                _ptr(ws), nws, _stream()), "pointwise_backward")
        return dx, dw.view(wshape), db, (dy if has_add else None)


def pointwise_conv_add(x, w, bias=None, addend=None):
    """y = conv1x1(x; w) + bias + addend  (x, addend (B, C, ...), w (C, C[, 1..]), C in {32, 64}) in one engine
    kernel each way; the gradient of `addend` is the incoming gradient itself."""
    return _PointwiseAddFn.apply(x, w, bias, addend)


class _PointwisePerSampleFn(torch.autograd.Function):
    """y[b] = conv1x1(x[b]; w) + bias[b]: the channel mix with a PER-SAMPLE bias row (B, C) - the Re-conditioning affine
    `B x + A re + bias` of the PINO observers (pinobserver.py:41-59), whose per-sample code `A re` would otherwise cost a
    broadcast-add pass over the whole tensor (and a reduction pass in backward).  One fno_pointwise_* call per sample."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _require_cuda(x, "x")
        x = x.contiguous()
        B, Cc = x.shape[0], x.shape[1]
        pw = x.numel() // (B * Cc)
        w2 = w.reshape(Cc, Cc).contiguous()
        bc = bias.contiguous()
        y = torch.empty_like(x)
        L = _lib.lib()
        with torch.cuda.device(x.device):
            for b in range(B):
                _lib.check(L.fno_pointwise_forward(1, Cc, pw, _ptr(x[b]), _ptr(w2), _ptr(bc[b]), None, 0, _ptr(y[b]), _stream()),
                           "pointwise_forward")
        ctx.save_for_backward(x, w2)
        ctx.meta = (B, Cc, pw, w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        B, Cc, pw, wshape = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dws = torch.empty((B, Cc, Cc), dtype=torch.float32, device=x.device)
        dbs = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
        nws = L.fno_pointwise_workspace_bytes(Cc)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            for b in range(B):
                _lib.check(L.fno_pointwise_backward(1, Cc, pw, _ptr(x[b]), _ptr(w2), _ptr(dy[b]), None, 0,
                                                    _ptr(dx[b]) if dx is not None else None, _ptr(dws[b]), _ptr(dbs[b]), _ptr(ws), nws,
                                                    _stream()), "pointwise_backward")
        return dx, dws.sum(0).view(wshape), dbs


def pointwise_conv_per_sample_bias(x, w, bias):
    """y[b] = conv1x1(x[b]; w) + bias[b] with bias (B, C); x (B, C, ...), C in {32, 64}."""
    return _PointwisePerSampleFn.apply(x, w, bias)


class _LiftingPerSampleFn(torch.autograd.Function):
    """lifting with a per-sample bias row (B, C): one fno_lifting_* call per sample (see _PointwisePerSampleFn)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _require_cuda(x, "x")
        if x.requires_grad:
            raise RuntimeError("fnoengine lifting: the input field is data (no gradient is produced for it)")
        x = x.contiguous()
        B, cin = x.shape[0], x.shape[1]
        cout = w.shape[0]
        pw = x.numel() // (B * cin)
        wc = w.reshape(cout, cin).contiguous()
        bc = bias.contiguous()
        y = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        with torch.cuda.device(x.device):
            for b in range(B):
                _lib.check(L.fno_lifting_forward(1, cin, cout, pw, _ptr(x[b]), _ptr(wc), _ptr(bc[b]), _ptr(y[b]), _stream()),
                           "lifting_forward")
        ctx.save_for_backward(x)
        ctx.meta = (B, cin, cout, pw, w.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, cin, cout, pw, wshape = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        dws = torch.empty((B, cout, cin), dtype=torch.float32, device=x.device)
        dbs = torch.empty((B, cout), dtype=torch.float32, device=x.device)
        nws = L.fno_lifting_workspace_bytes(cout)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            for b in range(B):
                _lib.check(L.fno_lifting_backward(1, cin, cout, pw, _ptr(x[b]), _ptr(dy[b]), _ptr(dws[b]), _ptr(dbs[b]), _ptr(ws), nws,
                                                  _stream()), "lifting_backward")
        return None, dws.sum(0).view(wshape), dbs


def lifting_per_sample_bias(x, w, bias):
    """y[b] = conv1x1(x[b]; w) + bias[b]: x (B, Cin <= 4, ...) data, w (C, Cin), bias (B, C)."""
    return _LiftingPerSampleFn.apply(x, w, bias)


class _ZeroLastPadsFn(torch.autograd.Function):
    """Zero the first p0 and last p1 entries of the last dimension IN PLACE, forward and backward: makes a tensor that was
    computed on a zero-padded INPUT equal to the zero-padded tensor the reference builds with F.pad after the fact
    (pinobserver.py:208-213), without the pad copy (forward) or the slice copy (backward)."""

    @staticmethod
    def forward(ctx, y, p0, p1):
        ctx.pads = (p0, p1)
        ctx.mark_dirty(y)
        if p0 > 0:
            y[..., :p0].zero_()
        if p1 > 0:
            y[..., y.shape[-1] - p1:].zero_()
        return y

    @staticmethod
    def backward(ctx, g):
        p0, p1 = ctx.pads
        g = g.contiguous()          # the block stack's input gradient: a fresh tensor with no other consumer
        if p0 > 0:
            g[..., :p0].zero_()
        if p1 > 0:
            g[..., g.shape[-1] - p1:].zero_()
        return g, None, None


def zero_last_pads_(y, p0, p1):
    return _ZeroLastPadsFn.apply(y, int(p0), int(p1))


# ----------------------------------------------------------------------------
# one layer of the observer stacks:  y = SpectralConv(a) + Conv1d_{k=1}(a) + bias,  a = gelu(u) or u
# ----------------------------------------------------------------------------
def spectral_layer_supported(u, n_spec_weights, modes, norm, weight_last_extent, input_gelu):
    """True when spectral_pointwise_layer can run this shape (result cached per shape)."""
    if not pointwise_supported(u) or n_spec_weights != 2 ** (u.dim() - 3):
        return False
    key = ("layer", u.dim() - 2, u.shape[1], tuple(u.shape[2:]), tuple(modes), weight_last_extent, norm, u.device.index, bool(input_gelu))
    ok = _spec_plans.get(key)
    if ok is None:
        try:
            spec_plan(u.dim() - 2, u.shape[1], u.shape[1], tuple(u.shape[2:]), modes, weight_last_extent, norm, u.device, input_gelu)
            ok = True
        except RuntimeError:
            ok = False
        _spec_plans[key] = ok
    return ok


class _SpectralLayerFn(torch.autograd.Function):
    """forward: sp = fno_spec_forward(u) [gelu on load], y = fno_pointwise_forward(u, addend = sp) [gelu on load].
    backward: (d_a from the spectral branch, dW_spec) = fno_spec_backward(dy); fno_pointwise_backward adds it to W^T dy,
    applies gelu'(u) and writes du: no separate activation, activation-derivative or gradient-accumulation pass."""

    @staticmethod
    def forward(ctx, u, w, bias, modes, norm, wle, input_gelu, direct, *spec_ws):
        _require_cuda(u, "u")
        u = u.contiguous()
        B, Cc = u.shape[0], u.shape[1]
        dims = tuple(u.shape[2:])
        pw = u.numel() // (B * Cc)
        ws_list, planes = _weights_ready(spec_ws)
        ctx.planes = planes
        L = _lib.lib()
        plan = spec_plan(len(dims), Cc, Cc, dims, modes, wle, norm, u.device, input_gelu, weight_planes=planes)
        sp = torch.empty_like(u)
        xhat = _bytes(L.fno_spec_xhat_bytes(plan, B), u.device)
        nws = L.fno_spec_workspace_bytes(plan, B)
        ws = _bytes(nws, u.device)
        wp = (C.c_void_p * 4)(*[t.data_ptr() for t in ws_list] + [0] * (4 - len(ws_list)))
        w2 = w.reshape(Cc, Cc).contiguous()
        b_c = bias.contiguous() if bias is not None else None
        y = torch.empty_like(u)
        with torch.cuda.device(u.device):
            _lib.check(L.fno_spec_forward(plan, B, _ptr(u), wp, None, _ptr(sp), _ptr(xhat), _ptr(ws), nws, _stream()), "spec_forward")
            _lib.check(L.fno_pointwise_forward(B, Cc, pw, _ptr(u), _ptr(w2), _ptr(b_c), _ptr(sp), 1 if input_gelu else 0, _ptr(y),
                                               _stream()), "pointwise_forward")
        ctx.plan, ctx.meta, ctx.direct = plan, (B, Cc, pw, w.shape, bias is not None, bool(input_gelu)), direct
        ctx.save_for_backward(u, w2, xhat, *ws_list)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, w2, xhat, *ws_list = ctx.saved_tensors
        B, Cc, pw, wshape, has_b, input_gelu = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        need_du = ctx.needs_input_grad[0]
        need_dws = any(ctx.needs_input_grad[8:])
        direct = ctx.direct if need_dws else None
        dws = (direct if direct is not None else _fresh_grads(ws_list, ctx.planes)) if need_dws else None
        da = torch.empty_like(u) if need_du else None           # gradient reaching gelu(u) through the spectral branch
        du = torch.empty_like(u) if need_du else None
        dw = torch.empty_like(w2)
        db = torch.empty(Cc, dtype=torch.float32, device=u.device) if has_b else None
        nws = L.fno_spec_workspace_bytes(ctx.plan, B)
        ws = _bytes(nws, u.device)
        nws2 = L.fno_pointwise_workspace_bytes(Cc)
        ws2 = _bytes(nws2, u.device)
        dwp = (C.c_void_p * 4)(*[t.data_ptr() for t in dws] + [0] * (4 - len(dws))) if need_dws else None
        with torch.cuda.device(u.device):
            _lib.check(L.fno_spec_backward(ctx.plan, B, _ptr(dy), _ptr(xhat), None, _ptr(da), dwp, None, _ptr(ws), nws, _stream()),
                       "spec_backward")
            _lib.check(L.fno_pointwise_backward(B, Cc, pw, _ptr(u), _ptr(w2), _ptr(dy), _ptr(da), 1 if input_gelu else 0, _ptr(du),
                                                _ptr(dw), _ptr(db), _ptr(ws2), nws2, _stream()), "pointwise_backward")
        gw = (None,) * len(ws_list) if (direct is not None or not need_dws) else tuple(dws)
        if direct is not None:
            _notify_direct(direct)
        return (du, dw.view(wshape), db, None, None, None, None, None) + gw


def spectral_pointwise_layer(u, spec_weights, modes, norm, w, bias, input_gelu=False, weight_last_extent=None,
                             direct_grads=False):
    """y = SpectralConv(a) + Conv1d_{k=1}(a; w) + bias with a = gelu(u) if input_gelu else u: one layer of the observer
    stacks (libs/models/pino_models/pinobserver.py:221-226) with the PREVIOUS layer's activation applied while u is
    loaded, so a stack is chained on pre-activation tensors.  Check spectral_layer_supported() first."""
    sw = [torch.view_as_real(t) if t.is_complex() else t for t in spec_weights]
    wle = int(weight_last_extent) if weight_last_extent is not None else int(sw[0].shape[-2])
    direct = _direct_views(direct_grads, spec_weights, last_dim=u.shape[-1])
    return _SpectralLayerFn.apply(u, w, bias, tuple(int(m) for m in modes), norm, wle, bool(input_gelu), direct, *sw)


# ----------------------------------------------------------------------------
# projection head  y = W2 gelu(W1 x + b1) + b2  on (B, C, ...) tensors
# ----------------------------------------------------------------------------
PROJ_MAXCO = 4        # k_projection.h


def projection_supported(x, hidden, cout, act="gelu"):
    return (pointwise_supported(x) and hidden in ((128, 256) if act == "gelu" else (256,))
            and (cout == 1 or (act == "gelu" and 1 <= cout <= PROJ_MAXCO))
            and act in _ACT_CODES and _lib.lib().fno_get_gemm_mode() == 1)


_ACT_CODES = {"gelu": 0, "relu": 1}      # FNO_ACT_* (include/fnoengine.h)


class _ProjectionHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act=0):
        _require_cuda(x, "x")
        x = x.contiguous()
        B, Cc = x.shape[0], x.shape[1]
        pw = x.numel() // (B * Cc)
        hid, co = w1.shape[0], w2.shape[0]
        w1c, b1c = w1.reshape(hid, Cc).contiguous(), b1.contiguous()
        w2c, b2c = w2.reshape(co, hid).contiguous(), b2.contiguous()
        if b1c.numel() != hid or b2c.numel() != co:
            raise RuntimeError(f"fnoengine projection_head: biases of {b1c.numel()} / {b2c.numel()} elements for {hid} hidden and {co} output channels")
        y = torch.empty((B, co) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().fno_projection_forward_act(B, Cc, hid, co, pw, _ptr(x), _ptr(w1c), _ptr(b1c), _ptr(w2c),
                                                             _ptr(b2c), act, _ptr(y), _stream()), "projection_forward")
        ctx.save_for_backward(x, w1c, b1c, w2c)
        ctx.meta = (B, Cc, hid, pw, w1.shape, w2.shape, act, co)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1c, b1c, w2c = ctx.saved_tensors
        B, Cc, hid, pw, w1shape, w2shape, act, co = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        dw1, db1, dw2 = torch.empty_like(w1c), torch.empty_like(b1c), torch.empty_like(w2c)
        db2 = torch.empty(co, dtype=torch.float32, device=x.device)
        nws = L.fno_projection_workspace_bytes(Cc, hid)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.fno_projection_backward_act(B, Cc, hid, co, pw, _ptr(x), _ptr(w1c), _ptr(b1c), _ptr(w2c), _ptr(dy), act,
                                                     _ptr(dx), _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2), _ptr(ws), nws,
                                                     _stream()), "projection_backward")
        return dx, dw1.view(w1shape), db1, dw2.view(w2shape), db2, None


def projection_head(x, w1, b1, w2, b2, act="gelu"):
    """(B, C, ...) -> (B, Cout, ...): fc2(act(fc1(x))) with fc1.weight (hidden, C), fc2.weight (Cout, hidden); act 'gelu'
    (FNO projection, PINO observer tails; Cout <= 4: PlanePredHead's out_dim * plane_num, pinobserver.py:257-273) or 'relu'
    (RNO2d's regressor head, rno.py:171-175; hidden 256, Cout 1)."""
    return _ProjectionHeadFn.apply(x, w1, b1, w2, b2, _ACT_CODES[act])


# ----------------------------------------------------------------------------
# lifting layer  y = W x + b,  (B, Cin <= 4, ...) -> (B, C, ...)
# ----------------------------------------------------------------------------
def lifting_supported(x, c_out):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and 1 <= x.shape[1] <= 4 and c_out in (32, 64)):
        return False
    pw = 1
    for s in x.shape[2:]:
        pw *= s
    return pw % 128 == 0


class _LiftingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        _require_cuda(x, "x")
        if x.requires_grad:
            raise RuntimeError("fnoengine lifting: the input field is data (no gradient is produced for it)")
        x = x.contiguous()
        B, cin = x.shape[0], x.shape[1]
        cout = w.shape[0]
        pw = x.numel() // (B * cin)
        wc = w.reshape(cout, cin).contiguous()
        bc = bias.contiguous() if bias is not None else None
        y = torch.empty((B, cout) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().fno_lifting_forward(B, cin, cout, pw, _ptr(x), _ptr(wc), _ptr(bc), _ptr(y), _stream()),
                       "lifting_forward")
        ctx.save_for_backward(x)
        ctx.meta = (B, cin, cout, pw, w.shape, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        B, cin, cout, pw, wshape, has_b = ctx.meta
        L = _lib.lib()
        dy = dy.contiguous()
        dw = torch.empty(cout, cin, dtype=torch.float32, device=x.device)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if has_b else None
        nws = L.fno_lifting_workspace_bytes(cout)
        ws = _bytes(nws, x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.fno_lifting_backward(B, cin, cout, pw, _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), _ptr(ws), nws, _stream()),
                       "lifting_backward")
        return None, dw.view(wshape), db


def lifting(x, w, bias=None):
    """conv1x1 from <= 4 input channels: x (B, Cin, ...) data, w (C, Cin[, 1..]), bias (C)."""
    return _LiftingFn.apply(x, w, bias)
