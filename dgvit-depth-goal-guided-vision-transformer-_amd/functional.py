"""Autograd wrappers around the C ABI (include/dgvit_hip.h).

``got_encoder`` is the whole of GoT.forward (reference GoalFormer.py:156-171) as ONE autograd node whose
forward and backward are each one call into libdgvit_hip.so; ``linear`` is an nn.Linear (+ReLU) of the
SAC heads (got_sac_network.py:111-121, 226-234).  The ``op_*`` helpers expose single operators for
operator-level parity tests.  Tensors must be fp32 on a ROCm device; anything else raises.
"""
import ctypes
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import DgvitError, dgvit_config


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise DgvitError(f"{name}: tensor is on {t.device}; the DGViT HIP path needs a ROCm device (no CPU fallback)")
    if t.dtype != torch.float32:
        raise DgvitError(f"{name}: dtype {t.dtype} unsupported, fp32 only")
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _table(tensors: Sequence[Optional[torch.Tensor]]):
    """device-pointer table; a None entry (an unused slot, e.g. to_out of a projection-less attention) is a NULL pointer"""
    return (ctypes.c_void_p * len(tensors))(*[0 if t is None else t.data_ptr() for t in tensors])


def _empty_batch(shape, tensors):
    """The reference's behaviour for an empty batch (its torch ops accept 0-row tensors): an empty result, no kernel launch, and a
    backward that hands every live parameter a zero gradient of its own shape."""
    for t in tensors:
        if isinstance(t, torch.Tensor):
            _dev(t, "input")          # still a device-only path: CPU tensors are refused as everywhere else
    like = next(t for t in tensors if isinstance(t, torch.Tensor))
    out = like.new_zeros(shape)
    if torch.is_grad_enabled():
        live = [t for t in tensors if isinstance(t, torch.Tensor) and t.requires_grad]
        if live:
            out = out + sum(t.reshape(-1)[:1].sum() for t in live) * 0.0
    return out


def make_config(image, patch, dim, depth, heads, dim_head, mlp_dim) -> dgvit_config:
    return dgvit_config(int(image[0]), int(image[1]), int(patch[0]), int(patch[1]), int(dim), int(depth), int(heads),
                        int(dim_head), int(mlp_dim), 0, 0)


# ------------------------------------------------------------------------------------------------ encoder
_N_NONPARAM_INPUTS = 7   # img, goal, cfg_tuple, keep, seed, need_grad, grad_hook precede *params in _GoTEncoder.apply


def _flat_grads(params, needs, dev):
    """Gradient tensors for the parameters autograd asks for, as views of ONE flat fp32 buffer laid out in parameter-table
    order (4-float aligned slots, the layout of optim._Block and parallel.GradSync): autograd adopts them as .grad without
    a copy.  Frozen parameters (``requires_grad`` off: ``needs[i]`` False) get None -- the C ABI then skips their
    weight-gradient work.  The buffer is zero-filled so that the padding lanes between slots stay finite for whoever
    consumes the flat buffer as a whole (Adam moments, all-reduce)."""
    offs, off = [], 0
    needs = [bool(need) and p is not None for p, need in zip(params, needs)]
    for p, need in zip(params, needs):
        offs.append(off)
        if need:
            off += (p.numel() + 3) & ~3
    flat = torch.zeros(off, dtype=torch.float32, device=dev) if off else None
    return [flat[o:o + p.numel()].view_as(p) if need else None for o, p, need in zip(offs, params, needs)]


_EVENT_POOL = {}    # (device index, depth) -> [event handles]: re-recorded by every backward that has a gradient-ready hook


def _layer_events(dev, depth):
    """`depth` hipEvent_t handles (dgvit_event_create) for the gradient-ready events of one backward on `dev`."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), depth)
    evs = _EVENT_POOL.get(key)
    if evs is None:
        lib = _lib.load()
        evs = []
        with torch.cuda.device(dev):
            for _ in range(depth):
                h = ctypes.c_void_p()
                _lib.check(lib.dgvit_event_create(ctypes.byref(h)), "dgvit_event_create")
                evs.append(h.value)
        _EVENT_POOL[key] = evs
    return evs


def _grad_events(depth, evs):
    """the dgvit_grad_events argument (and the ctypes array it points to, which must outlive the call)"""
    table = (ctypes.c_void_p * depth)(*evs)
    return _lib.dgvit_grad_events(depth, table, None), table


def _call_grad_hook(hook, grads, depth, evs):
    """hook(flat gradient buffer, [(lo, hi) element range of block i's gradients for i = depth-1 .. 0], [their events]): the order in
    which the backward finishes them.  Blocks whose parameters are all frozen are left out."""
    live = [g for g in grads if g is not None]
    if not live:
        return
    flat = torch.empty(0, dtype=torch.float32, device=live[0].device).set_(live[0].untyped_storage(), 0, (live[0].untyped_storage().nbytes() // 4,))
    groups, events = [], []
    for i in range(depth - 1, -1, -1):
        gs = [g for g in grads[_lib.NUM_GLOBAL_PARAMS + _lib.PARAMS_PER_LAYER * i:_lib.NUM_GLOBAL_PARAMS + _lib.PARAMS_PER_LAYER * (i + 1)] if g is not None]
        if gs:
            groups.append((min(g.storage_offset() for g in gs), max(g.storage_offset() + g.numel() for g in gs)))
            events.append(evs[i])
    hook(flat, groups, events)


def _grad_table(grads):
    return (ctypes.c_void_p * len(grads))(*[0 if g is None else g.data_ptr() for g in grads])


def _take_workspace(ctx, what):
    """The forward's activation workspace is released by the first backward: a second one cannot be served."""
    ws = ctx.ws
    if ws is None:
        raise DgvitError(f"{what}: backward called a second time; the fused encoder frees its activation workspace after "
                         "the first backward (retain_graph=True is not supported) -- run the forward again")
    return ws


class _GoTEncoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, goal, cfg_tuple, keep, seed, need_grad, grad_hook, *params):
        lib = _lib.load()
        cfg = dgvit_config(*cfg_tuple)
        img, goal = _dev(img, "img"), _dev(goal, "goal")
        params = [None if p is None else _dev(p, f"param[{i}]") for i, p in enumerate(params)]   # (None: unused to_out slots, dgvit_hip.h)
        nparam = _lib.NUM_GLOBAL_PARAMS + _lib.PARAMS_PER_LAYER * cfg.depth
        if len(params) != nparam:
            raise DgvitError(f"expected {nparam} parameter tensors, got {len(params)}")
        if img.dim() != 3 or img.shape[1] != cfg.image_h or img.shape[2] != cfg.image_w:
            raise DgvitError(f"img must be (B, {cfg.image_h}, {cfg.image_w}), got {tuple(img.shape)}")
        B = img.shape[0]
        if goal.shape != (B, cfg.dim):
            raise DgvitError(f"goal must be ({B}, {cfg.dim}), got {tuple(goal.shape)}")
        nws = lib.dgvit_got_workspace_floats(ctypes.byref(cfg), B, int(need_grad))
        if nws < 0:
            _lib.check(-1, "dgvit_got_workspace_floats")
        ws = torch.empty(nws, dtype=torch.float32, device=img.device)
        feat = torch.empty(B, cfg.dim, dtype=torch.float32, device=img.device)
        # `seed` is a host int, or a 1-element int64 DEVICE tensor (graph capture: the kernel reads it at run time)
        seed_dev = seed if isinstance(seed, torch.Tensor) else None
        seed_val = 0 if seed_dev is not None else int(seed)
        with torch.cuda.device(img.device):
            rc = lib.dgvit_got_forward(ctypes.byref(cfg), _table(params), _ptr(img), _ptr(goal), _ptr(feat), _ptr(ws), nws, B,
                                       int(need_grad), float(keep), seed_val, _ptr(seed_dev), _stream())
        _lib.check(rc, "dgvit_got_forward")
        if need_grad:
            ctx.cfg_tuple, ctx.keep, ctx.seed, ctx.batch = cfg_tuple, float(keep), seed_val, B
            ctx.seed_dev = seed_dev
            ctx.ws = ws
            ctx.grad_hook = grad_hook
            ctx.save_for_backward(*params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = _lib.load()
        cfg = dgvit_config(*ctx.cfg_tuple)
        ws = _take_workspace(ctx, "dgvit_got_backward")
        params = list(ctx.saved_tensors)
        dfeat = _dev(dfeat, "dfeat")
        B = ctx.batch
        dev = dfeat.device
        # all parameter gradients are views of ONE flat buffer: autograd adopts them as .grad without a copy, and
        # parallel.GradSync all-reduces the buffer in place (one large RCCL call instead of 70 small tensors)
        grads = _flat_grads(params, ctx.needs_input_grad[_N_NONPARAM_INPUTS:], dev)
        dgoal = torch.empty(B, cfg.dim, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        nsc = lib.dgvit_got_backward_scratch_floats(ctypes.byref(cfg), B)
        scratch = torch.empty(nsc, dtype=torch.float32, device=dev)
        evs = _layer_events(dev, cfg.depth) if ctx.grad_hook is not None else None
        events, keep_alive = _grad_events(cfg.depth, evs) if evs else (None, None)
        with torch.cuda.device(dev):
            rc = lib.dgvit_got_backward_ev(ctypes.byref(cfg), _table(params), _grad_table(grads), _ptr(dfeat), _ptr(dgoal), _ptr(ws),
                                           ws.numel(), _ptr(scratch), nsc, B, ctx.keep, ctx.seed, _ptr(ctx.seed_dev), _stream(),
                                           ctypes.byref(events) if events is not None else None)
        _lib.check(rc, "dgvit_got_backward")
        ctx.ws = None
        if evs:     # the kernels are queued, not finished: the hook orders its own stream behind the events (parallel.GradSync)
            _call_grad_hook(ctx.grad_hook, grads, cfg.depth, evs)
        return (None, dgoal, None, None, None, None, None, *grads)


def got_encoder(img, goal, cfg_tuple, params, dropout_keep=1.0, dropout_seed=0, grad_hook=None):
    """feat (B, D) = GoT.forward(img (B,H,W), goal (B,D)); params in the table order of dgvit_hip.h.
    ``grad_hook(flat, ranges, events)``: called inside the backward, right after its kernels are queued, with the flat gradient buffer,
    the element range of every transformer block's gradients (last block first) and the HIP event recorded where that range is final
    (include/dgvit_hip.h: dgvit_grad_events) -- parallel.GradSync(overlap=True) starts the blocks' all-reduces there."""
    if img.shape[0] == 0:
        return _empty_batch((0, int(cfg_tuple[4])), [img, goal, *params])
    need_grad = torch.is_grad_enabled() and (goal.requires_grad or any(p is not None and p.requires_grad for p in params))
    return _GoTEncoder.apply(img, goal, tuple(cfg_tuple), dropout_keep, dropout_seed, need_grad, grad_hook, *params)


# ------------------------------------------------------------------------------------------------ CNN feature stack
class _CnnStack(torch.autograd.Function):
    """conv1-relu-conv2-relu-conv3-relu-avgpool of QNetwork / GaussianPolicy (got_sac_network.py:151-155) as one node."""

    @staticmethod
    def forward(ctx, img, need_grad, *params):
        lib = _lib.load()
        img = _dev(img, "img")
        params = [_dev(p, f"conv param[{i}]") for i, p in enumerate(params)]
        if img.dim() != 3:
            raise DgvitError(f"img must be (B, H, W), got {tuple(img.shape)}")
        B, H, W = img.shape
        nws = lib.dgvit_cnn_workspace_floats(B, H, W)
        nsc = lib.dgvit_cnn_forward_scratch_floats(B, H, W)
        if nws < 0 or nsc < 0:
            _lib.check(-1, "dgvit_cnn_workspace_floats")
        ws = torch.empty(nws, dtype=torch.float32, device=img.device)
        scratch = torch.empty(nsc, dtype=torch.float32, device=img.device)
        feat = torch.empty(B, 256, dtype=torch.float32, device=img.device)
        with torch.cuda.device(img.device):
            rc = lib.dgvit_cnn_forward(_ptr(img), _table(params), _ptr(feat), _ptr(ws), nws, _ptr(scratch), nsc, B, H, W, _stream())
        _lib.check(rc, "dgvit_cnn_forward")
        if need_grad:
            ctx.ws = ws
            ctx.save_for_backward(img, *params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = _lib.load()
        ws = _take_workspace(ctx, "dgvit_cnn_backward")
        img, *params = ctx.saved_tensors
        dfeat = _dev(dfeat, "dfeat")
        B, H, W = img.shape
        grads = [torch.empty_like(p) for p in params]
        nsc = lib.dgvit_cnn_backward_scratch_floats(B, H, W)
        scratch = torch.empty(nsc, dtype=torch.float32, device=img.device)
        with torch.cuda.device(img.device):
            rc = lib.dgvit_cnn_backward(_ptr(img), _table(params), _table(grads), _ptr(dfeat), _ptr(ws), ws.numel(),
                                        _ptr(scratch), nsc, B, H, W, _stream())
        _lib.check(rc, "dgvit_cnn_backward")
        ctx.ws = None
        return (None, None, *grads)


def cnn_features(img, conv_params):
    """(B, H, W) frames -> (B, 256) pooled features; conv_params = [w1, b1, w2, b2, w3, b3] (reference layouts)."""
    if img.shape[0] == 0:
        return _empty_batch((0, 256), [img, *conv_params])
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in conv_params)
    return _CnnStack.apply(img, need_grad, *conv_params)


# ------------------------------------------------------------------------------------------------ head Linear
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act):
        lib = _lib.load()
        x, w = _dev(x, "x"), _dev(w, "weight")
        b = None if b is None else _dev(b, "bias")
        if x.dim() != 2 or w.dim() != 2 or x.shape[1] != w.shape[1]:
            raise DgvitError(f"linear: x {tuple(x.shape)} does not match weight {tuple(w.shape)}")
        M, K = x.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.dgvit_linear_forward(_ptr(x), _ptr(w), _ptr(b), _ptr(y), M, N, K, int(act), _stream())
        _lib.check(rc, "dgvit_linear_forward")
        ctx.act, ctx.has_bias = int(act), b is not None
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        dy = _dev(dy, "dy")
        M, K = x.shape
        N = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(N, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        nsc = lib.dgvit_linear_backward_scratch_floats(M, N, K)
        scratch = torch.empty(nsc, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.dgvit_linear_backward(_ptr(dy), _ptr(x), _ptr(w), _ptr(y), _ptr(dx), _ptr(dw), _ptr(db), _ptr(scratch), nsc,
                                           M, N, K, ctx.act, _stream())
        _lib.check(rc, "dgvit_linear_backward")
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    """y = x W^T + b, optionally ReLU'd, on the MFMA GEMM."""
    if isinstance(x, torch.Tensor) and x.dim() == 2 and x.shape[0] == 0 and isinstance(weight, torch.Tensor):
        return _empty_batch((0, weight.shape[0]), [x, weight, bias])
    return _Linear.apply(x, weight, bias, 1 if relu else 0)


# ------------------------------------------------------------------------------------------------ stand-alone RMSNorm
class _RmsNorm(torch.autograd.Function):
    """F.normalize(x, dim=-1) * sqrt(D) * g on the last dimension (GoalFormer.py:120-122) for a stand-alone RMSNorm module."""

    @staticmethod
    def forward(ctx, x, g):
        x2 = _dev(x, "x").reshape(-1, x.shape[-1])
        g = _dev(g, "g")
        y = op_rmsnorm_fwd(x2, g)
        ctx.save_for_backward(x2, g)
        ctx.shape = x.shape
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, g = ctx.saved_tensors
        dx, dg = op_rmsnorm_bwd(_dev(dy, "dy").reshape(x2.shape), x2, g)
        return dx.reshape(ctx.shape), dg


def rms_norm(x, g):
    return _RmsNorm.apply(x, g)


# ------------------------------------------------------------------------------------------------ fused MLP heads
def _mlp_desc(batch, xs, n1, n2, n3, towers, heads3):
    d = _lib.dgvit_mlp_desc()
    d.batch, d.nseg, d.n1, d.n2, d.n3, d.towers, d.heads3 = batch, len(xs), n1, n2, n3, towers, heads3
    for i, x in enumerate(xs):
        d.kx[i], d.ldx[i] = x.shape[1], x.stride(0)
    return d


class _MlpHead(torch.autograd.Function):
    """y[t, j] = W3[t][j] relu(W2[t] relu(W1[t] cat(xs) + b1[t]) + b2[t]) + b3[t][j] as one HIP launch forward and one backward
    (dgvit_mlp_head_forward / _backward): the SAC heads of got_sac_network.py:114-121, 230-235, 433-435."""

    @staticmethod
    def forward(ctx, nseg, towers, heads3, *tensors):
        lib = _lib.load()
        xs = [_dev(t, f"head input {i}") for i, t in enumerate(tensors[:nseg])]
        params = [_dev(t, f"head parameter {i}") for i, t in enumerate(tensors[nseg:])]
        per = 4 + 2 * heads3
        if len(params) != towers * per:
            raise DgvitError(f"mlp_head: expected {towers * per} parameter tensors, got {len(params)}")
        B = xs[0].shape[0]
        n1, n2, n3 = params[0].shape[0], params[2].shape[0], params[4].shape[0]
        K0 = sum(x.shape[1] for x in xs)
        for t in range(towers):
            q = params[t * per:(t + 1) * per]
            ok = (q[0].shape == (n1, K0) and q[1].shape == (n1,) and q[2].shape == (n2, n1) and q[3].shape == (n2,)
                  and all(q[4 + 2 * j].shape == (n3, n2) and q[5 + 2 * j].shape == (n3,) for j in range(heads3)))
            if not ok or any(x.dim() != 2 or x.shape[0] != B or x.stride(1) != 1 for x in xs):
                raise DgvitError("mlp_head: inconsistent input / parameter shapes")
        desc = _mlp_desc(B, xs, n1, n2, n3, towers, heads3)
        dev = xs[0].device
        h1 = torch.empty(towers, B, n1, dtype=torch.float32, device=dev)
        h2 = torch.empty(towers, B, n2, dtype=torch.float32, device=dev)
        y = torch.empty(towers, heads3, B, n3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.dgvit_mlp_head_forward(ctypes.byref(desc), _table(xs), _table(params), _ptr(h1), _ptr(h2), _ptr(y), _stream())
        _lib.check(rc, "dgvit_mlp_head_forward")
        ctx.meta = (nseg, towers, heads3, n1, n2, n3)
        ctx.save_for_backward(h1, h2, *xs, *params)
        ctx.set_materialize_grads(False)      # an unused output (e.g. log_std under a loss on the mean only) arrives as None
        return tuple(y[t, j] for t in range(towers) for j in range(heads3))

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.load()
        nseg, towers, heads3, n1, n2, n3 = ctx.meta
        h1, h2, *rest = ctx.saved_tensors
        xs, params = rest[:nseg], rest[nseg:]
        dys = [None if g is None else _dev(g, "dy") for g in dys]
        B, dev = xs[0].shape[0], xs[0].device
        need = ctx.needs_input_grad[3:]
        if all(g is None for g in dys):
            return (None,) * (3 + nseg + len(params))
        din = [torch.empty(B, x.shape[1], dtype=torch.float32, device=dev) if need[i] else None for i, x in enumerate(xs)]
        per = 4 + 2 * heads3
        dpar = []
        for i, p in enumerate(params):
            t, q = divmod(i, per)
            unused = q >= 4 and dys[t * heads3 + (q - 4) // 2] is None     # third layer whose output nobody used: no gradient,
            dpar.append(torch.empty_like(p) if need[nseg + i] and not unused else None)   # as for an nn.Linear outside the graph
        desc = _mlp_desc(B, xs, n1, n2, n3, towers, heads3)
        nsc = lib.dgvit_mlp_head_backward_scratch_floats(ctypes.byref(desc))
        scratch = torch.empty(max(int(nsc), 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.dgvit_mlp_head_backward(ctypes.byref(desc), _table(xs), _table(params), _ptr(h1), _ptr(h2), _grad_table(dys), _grad_table(din),
                                             _grad_table(dpar), _ptr(scratch), scratch.numel(), _stream())
        _lib.check(rc, "dgvit_mlp_head_backward")
        return (None, None, None, *din, *dpar)


def mlp_head_supported(xs, towers):
    """The fused head kernels take hidden widths that are multiples of 32 up to 128, at most 4 outputs and 512 inputs."""
    if not (1 <= len(xs) <= 3 and 1 <= len(towers) <= 2):
        return False
    l1, l2, l3s = towers[0]
    n1, n2, n3 = l1.weight.shape[0], l2.weight.shape[0], l3s[0].weight.shape[0]
    return (n1 % 32 == 0 and n2 % 32 == 0 and n1 <= 128 and n2 <= 128 and n3 <= 4 and 1 <= len(l3s) <= 2
            and sum(x.shape[1] for x in xs) <= 512 and all(x.shape[0] > 0 for x in xs))


def mlp_head(xs, towers):
    """Fused SAC head.  ``xs``: 1-3 (B, k_i) tensors that the reference concatenates along dim 1; ``towers``: 1 or 2 tuples
    ``(fc1, fc2, [third layers])`` of nn.Linear modules.  Returns [[y (B, n3) of third layer j of tower t]]."""
    flat = []
    for l1, l2, l3s in towers:
        flat += [l1.weight, l1.bias, l2.weight, l2.bias]
        for l3 in l3s:
            flat += [l3.weight, l3.bias]
    xs = [x if x.stride(-1) == 1 else x.contiguous() for x in xs]
    nh = len(towers[0][2])
    ys = _MlpHead.apply(len(xs), len(towers), nh, *xs, *flat)
    return [list(ys[t * nh:(t + 1) * nh]) for t in range(len(towers))]


class _TanhGaussian(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mean, log_std_raw, eps, scale, bias, lo, hi):
        lib = _lib.load()
        mean, lsr, eps = _dev(mean, "mean"), _dev(log_std_raw, "log_std"), _dev(eps, "eps")
        scale, bias = _dev(scale.reshape(-1), "action_scale"), _dev(bias.reshape(-1), "action_bias")
        B, A = mean.shape
        if lsr.shape != (B, A) or eps.shape != (B, A) or scale.numel() not in (1, A) or bias.numel() != scale.numel():
            raise DgvitError("tanh_gaussian_sample: mean, log_std and eps must be (B, A); scale / bias 1 or A values")
        action, tmean = torch.empty_like(mean), torch.empty_like(mean)
        logp = torch.empty(B, 1, dtype=torch.float32, device=mean.device)
        with torch.cuda.device(mean.device):
            rc = lib.dgvit_tanh_gaussian_forward(_ptr(mean), _ptr(lsr), _ptr(eps), _ptr(scale), _ptr(bias), scale.numel(), float(lo), float(hi),
                                                 _ptr(action), _ptr(logp), _ptr(tmean), B, A, _stream())
        _lib.check(rc, "dgvit_tanh_gaussian_forward")
        ctx.lohi = (float(lo), float(hi))
        ctx.save_for_backward(mean, lsr, eps, scale)
        return action, logp, tmean

    @staticmethod
    def backward(ctx, dact, dlp, dtm):
        lib = _lib.load()
        mean, lsr, eps, scale = ctx.saved_tensors
        B, A = mean.shape
        dact = None if dact is None else _dev(dact, "d_action")
        dlp = None if dlp is None else _dev(dlp, "d_log_prob")
        dtm = None if dtm is None else _dev(dtm, "d_tanh_mean")
        dmean, dls = torch.empty_like(mean), torch.empty_like(mean)
        with torch.cuda.device(mean.device):
            rc = lib.dgvit_tanh_gaussian_backward(_ptr(mean), _ptr(lsr), _ptr(eps), _ptr(scale), scale.numel(), ctx.lohi[0], ctx.lohi[1],
                                                  _ptr(dact), _ptr(dlp), _ptr(dtm), _ptr(dmean), _ptr(dls), B, A, _stream())
        _lib.check(rc, "dgvit_tanh_gaussian_backward")
        return dmean, dls, None, None, None, None, None


def tanh_gaussian_sample(mean, log_std_raw, eps, scale, bias, ls_min, ls_max):
    """(action, log_prob (B, 1), tanh_mean) of got_sac_network.py:238-251 from the head outputs and a standard-normal draw, one
    HIP launch forward and one backward; ``log_std_raw`` is the un-clamped log_std_linear output (the clamp is applied inside)."""
    return _TanhGaussian.apply(mean, log_std_raw, eps, scale, bias, ls_min, ls_max)


# ------------------------------------------------------------------------------------------------ operator-level helpers
def op_gemm(layout, epilogue, A, B, M, N, K, bias=None, res=None, aux=None, want_c2=False):
    lib = _lib.load()
    A, B = _dev(A, "A"), _dev(B, "B")
    C = torch.empty(M, N, dtype=torch.float32, device=A.device)
    C2 = torch.empty_like(C) if want_c2 else None
    nsc = lib.dgvit_gemm_scratch_floats(layout, M, N, K)
    scratch = torch.empty(max(nsc, 4), dtype=torch.float32, device=A.device)
    lda, ldb = A.shape[1], B.shape[1]
    rc = lib.dgvit_gemm(layout, epilogue, _ptr(A), lda, _ptr(B), ldb, _ptr(C), N, M, N, K, _ptr(bias), _ptr(res), N, _ptr(C2), N,
                        _ptr(aux), N, _ptr(scratch), scratch.numel(), _stream())
    _lib.check(rc, "dgvit_gemm")
    return (C, C2) if want_c2 else C


def op_layernorm_fwd(x, gamma, beta):
    lib = _lib.load()
    x = _dev(x, "x")
    T, D = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(T, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    _lib.check(lib.dgvit_layernorm_forward(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), T, D, _stream()),
               "dgvit_layernorm_forward")
    return y, mean, rstd


def op_layernorm_bwd(dy, x, mean, rstd, gamma, dres=None):
    lib = _lib.load()
    T, D = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(D, dtype=torch.float32, device=x.device)
    db = torch.empty_like(dg)
    nsc = lib.dgvit_layernorm_backward_scratch_floats(T, D)
    sc = torch.empty(nsc, dtype=torch.float32, device=x.device)
    _lib.check(lib.dgvit_layernorm_backward(_ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dres), _ptr(dx), _ptr(dg),
                                            _ptr(db), _ptr(sc), nsc, T, D, _stream()), "dgvit_layernorm_backward")
    return dx, dg, db


def op_rmsnorm_fwd(x, g):
    lib = _lib.load()
    x = _dev(x, "x")
    B, D = x.shape
    y = torch.empty_like(x)
    _lib.check(lib.dgvit_rmsnorm_forward(_ptr(x), D, _ptr(g), _ptr(y), B, D, _stream()), "dgvit_rmsnorm_forward")
    return y


def op_rmsnorm_bwd(dy, x, g):
    lib = _lib.load()
    B, D = x.shape
    dx = torch.empty_like(x)
    dg = torch.empty(D, dtype=torch.float32, device=x.device)
    nsc = lib.dgvit_rmsnorm_backward_scratch_floats(B, D)
    sc = torch.empty(nsc, dtype=torch.float32, device=x.device)
    _lib.check(lib.dgvit_rmsnorm_backward(_ptr(dy), _ptr(x), D, _ptr(g), _ptr(dx), D, _ptr(dg), _ptr(sc), nsc, B, D, _stream()),
               "dgvit_rmsnorm_backward")
    return dx, dg


def op_attention_fwd(qkv, heads, dim_head):
    lib = _lib.load()
    qkv = _dev(qkv, "qkv")
    B, N, W = qkv.shape
    assert W == 3 * heads * dim_head
    out = torch.empty(B, N, heads * dim_head, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=qkv.device)
    _lib.check(lib.dgvit_attention_forward(_ptr(qkv), _ptr(out), _ptr(lse), B, N, heads, dim_head, _stream()),
               "dgvit_attention_forward")
    return out, lse


def op_attention_bwd(qkv, out, dout, lse, heads, dim_head):
    lib = _lib.load()
    B, N, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    _lib.check(lib.dgvit_attention_backward(_ptr(qkv), _ptr(out), _ptr(_dev(dout, "dout")), _ptr(lse), _ptr(dqkv), B, N, heads,
                                            dim_head, _stream()), "dgvit_attention_backward")
    return dqkv


def op_patchify(img, patch):
    lib = _lib.load()
    img = _dev(img, "img")
    B, H, W = img.shape
    out = torch.empty(B, (H // patch[0]) * (W // patch[1]), patch[0] * patch[1], dtype=torch.float32, device=img.device)
    _lib.check(lib.dgvit_patchify(_ptr(img), _ptr(out), B, H, W, patch[0], patch[1], _stream()), "dgvit_patchify")
    return out


def op_dropout_(x, seed, keep):
    lib = _lib.load()
    _lib.check(lib.dgvit_dropout(_ptr(x), x.numel(), int(seed), float(keep), _stream()), "dgvit_dropout")
    return x


# ------------------------------------------------------------------------------------------------ bf16 configuration
def _dev_bf16(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.bfloat16:
        raise DgvitError(f"{name}: expected a bf16 tensor on a ROCm device")
    return t.contiguous()


def cast_bf16(x):
    """fp32 -> bf16 (round to nearest even) on the HIP path; numel must be a multiple of 4."""
    lib = _lib.load()
    x = _dev(x, "x")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.dgvit_cast_f32_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), "dgvit_cast_f32_bf16")
    return y


def op_gemm_bf16(epilogue, a, b, bias=None, res=None, aux=None, want_c2=False):
    """C = A B^T (+ epilogue) with A (M,K), B (N,K) bf16; see dgvit_gemm_bf16 in dgvit_hip.h."""
    lib = _lib.load()
    a, b = _dev_bf16(a, "a"), _dev_bf16(b, "b")
    M, K = a.shape
    N = b.shape[0]
    out_dtype = torch.float32 if epilogue in (2, 4) else torch.bfloat16
    c = torch.empty(M, N, dtype=out_dtype, device=a.device)
    c2 = torch.empty(M, N, dtype=torch.bfloat16, device=a.device) if want_c2 else None
    bias = None if bias is None else _dev(bias, "bias")
    res = None if res is None else _dev(res, "res")
    aux = None if aux is None else _dev_bf16(aux, "aux")
    if want_c2:
        epilogue = 5    # GELU with the pre-activation copy
    with torch.cuda.device(a.device):
        rc = lib.dgvit_gemm_bf16(int(epilogue), _ptr(a), K, _ptr(b), K, _ptr(c), N, M, N, K, _ptr(bias), _ptr(res), N, _ptr(c2), N,
                                 _ptr(aux), N, _stream())
    _lib.check(rc, "dgvit_gemm_bf16")
    return (c, c2) if want_c2 else c


def op_layernorm_bf16(x, gamma, beta):
    lib = _lib.load()
    x, gamma, beta = _dev(x, "x"), _dev(gamma, "gamma"), _dev(beta, "beta")
    rows, D = x.shape
    y = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.dgvit_layernorm_forward_bf16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), rows, D, _stream())
    _lib.check(rc, "dgvit_layernorm_forward_bf16")
    return y, mean, rstd


def op_attention_bf16(qkv, heads, dim_head=64, want_lse=False):
    lib = _lib.load()
    qkv = _dev_bf16(qkv, "qkv")
    B, N, W = qkv.shape
    if W != 3 * heads * dim_head:
        raise DgvitError(f"qkv last dim {W} != 3*{heads}*{dim_head}")
    out = torch.empty(B, N, heads * dim_head, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, heads, N, dtype=torch.float32, device=qkv.device) if want_lse else None
    with torch.cuda.device(qkv.device):
        rc = lib.dgvit_attention_forward_bf16(_ptr(qkv), _ptr(out), _ptr(lse), B, N, heads, dim_head, _stream())
    _lib.check(rc, "dgvit_attention_forward_bf16")
    return (out, lse) if want_lse else out


def op_attention_bwd_bf16(qkv, out, dout, lse, heads, dim_head=64):
    lib = _lib.load()
    qkv, out, dout, lse = _dev_bf16(qkv, "qkv"), _dev_bf16(out, "out"), _dev_bf16(dout, "dout"), _dev(lse, "lse")
    B, N, _ = qkv.shape
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B * heads * N, dtype=torch.float32, device=qkv.device)
    with torch.cuda.device(qkv.device):
        rc = lib.dgvit_attention_backward_bf16(_ptr(qkv), _ptr(out), _ptr(dout), _ptr(lse), _ptr(dqkv), _ptr(delta), B, N, heads, dim_head,
                                               _stream())
    _lib.check(rc, "dgvit_attention_backward_bf16")
    return dqkv


class Bf16Weights:
    """bf16 copies of an encoder's GEMM weights in one arena (dgvit_got_pack_weights_bf16).

    The copies are re-packed from the fp32 masters before EVERY forward (on the forward's stream, so a captured HIP graph
    replays the pack too).  Nothing cheaper is correct: fused optimiser steps, Polyak updates, ``param.data.copy_`` (the
    reference's utils.soft_update / hard_update, utils.py:31-37) and collective broadcasts all write parameters without
    touching autograd's version counters, so no host-side key can tell that a master has changed.  The pack is one HBM pass
    over the GEMM weights (0.1 ms for the 85 M-parameter ViT-Base variant against a 20 ms forward); transposed copies, which
    only the backward reads, are skipped under no_grad.

    ``frozen=True`` (``GoT.freeze_bf16_weights()``: serving with weights that no longer change) packs once and reuses the
    arena until ``invalidate()``; every in-tree writer of parameters (FlatAdam.step, soft_update, hard_update,
    GradSync.broadcast_parameters, load_state_dict) invalidates, and a moved parameter (``.to()``) is detected by address."""

    def __init__(self):
        self.arena = None
        self.frozen = False
        self.generation = 0          # bumped by invalidate(); packed_generation lags behind it while the arena is stale
        self._packed = None          # (generation, parameter addresses, with_transposes) of the arena's content
        self.packs = 0               # number of pack launches (tests)

    def invalidate(self):
        self.generation += 1

    def __getstate__(self):       # torch.save(module): the arena is a cache, not state
        return {"frozen": self.frozen}

    def __setstate__(self, state):
        self.__init__()
        self.frozen = bool(state.get("frozen", False))

    def get(self, cfg, params, with_transposes=True):
        lib = _lib.load()
        dev = params[0].device
        key = (self.generation, tuple(p.data_ptr() for p in params))
        if self.frozen and self.arena is not None and self.arena.device == dev and self._packed is not None \
                and self._packed[:2] == key and (self._packed[2] or not with_transposes):
            return self.arena
        n = lib.dgvit_got_bf16_weight_elems(ctypes.byref(cfg))
        if n < 0:
            _lib.check(-1, "dgvit_got_bf16_weight_elems")
        if self.arena is None or self.arena.numel() != n or self.arena.device != dev:
            self.arena = torch.empty(n, dtype=torch.bfloat16, device=dev)
        with torch.cuda.device(dev):
            rc = lib.dgvit_got_pack_weights_bf16(ctypes.byref(cfg), _table(params), _ptr(self.arena), n, int(with_transposes), _stream())
        _lib.check(rc, "dgvit_got_pack_weights_bf16")
        self._packed = (*key, bool(with_transposes))
        self.packs += 1
        if self.frozen:
            reg = _frozen_registry()
            for p in params:
                reg[p] = self
        return self.arena


_FROZEN_ARENAS = None   # parameter -> Bf16Weights holding a frozen copy of it (identity-keyed, weak)


def _frozen_registry():
    global _FROZEN_ARENAS
    if _FROZEN_ARENAS is None:
        from torch.utils.weak import WeakIdKeyDictionary
        _FROZEN_ARENAS = WeakIdKeyDictionary()
    return _FROZEN_ARENAS


def notify_parameters_changed(objs):
    """Tell the bf16 weight caches that fp32 masters were written behind autograd's back (see Bf16Weights).  ``objs``: a module,
    or an iterable of modules and / or parameters."""
    if isinstance(objs, torch.nn.Module):
        objs = [objs]
    reg = _frozen_registry()
    for o in objs:
        if isinstance(o, torch.nn.Module):
            for sub in o.modules():
                w = getattr(sub, "_bf16_weights", None)
                if isinstance(w, Bf16Weights):
                    w.invalidate()
        else:
            w = reg.get(o)
            if w is not None:
                w.invalidate()


def op_wgrad_bf16(dy, x, want_bias=True):
    """dW (Mo, Ko) fp32 = dY^T X, db = column sums of dY, for bf16 dY (T, Mo), X (T, Ko)."""
    lib = _lib.load()
    dy, x = _dev_bf16(dy, "dy"), _dev_bf16(x, "x")
    T, Mo = dy.shape
    Ko = x.shape[1]
    dev = dy.device
    dw = torch.empty(Mo, Ko, dtype=torch.float32, device=dev)
    db = torch.empty(Mo, dtype=torch.float32, device=dev) if want_bias else None
    ns = lib.dgvit_wgrad_bf16_scratch_floats(Mo, Ko, T)
    scratch = torch.empty(max(ns, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dgvit_wgrad_bf16(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), _ptr(scratch), ns, T, Mo, Ko, _stream())
    _lib.check(rc, "dgvit_wgrad_bf16")
    return (dw, db) if want_bias else dw


class _GoTEncoderBf16(torch.autograd.Function):
    """GoT.forward in the bf16 configuration as one autograd node (dgvit_got_forward_bf16 / dgvit_got_backward_bf16)."""

    @staticmethod
    def forward(ctx, img, goal, cfg_tuple, keep, seed, need_grad, grad_hook, weights, *params):
        lib = _lib.load()
        cfg = dgvit_config(*cfg_tuple)
        img, goal = _dev(img, "img"), _dev(goal, "goal")
        params = [_dev(p, f"param[{i}]") for i, p in enumerate(params)]
        nparam = _lib.NUM_GLOBAL_PARAMS + _lib.PARAMS_PER_LAYER * cfg.depth
        if len(params) != nparam:
            raise DgvitError(f"expected {nparam} parameter tensors, got {len(params)}")
        if img.dim() != 3 or img.shape[1] != cfg.image_h or img.shape[2] != cfg.image_w:
            raise DgvitError(f"img must be (B, {cfg.image_h}, {cfg.image_w}), got {tuple(img.shape)}")
        B = img.shape[0]
        if goal.shape != (B, cfg.dim):
            raise DgvitError(f"goal must be ({B}, {cfg.dim}), got {tuple(goal.shape)}")
        wpack = weights.get(cfg, params, with_transposes=bool(need_grad))
        nws = lib.dgvit_got_bf16_workspace_bytes(ctypes.byref(cfg), B, int(need_grad))
        if nws < 0:
            _lib.check(-1, "dgvit_got_bf16_workspace_bytes")
        ws = torch.empty(nws, dtype=torch.uint8, device=img.device)
        feat = torch.empty(B, cfg.dim, dtype=torch.float32, device=img.device)
        seed_dev = seed if isinstance(seed, torch.Tensor) else None
        seed_val = 0 if seed_dev is not None else int(seed)
        with torch.cuda.device(img.device):
            rc = lib.dgvit_got_forward_bf16(ctypes.byref(cfg), _table(params), _ptr(wpack), _ptr(img), _ptr(goal), _ptr(feat), _ptr(ws),
                                            nws, B, int(need_grad), float(keep), seed_val, _ptr(seed_dev), _stream())
        _lib.check(rc, "dgvit_got_forward_bf16")
        if need_grad:
            ctx.cfg_tuple, ctx.keep, ctx.seed, ctx.batch = cfg_tuple, float(keep), seed_val, B
            ctx.seed_dev, ctx.ws, ctx.wpack, ctx.img = seed_dev, ws, wpack, img
            ctx.grad_hook = grad_hook
            ctx.save_for_backward(*params)
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        lib = _lib.load()
        cfg = dgvit_config(*ctx.cfg_tuple)
        ws = _take_workspace(ctx, "dgvit_got_backward_bf16")
        params = list(ctx.saved_tensors)
        dfeat = _dev(dfeat, "dfeat")
        B, dev = ctx.batch, dfeat.device
        grads = _flat_grads(params, ctx.needs_input_grad[_N_NONPARAM_INPUTS + 1:], dev)   # one flat buffer (see _GoTEncoder.backward)
        dgoal = torch.empty(B, cfg.dim, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        nsc = lib.dgvit_got_bf16_backward_scratch_bytes(ctypes.byref(cfg), B)
        scratch = torch.empty(nsc, dtype=torch.uint8, device=dev)
        evs = _layer_events(dev, cfg.depth) if ctx.grad_hook is not None else None
        events, keep_alive = _grad_events(cfg.depth, evs) if evs else (None, None)
        with torch.cuda.device(dev):
            rc = lib.dgvit_got_backward_bf16_ev(ctypes.byref(cfg), _table(params), _ptr(ctx.wpack), _grad_table(grads), _ptr(dfeat), _ptr(dgoal),
                                                _ptr(ctx.img), _ptr(ws), ws.numel(), _ptr(scratch), nsc, B, ctx.keep, ctx.seed,
                                                _ptr(ctx.seed_dev), _stream(), ctypes.byref(events) if events is not None else None)
        _lib.check(rc, "dgvit_got_backward_bf16")
        ctx.ws = ctx.wpack = ctx.img = None
        if evs:
            _call_grad_hook(ctx.grad_hook, grads, cfg.depth, evs)
        return (None, dgoal, None, None, None, None, None, None, *grads)


def got_encoder_bf16(img, goal, cfg_tuple, params, weights: Bf16Weights, dropout_keep=1.0, dropout_seed=0, grad_hook=None):
    """GoT.forward in the bf16 configuration (bf16 storage of GEMM operands, fp32 master parameters and gradients)."""
    if any(p is None for p in params):
        raise NotImplementedError("the bf16 configuration needs an attention with an output projection (heads == 1 with dim_head == dim "
                                  "runs on the fp32 path only)")
    if img.shape[0] == 0:
        return _empty_batch((0, int(cfg_tuple[4])), [img, goal, *params])
    need_grad = torch.is_grad_enabled() and (goal.requires_grad or any(p.requires_grad for p in params))
    return _GoTEncoderBf16.apply(img, goal, tuple(cfg_tuple), dropout_keep, dropout_seed, need_grad, grad_hook, weights, *params)
