"""Flat-buffer optimiser step and Polyak target update for the DGViT networks.

The reference calls ``torch.optim.Adam`` over ~70 tensors per network (DRL.py:126-168, 401-403, 412-414) and a
per-parameter Python loop ``soft_update`` (utils.py:31-33).  Here the parameters of a network are re-homed ONCE into
one flat fp32 buffer (its *home*: each ``nn.Parameter`` keeps its identity, its ``.data`` becomes a view), and both
updates are one HBM-bound HIP kernel per contiguous run of that buffer (``dgvit_adam_step`` / ``dgvit_soft_update``).

There is one owner of parameter storage per module: ``flatten_parameters(module)`` creates (or returns) the module's
home and ``FlatAdam`` adopts it, so an optimiser and a Polyak update on the same network share the same buffer.
Layout of a home: for every ``GoT`` encoder inside the module its trainable parameters in the C ABI's table order --
exactly the layout of the flat gradient buffer the fused encoder backward produces, which Adam then consumes in place --
followed by all remaining parameters in ``parameters()`` order.  Two networks of the same architecture get the same
layout, which is what the one-kernel ``soft_update`` needs.
"""
import ctypes
import weakref
from typing import Iterable, List, Union

import torch
from torch.utils.weak import WeakIdKeyDictionary

from . import _lib
from . import functional as F_
from .goalformer import GoT


def _al4(n: int) -> int:
    return (n + 3) & ~3


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_HOME_OF = WeakIdKeyDictionary()   # parameter -> (home, index in home.params); keyed by identity (tensor == is elementwise)


def _no_home():
    return None


class _Home:
    """Parameters living back to back (4-float aligned slots) in one flat buffer, with room for Adam's moments."""

    def __init__(self, params: List[torch.Tensor], sections: List[int] = None):
        # sections[i]: which GoT encoder's table parameter i belongs to (-1: none); Adam runs never straddle two sections, so
        # the encoder run can pick up the fused backward's gradient buffer without a copy
        self.sections = list(sections) if sections is not None else [-1] * len(params)
        if not params:
            raise ValueError("no parameters to flatten")
        if not all(p.is_cuda for p in params):
            raise _lib.DgvitError("flat parameter buffers need the module on a ROCm device first (call .to(device) before)")
        self.params = params
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += _al4(p.numel())
        self.numel = off
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=params[0].device)   # zeros: padding lanes stay finite
        # Optimiser state follows a parameter into its new home: a parameter that has taken Adam steps in another (still intact)
        # home -- a loose FlatAdam(net.parameters()) home later adopted by the module's, a child module's home swallowed by its
        # parent's, a layout that changed -- keeps its moments and its step count (torch keeps them in `state[p]`, wherever p lives).
        carried, intact_cache = [], {}
        for i, p in enumerate(params):
            ent = _HOME_OF.get(p)
            if ent is None or ent[0] is self:
                continue
            old, j = ent
            if id(old) not in intact_cache:
                intact_cache[id(old)] = old.intact()
            if old.exp_avg is not None and old.steps[j] > 0:
                if not intact_cache[id(old)]:
                    raise _lib.DgvitError("parameter storage was replaced (e.g. by .to()) after optimiser state was built; rebuild the optimiser")
                carried.append((i, old, j))
        self.exp_avg = self.exp_avg_sq = self.gflat = None
        self.steps = [0] * len(params)          # per-parameter Adam step counts (torch keeps `state[p]["step"]`)
        self.owner = [None] * len(params)       # weak reference to the FlatAdam whose state lives in this home's slot (None: nobody's yet)
        for i, p in enumerate(params):
            ent = _HOME_OF.get(p)
            if ent is not None and ent[0] is not self:
                self.owner[i] = ent[0].owner[ent[1]]
        if carried:
            m, v = self.moments()
            for i, old, j in carried:
                n, o, oo = params[i].numel(), self.offsets[i], old.offsets[j]
                m[o:o + n].copy_(old.exp_avg[oo:oo + n])
                v[o:o + n].copy_(old.exp_avg_sq[oo:oo + n])
                self.steps[i] = old.steps[j]
        for i, (p, o) in enumerate(zip(params, self.offsets)):
            v = self.flat[o:o + p.numel()].view_as(p)
            v.copy_(p.data)
            p.data = v
            _HOME_OF[p] = (self, i)
        self.zero_copy_elems = self.copied_elems = 0   # gradient elements Adam consumed in place / had to gather (tests, tuning)
        self.signature = tuple((tuple(p.shape), o) for p, o in zip(params, self.offsets))

    def __deepcopy__(self, memo):
        return None     # a copied module's parameters are fresh tensors: its home is rebuilt on first use

    def __reduce__(self):
        return (_no_home, ())   # torch.save(module): parameters are pickled as ordinary tensors, the home is rebuilt on use

    def intact(self) -> bool:
        base = self.flat.data_ptr()
        return all(p.data.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def moments(self):
        if self.exp_avg is None:
            self.exp_avg = torch.zeros_like(self.flat)
            self.exp_avg_sq = torch.zeros_like(self.flat)
        return self.exp_avg, self.exp_avg_sq

    def grad_run(self, idx: List[int]) -> torch.Tensor:
        """Gradients of the parameters ``idx`` (adjacent slots) as one flat tensor laid out like their slots: zero copy when
        they already are views of one buffer with that layout (the fused encoder backward's), else one multi-tensor copy."""
        lo = self.offsets[idx[0]]
        hi = self.offsets[idx[-1]] + _al4(self.params[idx[-1]].numel())
        g0 = self.params[idx[0]].grad
        st, o0 = g0.untyped_storage().data_ptr(), g0.storage_offset()
        if o0 % 4 == 0 and g0.data_ptr() % 16 == 0 and (o0 + hi - lo) * 4 <= g0.untyped_storage().nbytes() and all(
                self.params[i].grad.untyped_storage().data_ptr() == st and self.params[i].grad.is_contiguous()
                and self.params[i].grad.storage_offset() - o0 == self.offsets[i] - lo for i in idx):
            self.zero_copy_elems += hi - lo
            return torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), o0, (hi - lo,))
        self.copied_elems += hi - lo
        if self.gflat is None:
            self.gflat = torch.zeros_like(self.flat)
        views = [self.gflat[self.offsets[i]:self.offsets[i] + self.params[i].numel()].view_as(self.params[i]) for i in idx]
        torch._foreach_copy_(views, [self.params[i].grad for i in idx])
        return self.gflat[lo:hi]


class _Moments:
    """Adam state (moments, step counts) of ONE optimiser over a home's slots, kept beside the home: used when another live
    optimiser already keeps its state in the home itself.  torch.optim.Adam holds its state per optimiser, and the reference
    builds two of them over the same policy (Imitation_learning.py:379 and :812): the second must not see, or wipe, the first's."""

    def __init__(self, home: _Home):
        self.home = weakref.ref(home)
        self.exp_avg = self.exp_avg_sq = None
        self.steps = [0] * len(home.params)

    def moments(self):
        if self.exp_avg is None:
            flat = self.home().flat
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
        return self.exp_avg, self.exp_avg_sq


def _layout(module: torch.nn.Module):
    order, sections, taken, nsec = [], [], set(), 0
    for sub in module.modules():
        if isinstance(sub, GoT):
            for p in sub.param_table():     # frozen parameters keep their slot: a frozen target has its source's layout
                if id(p) not in taken:
                    order.append(p)
                    sections.append(nsec)
                    taken.add(id(p))
            nsec += 1
    for p in module.parameters():
        if id(p) not in taken:
            order.append(p)
            sections.append(-1)
            taken.add(id(p))
    return order, sections


def flatten_parameters(module: torch.nn.Module) -> torch.Tensor:
    """Re-home ALL parameters of ``module`` into one flat buffer (see the module docstring for the layout) and return it;
    a module that already has an intact home keeps it.  Call after ``.to(device)``; ``FlatAdam`` and ``soft_update`` do it
    themselves when needed."""
    return home_of(module).flat


def home_of(module: torch.nn.Module) -> _Home:
    h = getattr(module, "_dgvit_home", None)
    lay, sections = _layout(module)
    if isinstance(h, _Home) and len(h.params) == len(lay) and all(a is b for a, b in zip(h.params, lay)) and h.intact():
        return h
    h = _Home(lay, sections)        # (carries over the Adam state of parameters that lived in another intact home; raises if that
                                    #  home's storage was replaced, e.g. by .to(), after optimiser state was built)
    module._dgvit_home = h
    return h


class FlatAdam:
    """``torch.optim.Adam`` semantics (amsgrad=False) with one HIP kernel per contiguous run of a flat parameter buffer.

    ``params``: modules (all their parameters) and / or an iterable of parameters, like the reference's optimisers over
    sub-sets of a network (DRL.py:107-111, 145-148).  As with torch's optimisers a parameter whose ``.grad`` is None at
    ``step()`` is skipped (no moment update, its own step count does not advance), and a parameter that first receives a
    gradient later starts its bias correction at 1 then.  Build it after ``module.to(device)``.  ``capturable=True`` keeps
    the step counter in device memory (like torch's ``capturable`` optimisers) so that ``step()`` can be recorded into a HIP
    graph; every replay must then see the same set of parameters with gradients.
    """

    def __init__(self, params: Union[torch.nn.Module, Iterable], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 capturable: bool = False):
        self.capturable, self._step_dev = bool(capturable), None
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        items = [params] if isinstance(params, torch.nn.Module) else list(params)
        self.modules = [m for m in items if isinstance(m, torch.nn.Module)]
        chosen, seen = [], set()
        for m in self.modules:
            home_of(m)
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    chosen.append(p)
        loose = []
        for p in items:
            if isinstance(p, torch.nn.Module):
                continue
            if not (isinstance(p, torch.Tensor) and p.is_leaf and p.dtype == torch.float32):   # nn.Parameter or a plain leaf such as DRL.py's log_alpha
                raise TypeError(f"FlatAdam: expected modules, parameters or fp32 leaf tensors, got {type(p).__name__}")
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                chosen.append(p)
                ent = _HOME_OF.get(p)
                if ent is None or not ent[0].intact():
                    loose.append(p)
        if loose:
            _Home(loose)               # parameters handed over one by one that no module home owns yet
        self.params = chosen
        self.step_count = 0
        # Like a new torch.optim.Adam, a new FlatAdam starts from empty state.  Moments and step counts normally live with the
        # parameters' home (they follow the parameters when a home is rebuilt), owned by ONE optimiser per slot.  A slot whose owner
        # is still alive keeps serving that optimiser -- this one then keeps its own state beside the home (`_Moments`), exactly as
        # two torch.optim.Adam instances over the same parameters would; a slot whose owner is gone is taken over and whatever it
        # left there is cleared.  load_state_dict() restores a saved state.
        self._private = {}                 # id(home) -> _Moments
        for p in chosen:
            home, i = _HOME_OF[p]
            other = home.owner[i]() if home.owner[i] is not None else None
            if other is not None and other is not self and id(home) not in self._private:
                self._private[id(home)] = _Moments(home)
        for p in chosen:
            home, i = _HOME_OF[p]
            if id(home) in self._private:
                continue
            home.owner[i] = weakref.ref(self)
            if home.steps[i]:
                home.steps[i] = 0
                if home.exp_avg is not None:
                    o, n = home.offsets[i], p.numel()
                    home.exp_avg[o:o + n].zero_()
                    home.exp_avg_sq[o:o + n].zero_()

    def _all_params(self):
        return self.params

    def _state(self, home: _Home):
        """where this optimiser's moments / step counts for `home`'s slots live: the home itself, or its own `_Moments`"""
        if not self._private:
            return home
        st = self._private.get(id(home))
        if st is not None and st.home() is home:
            return st
        if any(m.home() is None or not m.home().intact() for m in self._private.values()):
            raise _lib.DgvitError("FlatAdam: parameter storage was re-homed after this (second) optimiser over the same parameters was "
                                  "built; rebuild the optimiser")
        return home

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def _runs(self):
        """[(home, [param indices])]: maximal runs of adjacent slots whose parameters have a gradient and equal step counts."""
        by_home = {}
        for p in self.params:
            if p.grad is None:
                continue
            ent = _HOME_OF.get(p)
            if ent is None or not ent[0].intact():
                raise _lib.DgvitError("FlatAdam: parameter storage was replaced (e.g. by .to() or deepcopy); rebuild the optimiser")
            by_home.setdefault(id(ent[0]), (ent[0], []))[1].append(ent[1])
        runs = []
        for home, idx in by_home.values():
            idx.sort()
            cur = [idx[0]]
            steps = self._state(home).steps
            for a, b in zip(idx, idx[1:]):
                if b == a + 1 and steps[b] == steps[a] and home.sections[b] == home.sections[a]:
                    cur.append(b)
                else:
                    runs.append((home, cur))
                    cur = [b]
            runs.append((home, cur))
        return runs

    @torch.no_grad()
    def step(self) -> None:
        lib = _lib.load()
        runs = self._runs()
        if not runs:
            return
        self.step_count += 1
        step_ptr = ctypes.c_void_p(0)
        if self.capturable:
            if len({self._state(home).steps[idx[0]] for home, idx in runs}) != 1:
                raise _lib.DgvitError("FlatAdam(capturable=True): every parameter must receive a gradient on every step")
            if self._step_dev is None:
                self._step_dev = torch.full((1,), self._state(runs[0][0]).steps[runs[0][1][0]], dtype=torch.int64, device=runs[0][0].flat.device)
            self._step_dev += 1                      # a device op: replayed with the graph
            step_ptr = ctypes.c_void_p(self._step_dev.data_ptr())
        for home, idx in runs:
            lo = home.offsets[idx[0]]
            n = home.offsets[idx[-1]] + _al4(home.params[idx[-1]].numel()) - lo
            g = home.grad_run(idx)
            st = self._state(home)
            m, v = st.moments()
            t = st.steps[idx[0]] + 1
            for i in idx:
                st.steps[i] = t
            with torch.cuda.device(home.flat.device):
                rc = lib.dgvit_adam_step(ctypes.c_void_p(home.flat.data_ptr() + 4 * lo), ctypes.c_void_p(g.data_ptr()),
                                         ctypes.c_void_p(m.data_ptr() + 4 * lo), ctypes.c_void_p(v.data_ptr() + 4 * lo), n,
                                         self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, t, step_ptr, _stream())
            _lib.check(rc, "dgvit_adam_step")
        # the kernel wrote the parameters through raw pointers: no autograd version counter moved
        F_.notify_parameters_changed(self.params)

    def _homes(self):
        out, seen = [], set()
        for p in self.params:
            ent = _HOME_OF.get(p)
            if ent is not None and id(ent[0]) not in seen:
                seen.add(id(ent[0]))
                out.append(ent[0])
        return out

    def state_dict(self):
        """Per parameter (in the optimiser's parameter order): step count and both moments, like torch.optim.Adam's state."""
        state = []
        if self.capturable and self._step_dev is not None:
            # graph replays advance only the device counter: bring the host mirror up to date (one device -> host read)
            dev_step = int(self._step_dev.item())
            for p in self.params:
                home, i = _HOME_OF[p]
                st = self._state(home)
                if st.steps[i] > 0:
                    st.steps[i] = dev_step
            self.step_count = max(self.step_count, dev_step)
        for p in self.params:
            home, i = _HOME_OF[p]
            st = self._state(home)
            o, n = home.offsets[i], p.numel()
            if st.exp_avg is None or st.steps[i] == 0:
                state.append(None)
            else:
                state.append({"step": st.steps[i], "exp_avg": st.exp_avg[o:o + n].view_as(p).clone(),
                              "exp_avg_sq": st.exp_avg_sq[o:o + n].view_as(p).clone()})
        return {"step": self.step_count, "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay,
                "state": state}

    def load_state_dict(self, sd) -> None:
        if len(sd["state"]) != len(self.params):
            raise ValueError("optimizer state does not match this optimiser's parameters")
        self.step_count = int(sd["step"])
        for p, st in zip(self.params, sd["state"]):
            home, i = _HOME_OF[p]
            mine = self._state(home)
            o, n = home.offsets[i], p.numel()
            m, v = mine.moments()
            if st is None:
                mine.steps[i] = 0
                m[o:o + n].zero_()
                v[o:o + n].zero_()
            else:
                mine.steps[i] = int(st["step"])
                m[o:o + n].view_as(p).copy_(st["exp_avg"])
                v[o:o + n].view_as(p).copy_(st["exp_avg_sq"])
        if self._step_dev is not None:
            self._step_dev.fill_(max((self._state(_HOME_OF[p][0]).steps[_HOME_OF[p][1]] for p in self.params), default=0))


def _flat_pair(target, source):
    tb, sb = home_of(target), home_of(source)
    if tb.signature != sb.signature:
        raise _lib.DgvitError("soft_update: target and source networks do not have the same parameter layout")
    return tb, sb


@torch.no_grad()
def soft_update(target: torch.nn.Module, source: torch.nn.Module, tau: float) -> None:
    """target <- target*(1-tau) + source*tau (utils.py:31-33) as ONE HIP kernel over the two networks' flat buffers (both are
    flattened on first use; an optimiser built on either keeps working because it adopts the same home)."""
    tb, sb = _flat_pair(target, source)
    lib = _lib.load()
    with torch.cuda.device(tb.flat.device):
        rc = lib.dgvit_soft_update(ctypes.c_void_p(tb.flat.data_ptr()), ctypes.c_void_p(sb.flat.data_ptr()), tb.numel, float(tau),
                                   _stream())
    _lib.check(rc, "dgvit_soft_update")
    F_.notify_parameters_changed(target)


@torch.no_grad()
def hard_update(target: torch.nn.Module, source: torch.nn.Module) -> None:
    """target <- source (utils.py:35-37): one device copy of the flat buffer."""
    tb, sb = _flat_pair(target, source)
    tb.flat.copy_(sb.flat)
    F_.notify_parameters_changed(target)
