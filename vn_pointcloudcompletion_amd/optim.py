"""Flat parameter / gradient buffers and the fused Adam step.

The reference trains with ``torch.optim.Adam(model.parameters(), lr, betas=(0.9, 0.999))`` (train.py:70).  Here every
trainable parameter, its gradient and both Adam moments are views into four flat fp32 buffers, so that
  * the optimizer step is ONE HBM-bound kernel (``vnp_adam_step``) instead of ~25 small launches, and
  * the data-parallel gradient exchange is ONE RCCL all-reduce over the flat gradient buffer (distributed.py).
``FusedAdam`` is a ``torch.optim.Optimizer``: ``state_dict()`` / ``load_state_dict()`` keep torch-Adam's format
(``state[i] = {step, exp_avg, exp_avg_sq}``), so the reference's ``optim_{last,best}.pth`` files round-trip
(train.py:255-277), and ``StepLR`` works unchanged (train.py:93).
Parameters that never receive a gradient (``maxpool*.map_to_dir``) keep a zero gradient slice and are
therefore never moved, exactly as torch's Adam skips ``grad is None`` parameters.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib

_ALIGN = 4  # elements: keeps every slice 16-byte aligned for the float4 kernels


class FlatBuffers:
    """Re-homes ``params`` (and their ``.grad``) into one flat parameter buffer and one flat gradient buffer.

    Two ways for the gradients to reach the flat buffer:
      * in place (CPU tensors; ``gathered=False``): every ``p.grad`` IS its slice of ``flat_grad``, the buffer is zeroed per step and
        autograd adds each incoming gradient into the slice — one small add launch per parameter per step (config 5: 190 of them);
      * gathered (device tensors, the default): ``zero_grad`` empties ``p.grad``, autograd keeps each incoming gradient tensor as it is
        (no add, no zero-fill), and ``gather`` copies them to their slices with one launch per 48 tensors (``vnp_pack_tensors``), after
        which ``p.grad`` is the slice again.  A parameter that received nothing gets zeros (once: a slice known to be zero is left alone).
    """

    def __init__(self, params, gathered=None):
        self.params = self._layout_order([p for p in params if p.requires_grad])
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.offsets = []
        off = 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("all parameters must be fp32 on one device")
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.flat_param = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(off, device=dev, dtype=torch.float32)
        self.gathered = (dev.type == "cuda") if gathered is None else bool(gathered)
        if self.gathered and dev.type != "cuda":
            raise ValueError("gathered gradients need the device library (vnp_pack_tensors)")
        self._zero = [True] * len(self.params)            # slice i of flat_grad is known to hold zeros
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                self.flat_param[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat_param[o:o + p.numel()].view(p.shape)
        # the gradient views and their addresses, made once: gather() runs once per step over every parameter, and building two tensor
        # views per parameter there was 3 ms of host time per step at VN-PoinTr's 190 parameters
        self._gviews = [self.flat_grad[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]
        self._homes = [self.flat_grad.data_ptr() + o * 4 for o in self.offsets]
        self._numels = [p.numel() for p in self.params]
        for p, v in zip(self.params, self._gviews):
            p.grad = v

    @staticmethod
    def _layout_order(params):
        """The order in which the parameters are laid out (``self.params``; offsets ascend along it): the given order, except that a
        parameter another one names as its pack successor (ops.pack_hint: a layer's map_to_dir after its map_to_feat, an attention's
        Wk, Wv after Wq) moves up right behind it, so that [Wf; Wd] is one contiguous block of the flat buffer and the layers take it
        as a view instead of concatenating the two every step.  A chain is cut at a parameter whose size would need padding."""
        ids = {id(p) for p in params}
        placed, order = set(), []
        for p in params:
            if id(p) in placed:
                continue
            q = p
            while True:
                order.append(q)
                placed.add(id(q))
                nxt = getattr(q, "_vnp_pack_next", None)
                if nxt is None or id(nxt) not in ids or id(nxt) in placed or q.numel() % _ALIGN or nxt.dtype != q.dtype:
                    break
                q = nxt
        return order

    def view(self, flat, i):
        p, o = self.params[i], self.offsets[i]
        return flat[o:o + p.numel()].view(p.shape)

    def _home(self, i):
        return self._homes[i]

    def zero_grad(self):
        if self.gathered:
            for p in self.params:                     # autograd will keep the incoming gradient tensors; gather() brings them home
                p.grad = None
            return
        self.flat_grad.zero_()
        for i, p in enumerate(self.params):          # re-attach views a caller may have dropped (set_to_none=True)
            if p.grad is None or p.grad.data_ptr() != self._homes[i]:
                p.grad = self._gviews[i]

    def gather(self, indices=None):
        """Bring the gradients of ``indices`` (default: all parameters) home: afterwards ``p.grad`` is the parameter's slice of
        ``flat_grad`` and holds what autograd produced (zeros if nothing).  No-op for parameters that are at home already."""
        idx = range(len(self.params)) if indices is None else indices
        src, off, num, keep = [], [], [], []
        params, homes, gviews, zero, gathered = self.params, self._homes, self._gviews, self._zero, self.gathered
        dev = self.flat_grad.device
        for i in idx:
            p = params[i]
            g = p.grad
            if g is None:
                if not zero[i]:
                    src.append(None); off.append(self.offsets[i]); num.append(self._numels[i])
                    zero[i] = True
            elif g.data_ptr() == homes[i]:
                if gathered:
                    zero[i] = False                   # autograd may have accumulated into the view in place
                continue
            elif gathered and g.is_contiguous() and g.dtype == torch.float32 and g.numel() == self._numels[i] and g.device == dev:
                src.append(g.data_ptr()); off.append(self.offsets[i]); num.append(self._numels[i]); keep.append(g)
                zero[i] = False
            else:
                gviews[i].copy_(g)
                zero[i] = False
            p.grad = gviews[i]
        if src:
            if not self.gathered:                     # in-place mode only ever sees a dropped view: zero-fill
                for o, n in zip(off, num):
                    self.flat_grad[o:o + n].zero_()
            else:
                n = len(src)
                a_src = (ctypes.c_void_p * n)(*src)
                a_off = (ctypes.c_longlong * n)(*off)
                a_num = (ctypes.c_longlong * n)(*num)
                _lib.call("vnp_pack_tensors", n, ctypes.addressof(a_src), ctypes.addressof(a_off), ctypes.addressof(a_num),
                          self.flat_grad.data_ptr())
        del keep                                      # the sources stayed alive until the launch was queued on their stream

    def mark_written(self, indices=None):
        """Something other than ``gather`` wrote into the slices of ``indices`` (default: all) — a collective that left the average of
        the ranks' gradients there, a host-staged transport, a test.  A slice that ``gather`` had left as known zeros (its parameter
        received no gradient on THIS rank) may hold another rank's contribution now, so it loses the mark and is zero-filled again the
        next time its parameter comes up without a gradient; otherwise the stale average would be reduced and applied once more."""
        zero = self._zero
        for i in (range(len(zero)) if indices is None else indices):
            zero[i] = False

    def check_views(self):
        """Every gradient produced by autograd must be in the flat buffer before it is reduced or applied."""
        self.gather()


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        self._flat = []
        for group in self.param_groups:
            fb = FlatBuffers(group["params"])
            fb.exp_avg = torch.zeros_like(fb.flat_param)
            fb.exp_avg_sq = torch.zeros_like(fb.flat_param)
            fb.step = 0
            self._flat.append(fb)

    @property
    def flat_buffers(self):
        return self._flat

    def enable_overlapped_allreduce(self, bucket_bytes=32 << 20, transport=None):
        """Data parallel: average gradients in large buckets while backward is still running (distributed.py)."""
        from .distributed import OverlappedGradReducer
        for r in getattr(self, "grad_reducers", []):
            r.remove()
        self.grad_reducers = [OverlappedGradReducer(fb, bucket_bytes, transport) for fb in self._flat]
        return self.grad_reducers

    def zero_grad(self, set_to_none: bool = True):
        for fb in self._flat:
            fb.zero_grad()

    def _publish_state(self, fb):
        for i, p in enumerate(fb.params):
            self.state[p] = {"step": torch.tensor(float(fb.step)), "exp_avg": fb.view(fb.exp_avg, i),
                             "exp_avg_sq": fb.view(fb.exp_avg_sq, i)}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, fb in zip(self.param_groups, self._flat):
            fb.check_views()
            fb.step += 1
            b1, b2 = group["betas"]
            _lib.call("vnp_adam_step", fb.flat_param.data_ptr(), fb.flat_grad.data_ptr(), fb.exp_avg.data_ptr(),
                      fb.exp_avg_sq.data_ptr(), fb.numel, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                      fb.step, float(group["weight_decay"]))
        return loss

    def state_dict(self):
        for fb in self._flat:
            self._publish_state(fb)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        with torch.no_grad():
            for fb in self._flat:
                steps = []
                for i, p in enumerate(fb.params):
                    st = self.state.get(p)
                    if st:                                   # torch's Adam has no state for grad-is-None parameters
                        fb.view(fb.exp_avg, i).copy_(st["exp_avg"])
                        fb.view(fb.exp_avg_sq, i).copy_(st["exp_avg_sq"])
                        steps.append(int(float(st["step"])))
                fb.step = max(steps) if steps else 0
                self._publish_state(fb)
