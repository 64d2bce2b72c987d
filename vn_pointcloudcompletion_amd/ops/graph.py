"""kNN, FPS, point gathers and the EdgeConv Functions of configs 3 and 5 (replace KNN_CUDA / pointnet2_ops and
/root/reference/models/dgcnn.py:201-278)."""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib
from .._lib import call, ptr
from ._state import HK, SW, POOL_PATH_COUNTS, prof_arm, trace
from .core import (BN_EPS, BN_MOMENTUM, _bn_leaky_backward, _bn_leaky_forward, _bn_statistics, _chunk_pc, _dgrad, _ld, _nchunk,
                   _per_sample_colsum, _rows, _wgrad, cat_rows, gemm, gemm_pd_stats, reduce_pairs, reduce_rows)


def knn_indices(ref, query, k, return_dist=False):
    """ref [B,Nr,D], query [B,Nq,D] (point-major, fp32, contiguous) -> idx [B,Nq,k] int32 local indices, sorted by
    (squared distance, index).  Replaces KNN_CUDA (models/dgcnn.py:11,236)."""
    _lib.require_gpu(ref, query)
    ref, query = ref.contiguous().float(), query.contiguous().float()
    B, Nr, D = ref.shape
    Nq = query.shape[1]
    idx = torch.empty((B, Nq, k), device=ref.device, dtype=torch.int32)
    dist = torch.empty((B, Nq, k), device=ref.device, dtype=torch.float32) if return_dist else None
    forced = HK.knn_force
    if forced is not None and not return_dist:
        # TEST HOOK (teacher forcing, tests/test_gpu_grad_parity.py): the next recorded neighbour table instead of this search
        f = forced.pop(0).to(device=ref.device, dtype=torch.int32).contiguous()
        if f.shape != idx.shape:
            raise _lib.VnpError(f"knn_force: table {tuple(f.shape)} offered to a search of shape {tuple(idx.shape)}")
        trace("knn_forced", f, key="knn_idx")
        return f
    if D == 3 and Nr <= 4096 and k <= 64:
        # coordinate graphs: one wave per query on register-resident distances with a threshold selection, no distance matrix
        # (vnp_knn_coords_kernel behind vnp_knn; same bits and order)
        call("vnp_knn", B, D, ptr(ref), Nr, ptr(query), Nq, k, ptr(idx), ptr(dist))
    elif Nr <= 16384 and Nq * Nr >= 4096 and B * Nq * Nr <= (1 << 29):
        # distance matrix once + per-row selection instead of k full passes per query (same bits as vnp_knn; ~50x faster for the
        # feature-space graphs at D = 192, ~3x for coordinates)
        ws = torch.empty(B * Nq * Nr, device=ref.device, dtype=torch.float32)
        call("vnp_knn_matrix", B, D, ptr(ref), Nr, ptr(query), Nq, k, ptr(idx), ptr(dist), ptr(ws))
    else:
        call("vnp_knn", B, D, ptr(ref), Nr, ptr(query), Nq, k, ptr(idx), ptr(dist))
    if not return_dist:
        trace("knn", idx, key="knn_idx")
    return (idx, dist) if return_dist else idx


def fps_indices(xyz, m, skip_near_origin=None):
    """furthest_point_sample(xyz[B,N,3], m) -> idx [B,m] int32 (starts at 0).  Replaces pointnet2_ops (models/pcn.py:7).
    pointnet2_ops' furthest_point_sampling_kernel skips candidates with |p|^2 <= 1e-3 (`if (mag <= 1e-3) continue;`): on clouds
    centred at the origin ~0.3 points per 2048 fall inside that ball and would be picked by plain FPS but never by the package the
    reference calls.  Default: the package's convention (so reference checkpoints see the centres they were trained with);
    VNP_FPS_PLAIN=1 (or skip_near_origin=False) selects plain FPS.  PARITY UNPINNED either way (third-party source absent)."""
    _lib.require_gpu(xyz)
    xyz = xyz.contiguous().float()
    B, N, _ = xyz.shape
    idx = torch.empty((B, m), device=xyz.device, dtype=torch.int32)
    skip = SW.FPS_SKIP_NEAR_ORIGIN if skip_near_origin is None else bool(skip_near_origin)
    call("vnp_fps_ex", B, N, ptr(xyz), m, ptr(idx), 1 if skip else 0)
    trace("fps", idx, key="fps_idx")
    return idx


_SIDE_STREAMS = {}


def _side_stream(device):
    s = _SIDE_STREAMS.get(device.index)
    if s is None:
        s = _SIDE_STREAMS[device.index] = torch.cuda.Stream(device=device)
    return s


class FpsPlan:
    """Every furthest-point sampling of an encoder pass (models/dgcnn.py:201-222, 291-314; dgcnn_group.py:141-160; models/pcn.py:6-9)
    depends on the INPUT COORDINATES alone: level i+1 samples the points level i picked, the 224 extra points of a `num_coarse == 448`
    head sample the input.  A sampling is M dependent rounds of one workgroup per cloud — latency, 32 of 256 CUs busy — so the whole
    chain is launched on a side HIP stream as soon as the coordinates exist and runs beside the kNN searches and EdgeConv layers of the
    main stream; a consumer waits for the event of the level it needs (``level``).  All buffers are allocated on the main stream before
    the side stream starts and every one of them (the input coordinates included) is marked ``record_stream(side)``: the caching
    allocator does not hand a block out again before the side stream's work queued up to its release has finished, also when a forward
    raises between the plan's construction and ``close``.  ``SW.FPS_AHEAD = False``: the same launches in the same order on the main
    stream.  Each level also yields its coordinates and the global row ids of its picks (vnp_gather_coords)."""

    def __init__(self, coords, chain, extra=()):
        _lib.require_gpu(coords)
        coords = coords.contiguous().float()
        B, N, _ = coords.shape
        dev = coords.device
        skip = 1 if SW.FPS_SKIP_NEAR_ORIGIN else 0
        specs = [(True, int(m)) for m in chain] + [(False, int(m)) for m in extra]
        self._items, self._events, self._keep = [], [], [coords]
        for _, m in specs:
            self._items.append((torch.empty((B, m), device=dev, dtype=torch.int32), torch.empty((B, m, 3), device=dev, dtype=torch.float32),
                                torch.empty(B * m, device=dev, dtype=torch.int32)))
        self._dev = dev
        self._main = torch.cuda.current_stream(dev)
        ahead = SW.FPS_AHEAD and HK.trace is None          # a traced pass reads its picks on the host while it runs: keep it on one stream
        self._side = _side_stream(dev) if ahead else None
        if self._side is not None:
            self._side.wait_stream(self._main)
            for t in (coords, *(t for item in self._items for t in item)):
                t.record_stream(self._side)
        cur, n = coords, N
        with torch.cuda.stream(self._side if self._side is not None else self._main):
            for (chained, m), (idx, xyz_m, gid) in zip(specs, self._items):
                src, ns = (cur, n) if chained else (coords, N)
                call("vnp_fps_ex", B, ns, ptr(src), m, ptr(idx), skip)
                call("vnp_gather_coords", B, ns, ptr(src), ptr(idx), m, ptr(xyz_m), ptr(gid))
                if self._side is not None:
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    self._events.append(ev)
                if chained:
                    cur, n = xyz_m, m
        self._waited = -1

    def level(self, i):
        """(idx [B, m] int32 local picks, coordinates [B, m, 3], global row ids [B*m] int32) of item i, ready on the current stream."""
        if self._side is not None and i > self._waited:
            torch.cuda.current_stream(self._dev).wait_event(self._events[i])
            self._waited = i
        idx, xyz_m, gid = self._items[i]
        trace("fps", idx, key="fps_idx")
        return idx, xyz_m, gid

    def close(self):
        if self._side is not None and self._events and self._waited < len(self._events) - 1:
            torch.cuda.current_stream(self._dev).wait_event(self._events[-1])
            self._waited = len(self._events) - 1


class GatherPointsFn(Function):
    """rows layout: out point i <- x point src[i] (global point ids, int32 [Pout]).  gather_operation of pointnet2_ops
    (models/dgcnn.py:214-218) on VN features; backward = scatter-add."""

    @staticmethod
    def forward(ctx, x, src):
        x = _rows(x)
        src = src.reshape(-1).contiguous()
        P_out, W = src.numel(), x.shape[1]
        out = torch.empty((P_out * 3, W), device=x.device, dtype=torch.float32)
        call("vnp_gather_points", ptr(x), _ld(x), ptr(src), P_out, W, ptr(out), W)
        ctx.save_for_backward(src)
        ctx.cfg = (x.shape[0] // 3, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (src,) = ctx.saved_tensors
        P_in, W = ctx.cfg
        g = _rows(g)
        gx = torch.empty((P_in * 3, W), device=g.device, dtype=torch.float32)
        ws = torch.empty(2 * (P_in + 1) + 2 * src.numel(), device=g.device, dtype=torch.int32)
        call("vnp_scatter_add_points", ptr(g), _ld(g), ptr(src), src.numel(), W, ptr(gx), W, P_in, ptr(ws))
        return gx, None


class EdgeGatherAddFn(Function):
    """Edge tensor of vn_get_graph_feature (models/dgcnn.py:250-278) after the layer's linear maps:
    out[(p,j)] = u[nbr[p,j]] + v[p], with u = x W1^T and v = x (W2-W1)^T computed per POINT (k times fewer GEMM rows than
    applying W to cat(feature - x, x) per EDGE).  nbr: global point ids [P,k] int32."""

    @staticmethod
    def forward(ctx, u, v, nbr):
        u, v = _rows(u), _rows(v)
        nbr = nbr.contiguous()
        P, k = nbr.shape
        W = u.shape[1]
        out = torch.empty((P * k * 3, W), device=u.device, dtype=torch.float32)
        call("vnp_edge_gather_add", ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), P, k, W, ptr(out), W)
        ctx.save_for_backward(nbr)
        ctx.cfg = (u.shape[0] // 3, P, k, W)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (nbr,) = ctx.saved_tensors
        Pu, P, k, W = ctx.cfg
        g = _rows(g)
        dev = g.device
        gu = torch.empty((Pu * 3, W), device=dev, dtype=torch.float32)
        ws = torch.empty(2 * (Pu + 1) + 2 * P * k, device=dev, dtype=torch.int32)
        call("vnp_scatter_add_points", ptr(g), _ld(g), ptr(nbr), P * k, W, ptr(gu), W, Pu, ptr(ws))
        gv = torch.empty((P * 3, W), device=dev, dtype=torch.float32)
        call("vnp_vn_mean_pool_fwd", ptr(g), _ld(g), P, k, W, ptr(gv), W)          # mean over k ...
        gv.mul_(float(k))                                                          # ... times k = sum
        return gu, gv, None




def _edge_reverse_launch(nbr, Pu):
    Pq, k = nbr.shape
    start = torch.empty(Pu + 1, device=nbr.device, dtype=torch.int32)
    rev = torch.empty(Pq * k, device=nbr.device, dtype=torch.int32)
    ws = torch.empty(Pu + 1 + Pq * k, device=nbr.device, dtype=torch.int32)
    return start, rev, ws


def edge_reverse(nbr, Pu):
    """Reverse adjacency (start [Pu + 1], rev [Pq * k]) of a neighbour table, cached on the tensor: layers that share a graph share it.
    A table whose lists were started ahead of time (edge_reverse_ahead) is waited for here, once."""
    cached = getattr(nbr, "_vnp_rev", None)
    if cached is not None and cached[0].shape[0] == Pu + 1:
        if len(cached) == 3:                                       # built on the side stream: this stream waits for it, then forgets the event
            torch.cuda.current_stream(nbr.device).wait_event(cached[2])
            nbr._vnp_rev = cached[:2]
        return cached[0], cached[1]
    start, rev, ws = _edge_reverse_launch(nbr, Pu)
    call("vnp_edge_reverse", ptr(nbr), nbr.shape[0], nbr.shape[1], Pu, ptr(start), ptr(rev), ptr(ws))
    nbr._vnp_rev = (start, rev)
    return start, rev


def edge_reverse_ahead(nbr, Pu):
    """Start the reverse adjacency of a graph whose backward pass will need it (five small launches, ~55 us, one of them a single
    workgroup) on the side HIP stream while the forward pass goes on; edge_reverse waits for its event.  The buffers are allocated on the
    calling stream and marked record_stream(side), like FpsPlan's."""
    if not SW.REV_AHEAD or HK.trace is not None or getattr(nbr, "_vnp_rev", None) is not None:
        return
    dev = nbr.device
    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
    start, rev, ws = _edge_reverse_launch(nbr, Pu)
    side.wait_stream(main)                                         # the table is written by work queued on the calling stream
    for t in (nbr, start, rev, ws):
        t.record_stream(side)
    with torch.cuda.stream(side):
        call("vnp_edge_reverse", ptr(nbr), nbr.shape[0], nbr.shape[1], Pu, ptr(start), ptr(rev), ptr(ws))
        ev = torch.cuda.Event()
        ev.record(side)
    nbr._vnp_rev = (start, rev, ev)


def _edge_pass(mode, u, v, r1, nbr, C, bn, ns, partial=None, out=None, go=None, sums=None, gv=None):
    """One pass of the fused EdgeConv (vnp_edge_fused modes 0-3) on the per-point matrices u | v, or — ``r1 = (wu, wv)`` — on the
    coordinates themselves (u = v = x [P*3, 1], vnp_edge_fused_r1).  ``bn`` = (mean, rstd, gamma, beta) or None (mode 0)."""
    Pq, k = nbr.shape
    mean, rstd, gamma, beta = bn if bn is not None else (None, None, None, None)
    tail = (ptr(partial), ptr(out), out.shape[1] if out is not None else 0, ptr(go), _ld(go) if go is not None else 0, sums,
            Pq * k if mode == 3 else 0)
    if r1 is None:
        call("vnp_edge_fused", mode, ptr(u), _ld(u), ptr(v), _ld(v), ptr(nbr), Pq, k, C, ptr(mean), ptr(rstd), ptr(gamma), ptr(beta),
             float(ns), *tail, None, 0, 0, ptr(gv), _ld(gv) if gv is not None else 0)
    else:
        call("vnp_edge_fused_r1", mode, ptr(u), ptr(v), ptr(r1[0]), ptr(r1[1]), ptr(nbr), Pq, k, C, ptr(mean), ptr(rstd), ptr(gamma),
             ptr(beta), float(ns), *tail, ptr(gv), _ld(gv) if gv is not None else 0)


def _edge_conv_forward(u, v, nbr, gamma, beta, running_mean, running_var, training, ns, r1=None, C=None, ahead=False):
    """The two forward passes of the fused EdgeConv on u [Pu*3, >= 2C], v [Pq*3, >= 2C] (row strides free) — or, with ``r1`` = (wu, wv)
    [2C] and ``C`` given, on the coordinates u = v = x [P*3, 1]: (out, mean, rstd, nblk)."""
    Pq, k = nbr.shape
    C = u.shape[1] // 2 if C is None else C
    dev = u.device
    if ahead:
        edge_reverse_ahead(nbr, u.shape[0] // 3)
    qb = _lib.lib().vnp_edge_block_queries()
    nblk = (Pq + qb - 1) // qb
    if training:
        part = torch.empty((nblk, C, 2), device=dev, dtype=torch.float32)
        _edge_pass(0, u, v, r1, nbr, C, None, ns, partial=part)
        mean = torch.empty(C, device=dev, dtype=torch.float32)
        rstd = torch.empty(C, device=dev, dtype=torch.float32)
        call("vnp_vn_bn_finalize_chunked", ptr(part), nblk, qb * k, C, Pq * k, ptr(mean), ptr(rstd), ptr(running_mean),
             ptr(running_var), BN_MOMENTUM, BN_EPS)
    else:
        mean = running_mean
        rstd = torch.empty(C, device=dev, dtype=torch.float32)
        call("vnp_rstd_from_var", ptr(running_var), ptr(rstd), C, BN_EPS)
    out = torch.empty((Pq * 3, C), device=dev, dtype=torch.float32)
    _edge_pass(1, u, v, r1, nbr, C, (mean, rstd, gamma, beta), ns, out=out)
    return out, mean, rstd, nblk


def _edge_conv_backward(go, u, v, nbr, gamma, beta, mean, rstd, training, ns, nblk, gu, gv, r1=None, C=None):
    """The backward passes: fills gu [Pu*3, >= 2C] and gv [Pq*3, >= 2C] (row strides free: column halves of one buffer in the packed
    form) and returns the per-channel sums as planar [2, C] rows (dL/dbeta, dL/dgamma).  ``r1`` / ``C`` as in _edge_conv_forward."""
    Pq, k = nbr.shape
    C = u.shape[1] // 2 if C is None else C
    Pu = u.shape[0] // 3
    bn = (mean, rstd, gamma, beta)
    part = torch.empty((nblk, C, 2), device=go.device, dtype=torch.float32)
    _edge_pass(2, u, v, r1, nbr, C, bn, ns, partial=part, go=go)
    S, planar = reduce_pairs(part, nblk, C)
    sums = ptr(S) if training else None
    # gv per query; gu per SOURCE point through the reverse adjacency (built once per graph: conv4 / conv5 share theirs) — a gather
    # with plain stores instead of float atomics
    start, rev = edge_reverse(nbr, Pu)
    _edge_pass(3, u, v, r1, nbr, C, bn, ns, go=go, sums=sums, gv=gv)
    if r1 is None:
        call("vnp_edge_fused_gather", ptr(u), _ld(u), ptr(v), _ld(v), ptr(start), ptr(rev), Pu, k, C, ptr(mean), ptr(rstd), ptr(gamma),
             ptr(beta), ns, ptr(go), _ld(go), sums, Pq * k, ptr(gu), _ld(gu))
    else:
        call("vnp_edge_fused_gather_r1", ptr(u), ptr(v), ptr(r1[0]), ptr(r1[1]), ptr(start), ptr(rev), Pu, k, C, ptr(mean), ptr(rstd),
             ptr(gamma), ptr(beta), ns, ptr(go), _ld(go), sums, Pq * k, ptr(gu), _ld(gu))
    return planar


class EdgeConvFn(Function):
    """VNLinearLeakyReLU on the kNN edge tensor followed by mean_pool over the k neighbours, with the edge tensor never
    materialised (vnp_edge_fused): u, v = the layer's linear maps applied per point ([P*3, 2C'], columns feat | dir),
    nbr [Pq, k] global neighbour ids.  Equivalent to EdgeGatherAddFn -> VNBnLeakyFn -> VNMeanPoolFn."""

    @staticmethod
    def forward(ctx, u, v, nbr, gamma, beta, running_mean, running_var, training, ns):
        u, v = _rows(u), _rows(v)
        nbr = nbr.contiguous()
        out, mean, rstd, nblk = _edge_conv_forward(u, v, nbr, gamma, beta, running_mean, running_var, training, ns, ahead=any(ctx.needs_input_grad))
        ctx.save_for_backward(u, v, nbr, gamma, beta, mean, rstd)
        ctx.cfg = (bool(training), float(ns), nblk)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        u, v, nbr, gamma, beta, mean, rstd = ctx.saved_tensors
        training, ns, nblk = ctx.cfg
        go = _rows(go)
        C = u.shape[1] // 2
        gu = torch.empty((u.shape[0], 2 * C), device=go.device, dtype=torch.float32)
        gv = torch.empty((v.shape[0], 2 * C), device=go.device, dtype=torch.float32)
        S = _edge_conv_backward(go, u, v, nbr, gamma, beta, mean, rstd, training, ns, nblk, gu, gv)
        return gu, gv, None, S[1], S[0], None, None, None, None


class EdgeConvPackedFn(Function):
    """EdgeConvFn for a graph on ONE point set with u and v side by side in one matrix uv [P*3, 4C'] = x [Wu; Wv]^T (one GEMM instead
    of two, forward and backward): u = uv[:, :2C'], v = uv[:, 2C':]; the gradient comes back as one [P*3, 4C'] matrix whose column halves
    the backward kernels fill in place (no zero-padded slice gradients, no add)."""

    @staticmethod
    def forward(ctx, uv, nbr, gamma, beta, running_mean, running_var, training, ns):
        uv = _rows(uv)
        nbr = nbr.contiguous()
        C2 = uv.shape[1] // 2
        out, mean, rstd, nblk = _edge_conv_forward(uv[:, :C2], uv[:, C2:], nbr, gamma, beta, running_mean, running_var, training, ns,
                                                   ahead=any(ctx.needs_input_grad))
        ctx.save_for_backward(uv, nbr, gamma, beta, mean, rstd)
        ctx.cfg = (bool(training), float(ns), nblk)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        uv, nbr, gamma, beta, mean, rstd = ctx.saved_tensors
        training, ns, nblk = ctx.cfg
        go = _rows(go)
        C2 = uv.shape[1] // 2
        guv = torch.empty_like(uv)
        S = _edge_conv_backward(go, uv[:, :C2], uv[:, C2:], nbr, gamma, beta, mean, rstd, training, ns, nblk, guv[:, :C2], guv[:, C2:])
        return guv, None, S[1], S[0], None, None, None, None


class EdgeConvRank1Fn(Function):
    """EdgeConvPackedFn for a layer whose input is ONE VN channel, the points themselves (conv1 of VN_DGCNN_fps and of the VN-PoinTr
    grouper, models/dgcnn.py:291, dgcnn_group.py:225): x [P*3, 1] coordinates in the rows layout, Wuv [4C', 1] = [Wu; Wv] (EdgeWeightsFn).
    The per-point matrices u | v = x Wuv^T ([P*3, 4C'], 100 MB at 65536 points) are never formed: every pass rebuilds an edge's
    pre-activations from the neighbour's and the centre's 12 bytes of coordinates (vnp_edge_fused_r1) instead of gathering 24 C' bytes per
    edge and channel block — the gathers were what bounded the passes.  Forward: the same bits as the general form; gradients: to rounding."""

    @staticmethod
    def forward(ctx, x, Wuv, nbr, gamma, beta, running_mean, running_var, training, ns):
        x = _rows(x).contiguous()
        nbr = nbr.contiguous()
        Wuv = Wuv.contiguous()
        C = Wuv.shape[0] // 4
        if x.shape[1] != 1 or Wuv.shape[1] != 1 or x.shape[0] != nbr.shape[0] * 3:
            raise _lib.VnpError("EdgeConvRank1Fn: one input channel and a graph on the layer's own points expected")
        out, mean, rstd, nblk = _edge_conv_forward(x, x, nbr, gamma, beta, running_mean, running_var, training, ns,
                                                   r1=(Wuv[:2 * C], Wuv[2 * C:]), C=C, ahead=any(ctx.needs_input_grad))
        trace("edge_conv_rank1")
        ctx.save_for_backward(x, Wuv, nbr, gamma, beta, mean, rstd)
        ctx.cfg = (bool(training), float(ns), nblk)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, go):
        x, Wuv, nbr, gamma, beta, mean, rstd = ctx.saved_tensors
        training, ns, nblk = ctx.cfg
        go = _rows(go)
        C = Wuv.shape[0] // 4
        guv = torch.empty((x.shape[0], 4 * C), device=go.device, dtype=torch.float32)
        S = _edge_conv_backward(go, x, x, nbr, gamma, beta, mean, rstd, training, ns, nblk, guv[:, :2 * C], guv[:, 2 * C:],
                                r1=(Wuv[:2 * C], Wuv[2 * C:]), C=C)
        gW = _wgrad(guv, x) if ctx.needs_input_grad[1] else None                 # [4C', 1] = guv^T x
        gx = _dgrad(guv, Wuv) if ctx.needs_input_grad[0] else None               # [P*3, 1] (the input cloud rarely asks for it)
        return gx, gW, None, S[1], S[0], None, None, None, None


class EdgeWeightsFn(Function):
    """[Wu; Wv] [4C', C] of an EdgeConv layer from its VNLinearLeakyReLU weights Wf, Wd [C', 2C] (input channels: C of feature - x, then C
    of x): W cat(f_j - x, x) = W1 f_j + (W2 - W1) x, so Wu = [Wf1; Wd1] acts on the neighbour and Wv = [Wf2 - Wf1; Wd2 - Wd1] on the centre.
    One Function instead of eight autograd nodes of slices, subtractions and concatenations per layer and step."""

    @staticmethod
    def forward(ctx, Wf, Wd):
        Cp, C = Wf.shape[0], Wf.shape[1] // 2
        Wf, Wd = Wf.contiguous(), Wd.contiguous()
        out = torch.empty((4 * Cp, C), device=Wf.device, dtype=torch.float32)
        call("vnp_edge_weights", 0, ptr(Wf), ptr(Wd), ptr(out), Cp, C)
        ctx.cfg = (Cp, C)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        Cp, C = ctx.cfg
        g = g.contiguous()
        gWf = torch.empty((Cp, 2 * C), device=g.device, dtype=torch.float32)
        gWd = torch.empty((Cp, 2 * C), device=g.device, dtype=torch.float32)
        call("vnp_edge_weights", 1, ptr(gWf), ptr(gWd), ptr(g), Cp, C)
        return gWf, gWd
