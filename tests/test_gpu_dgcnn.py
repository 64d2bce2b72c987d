"""BASELINE config 3 path (vn_dgcnn_fps + vn_foldingnet): HIP kNN / FPS / gathers against the oracle's conventions and
the VN_DGCNN_fps encoder against tests/golden/dgcnn_small.npz (reference run with functional stand-ins for the
CUDA-only KNN_CUDA / pointnet2_ops packages: parity for their tie order is unpinned; random fp32 data
has no ties).  Index outputs are compared exactly; floats at 2e-4 of the tensor's max."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import load_golden, relerr, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def D(lib_built):
    return load_golden("dgcnn_small.npz")


def test_knn_matches_oracle(D):
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    x = t(D["x"])
    idx = ops.knn_indices(x, x, 16)
    assert np.array_equal(idx.cpu().numpy(), D["knn_idx"])
    # query != ref, k = 8, ragged sizes; generic-D path (feature-space graphs of VN-PoinTr: D = 3*C)
    g = torch.Generator().manual_seed(3)
    for Dm, Nr, Nq, k in ((3, 700, 300, 8), (12, 130, 130, 8), (96, 128, 128, 8), (192, 70, 33, 5)):
        r, q = torch.randn(2, Nr, Dm, generator=g), torch.randn(2, Nq, Dm, generator=g)
        got, dist = ops.knn_indices(r.cuda(), q.cuda(), k, return_dist=True)
        want = O.knn_torch(r, q, k)
        assert np.array_equal(got.cpu().numpy(), want.numpy()), (Dm, Nr, Nq, k)
        dref = ((q[:, :, None] - r[:, None]) ** 2).sum(-1).gather(2, want)
        assert relerr(dist, dref) < 1e-5
    # exact duplicates: (distance, index) order
    r = torch.randn(1, 40, 3).repeat(1, 3, 1)
    got = ops.knn_indices(r.cuda(), r[:, :40].cuda(), 6)
    assert np.array_equal(got.cpu().numpy(), O.knn_torch(r, r[:, :40], 6).numpy())


def test_knn_module_surface(lib_built):
    from vn_pointcloudcompletion_amd.sampling import KNN
    x = torch.randn(2, 3, 100).cuda()
    dist, idx = KNN(k=16, transpose_mode=False)(x, x)
    assert idx.shape == (2, 16, 100) and dist.shape == (2, 16, 100)          # "bs k np", dgcnn.py:236-237
    assert torch.equal(idx[:, 0], torch.arange(100, device="cuda").expand(2, -1))


def test_fps_matches_oracle(D):
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    idx = ops.fps_indices(t(D["x"]), 512)
    assert np.array_equal(idx.cpu().numpy(), D["fps_idx"])
    x = torch.rand(3, 5000, 3)
    assert np.array_equal(ops.fps_indices(x.cuda(), 224).cpu().numpy(), O.fps_torch(x, 224).numpy())
    assert np.array_equal(ops.fps_indices(x[:, :7].cuda(), 7).cpu().numpy(), O.fps_torch(x[:, :7], 7).numpy())


def test_fps_sizes_and_exact_ties(lib_built):
    """Both FPS kernels (four waves per sample up to 4096 points, sixteen above) on every slots-per-thread instantiation, with
    duplicated points (exact ties in the running distance: the smaller index must win) — against the oracle's sequential restatement."""
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(12)
    for N, m in ((70, 70), (300, 37), (1000, 100), (2048, 512), (3000, 40), (4096, 64), (4097, 30), (9000, 24)):
        x = torch.rand(2, N, 3, generator=g) - 0.5
        x[:, N // 2] = x[:, N // 3]                               # a duplicate: two candidates always tie exactly
        x[:, N - 1] = x[:, 1]
        got = ops.fps_indices(x.cuda(), m, skip_near_origin=False).cpu().numpy()
        assert np.array_equal(got, O.fps_torch(x, m, skip_near_origin=False).numpy()), (N, m)


def test_fps_pointnet2_near_origin_convention(lib_built):
    """pointnet2_ops' kernel never selects a candidate with |p|^2 <= 1e-3 (default here; VNP_FPS_PLAIN=1 / skip_near_origin=False
    is plain FPS).  Both conventions against the oracle's restatement, on clouds seeded with near-origin points."""
    from oracle import vn_oracle as O
    from vn_pointcloudcompletion_amd import ops
    assert ops.SW.FPS_SKIP_NEAR_ORIGIN is True
    g = torch.Generator().manual_seed(4)
    x = torch.rand(3, 2048, 3, generator=g) - 0.5
    x[:, 5::97] *= 0.04                                           # ~21 points per cloud inside the r = 0.0316 ball
    x[1, 0] = 0.0                                                 # the start point itself is near the origin: still sample 0
    near = (x * x).sum(-1) <= 1e-3
    assert near.sum().item() >= 40
    for m in (224, 1024, 2048):
        skip = ops.fps_indices(x.cuda(), m).cpu().long()
        plain = ops.fps_indices(x.cuda(), m, skip_near_origin=False).cpu().long()
        assert torch.equal(skip, O.fps_torch(x, m, skip_near_origin=True)), m
        assert torch.equal(plain, O.fps_torch(x, m, skip_near_origin=False)), m
        picked_near = torch.gather(near, 1, skip)
        picked_near[:, 0] = False                                 # sample 0 is index 0 by definition
        n_valid = (~near).sum(1)
        for b in range(3):                                        # once the far candidates are exhausted the kernel yields index 0
            assert not picked_near[b, :min(m, int(n_valid[b]))].any() or bool(near[b, 0])
        assert torch.gather(near, 1, plain)[:, 1:].any()          # plain FPS does pick them (m >= 224 of 2048)
    tiny = torch.full((1, 64, 3), 0.01)                           # every candidate skipped: the pick is index 0 throughout
    assert torch.equal(ops.fps_indices(tiny.cuda(), 8).cpu().long(), torch.zeros(1, 8, dtype=torch.long))


def test_gather_and_edge_ops(lib_built):
    from vn_pointcloudcompletion_amd import ops
    P, k, W = 50, 4, 70
    u = torch.randn(P * 3, W, device="cuda", requires_grad=True)
    v = torch.randn(P * 3, W, device="cuda", requires_grad=True)
    nbr = torch.randint(0, P, (P, k), device="cuda", dtype=torch.int32)
    out = ops.EdgeGatherAddFn.apply(u, v, nbr)
    u3, v3 = u.detach().view(P, 3, W), v.detach().view(P, 3, W)
    ref = (u3[nbr.long()] + v3[:, None]).reshape(P * k * 3, W)
    assert torch.equal(out, ref)
    g = torch.randn_like(out)
    out.backward(g)
    g4 = g.view(P, k, 3, W)
    assert relerr(v.grad, g4.sum(1).reshape(P * 3, W)) < 1e-6
    gu = torch.zeros(P, 3, W, device="cuda").index_add_(0, nbr.long().reshape(-1), g4.reshape(P * k, 3, W))
    assert relerr(u.grad, gu.reshape(P * 3, W)) < 1e-5
    src = torch.randperm(P, device="cuda")[:20].int()
    x = torch.randn(P * 3, W, device="cuda", requires_grad=True)
    y = ops.GatherPointsFn.apply(x, src)
    assert torch.equal(y, x.detach().view(P, 3, W)[src.long()].reshape(60, W))
    y.backward(torch.ones_like(y))
    assert x.grad.view(P, 3, W)[src.long()].min().item() == 1.0 and x.grad.sum().item() == 20 * 3 * W


def _build(D):
    from vn_pointcloudcompletion_amd.model import PCNNet
    torch.manual_seed(int(D["seed"]))
    cfg = SimpleNamespace(num_coarse=32, latent_dim=512, only_coarse=False, enc_pretrained="none", device="cuda")
    net = PCNNet(cfg, "vn_dgcnn_fps", "vn_foldingnet")
    net.decoder.num_dense = 512
    return net.train()


def test_vn_dgcnn_fps_matches_reference(D):
    net = _build(D)
    names = [k for k, _ in net.encoder.named_parameters()]
    assert names == list(D["grad_names"])
    for (k, p), s, a in zip(net.encoder.named_parameters(), D["weight_sum"], D["weight_abssum"]):
        assert abs(p.double().sum().item() - s) <= 1e-9 * max(1.0, a), k
    x = t(D["x"])
    coarse, g = net.encoder(x)
    assert coarse.shape == (2, 32, 3) and g.shape == (2, 512, 3, 1)
    assert relerr(coarse, D["coarse"]) < 2e-4
    flipped = ((g.cpu() - torch.as_tensor(D["global"])).abs().amax(dim=(2, 3)) > 1e-4).sum().item()
    assert flipped <= 0.01 * 1024
    (coarse * t(D["gc"])).sum().backward(retain_graph=True)
    (g * t(D["gg"])).sum().backward()
    bad = []
    for (k, prm), n in zip(net.encoder.named_parameters(), D["grad_norms"]):
        if n < 0:
            assert prm.grad is None, k
            continue
        if abs(prm.grad.double().norm().item() - n) > 5e-3 * n:
            bad.append((k, prm.grad.double().norm().item(), n))
    assert not bad, bad
    sd = net.state_dict()
    for k in D:
        if k.startswith("after."):
            assert relerr(sd[k[6:]], D[k]) < 1e-4, k


def test_pcnnet_dgcnn_foldingnet_end_to_end(D):
    net = _build(D)
    with torch.no_grad():
        coarse, fine = net(t(D["x"]))
    assert fine.shape == (2, 512, 3)
    assert (fine.cpu() - torch.as_tensor(D["fine"])).abs().max().item() < 2e-3


@pytest.mark.parametrize("Dm,Nr,Nq,k", [(12, 130, 130, 8), (96, 512, 512, 16), (192, 70, 33, 5), (33, 1000, 257, 20), (384, 256, 256, 16)])
def test_knn_matrix_path_equals_per_query_kernel(lib_built, Dm, Nr, Nq, k):
    """Feature-space graphs go through vnp_knn_matrix (distance matrix once + per-row selection): indices and distances are
    bit-identical to the one-thread-per-query kernel vnp_knn (same fmaf chain per entry, same (distance, index) order)."""
    from vn_pointcloudcompletion_amd._lib import call, ptr
    g = torch.Generator().manual_seed(Dm + Nr)
    r, q = torch.randn(2, Nr, Dm, generator=g).cuda(), torch.randn(2, Nq, Dm, generator=g).cuda()
    r[:, 5] = r[:, 3]                                               # exact duplicates: ties
    out = []
    for matrix in (False, True):
        idx = torch.empty(2, Nq, k, device="cuda", dtype=torch.int32)
        dist = torch.empty(2, Nq, k, device="cuda")
        if matrix:
            ws = torch.empty(2 * Nq * Nr, device="cuda")
            call("vnp_knn_matrix", 2, Dm, ptr(r), Nr, ptr(q), Nq, k, ptr(idx), ptr(dist), ptr(ws))
        elif Dm <= 320:
            call("vnp_knn", 2, Dm, ptr(r), Nr, ptr(q), Nq, k, ptr(idx), ptr(dist))
        else:
            continue                                                 # the per-query kernel stops at D = 320 (LDS)
        out.append((idx, dist))
    if len(out) == 2:
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1].view(torch.int32), out[1][1].view(torch.int32))
    # and it IS the k-nearest set in ascending order (checked in fp64; robust to the summation order over D, which at D = 384
    # legitimately reorders neighbours whose distances agree to 1e-7)
    idx, dist = out[-1]
    d64 = ((q.double()[:, :, None] - r.double()[:, None]) ** 2).sum(-1)                       # [2,Nq,Nr]
    mine = d64.gather(2, idx.long())
    assert (mine[..., 1:] >= mine[..., :-1] * (1 - 1e-6)).all()
    assert torch.allclose(dist.double(), mine, rtol=1e-5)
    kth = d64.sort(dim=-1)[0][..., k - 1]
    assert (mine[..., -1] <= kth * (1 + 1e-6)).all()
    assert all(len(set(row.tolist())) == k for row in idx[0, :8].cpu())


@pytest.mark.parametrize("P,k,C,Pu", [(130, 16, 32, 130), (512, 20, 64, 512), (70, 8, 96, 200), (1000, 16, 128, 1000)])
@pytest.mark.parametrize("training", [True, False])
def test_edge_conv_fused_equals_unfused(lib_built, P, k, C, Pu, training):
    """EdgeConv with the edge tensor recomputed inside each pass (ops.EdgeConvFn / vnp_edge_fused) against the materialising
    chain EdgeGatherAddFn -> VNBnLeakyFn -> VNMeanPoolFn: outputs, all gradients and the BatchNorm running statistics."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(P + k + C)
    u0 = torch.randn(Pu * 3, 2 * C, generator=g)
    v0 = torch.randn(P * 3, 2 * C, generator=g)
    nbr = torch.randint(0, Pu, (P, k), generator=g, dtype=torch.int32).cuda()
    gam0, bet0 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    gout = torch.randn(P * 3, C, generator=g).cuda()
    res = []
    for fused in (True, False):
        u, v = u0.cuda().requires_grad_(True), v0.cuda().requires_grad_(True)
        gam, bet = gam0.cuda().requires_grad_(True), bet0.cuda().requires_grad_(True)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        if fused:
            out = ops.EdgeConvFn.apply(u, v, nbr, gam, bet, rm, rv, training, 0.2)
        else:
            pd = ops.EdgeGatherAddFn.apply(u, v, nbr)
            f = ops.VNBnLeakyFn.apply(pd, True, gam, bet, rm, rv, training, 0.2)
            out = ops.VNMeanPoolFn.apply(f, P, k)
        (out * gout).sum().backward()
        res.append((out.detach(), u.grad, v.grad, gam.grad, bet.grad, rm, rv))
    names = ("out", "gu", "gv", "ggamma", "gbeta", "running_mean", "running_var")
    for name, a, b in zip(names, res[0], res[1]):
        scale = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= 2e-5 * scale, (name, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("Pq,k,Pu", [(130, 16, 130), (70, 8, 200), (4096, 20, 4096), (5, 3, 1), (1000, 1, 3000), (600, 16, 50)])
def test_edge_reverse_is_the_reverse_adjacency(lib_built, Pq, k, Pu):
    """vnp_edge_reverse: start = exclusive prefix of the in-degrees, and rev = the queries grouped by the source they name, ASCENDING inside
    a group (a query naming a source twice is listed twice; sources nobody names have empty lists) — the same array on every run: lists of
    up to 64 entries (sorted by one wave), a hub beyond 64 (workgroup sort in LDS) and one beyond 4096 (in place in global memory)."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(Pq * 7 + k)
    nbr = torch.randint(0, Pu, (Pq, k), generator=g, dtype=torch.int32)
    nbr[: Pq // 3] = nbr[0, 0]                                       # a hub: a third of the queries name one source k times each
    start, rev = ops.edge_reverse(nbr.cuda(), Pu)
    start, rev = start.cpu().numpy(), rev.cpu().numpy()
    flat = nbr.numpy().reshape(-1)
    deg = np.bincount(flat, minlength=Pu)
    assert np.array_equal(start, np.concatenate([[0], np.cumsum(deg)]))
    want_q = np.argsort(flat, kind="stable") // k                    # queries grouped by source, ascending inside a group
    assert np.array_equal(rev, want_q)
    again = ops.edge_reverse(nbr.clone().cuda(), Pu)[1].cpu().numpy()
    assert np.array_equal(again, rev)


@pytest.mark.parametrize("P,k,C", [(512, 20, 64), (1000, 16, 128), (2048, 16, 32)])
def test_edge_grad_gather_is_deterministic_and_equals_the_unfused_form(lib_built, P, k, C):
    """The neighbour-feature gradient of the fused EdgeConv as a gather over the reverse adjacency (vnp_edge_fused_gather): the same bits
    on every run (sums taken in double over a list in edge order), and equal to the layer evaluated as gather-add -> BatchNorm / leaky ->
    mean pool on the materialised edge tensor to summation rounding; gv and the parameter gradients are the same kernels' output."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(P + C)
    u0, v0 = torch.randn(P * 3, 2 * C, generator=g), torch.randn(P * 3, 2 * C, generator=g)
    nbr = torch.randint(0, P, (P, k), generator=g, dtype=torch.int32)
    nbr[:, 0] = 7                                                    # a hub with P incoming edges
    gam0, bet0 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    gout = torch.randn(P * 3, C, generator=g).cuda()

    def run(fused):
        u, v = u0.cuda().requires_grad_(True), v0.cuda().requires_grad_(True)
        gam, bet = gam0.cuda().requires_grad_(True), bet0.cuda().requires_grad_(True)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        if fused:
            out = ops.EdgeConvFn.apply(u, v, nbr.cuda(), gam, bet, rm, rv, True, 0.2)
        else:
            pd = ops.EdgeGatherAddFn.apply(u, v, nbr.cuda())
            out = ops.VNMeanPoolFn.apply(ops.VNBnLeakyFn.apply(pd, True, gam, bet, rm, rv, True, 0.2), P, k)
        (out * gout).sum().backward()
        return u.grad, v.grad, gam.grad, bet.grad
    a, b, c = run(True), run(True), run(False)
    for x, y in zip(a, b):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))             # bit-for-bit repeatable
    for name, x, y in zip(("gu", "gv", "ggamma", "gbeta"), a, c):
        scale = y.abs().max().item() + 1e-12
        assert (x - y).abs().max().item() <= 2e-5 * scale, (name, (x - y).abs().max().item(), scale)


@pytest.mark.parametrize("Nr,Nq,k", [(2048, 2048, 20), (512, 512, 16), (128, 128, 16), (700, 300, 8), (4096, 100, 32), (40, 40, 40), (70, 33, 64)])
def test_knn_coords_kernel_equals_matrix_path(lib_built, Nr, Nq, k):
    """vnp_knn_coords_kernel (coordinate graphs: one wave per query, distances in registers, threshold selection) against the
    distance-matrix path: identical indices and distance bits, including exact duplicates (more candidates than the 64-entry list:
    the k-pass fallback), a reference set smaller than the wave, and k equal to the set size."""
    from vn_pointcloudcompletion_amd._lib import call, ptr
    k = min(k, Nr)
    g = torch.Generator().manual_seed(Nr + Nq + k)
    r = (torch.rand(3, Nr, 3, generator=g) - 0.5).cuda()
    q = (r[:, :Nq] if Nq <= Nr else torch.rand(3, Nq, 3, generator=g) - 0.5).contiguous().cuda()
    r[1, : Nr // 2] = r[1, Nr // 2: 2 * (Nr // 2)]                  # sample 1: every point twice -> exact ties everywhere
    r[2] = r[2, :1]                                                # sample 2: all points identical -> every distance ties
    q = q.contiguous()
    out = []
    for matrix in (False, True):
        idx = torch.empty(3, Nq, k, device="cuda", dtype=torch.int32)
        dist = torch.empty(3, Nq, k, device="cuda")
        if matrix:
            ws = torch.empty(3 * Nq * Nr, device="cuda")
            call("vnp_knn_matrix", 3, 3, ptr(r), Nr, ptr(q), Nq, k, ptr(idx), ptr(dist), ptr(ws))
        else:
            call("vnp_knn", 3, 3, ptr(r), Nr, ptr(q), Nq, k, ptr(idx), ptr(dist))
        out.append((idx, dist))
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1].view(torch.int32), out[1][1].view(torch.int32))
    # independent of either kernel: (distance, index) strictly ascending, and it IS the k-nearest set (fp64, as above)
    idx, dist = out[0]
    li = idx.long()
    lex = (dist[..., 1:] > dist[..., :-1]) | ((dist[..., 1:] == dist[..., :-1]) & (li[..., 1:] > li[..., :-1]))
    assert lex.all()
    d64 = ((q.double()[:, :, None] - r.double()[:, None]) ** 2).sum(-1)
    mine = d64.gather(2, li)
    assert torch.allclose(dist.double(), mine, rtol=1e-5, atol=1e-12)
    kth = d64.sort(dim=-1)[0][..., k - 1]
    assert (mine[..., -1] <= kth * (1 + 1e-6) + 1e-12).all()
    # exact ties (samples 1 and 2) resolve to the lowest indices: among equal distances no unselected index lies below a selected one
    same = d64[2, 0]                                               # sample 2: every reference at the same distance
    assert torch.equal(idx[2, 0].cpu(), torch.arange(k, dtype=torch.int32)) and float(same.max() - same.min()) == 0.0


@pytest.mark.parametrize("ahead", [True, False])
def test_fps_plan_equals_the_samplings_one_after_the_other(lib_built, ahead):
    """ops.FpsPlan (the encoder's furthest-point samplings, chained 512 -> 128 plus 224 extra picks of the input, launched on a side
    stream beside other work) against ops.fps_indices + gathers on the main stream: the same picks, coordinates and global row ids, with
    main-stream work queued before (the plan must wait for it: the coordinates are written by that work) and after."""
    from vn_pointcloudcompletion_amd import ops
    B, N = 5, 2048
    g = torch.Generator().manual_seed(11)
    base = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    with ops.switches(FPS_AHEAD=ahead):
        busy = torch.randn(4096, 4096, device="cuda")
        for _ in range(3):
            busy = busy @ busy * 1e-3                       # main-stream work the side stream has to queue behind
        coords = base * 1.0 + 0.0                           # produced on the main stream just before the plan starts
        plan = ops.FpsPlan(coords, (512, 128), (224,))
        for _ in range(3):
            busy = busy @ busy * 1e-3                       # and work that runs beside the samplings
        got = [plan.level(i) for i in range(3)]
        plan.close()
    torch.cuda.synchronize()
    i512 = ops.fps_indices(base, 512)
    c512 = torch.gather(base, 1, i512.long().unsqueeze(-1).expand(-1, -1, 3))
    i128 = ops.fps_indices(c512, 128)
    c128 = torch.gather(c512, 1, i128.long().unsqueeze(-1).expand(-1, -1, 3))
    i224 = ops.fps_indices(base, 224)
    c224 = torch.gather(base, 1, i224.long().unsqueeze(-1).expand(-1, -1, 3))
    off = torch.arange(B, device="cuda", dtype=torch.int32).view(B, 1)
    for (idx, xyz, gid), (ri, rc, n_src) in zip(got, ((i512, c512, N), (i128, c128, 512), (i224, c224, N))):
        assert torch.equal(idx, ri) and torch.equal(xyz, rc)
        assert torch.equal(gid, (ri + off * n_src).reshape(-1))


@pytest.mark.parametrize("B,N,k,Cp,training", [(2, 300, 16, 32, True), (3, 128, 8, 128, True), (1, 70, 5, 24, False), (2, 2048, 20, 32, True)])
def test_edge_conv_on_the_cloud_itself_equals_the_general_form(lib_built, B, N, k, Cp, training):
    """ops.EdgeConvRank1Fn (EdgeConv whose input is ONE VN channel, the coordinates: the per-point matrices u | v = x [Wu; Wv]^T are rebuilt
    from 12 bytes per point inside the passes, vnp_edge_fused_r1) against the general fused form on the formed matrices
    (VNLinearFn -> EdgeConvPackedFn): output, running statistics and the gradients of the weights, of gamma / beta and of the cloud, bit for
    bit (the rebuilt products are rounded once and not contracted with the additions behind them)."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(N + k)
    xyz = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    idx = ops.knn_indices(xyz, xyz, k)
    nbr = (idx + (torch.arange(B, device="cuda", dtype=torch.int32) * N).view(B, 1, 1)).reshape(B * N, k)
    W0 = (torch.randn(4 * Cp, 1, generator=g) * 0.7).cuda()
    gam0, bet0 = (torch.rand(Cp, generator=g) + 0.5).cuda(), (torch.randn(Cp, generator=g) * 0.2).cuda()
    go = torch.randn(B * N * 3, Cp, generator=g).cuda()

    def run(rank1):
        x = xyz.reshape(B * N * 3, 1).clone().requires_grad_(True)
        W, gam, bet = W0.clone().requires_grad_(True), gam0.clone().requires_grad_(True), bet0.clone().requires_grad_(True)
        rm, rv = torch.zeros(Cp, device="cuda"), torch.ones(Cp, device="cuda")
        if rank1:
            out = ops.EdgeConvRank1Fn.apply(x, W, nbr, gam, bet, rm, rv, training, 0.2)
        else:
            out = ops.EdgeConvPackedFn.apply(ops.VNLinearFn.apply(x, W, None, 0), nbr, gam, bet, rm, rv, training, 0.2)
        out.backward(go)
        return [out.detach(), rm, rv, W.grad, gam.grad, bet.grad, x.grad]
    names = ("out", "running_mean", "running_var", "gW", "ggamma", "gbeta", "gx")
    for nm, a, b in zip(names, run(True), run(False)):
        if nm in ("gW", "gx"):
            # products over the source points' gradient gu: the two instantiations of the gather kernel are free to contract a * b + c
            # differently (one rounding more or less per edge term), so these agree to rounding, not to the bit
            assert relerr(a, b) < 2e-6, (nm, relerr(a, b))
        else:
            assert torch.equal(a, b), nm


@pytest.mark.parametrize("enc", ["vn_dgcnn_fps", "vn_pointr"])
def test_extra_coarse_points_carry_the_input_gradient(lib_built, enc):
    """The 224 furthest-point picks of the input that both graph encoders append to their coarse cloud (dgcnn.py:320-323,
    vn_pointr.py:720-722) are a differentiable gather of the INPUT in the reference: a cloud that asks for a gradient must receive
    the cotangent of those rows at the picked points (the plan's own gather kernel has no autograd)."""
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.model import PCNNet
    torch.manual_seed(3)
    B, N = 2, 2048
    if enc == "vn_dgcnn_fps":
        cfg = SimpleNamespace(num_coarse=448, latent_dim=512, only_coarse=True, enc_pretrained="none", device="cuda")
        tail = lambda out: out[1][:, 224:]                   # encoder returns (coarse224, cat(coarse224, fps224)) at num_coarse == 448
    else:
        cfg = SimpleNamespace(num_coarse=1248, latent_dim=1024, only_coarse=True, enc_pretrained="none", device="cuda")
        tail = lambda out: out[:, 1024:]
    net = PCNNet(cfg, enc, "vn_foldingnet").train()
    x = ((torch.rand(B, N, 3) - 0.5).cuda()).requires_grad_(True)
    coarse, _ = net.encoder(x)
    picks = tail(coarse)
    assert picks.shape == (B, 224, 3)
    idx = ops.fps_indices(x.detach(), 224).long()
    assert torch.equal(picks.detach(), torch.gather(x.detach(), 1, idx.unsqueeze(-1).expand(-1, -1, 3)))
    cot = torch.randn_like(picks)
    (gx,) = torch.autograd.grad((picks * cot).sum(), x)
    want = torch.zeros_like(x).scatter_add_(1, idx.unsqueeze(-1).expand(-1, -1, 3), cot)
    assert torch.equal(gx, want)


def test_reverse_adjacency_started_ahead_equals_the_direct_build(lib_built):
    """ops.edge_reverse_ahead (the lists built on the side stream during the forward pass, the backward pass waits for their event) against
    the direct build on the calling stream, with main-stream work queued before (the table is produced by it) and after."""
    from vn_pointcloudcompletion_amd import ops
    g = torch.Generator().manual_seed(4)
    B, N, k = 6, 512, 16
    local = torch.stack([torch.stack([torch.randperm(N, generator=g)[:k] for _ in range(N)]) for _ in range(B)]).int()
    base = (torch.arange(B, dtype=torch.int32) * N).view(B, 1, 1)
    busy = torch.randn(2048, 2048, device="cuda")
    for _ in range(3):
        busy = busy @ busy * 1e-3
    nbr = ((local + base).reshape(B * N, k).cuda() + 0)              # produced on the main stream just before
    ops.edge_reverse_ahead(nbr, B * N)
    for _ in range(3):
        busy = busy @ busy * 1e-3
    s1, r1 = ops.edge_reverse(nbr, B * N)
    s2, r2 = ops.edge_reverse(nbr.clone(), B * N)
    torch.cuda.synchronize()
    assert torch.equal(s1, s2) and torch.equal(r1, r2)
    with ops.switches(REV_AHEAD=False):
        fresh = nbr.clone()
        ops.edge_reverse_ahead(fresh, B * N)
        assert getattr(fresh, "_vnp_rev", None) is None
