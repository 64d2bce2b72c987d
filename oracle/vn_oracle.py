"""Functional torch-CPU restatement of the reference's Vector-Neuron layers and the
VN-PointNet / VN-FoldingNet networks.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Tensors use the reference layout ``[B, C, 3, N, ...]`` (models/vn_layers.py:19).
Parameters are passed as plain tensors / a dict keyed by the reference's
``state_dict`` names, so the same weights drive the reference (when generating the
golden fixtures), this oracle and the HIP path.

Parity status: PINNED — every function here is checked against the imported
reference by ``tests/golden/make_golden.py`` (run in the build container) and against
the committed fixtures by ``tests/test_oracle_golden.py`` (layers, both networks, Chamfer,
and since round 2 calc_dcd / fscore against the reference's utils/loss.py + fscore.py:
tests/golden/dcd.npz) — EXCEPT ``knn_torch`` / ``fps_torch``, which stand for third-party
CUDA packages whose sources are absent: PARITY UNPINNED, conventions stated at each function.
"""
from __future__ import annotations

import torch

EPS = 1e-6          # models/vn_layers.py:10
BN_EPS = 1e-5       # torch.nn.BatchNorm default (models/vn_layers.py:111-114)
BN_MOMENTUM = 0.1
LN_EPS = 1e-5       # torch.nn.LayerNorm default (models/vn_layers.py:135)


# --------------------------------------------------------------------------- layers
def vn_linear(x, w):
    """models/vn_layers.py:12-22: y[b,o,t,...] = sum_c w[o,c] x[b,c,t,...]."""
    return torch.einsum("oc,bc...->bo...", w, x)


def _leaky_project(p, d, negative_slope):
    """Shared tail of models/vn_layers.py:38-42 and :70-73."""
    dot = (p * d).sum(2, keepdim=True)
    dsq = (d * d).sum(2, keepdim=True)
    neg = (dot < 0).to(p.dtype)
    return p - (1.0 - negative_slope) * neg * (dot / (dsq + EPS)) * d


def vn_leaky_relu(x, w_dir, negative_slope=0.2):
    """models/vn_layers.py:25-43 (share_nonlinearity handled by w_dir having 1 row)."""
    d = vn_linear(x, w_dir)
    return _leaky_project(x, d, negative_slope)


def vn_batchnorm(x, gamma, beta, running_mean=None, running_var=None, training=True):
    """models/vn_layers.py:107-127.  Returns (y, new_running_mean, new_running_var).

    Norm over the 3-axis plus EPS, BatchNorm over every axis except channels,
    x / norm * norm_bn.  Running stats follow torch: momentum 0.1, unbiased variance.
    """
    norm = torch.sqrt((x * x).sum(2)) + EPS                      # [B, C, N, ...]
    red = [i for i in range(norm.dim()) if i != 1]
    shape = [1, -1] + [1] * (norm.dim() - 2)
    if training:
        mean = norm.mean(red)
        var = norm.var(red, unbiased=False)
        count = norm.numel() // norm.shape[1]
        if running_mean is not None:
            new_rm = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean.detach()
            new_rv = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * var.detach() * count / max(count - 1, 1)
        else:
            new_rm = new_rv = None
    else:
        mean, var = running_mean, running_var
        new_rm, new_rv = running_mean, running_var
    norm_bn = (norm - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * gamma.view(shape) + beta.view(shape)
    y = x / norm.unsqueeze(2) * norm_bn.unsqueeze(2)
    return y, new_rm, new_rv


def vn_linear_leaky_relu(x, w_feat, w_dir, gamma, beta, running_mean=None, running_var=None,
                         training=True, negative_slope=0.2):
    """models/vn_layers.py:46-74: direction comes from the layer INPUT (:69)."""
    p = vn_linear(x, w_feat)
    p, rm, rv = vn_batchnorm(p, gamma, beta, running_mean, running_var, training)
    d = vn_linear(x, w_dir)
    return _leaky_project(p, d, negative_slope), rm, rv


def vn_linear_and_leaky_relu(x, w_lin, w_dir, negative_slope=0.2):
    """models/vn_layers.py:77-104 with use_batchnorm='none': direction from the linear OUTPUT."""
    y = vn_linear(x, w_lin)
    return vn_leaky_relu(y, w_dir, negative_slope)


def vn_max_pool(x, w_dir, return_index=False, forced_idx=None):
    """models/vn_layers.py:153-167: argmax over the last axis of <x, W_d x>.
    ``forced_idx`` [B,C] (teacher forcing for parity tests): gather at these indices instead of the arg-max; the scores are
    still computed and returned so that the caller can check that the forced choice is a (near-)maximum."""
    d = vn_linear(x, w_dir)
    dot = (x * d).sum(2, keepdim=True)                            # [B,C,1,N]
    idx = dot.argmax(dim=-1, keepdim=True)                        # first maximum
    if forced_idx is not None:
        idx = forced_idx.to(idx.dtype).reshape(idx.shape)
    shape = list(idx.shape)
    shape[2] = 3
    out = torch.gather(x, -1, idx.expand(shape)).squeeze(-1)
    if return_index:
        return out, idx.squeeze(-1).squeeze(2), dot.squeeze(2)
    return out


def mean_pool(x, dim=-1, keepdim=False):
    """models/vn_layers.py:170-171."""
    return x.mean(dim=dim, keepdim=keepdim)


def vn_layer_norm(x, gamma, beta):
    """models/vn_layers.py:129-150: LayerNorm over the channel axis of the vector norms."""
    norm = torch.sqrt((x * x).sum(2)) + EPS                      # [B, C, N]
    nt = norm.transpose(2, 1)                                     # [B, N, C]
    mu = nt.mean(-1, keepdim=True)
    var = nt.var(-1, unbiased=False, keepdim=True)
    nl = (nt - mu) / torch.sqrt(var + LN_EPS) * gamma + beta      # [B, N, C]
    nl = nl.transpose(2, 1)                                       # [B, C, N]
    return x / norm.unsqueeze(2) * nl.unsqueeze(2)


# ------------------------------------------------------------------------- networks
def _bn(sd, prefix):
    return (sd[prefix + ".batchnorm.bn.weight"], sd[prefix + ".batchnorm.bn.bias"],
            sd.get(prefix + ".batchnorm.bn.running_mean"), sd.get(prefix + ".batchnorm.bn.running_var"))


def _lin_leaky(sd, prefix, x, training, new_stats):
    g, b, rm, rv = _bn(sd, prefix)
    y, nrm, nrv = vn_linear_leaky_relu(x, sd[prefix + ".map_to_feat.weight"], sd[prefix + ".map_to_dir.weight"],
                                       g, b, rm, rv, training)
    if new_stats is not None and nrm is not None:
        new_stats[prefix + ".batchnorm.bn.running_mean"] = nrm
        new_stats[prefix + ".batchnorm.bn.running_var"] = nrv
    return y


def vn_pointnet_forward(sd, xyz, training=True, prefix="encoder", new_stats=None, pool_idx=None, trace=None):
    """models/pcn.py:165-184 (VN_PointNet.forward), num_coarse != 224 branch.

    Returns (coarse [B,Mc,3], feature_global [B,C,3,1]).  ``pool_idx`` = (idx1 [B,512], idx2 [B,2*latent]) teacher-forces
    the two arg-max pools (parity tests at sizes where a 1-ulp score difference can flip an index); ``trace`` (a dict)
    receives the pool scores ``dot1``/``dot2`` [B,C,N], the oracle's own arg-max ``idx1``/``idx2`` and the stage tensors."""
    B, N, _ = xyz.shape
    p = prefix
    x = xyz.transpose(2, 1).unsqueeze(1)                          # [B,1,3,N]
    f = _lin_leaky(sd, f"{p}.first_conv.0", x, training, new_stats)
    f = vn_linear(f, sd[f"{p}.first_conv.1.map_to_feat.weight"])
    g, i1, dot1 = vn_max_pool(f, sd[f"{p}.maxpool1.map_to_dir.weight"], True, None if pool_idx is None else pool_idx[0])
    g = g.unsqueeze(-1)
    if trace is not None:
        trace.update(f1=f.detach(), g1=g.detach(), dot1=dot1.detach(), idx1=dot1.detach().argmax(-1))
    f = torch.cat([g.expand(-1, -1, -1, N), f], dim=1)
    f = _lin_leaky(sd, f"{p}.second_conv.0", f, training, new_stats)
    if trace is not None:
        trace.update(f2=f.detach())
    f = vn_linear(f, sd[f"{p}.second_conv.1.map_to_feat.weight"])
    g, i2, dot2 = vn_max_pool(f, sd[f"{p}.maxpool2.map_to_dir.weight"], True, None if pool_idx is None else pool_idx[1])
    g = g.unsqueeze(-1)                                                       # [B,C,3,1]
    if trace is not None:
        trace.update(dot2=dot2.detach(), idx2=dot2.detach().argmax(-1))
    h = vn_linear_and_leaky_relu(g, sd[f"{p}.mlp.0.linear.map_to_feat.weight"],
                                 sd[f"{p}.mlp.0.leaky_relu.map_to_dir.weight"])
    h = vn_linear_and_leaky_relu(h, sd[f"{p}.mlp.1.linear.map_to_feat.weight"],
                                 sd[f"{p}.mlp.1.leaky_relu.map_to_dir.weight"])
    h = vn_linear(h, sd[f"{p}.mlp.2.map_to_feat.weight"])          # [B,Mc,3,1]
    mc = h.shape[1]
    coarse = h.reshape(-1, mc, 3)
    return coarse.contiguous(), g


def folding_seed(grid_size=4, lo=-0.05, hi=0.05):
    """models/pcn.py:359-362: [1,1,3,grid^2], z = 0."""
    lin = torch.linspace(lo, hi, steps=grid_size, dtype=torch.float)
    a = lin.view(1, grid_size).expand(grid_size, grid_size).reshape(1, -1)
    b = lin.view(grid_size, 1).expand(grid_size, grid_size).reshape(1, -1)
    c = torch.zeros_like(a)
    return torch.cat([a, b, c], dim=0).reshape(1, 1, 3, -1)


def vn_foldingnet_forward(sd, coarse, feature_global, rot_R=None, training=True, grid_size=4,
                          prefix="decoder", new_stats=None):
    """models/pcn.py:365-389 (VN_FoldingNet.forward).  ``rot_R`` is a [B,3,3] rotation in the
    pytorch3d row-vector convention (points @ R), or None."""
    p = prefix
    B, mc, _ = coarse.shape
    S = grid_size * grid_size
    md = mc * S
    seed = folding_seed(grid_size).to(coarse)
    if rot_R is not None:
        s = seed.squeeze(1).transpose(1, 2)                      # [1,S,3]
        s = torch.matmul(s.expand(B, -1, -1), rot_R)              # [B,S,3]
        seed = s.transpose(1, 2).unsqueeze(1)                     # [B,1,3,S]
    point_feat = coarse.unsqueeze(2).expand(-1, -1, S, -1).reshape(-1, md, 3).transpose(2, 1).unsqueeze(1)
    seed = seed.unsqueeze(3).expand(B, -1, -1, mc, -1).reshape(B, -1, 3, md)
    fg = feature_global.expand(-1, -1, -1, md)
    feat = torch.cat([fg, seed, point_feat], dim=1)
    f = _lin_leaky(sd, f"{p}.final_conv.0", feat, training, new_stats)
    f = _lin_leaky(sd, f"{p}.final_conv.1", f, training, new_stats)
    f = vn_linear(f, sd[f"{p}.final_conv.2.map_to_feat.weight"])
    fine = f + point_feat
    return fine.squeeze(1).transpose(1, 2).contiguous()


def pcnnet_forward(sd, xyz, rot_R=None, training=True, new_stats=None):
    """models/model.py:52-64 for enc_type='vn_pointnet', dec_type='vn_foldingnet'."""
    coarse, g = vn_pointnet_forward(sd, xyz, training, new_stats=new_stats)
    fine = vn_foldingnet_forward(sd, coarse, g, rot_R, training, new_stats=new_stats)
    return coarse, fine, g


# ------------------------------------------------------------------- graph helpers / VN-DGCNN
class Teacher:
    """Teacher forcing for the full-resolution / gradient parity tests (tests/test_gpu_grad_parity.py): the DISCRETE choices of a network
    — kNN tables [B,Nq,k], FPS picks [B,m], arg-max pool indices [B,C] — recorded from the path under test, consumed here in call order
    instead of this oracle's own searches.  Every search still runs; for each forced choice ``report`` receives how far it is from the
    oracle's own optimum in the oracle's arithmetic, so that the test can assert that each forced choice is one the reference's
    arithmetic could have made (a rounding-level tie), and how many choices are identical outright."""

    def __init__(self, knn=None, fps=None, pool=None):
        self.knn = [t.long() for t in knn] if knn is not None else None
        self.fps = [t.long() for t in fps] if fps is not None else None
        self.pool = [t.long() for t in pool] if pool is not None else None
        self.report = []                                            # (kind, shape, identical fraction, worst relative gap)
        self.taken = {"knn": [], "fps": [], "pool": []}             # what was used, in call order (a free run records its own choices)

    def take_knn(self, ref, query, k):
        own = knn_torch(ref, query, k)
        if self.knn is None:
            self.taken["knn"].append(own)
            return own
        idx = self.knn.pop(0)
        self.taken["knn"].append(idx)
        assert idx.shape == own.shape, (idx.shape, own.shape)
        with torch.no_grad():
            def kth(i):                                             # largest of the k selected squared distances, per query
                sel = torch.gather(ref.unsqueeze(1).expand(-1, query.shape[1], -1, -1), 2, i[..., None].expand(-1, -1, -1, ref.shape[-1]))
                return ((sel - query.unsqueeze(2)) ** 2).sum(-1).max(-1)[0]
            mine, best = kth(idx), kth(own)
            gap = ((mine - best) / best.clamp_min(1e-30)).max().item()
            same = (torch.sort(idx, -1)[0] == torch.sort(own, -1)[0]).all(-1).float().mean().item()
        self.report.append(("knn", tuple(idx.shape), same, gap))
        return idx

    def take_fps(self, xyz, m):
        if self.fps is None:
            self.taken["fps"].append(fps_torch(xyz, m))
            return self.taken["fps"][-1]
        idx = self.fps.pop(0)
        self.taken["fps"].append(idx)
        own = fps_torch(xyz, m)
        assert idx.shape == own.shape, (idx.shape, own.shape)
        self.report.append(("fps", tuple(idx.shape), (idx == own).float().mean().item(), float("nan")))
        return idx

    def take_pool(self, dot_shape):
        return self.pool.pop(0) if self.pool is not None else None

    def note_pool(self, idx, dot):
        """dot [B,C,N]: the oracle's own scores of the pool whose indices were forced."""
        with torch.no_grad():
            chosen = dot.gather(-1, idx.unsqueeze(-1)).squeeze(-1)
            best, lo = dot.max(-1)[0], dot.min(-1)[0]
            gap = ((best - chosen) / (best - lo).clamp_min(1e-30)).max().item()
            same = (dot.argmax(-1) == idx).float().mean().item()
        self.report.append(("pool", tuple(idx.shape), same, gap))

    def exhausted(self):
        return not (self.knn or self.fps or self.pool)


_FREE = Teacher()


def knn_torch(ref, query, k):
    """ref [B,Nr,D], query [B,Nq,D] -> idx [B,Nq,k]: k smallest squared-L2 distances, ascending (direct-form distances).
    Stands for KNN_CUDA 0.2 (third party, source absent; call sites models/dgcnn.py:236,257-259): PARITY UNPINNED for
    tie order — the convention (distance, then index) is ours."""
    d = ((query[:, :, None, :] - ref[:, None, :, :]) ** 2).sum(-1)
    # lexicographic (distance, index): stable sort on the distances
    return torch.sort(d, dim=-1, stable=True)[1][..., :k]


def fps_torch(xyz, m, skip_near_origin=True):
    """xyz [B,N,3] -> idx [B,m] (int64): iterative furthest point sampling starting at index 0, ties to the lowest index.
    Stands for pointnet2_ops.furthest_point_sample (third party, source absent; models/dgcnn.py:210): PARITY UNPINNED.
    ``skip_near_origin`` restates the one documented quirk of that package's published kernel
    (furthest_point_sampling_kernel: ``if (mag <= 1e-3) continue;``): a candidate with x^2+y^2+z^2 <= 1e-3 is never selected and
    its running distance is not updated; if every candidate is skipped the pick is index 0."""
    B, N, _ = xyz.shape
    idx = torch.zeros(B, m, dtype=torch.long)
    md = torch.full((B, N), 1e10, dtype=xyz.dtype)
    skipped = (xyz * xyz).sum(-1) <= 1e-3 if skip_near_origin else torch.zeros(B, N, dtype=torch.bool)
    cur = torch.zeros(B, dtype=torch.long)
    ar = torch.arange(B)
    for r in range(1, m):
        c = xyz[ar, cur]                                          # [B,3]
        d = ((xyz - c[:, None, :]) ** 2).sum(-1)
        md = torch.where(skipped, md, torch.minimum(md, d))
        cand = md.masked_fill(skipped, -1.0)
        cur = cand.argmax(dim=1)                                  # first maximum; all skipped -> all -1 -> index 0
        idx[:, r] = cur
    return idx


def vn_graph_feature(x, idx):
    """models/dgcnn.py:250-278: x [B,C,3,N], idx [B,N,k] -> cat(feature - x, x) [B,2C,3,N,k]."""
    B, C, _, N = x.shape
    k = idx.shape[-1]
    xt = x.permute(0, 3, 1, 2)                                    # [B,N,C,3]
    feat = torch.gather(xt.unsqueeze(1).expand(-1, N, -1, -1, -1), 2,
                        idx[..., None, None].expand(-1, -1, -1, C, 3))          # [B,N,k,C,3]
    xc = xt.unsqueeze(2).expand(-1, -1, k, -1, -1)
    return torch.cat([feat - xc, xc], dim=3).permute(0, 3, 4, 1, 2)


def vn_dgcnn_fps_forward(sd, x, training=True, k=16, prefix="encoder", new_stats=None, pool_idx=None, trace=None):
    """models/dgcnn.py:282-324 (VN_DGCNN_fps.forward, num_coarse != 224).  x [B,N,3] -> (coarse [B,Mc,3], global [B,512,3,1]).
    ``pool_idx`` [B,512] teacher-forces pool5's arg-max (full-resolution parity tests); ``trace`` receives its scores / own arg-max."""
    p = prefix
    B, N, _ = x.shape
    coor = x                                                       # [B,N,3]
    f = x.transpose(1, 2).unsqueeze(1)                             # [B,1,3,N]

    def conv(name, f, coords):
        e = vn_graph_feature(f, knn_torch(coords, coords, k))
        return mean_pool(_lin_leaky(sd, f"{p}.{name}", e, training, new_stats))

    def down(coords, f, m):
        i = fps_torch(coords, m)
        nc = torch.gather(coords, 1, i[..., None].expand(-1, -1, 3))
        nf = torch.gather(f, 3, i[:, None, None, :].expand(-1, f.shape[1], 3, -1))
        return nc, nf

    f = conv("conv1.0", f, coor)
    cq, fq = down(coor, f, 512)
    f = conv("conv4", fq, cq)
    f = conv("conv5", f, cq)
    cq, fq = down(cq, f, 128)
    f = conv("conv6", fq, cq)
    g, _, dot5 = vn_max_pool(f, sd[f"{p}.pool5.map_to_dir.weight"], True, pool_idx)
    g = g.unsqueeze(-1)
    if trace is not None:
        trace.update(dot5=dot5.detach(), idx5=dot5.detach().argmax(-1))
    h = _lin_leaky(sd, f"{p}.conv7.0", g, training, new_stats)
    h = vn_linear(h, sd[f"{p}.conv7.1.map_to_feat.weight"])
    return h.squeeze(-1).contiguous(), g


# ------------------------------------------------------------- non-equivariant baselines
def _bn1d(sd, prefix, x, training, new_stats):
    """nn.BatchNorm1d on [B, C, N] (momentum 0.1, eps 1e-5)."""
    w, b, rm, rv = sd[f"{prefix}.weight"], sd[f"{prefix}.bias"], sd[f"{prefix}.running_mean"], sd[f"{prefix}.running_var"]
    if not training:
        return (x - rm[None, :, None]) / torch.sqrt(rv[None, :, None] + 1e-5) * w[None, :, None] + b[None, :, None]
    mean = x.mean(dim=(0, 2))
    var = x.var(dim=(0, 2), unbiased=False)
    if new_stats is not None:
        n = x.shape[0] * x.shape[2]
        new_stats[f"{prefix}.running_mean"] = 0.9 * rm + 0.1 * mean
        new_stats[f"{prefix}.running_var"] = 0.9 * rv + 0.1 * var * n / max(n - 1, 1)
    return (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5) * w[None, :, None] + b[None, :, None]


def foldingnet_forward(sd, coarse, feature_global, training=True, grid_size=4, prefix="decoder", new_stats=None):
    """models/pcn.py:274-317 (FoldingNet.forward; `rot` is accepted and ignored there).  coarse [B,Mc,3],
    feature_global [B,2048,3] (flattened to 6144) -> fine [B, Mc*S, 3]."""
    B, Mc, _ = coarse.shape
    S = grid_size ** 2
    Md = Mc * S
    g = feature_global.reshape(B, -1)
    point_feat = coarse.unsqueeze(2).expand(-1, -1, S, -1).reshape(B, Md, 3).transpose(2, 1)        # [B,3,Md]
    lin = torch.linspace(-0.05, 0.05, steps=grid_size, dtype=coarse.dtype)
    a = lin.view(1, grid_size).expand(grid_size, grid_size).reshape(1, -1)
    b = lin.view(grid_size, 1).expand(grid_size, grid_size).reshape(1, -1)
    seed = torch.cat([a, b], dim=0).view(1, 2, 1, S).expand(B, -1, Mc, -1).reshape(B, 2, Md)
    feat = torch.cat([g.unsqueeze(2).expand(-1, -1, Md), seed, point_feat], dim=1)

    def conv(i, x):
        return torch.einsum("oc,bcn->bon", sd[f"{prefix}.final_conv.{i}.weight"][:, :, 0], x) + sd[f"{prefix}.final_conv.{i}.bias"][None, :, None]
    h = torch.relu(_bn1d(sd, f"{prefix}.final_conv.1", conv(0, feat), training, new_stats))
    h = torch.relu(_bn1d(sd, f"{prefix}.final_conv.4", conv(3, h), training, new_stats))
    fine = conv(6, h) + point_feat
    return fine.transpose(1, 2).contiguous()


def graph_feature(coor_q, x_q, coor_k, x_k, k=16):
    """models/dgcnn.py:101-124 (DGCNN_fps.get_graph_feature).  coor [B,3,N], x [B,C,N] -> [B,2C,Nq,k]."""
    B, C, Nk = x_k.shape
    Nq = x_q.shape[2]
    idx = knn_torch(coor_k.transpose(1, 2), coor_q.transpose(1, 2), k)                 # [B,Nq,k]
    flat = x_k.transpose(1, 2).reshape(B * Nk, C)
    gi = (idx + torch.arange(B).view(B, 1, 1) * Nk).reshape(-1)
    feature = flat[gi].view(B, Nq, k, C).permute(0, 3, 1, 2)                             # [B,C,Nq,k]
    xq = x_q.unsqueeze(-1).expand(-1, -1, -1, k)
    return torch.cat((feature - xq, xq), dim=1)


def dgcnn_fps_forward(sd, x, training=True, k=16, prefix="encoder", new_stats=None):
    """models/dgcnn.py:126-161 (DGCNN_fps.forward, num_coarse != 224).  x [B,N,3] -> (coarse [B,Mc,3], feature_global [B,1024])."""
    p = prefix
    B = x.shape[0]
    coor = x.transpose(1, 2)                                                             # [B,3,N]
    f = torch.einsum("oc,bcn->bon", sd[f"{p}.input_trans.weight"][:, :, 0], coor) + sd[f"{p}.input_trans.bias"][None, :, None]

    def layer(name, e):
        y = torch.einsum("oc,bcnk->bonk", sd[f"{p}.{name}.0.weight"][:, :, 0, 0], e)
        y = torch.nn.functional.group_norm(y, 4, sd[f"{p}.{name}.1.weight"], sd[f"{p}.{name}.1.bias"], 1e-5)
        return torch.nn.functional.leaky_relu(y, 0.2).max(dim=-1)[0]

    def down(coor, f, m):
        i = fps_torch(coor.transpose(1, 2), m)
        return (torch.gather(coor, 2, i[:, None, :].expand(-1, 3, -1)), torch.gather(f, 2, i[:, None, :].expand(-1, f.shape[1], -1)))

    f = layer("layer1", graph_feature(coor, f, coor, f, k))
    cq, fq = down(coor, f, 512)
    f = layer("layer2", graph_feature(cq, fq, coor, f, k))
    coor = cq
    f = layer("layer3", graph_feature(coor, f, coor, f, k))
    cq, fq = down(coor, f, 128)
    f = layer("layer4", graph_feature(cq, fq, coor, f, k))
    h = torch.einsum("oc,bcn->bon", sd[f"{p}.increase_dim.0.weight"][:, :, 0], f) + sd[f"{p}.increase_dim.0.bias"][None, :, None]
    h = torch.nn.functional.leaky_relu(_bn1d(sd, f"{p}.increase_dim.1", h, training, new_stats), 0.2)
    h = torch.einsum("oc,bcn->bon", sd[f"{p}.increase_dim.3.weight"][:, :, 0], h) + sd[f"{p}.increase_dim.3.bias"][None, :, None]
    g = h.max(dim=-1)[0]                                                                 # [B,1024]
    c = torch.relu(g @ sd[f"{p}.coarse_pred.0.weight"].t() + sd[f"{p}.coarse_pred.0.bias"])
    c = c @ sd[f"{p}.coarse_pred.2.weight"].t() + sd[f"{p}.coarse_pred.2.bias"]
    return c.reshape(B, -1, 3).contiguous(), g


# ------------------------------------------------------------- VN transformer / attention decoder
def vn_attention(sd, prefix, vn_x, num_heads, scale):
    """models/transformer.py:90-106 (Attention.forward).  vn_x [B,C,3,N]."""
    B, C, _, N = vn_x.shape
    H = num_heads

    def heads(w):
        return vn_linear(vn_x, sd[f"{prefix}.{w}.map_to_feat.weight"]).reshape(B, H, C // H, 3, N).permute(0, 1, 4, 2, 3).reshape(B, H, N, -1)
    q, k, v = heads("proj_vnq"), heads("proj_vnk"), heads("proj_vnv")
    attn = ((q @ k.transpose(-2, -1)) * scale).softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B, N, C, 3).permute(0, 2, 3, 1)
    return vn_linear(x, sd[f"{prefix}.proj_vn.map_to_feat.weight"])


def vn_block(sd, prefix, x, num_heads, scale, training=True, new_stats=None):
    """models/transformer.py:43-74 (VN_Block.forward, knn_index=None).  x [B,N,C*3] tokens."""
    B, N, _ = x.shape
    vn_x = x.transpose(1, 2).reshape(B, -1, 3, N)
    norm_x = vn_layer_norm(vn_x, sd[f"{prefix}.norm1.layer_norm.weight"], sd[f"{prefix}.norm1.layer_norm.bias"])
    vn_x = vn_x + vn_attention(sd, f"{prefix}.attn", norm_x, num_heads, scale)
    n2 = vn_layer_norm(vn_x, sd[f"{prefix}.norm2.layer_norm.weight"], sd[f"{prefix}.norm2.layer_norm.bias"])
    x_2 = _lin_leaky(sd, f"{prefix}.conv4", _lin_leaky(sd, f"{prefix}.conv3", n2, training, new_stats), training, new_stats)
    vn_x = vn_x + x_2
    return vn_x.reshape(B, -1, N).transpose(1, 2)


def attention_vn_foldingnet_forward(sd, coarse, feature_global, training=True, grid_size=4, prefix="decoder", new_stats=None):
    """models/pcn.py:462-520 (Attention_VN_FoldingNet.forward; rot is ignored by the reference)."""
    p = prefix
    bs, N, _ = coarse.shape
    S = grid_size * grid_size
    centers = coarse.unsqueeze(1).expand(-1, 384, -1, -1).reshape(bs, -1, N).transpose(2, 1)
    fg = vn_linear(feature_global, sd[f"{p}.downsize_global.map_to_feat.weight"]).expand(-1, -1, -1, N).reshape(bs, -1, N).transpose(2, 1)
    fg = fg + centers
    for i in range(2):
        fg = vn_block(sd, f"{p}.transformer.{i}", fg, 8, 1, training, new_stats)
    fg = fg.reshape(bs, N, -1, 3).permute(0, 2, 3, 1)
    dim = fg.shape[1]
    x = fg.permute(0, 3, 1, 2).reshape(bs * N, -1, 3)
    features = x.reshape(bs * N, dim, 3, 1).expand(-1, -1, -1, S)
    lin = torch.linspace(-1., 1., steps=grid_size, dtype=torch.float)
    a = lin.view(1, grid_size).expand(grid_size, grid_size).reshape(1, -1)
    b = lin.view(grid_size, 1).expand(grid_size, grid_size).reshape(1, -1)
    seed = torch.cat([a, b, torch.zeros_like(a)], dim=0).to(coarse).view(1, 1, 3, S).expand(bs * N, -1, -1, -1)

    def fold(name, x):
        h = _lin_leaky(sd, f"{p}.{name}.0", x, training, new_stats)
        h = _lin_leaky(sd, f"{p}.{name}.1", h, training, new_stats)
        return vn_linear(h, sd[f"{p}.{name}.2.map_to_feat.weight"])
    fd1 = fold("vn_folding1", torch.cat([seed, features], dim=1))
    fd2 = fold("vn_folding2", torch.cat([fd1, features], dim=1))
    rel = fd2.reshape(bs, N, 3, -1)
    return (rel + coarse.unsqueeze(-1)).transpose(2, 3).reshape(bs, -1, 3)


# ------------------------------------------------------------------------ VN-PoinTr encoder
def _lin_and_leaky_bn(sd, prefix, x, training, new_stats):
    """VNLinearAndLeakyReLU with use_batchnorm='norm' (models/vn_layers.py:92-104)."""
    y = vn_linear(x, sd[f"{prefix}.linear.map_to_feat.weight"])
    g, b = sd[f"{prefix}.batchnorm.bn.weight"], sd[f"{prefix}.batchnorm.bn.bias"]
    rm, rv = sd.get(f"{prefix}.batchnorm.bn.running_mean"), sd.get(f"{prefix}.batchnorm.bn.running_var")
    y, nrm, nrv = vn_batchnorm(y, g, b, rm, rv, training)
    if new_stats is not None and nrm is not None:
        new_stats[f"{prefix}.batchnorm.bn.running_mean"], new_stats[f"{prefix}.batchnorm.bn.running_var"] = nrm, nrv
    return vn_leaky_relu(y, sd[f"{prefix}.leaky_relu.map_to_dir.weight"])


def vn_dgcnn_grouper_forward(sd, prefix, coords, training=True, k=16, new_stats=None, teacher=None):
    """models/pointr/utils/dgcnn_group.py:214-247: every kNN graph is built on the flattened VN FEATURES.
    coords [B,N,3] -> (proxy coords [B,128,3], features [B,128,3,128]).  ``teacher``: see Teacher."""
    p = prefix
    T = teacher or _FREE

    def conv(name, f):
        B, C, _, N = f.shape
        feats = f.permute(0, 3, 1, 2).reshape(B, N, C * 3)
        e = vn_graph_feature(f, T.take_knn(feats.detach(), feats.detach(), k))
        return mean_pool(_lin_leaky(sd, f"{p}.{name}", e, training, new_stats))

    def down(c, f, m):
        i = T.take_fps(c, m)
        return (torch.gather(c, 1, i[..., None].expand(-1, -1, 3)),
                torch.gather(f, 3, i[:, None, None, :].expand(-1, f.shape[1], 3, -1)))

    f = conv("conv1.0", coords.transpose(1, 2).unsqueeze(1))
    cq, fq = down(coords, f, 512)
    f = conv("conv4", fq)
    f = conv("conv5", f)
    cq, fq = down(cq, f, 128)
    return cq, conv("conv6", fq)


def pointr_vn_block(sd, prefix, x, knn_idx, training=True, new_stats=None):
    """models/pointr/vn_pointr.py:384-412 (VN_Block.forward).  x [B,N,384] tokens; knn_idx [B,N,8] or None."""
    B, N, _ = x.shape
    vn_x = x.transpose(1, 2).reshape(B, -1, 3, N)                          # [B,128,3,N]
    norm_x = vn_layer_norm(vn_x, sd[f"{prefix}.norm1.layer_norm.weight"], sd[f"{prefix}.norm1.layer_norm.bias"])
    C, H = 128, 4
    scale = (384 // H) ** -0.5

    def heads(w):
        return vn_linear(norm_x, sd[f"{prefix}.attn.{w}.map_to_feat.weight"]).reshape(B, H, 3 * C // H, 3, N).permute(0, 1, 4, 2, 3).reshape(B, H, N, -1)
    q, kk, v = heads("proj_vnq"), heads("proj_vnk"), heads("proj_vnv")
    attn = ((q @ kk.transpose(-2, -1)) * scale).softmax(dim=-1)
    a = (attn @ v).transpose(1, 2).reshape(B, N, 3 * C, 3).permute(0, 2, 3, 1)
    x_1 = vn_linear(a, sd[f"{prefix}.attn.proj_vn.map_to_feat.weight"])
    if knn_idx is not None:
        f = mean_pool(_lin_leaky(sd, f"{prefix}.conv1", vn_graph_feature(norm_x, knn_idx), training, new_stats))
        x_1 = vn_linear(torch.cat([x_1, f], dim=1), sd[f"{prefix}.conv2.map_to_feat.weight"])
    vn_x = vn_x + x_1
    n2 = vn_layer_norm(vn_x, sd[f"{prefix}.norm2.layer_norm.weight"], sd[f"{prefix}.norm2.layer_norm.bias"])
    vn_x = vn_x + _lin_leaky(sd, f"{prefix}.conv4", _lin_leaky(sd, f"{prefix}.conv3", n2, training, new_stats), training, new_stats)
    return vn_x.reshape(B, -1, N).transpose(1, 2)


def pointr_embed(sd, prefix, f, coor, inpc, training=True, new_stats=None):
    """models/vn_pointr.py:585-637 (vn_input_proj on the grouper's features; fourth_vn_pos_embed on [proxy coordinate, cloud centre]).
    f [B,128,3,nc], coor [B,nc,3], inpc [B,N,3] -> (tokens x [B,nc,384], pos [B,nc,384])."""
    p = prefix
    B, nc = f.shape[0], f.shape[-1]
    f = vn_linear(_lin_leaky(sd, f"{p}.vn_input_proj.0", f, training, new_stats), sd[f"{p}.vn_input_proj.1.map_to_feat.weight"])
    x = f.reshape(B, -1, nc).transpose(1, 2)
    centers = inpc.mean(dim=1)
    pos_in = torch.cat([coor.transpose(1, 2).unsqueeze(1), centers.unsqueeze(-1).expand(-1, -1, nc).unsqueeze(1)], dim=1)   # [B,2,3,nc]
    pos = vn_linear(_lin_and_leaky_bn(sd, f"{p}.fourth_vn_pos_embed.0", pos_in, training, new_stats),
                    sd[f"{p}.fourth_vn_pos_embed.1.map_to_feat.weight"]).reshape(B, -1, nc).transpose(1, 2)
    return x, pos


def pointr_head(sd, prefix, x, training=True, new_stats=None, teacher=None, trace=None):
    """models/vn_pointr.py:655-668, 700-719: vn_increase_dim -> vn_global_pool -> vn_coarse_pred.
    x [B,nc,384] tokens -> (coarse [B,1024,3], global_feature [B,1024,3,1])."""
    p = prefix
    T = teacher or _FREE
    B, nc, _ = x.shape
    vn_x = x.transpose(1, 2).reshape(B, -1, 3, nc)
    g = vn_linear(_lin_and_leaky_bn(sd, f"{p}.vn_increase_dim.0", vn_x, training, new_stats), sd[f"{p}.vn_increase_dim.1.map_to_feat.weight"])
    if trace is not None:
        trace["pre_pool"] = g.detach()
    forced = T.take_pool(None)
    g, own_idx, dot = vn_max_pool(g, sd[f"{p}.vn_global_pool.map_to_dir.weight"], True, forced)
    T.taken["pool"].append(own_idx)
    if forced is not None:
        T.note_pool(forced, dot)
    g = g.unsqueeze(-1)                                                                         # [B,1024,3,1]
    h = vn_linear(g, sd[f"{p}.vn_coarse_pred.0.map_to_feat.weight"])
    h = vn_leaky_relu(h, sd[f"{p}.vn_coarse_pred.1.map_to_dir.weight"])
    h = vn_linear(h, sd[f"{p}.vn_coarse_pred.2.map_to_feat.weight"]).squeeze(-1)                  # [B,1024,3]
    return h, g


def vn_pctransformer_forward(sd, inpc, training=True, knn_layer=1, prefix="encoder", new_stats=None, teacher=None, trace=None):
    """models/pointr/vn_pointr.py:575-722 (VN_PCTransformer.forward, dgcnn='vn_dgcnn', trans='vn_trans', only_coarse=True).
    inpc [B,N,3] -> (coarse [B,1248,3], global_feature [B,1024,3,1]).  ``teacher``: see Teacher (call order: the grouper's four graphs
    and two FPS stages, the k = 8 coordinate graph, the global pool, the FPS-224 subset).  ``trace`` receives the stage tensors
    (grouper_coor / grouper_f, tok_in, pos, knn8, blk0 .. blk5, pre_pool) for the stage-wise parity tests."""
    p = prefix
    T = teacher or _FREE
    cap = (lambda k, v: trace.__setitem__(k, v.detach())) if trace is not None else (lambda k, v: None)
    coor, f = vn_dgcnn_grouper_forward(sd, f"{p}.grouper", inpc, training, 16, new_stats, teacher=T)      # [B,128,3], [B,128,3,128]
    cap("grouper_coor", coor)
    cap("grouper_f", f)
    x, pos = pointr_embed(sd, p, f, coor, inpc, training, new_stats)
    cap("tok_in", x)
    cap("pos", pos)
    knn_idx = T.take_knn(coor, coor, 8)
    cap("knn8", knn_idx)
    for i in range(6):
        x = pointr_vn_block(sd, f"{p}.encoder.{i}", x + pos, knn_idx if i < knn_layer else None, training, new_stats)
        cap(f"blk{i}", x)
    h, g = pointr_head(sd, p, x, training, new_stats, T, trace)
    i = T.take_fps(inpc, 224)
    sparse = torch.gather(inpc, 1, i[..., None].expand(-1, -1, 3))
    return torch.cat([h, sparse], dim=1).contiguous(), g


# --------------------------------------------------------------------------- losses
def chamfer_torch(a, b, chunk=1024):
    """Direct-form squared NN distances (same formula as chamfer3D.cu:33-36) in torch, per sample and in row chunks
    (bounded memory), differentiable.  Returns dist1[B,n], dist2[B,m], idx1, idx2 (int32; first minimum wins)."""
    d1, d2, i1, i2 = [], [], [], []
    for x, y in zip(a, b):
        v1, j1, colmin, colidx = [], [], None, None
        for s in range(0, x.shape[0], chunk):
            diff = x[s:s + chunk, None, :] - y[None, :, :]
            P = (diff * diff).sum(-1)
            v, j = P.min(1)
            v1.append(v), j1.append(j)
            cv, cj = P.min(0)
            cj = cj + s
            if colmin is None:
                colmin, colidx = cv, cj
            else:
                better = cv < colmin                      # strict: the earlier chunk keeps ties (first minimum)
                colmin = torch.where(better, cv, colmin)
                colidx = torch.where(better, cj, colidx)
        d1.append(torch.cat(v1)), i1.append(torch.cat(j1).int()), d2.append(colmin), i2.append(colidx.int())
    return torch.stack(d1), torch.stack(d2), torch.stack(i1), torch.stack(i2)


def chamfer_from_idx(a, b, idx1, idx2):
    """Differentiable squared NN distances given the nearest-neighbour indices (e.g. from oracle/chamfer_oracle.c, which
    finds them ~50x faster than chamfer_torch): dist1[b,i] = |a[b,i] - b[b,idx1[b,i]]|^2 in the direct form of
    chamfer3D.cu:33-36.  Its autograd gradient is chamfer3D.cu:155-174's (each pair contributes 2 g (x1 - x2) to both sides)."""
    i1 = torch.as_tensor(idx1).long().unsqueeze(-1).expand(-1, -1, 3)
    i2 = torch.as_tensor(idx2).long().unsqueeze(-1).expand(-1, -1, 3)
    d1 = ((a - torch.gather(b, 1, i1)) ** 2).sum(-1)
    d2 = ((b - torch.gather(a, 1, i2)) ** 2).sum(-1)
    return d1, d2


def cd_loss_l1(dist1, dist2):
    """metrics/loss.py:28-31."""
    return (torch.sqrt(dist1).mean() + torch.sqrt(dist2).mean()) / 2.0


def cd_loss_l2(dist1, dist2):
    """metrics/loss.py:42-43."""
    return dist1.mean() + dist2.mean()


def l1_cd_metric(dist1, dist2):
    """metrics/metric.py:19-23 (sum over the batch)."""
    return (torch.sqrt(dist1).mean(1) + torch.sqrt(dist2).mean(1)).sum() / 2


def l2_cd_metric(dist1, dist2):
    """metrics/metric.py:12-16."""
    return (dist1.mean(1) + dist2.mean(1)).sum()


def calc_dcd_oracle(x, gt, alpha=1000, n_lambda=1, non_reg=False):
    """metrics/loss.py:77-118 (density-aware Chamfer distance) with the reference's per-sample bincount loop.
    Returns (loss [B], cd_p [B], cd_t [B])."""
    B, n_x, _ = x.shape
    n_gt = gt.shape[1]
    frac_12 = max(1, n_x / n_gt) if non_reg else n_x / n_gt
    frac_21 = max(1, n_gt / n_x) if non_reg else n_gt / n_x
    dist1, dist2, idx1, idx2 = chamfer_torch(gt, x)                 # calc_cd passes (gt, output), loss.py:61
    cd_p = (torch.sqrt(dist1).mean(1) + torch.sqrt(dist2).mean(1)) / 2
    cd_t = dist1.mean(1) + dist2.mean(1)
    e1, e2 = torch.exp(-dist1 * alpha), torch.exp(-dist2 * alpha)
    l1, l2 = [], []
    for b in range(B):
        c1 = torch.bincount(idx1[b].long())
        w1 = (c1[idx1[b].long()].float().detach() ** n_lambda + 1e-6) ** (-1) * frac_21
        l1.append((-e1[b] * w1 + 1.).mean())
        c2 = torch.bincount(idx2[b].long())
        w2 = (c2[idx2[b].long()].float().detach() ** n_lambda + 1e-6) ** (-1) * frac_12
        l2.append((-e2[b] * w2 + 1.).mean())
    return (torch.stack(l1) + torch.stack(l2)) / 2, cd_p, cd_t


def fscore_oracle(dist1, dist2, threshold=0.001):
    """extensions/ChamferDistancePytorch/fscore.py:3-16."""
    p1 = (dist1 < threshold).float().mean(1)
    p2 = (dist2 < threshold).float().mean(1)
    f = 2 * p1 * p2 / (p1 + p2)
    f[torch.isnan(f)] = 0
    return f, p1, p2


# ----------------------------------------------------------------- deterministic fill
def closed_form_fill(state_dict):
    """Fill a state_dict in place with a closed-form rule keyed by (position, flat index), so the
    reference, the oracle and the HIP modules can be given identical weights without storing them.
    Linear weights: amplitude sqrt(3/fan_in) (variance ~ 1.5/fan_in); BN gamma 1+0.1 sin, beta 0.1 sin."""
    import math
    for k, (name, t) in enumerate(state_dict.items()):
        if not t.dtype.is_floating_point:
            continue
        i = torch.arange(t.numel(), dtype=torch.float64)
        s = torch.sin(0.37 * i + 1.0 * k + 0.11 * (i % 7))
        if name.endswith("running_mean") or name.endswith("running_var"):
            continue
        if name.endswith("bn.weight") or name.endswith("norm.weight") or name.endswith("layer_norm.weight"):
            v = 1.0 + 0.1 * s
        elif name.endswith(".bias"):
            v = 0.1 * s
        elif t.dim() >= 2:
            v = math.sqrt(3.0 / t.shape[1]) * s
        else:
            v = 0.5 * s
        t.copy_(v.to(t.dtype).view_as(t))
    return state_dict
