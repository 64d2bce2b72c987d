#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference
(/root/reference) on CPU in the build container, and cross-check the oracle against it.

Run:  python tests/golden/make_golden.py        (needs /root/reference; never run on the GPU box)

The reference cannot travel, so only inputs / outputs (data) are written here.  Third-party
CUDA-only imports of the reference (pointnet2_ops, knn_cuda, timm) get empty stand-ins in
sys.modules before the import, exactly as SURVEY.md §8c describes; nothing of the reference
is executed through them on the vn_pointnet / vn_foldingnet path.
"""
import os
import sys
import types
import importlib.util

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)


def _standins():
    """Functional stand-ins for the CUDA-only third-party imports: pure-torch kNN (distance, then index),
    sequential FPS starting at index 0, torch.gather — the conventions of oracle/vn_oracle.py."""
    from oracle import vn_oracle as O
    m = types.ModuleType("pointnet2_ops")
    m.pointnet2_utils = types.ModuleType("pointnet2_ops.pointnet2_utils")
    m.pointnet2_utils.furthest_point_sample = lambda xyz, n: O.fps_torch(xyz, n).int()
    m.pointnet2_utils.gather_operation = lambda feat, idx: torch.gather(feat, 2, idx.long().unsqueeze(1).expand(-1, feat.shape[1], -1))
    sys.modules["pointnet2_ops"] = m
    sys.modules["pointnet2_ops.pointnet2_utils"] = m.pointnet2_utils
    k = types.ModuleType("knn_cuda")

    class KNN:                                                # KNN(k, transpose_mode=False)(ref[B,D,Nr], query[B,D,Nq])
        def __init__(self, k, transpose_mode=False):
            self.k, self.t = k, transpose_mode

        def __call__(self, ref, query):
            if not self.t:
                ref, query = ref.transpose(1, 2), query.transpose(1, 2)
            idx = O.knn_torch(ref, query, self.k)             # [B,Nq,k]
            return None, (idx if self.t else idx.transpose(1, 2).contiguous())
    k.KNN = KNN
    sys.modules["knn_cuda"] = k
    _arange = torch.arange

    def arange(*a, **kw):                                     # dgcnn.py:260 hard-codes device='cuda'
        if "device" in kw and str(kw["device"]).startswith("cuda") and not torch.cuda.is_available():
            kw.pop("device")
        return _arange(*a, **kw)
    torch.arange = arange
    t, tm, tl = types.ModuleType("timm"), types.ModuleType("timm.models"), types.ModuleType("timm.models.layers")
    tl.DropPath = torch.nn.Identity
    tl.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules["timm"], sys.modules["timm.models"], sys.modules["timm.models.layers"] = t, tm, tl
    torch.Tensor.cuda = lambda self, *a, **k: self          # folding_seed(...).cuda(), pcn.py:362


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Rot:
    """Minimal pytorch3d.transforms.Rotate look-alike: row-vector convention p @ R."""
    def __init__(self, R):
        self.R = R

    def transform_points(self, p):
        return torch.matmul(p.expand(self.R.shape[0], -1, -1) if p.shape[0] == 1 else p, self.R)


def random_rotations(n, gen):
    q = torch.randn(n, 4, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).view(n, 3, 3)
    return R


def npz(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


def check(a, b, what, tol=2e-5):
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-6)
    assert err <= tol * max(scale, 1.0), f"oracle != reference for {what}: {err} (scale {scale})"
    return err


def main():
    _standins()
    from oracle import vn_oracle as O
    from oracle import chamfer_oracle as CO
    vl = _load(os.path.join(REF, "models/vn_layers.py"), "ref_vn_layers")
    chp = _load(os.path.join(REF, "extensions/ChamferDistancePytorch/chamfer_python.py"), "ref_chamfer_python")
    from models.model import PCNNet
    from models.pcn import VN_FoldingNet
    from types import SimpleNamespace

    gen = torch.Generator().manual_seed(20240917)
    R = lambda *s: torch.randn(*s, generator=gen)

    # ------------------------------------------------------------------ 1. layers
    out = {}
    B, Ci, Co, N, K = 2, 6, 8, 16, 4
    for tag, shape in (("d4", (B, Ci, 3, N)), ("d5", (B, Ci, 3, N, K))):
        dim = len(shape)
        x = R(*shape)
        # VNLinear
        m = vl.VNLinear(Ci, Co)
        O.closed_form_fill(m.state_dict())
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        g = R(*y.shape)
        y.backward(g)
        out.update({f"lin_{tag}_x": x, f"lin_{tag}_w": m.map_to_feat.weight, f"lin_{tag}_y": y, f"lin_{tag}_g": g,
                    f"lin_{tag}_gx": xr.grad, f"lin_{tag}_gw": m.map_to_feat.weight.grad})
        check(O.vn_linear(x, m.map_to_feat.weight.detach()), y, f"vn_linear {tag}")
        # VNLeakyReLU
        m = vl.VNLeakyReLU(Ci)
        O.closed_form_fill(m.state_dict())
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        g = R(*y.shape)
        y.backward(g)
        out.update({f"leaky_{tag}_w": m.map_to_dir.weight, f"leaky_{tag}_y": y, f"leaky_{tag}_g": g,
                    f"leaky_{tag}_gx": xr.grad, f"leaky_{tag}_gw": m.map_to_dir.weight.grad})
        check(O.vn_leaky_relu(x, m.map_to_dir.weight.detach()), y, f"vn_leaky {tag}")
        # VNBatchNorm train + eval
        m = vl.VNBatchNorm(Ci, dim=dim)
        O.closed_form_fill(m.state_dict())
        m.train()
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        g = R(*y.shape)
        y.backward(g)
        out.update({f"bn_{tag}_gamma": m.bn.weight, f"bn_{tag}_beta": m.bn.bias, f"bn_{tag}_y": y, f"bn_{tag}_g": g,
                    f"bn_{tag}_gx": xr.grad, f"bn_{tag}_ggamma": m.bn.weight.grad, f"bn_{tag}_gbeta": m.bn.bias.grad,
                    f"bn_{tag}_rm": m.bn.running_mean.clone(), f"bn_{tag}_rv": m.bn.running_var.clone()})
        yo, rm, rv = O.vn_batchnorm(x, m.bn.weight.detach(), m.bn.bias.detach(), torch.zeros(Ci), torch.ones(Ci), True)
        check(yo, y, f"vn_bn train {tag}"), check(rm, m.bn.running_mean, "bn rm"), check(rv, m.bn.running_var, "bn rv")
        m.eval()
        ye = m(x)
        out[f"bn_{tag}_y_eval"] = ye
        check(O.vn_batchnorm(x, m.bn.weight.detach(), m.bn.bias.detach(), m.bn.running_mean, m.bn.running_var, False)[0],
              ye, f"vn_bn eval {tag}")
        # VNLinearLeakyReLU train (+ eval)
        m = vl.VNLinearLeakyReLU(Ci, Co, dim=dim)
        O.closed_form_fill(m.state_dict())
        m.train()
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        g = R(*y.shape)
        y.backward(g)
        out.update({f"ll_{tag}_wf": m.map_to_feat.weight, f"ll_{tag}_wd": m.map_to_dir.weight,
                    f"ll_{tag}_gamma": m.batchnorm.bn.weight, f"ll_{tag}_beta": m.batchnorm.bn.bias,
                    f"ll_{tag}_y": y, f"ll_{tag}_g": g, f"ll_{tag}_gx": xr.grad,
                    f"ll_{tag}_gwf": m.map_to_feat.weight.grad, f"ll_{tag}_gwd": m.map_to_dir.weight.grad,
                    f"ll_{tag}_ggamma": m.batchnorm.bn.weight.grad, f"ll_{tag}_gbeta": m.batchnorm.bn.bias.grad,
                    f"ll_{tag}_rm": m.batchnorm.bn.running_mean.clone(), f"ll_{tag}_rv": m.batchnorm.bn.running_var.clone()})
        yo, _, _ = O.vn_linear_leaky_relu(x, m.map_to_feat.weight.detach(), m.map_to_dir.weight.detach(),
                                          m.batchnorm.bn.weight.detach(), m.batchnorm.bn.bias.detach(),
                                          torch.zeros(Co), torch.ones(Co), True)
        check(yo, y, f"vn_linear_leaky {tag}")
        m.eval()
        ye = m(x)
        out[f"ll_{tag}_y_eval"] = ye
        # VNLinearAndLeakyReLU (no BN)
        m = vl.VNLinearAndLeakyReLU(Ci, Co, dim=dim, use_batchnorm="none")
        O.closed_form_fill(m.state_dict())
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        g = R(*y.shape)
        y.backward(g)
        out.update({f"lal_{tag}_wl": m.linear.map_to_feat.weight, f"lal_{tag}_wd": m.leaky_relu.map_to_dir.weight,
                    f"lal_{tag}_y": y, f"lal_{tag}_g": g, f"lal_{tag}_gx": xr.grad,
                    f"lal_{tag}_gwl": m.linear.map_to_feat.weight.grad, f"lal_{tag}_gwd": m.leaky_relu.map_to_dir.weight.grad})
        check(O.vn_linear_and_leaky_relu(x, m.linear.map_to_feat.weight.detach(), m.leaky_relu.map_to_dir.weight.detach()),
              y, f"vn_linear_and_leaky {tag}")
        # VNMaxPool
        m = vl.VNMaxPool(Ci)
        O.closed_form_fill(m.state_dict())
        xr = x.clone().requires_grad_(True)
        y = m(xr)
        g = R(*y.shape)
        y.backward(g)
        out.update({f"pool_{tag}_w": m.map_to_dir.weight, f"pool_{tag}_y": y, f"pool_{tag}_g": g, f"pool_{tag}_gx": xr.grad})
        assert m.map_to_dir.weight.grad is None          # argmax detaches the direction weights
        check(O.vn_max_pool(x, m.map_to_dir.weight.detach()), y, f"vn_max_pool {tag}")
        if dim == 4:
            m = vl.VNLayerNorm(Ci)
            O.closed_form_fill(m.state_dict())
            xr = x.clone().requires_grad_(True)
            y = m(xr)
            g = R(*y.shape)
            y.backward(g)
            out.update({"ln_gamma": m.layer_norm.weight, "ln_beta": m.layer_norm.bias, "ln_y": y, "ln_g": g,
                        "ln_gx": xr.grad, "ln_ggamma": m.layer_norm.weight.grad, "ln_gbeta": m.layer_norm.bias.grad})
            check(O.vn_layer_norm(x, m.layer_norm.weight.detach(), m.layer_norm.bias.detach()), y, "vn_layer_norm")
        out[f"x_{tag}"] = x
    # VNStdFeature (unused by PCNNet; kept for layer-library completeness): C = 8 -> 4 -> 2 -> 3 frame vectors
    for nf in (False, True):
        m = vl.VNStdFeature(8, dim=4, normalize_frame=nf)
        O.closed_form_fill(m.state_dict())
        m.train()
        xs = torch.randn(2, 8, 3, 16, generator=torch.Generator().manual_seed(99))      # own stream: later sections keep theirs
        ys, zs = m(xs)
        tagf = "std_nf" if nf else "std"
        out.update({f"{tagf}_x": xs, f"{tagf}_y": ys, f"{tagf}_z": zs})
    npz("layers.npz", **out)

    # ------------------------------------------------------- 2. small VN_FoldingNet
    cfg = SimpleNamespace(num_coarse=4, latent_dim=8, only_coarse=False, enc_pretrained="none", device="cpu")
    dec = VN_FoldingNet(cfg)
    dec.num_dense = 64
    O.closed_form_fill(dec.state_dict())
    dec.train()
    coarse = 0.3 * R(2, 4, 3)
    fg = R(2, 8, 3, 1)
    Rm = random_rotations(2, gen)
    out = {"coarse": coarse, "fg": fg, "R": Rm}
    sd0 = {"decoder." + k: v.clone() for k, v in dec.state_dict().items()}     # closed-form: not stored
    for tag, rot in (("norot", None), ("rot", Rot(Rm))):
        dec.load_state_dict({k[len("decoder."):]: v for k, v in sd0.items()})
        dec.zero_grad()
        cr, fr = coarse.clone().requires_grad_(True), fg.clone().requires_grad_(True)
        fine = dec(cr, fr, rot)
        g = R(*fine.shape)
        fine.backward(g)
        out.update({f"{tag}_fine": fine, f"{tag}_g": g, f"{tag}_gcoarse": cr.grad, f"{tag}_gfg": fr.grad})
        for k, p in dec.named_parameters():
            if p.numel() <= 4096:
                out[f"{tag}_grad.decoder.{k}"] = p.grad.clone()
            else:
                out[f"{tag}_gradnorm.decoder.{k}"] = p.grad.double().norm()
                out[f"{tag}_gradhead.decoder.{k}"] = p.grad.flatten()[:256].clone()
        for k, v in dec.state_dict().items():
            if "running" in k:
                out[f"{tag}_after.decoder.{k}"] = v.clone()
        fo = O.vn_foldingnet_forward(sd0, coarse, fg, None if rot is None else Rm, True)
        check(fo, fine, f"vn_foldingnet {tag}")
    npz("foldingnet_small.npz", **out)

    # ------------------------------------------------- 3. config 1 end-to-end (B=2)
    # Train-mode BatchNorm over a batch of 2 plus the arg-max pools make this network chaotic for most
    # weight draws: the reference's own fp32 and fp64 runs then disagree by O(0.1) in `fine`.  Parity can
    # only be pinned where the reference itself is reproducible, so scan seeds of torch's DEFAULT init
    # (the reference's init; our modules draw the same stream) for a well-conditioned instance, and store
    # the fp64 run as the yardstick.  Weights are not stored: seed + per-tensor checksums pin them.
    cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cpu")

    def cd64(x, y):
        P = ((x[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)
        return (P.min(2)[0].sqrt().mean() + P.min(1)[0].sqrt().mean()) / 2

    def build(seed, dtype=torch.float32, sd=None):
        torch.manual_seed(seed)
        n_ = PCNNet(cfg, "vn_pointnet", "vn_foldingnet")
        n_.decoder.num_dense = 64 * 16                    # 16384 is hard-coded at pcn.py:332
        if sd is not None:
            n_ = n_.to(dtype)
            n_.decoder.folding_seed = n_.decoder.folding_seed.to(dtype)
            n_.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
        n_.train()
        return n_

    p = torch.rand(2, 256, 3, generator=gen) - 0.5
    c = torch.rand(2, 1024, 3, generator=gen) - 0.5
    chosen = None
    for seed in range(32):
        net = build(seed)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        coarse, fine = net(p)
        net64 = build(seed, torch.float64, sd0)
        c64, f64 = net64(p.double())
        dfine = (fine.double() - f64).abs().max().item()
        print(f"cfg1 seed {seed}: reference fp32 vs fp64 max |fine| diff {dfine:.2e}")
        if dfine < 1e-4:
            chosen = seed
            break
    assert chosen is not None
    d1c, d2c, i1c, i2c = chp.distChamfer(coarse, c)
    d1f, d2f, i1f, i2f = chp.distChamfer(fine, c)
    loss1 = (torch.sqrt(d1c).mean() + torch.sqrt(d2c).mean()) / 2          # metrics/loss.py:28-31
    loss2 = (torch.sqrt(d1f).mean() + torch.sqrt(d2f).mean()) / 2
    (loss1 + loss2).backward()
    l1_64, l2_64 = cd64(c64, c.double()), cd64(f64, c.double())
    (l1_64 + l2_64).backward()
    out = {"seed": chosen, "p": p, "c": c, "coarse": coarse, "fine": fine, "loss1": loss1, "loss2": loss2,
           "coarse64": c64, "fine64": f64, "loss1_64": l1_64, "loss2_64": l2_64,
           "d1c": d1c, "d2c": d2c, "i1c": i1c, "i2c": i2c, "d1f": d1f, "d2f": d2f, "i1f": i1f, "i2f": i2f}
    names, norms, norms64, heads, heads64, wsum, wabs = [], [], [], [], [], [], []
    for (k, prm), (_, prm64) in zip(net.named_parameters(), net64.named_parameters()):
        names.append(k)
        wsum.append(sd0[k].double().sum().item())
        wabs.append(sd0[k].double().abs().sum().item())
        if prm.grad is None:
            norms.append(-1.0), norms64.append(-1.0)
            heads.append(np.zeros(16, np.float32)), heads64.append(np.zeros(16))
        else:
            norms.append(prm.grad.double().norm().item()), norms64.append(prm64.grad.norm().item())
            hh = torch.zeros(16); hh[:min(16, prm.numel())] = prm.grad.flatten()[:16]
            h64 = torch.zeros(16, dtype=torch.float64); h64[:min(16, prm.numel())] = prm64.grad.flatten()[:16]
            heads.append(hh.numpy()), heads64.append(h64.numpy())
    out.update({"grad_names": np.array(names), "grad_norms": np.array(norms), "grad_norms64": np.array(norms64),
                "grad_heads": np.stack(heads), "grad_heads64": np.stack(heads64),
                "weight_sum": np.array(wsum), "weight_abssum": np.array(wabs)})
    for k, v in net.state_dict().items():
        if "running" in k:
            out["after." + k] = v.clone()
    print("cfg1: loss1 32/64", loss1.item(), l1_64.item(), "loss2 32/64", loss2.item(), l2_64.item())
    # oracle cross-check (teacher-forced at the arg-max pools, see SURVEY.md §7 'Hard parts')
    net.load_state_dict(sd0)
    with torch.no_grad():
        c_ref, g_ref = net.encoder(p)
    out["feature_global"] = g_ref
    net.load_state_dict(sd0)
    co, go = O.vn_pointnet_forward(sd0, p, True)
    flipped = ((go - g_ref).abs().amax(dim=(2, 3)) > 1e-4).sum().item()
    print(f"cfg1: oracle vs reference feature_global: {flipped} of {go.shape[0] * go.shape[1]} channels picked another point")
    assert flipped <= 0.005 * go.shape[0] * go.shape[1]
    h = O.vn_linear_and_leaky_relu(g_ref, sd0["encoder.mlp.0.linear.map_to_feat.weight"], sd0["encoder.mlp.0.leaky_relu.map_to_dir.weight"])
    h = O.vn_linear_and_leaky_relu(h, sd0["encoder.mlp.1.linear.map_to_feat.weight"], sd0["encoder.mlp.1.leaky_relu.map_to_dir.weight"])
    h = O.vn_linear(h, sd0["encoder.mlp.2.map_to_feat.weight"]).reshape(-1, 64, 3)
    check(h, coarse, "cfg1 coarse (teacher-forced mlp)", 2e-5)
    fo = O.vn_foldingnet_forward(sd0, coarse.detach(), g_ref, None, True)
    check(fo, fine, "cfg1 fine (teacher-forced)", 2e-4)
    # eval-mode forward with the updated running stats
    net.load_state_dict(sd0)
    net.train()
    with torch.no_grad():
        net(p)
    net.eval()
    with torch.no_grad():
        ce, fe = net(p)
    out["coarse_eval"], out["fine_eval"] = ce, fe
    # rotated input + rot passed to the decoder (equivariance fixture)
    Rm = random_rotations(2, gen)
    net.load_state_dict(sd0)
    net.train()
    with torch.no_grad():
        cr_, fr_ = net(torch.matmul(p, Rm), Rot(Rm))
    out["R"], out["coarse_rot"], out["fine_rot"] = Rm, cr_, fr_
    print("cfg1: equivariance of the reference itself: max |fine_rot - fine@R|",
          (fr_ - torch.matmul(fine.detach(), Rm)).abs().max().item())
    npz("cfg1_e2e.npz", **out)

    # -------------------------------------------------------------- 4. chamfer goldens
    a = torch.rand(4, 100, 3, generator=gen)               # unit_test.py:16-17 shapes
    b = torch.rand(4, 200, 3, generator=gen)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    d1, d2, i1, i2 = chp.distChamfer(ar, br)
    g1, g2 = R(*d1.shape), R(*d2.shape)
    ((d1 * g1).sum() + (d2 * g2).sum()).backward()
    out = {"a": a, "b": b, "d1": d1, "d2": d2, "i1": i1, "i2": i2, "g1": g1, "g2": g2, "ga": ar.grad, "gb": br.grad}
    od1, od2, oi1, oi2 = CO.chamfer_forward(a.numpy(), b.numpy())
    assert np.abs(od1 - d1.detach().numpy()).max() < 1e-6 and (oi1 == i1.numpy()).all() and (oi2 == i2.numpy()).all()
    oga, ogb = CO.chamfer_backward(a.numpy(), b.numpy(), g1.numpy(), g2.numpy(), oi1, oi2)
    assert np.abs(oga - ar.grad.numpy()).max() < 1e-5 and np.abs(ogb - br.grad.numpy()).max() < 1e-5
    a2 = torch.rand(1, 1024, 3, generator=gen) - 0.5
    b2 = torch.rand(1, 16384, 3, generator=gen) - 0.5
    e1, e2, j1, j2 = chp.distChamfer(a2, b2)
    out.update({"a2": a2, "b2": b2, "e1": e1, "e2": e2, "j1": j1, "j2": j2})
    od1, od2, oi1, oi2 = CO.chamfer_forward(a2.numpy(), b2.numpy())
    print("chamfer 1024x16384: max |d| diff", np.abs(od1 - e1.numpy()).max(), np.abs(od2 - e2.numpy()).max(),
          "idx mismatches", int((oi1 != j1.numpy()).sum()), int((oi2 != j2.numpy()).sum()))
    npz("chamfer.npz", **out)


def main_dgcnn():
    """Section 5: VN_DGCNN_fps encoder (BASELINE config 3 at reduced size: B=2, N=640 -> 512 -> 128, k=16, Mc=32)."""
    _standins()
    from oracle import vn_oracle as O
    from models.model import PCNNet
    from types import SimpleNamespace
    gen = torch.Generator().manual_seed(77)
    cfg = SimpleNamespace(num_coarse=32, latent_dim=512, only_coarse=False, enc_pretrained="none", device="cpu")
    x = torch.rand(2, 640, 3, generator=gen) - 0.5
    gc, gg = torch.randn(2, 32, 3, generator=gen), torch.randn(2, 512, 3, 1, generator=gen)

    def build(seed, dtype, sd=None):
        torch.manual_seed(seed)
        n_ = PCNNet(cfg, "vn_dgcnn_fps", "vn_foldingnet")
        n_.decoder.num_dense = 512
        if sd is not None:
            n_ = n_.to(dtype)
            n_.decoder.folding_seed = n_.decoder.folding_seed.to(dtype)
            n_.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
        return n_.train()

    chosen = None
    for seed in range(40):
        net = build(seed, torch.float32)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        coarse, g = net.encoder(x)
        c64, g64 = build(seed, torch.float64, sd0).encoder(x.double())
        dc, dg = (coarse.double() - c64).abs().max().item(), (g.double() - g64).abs().max().item()
        print(f"dgcnn seed {seed}: reference fp32 vs fp64: coarse {dc:.2e} global {dg:.2e}")
        if dc < 2e-5 and dg < 2e-5:
            chosen = seed
            break
    assert chosen is not None
    (coarse * gc).sum().backward(retain_graph=True)
    (g * gg).sum().backward()
    out = {"seed": chosen, "x": x, "coarse": coarse, "global": g, "coarse64": c64, "global64": g64, "gc": gc, "gg": gg}
    names, norms, heads, wsum, wabs = [], [], [], [], []
    for k, prm in net.encoder.named_parameters():
        names.append(k)
        wsum.append(sd0["encoder." + k].double().sum().item())
        wabs.append(sd0["encoder." + k].double().abs().sum().item())
        if prm.grad is None:
            norms.append(-1.0)
            heads.append(np.zeros(16, np.float32))
        else:
            norms.append(prm.grad.double().norm().item())
            hh = torch.zeros(16)
            hh[:min(16, prm.numel())] = prm.grad.flatten()[:16]
            heads.append(hh.numpy())
    out.update({"grad_names": np.array(names), "grad_norms": np.array(norms), "grad_heads": np.stack(heads),
                "weight_sum": np.array(wsum), "weight_abssum": np.array(wabs)})
    for k, v in net.state_dict().items():
        if "running" in k and k.startswith("encoder."):
            out["after." + k] = v.clone()
    # end-to-end forward through the decoder as well (PCNNet, latent_dim = 512)
    net.load_state_dict(sd0)
    with torch.no_grad():
        c2, f2 = net(x)
    out["fine"] = f2
    # oracle cross-check
    co, go = O.vn_dgcnn_fps_forward(sd0, x, True)
    check(co, coarse, "dgcnn coarse", 1e-4), check(go, g, "dgcnn global", 1e-4)
    # the stand-in conventions themselves (pins our HIP kNN / FPS against the oracle's)
    out["knn_idx"] = O.knn_torch(x, x, 16)
    out["fps_idx"] = O.fps_torch(x, 512)
    npz("dgcnn_small.npz", **out)


def main_baselines():
    """Section 8: the non-equivariant baselines of the plugin surface: FoldingNet decoder
    (models/pcn.py:274-317) on a VN encoder's [B,2048,3] feature, and the DGCNN_fps encoder (models/dgcnn.py:19-161)."""
    _standins()
    from oracle import vn_oracle as O
    from types import SimpleNamespace
    import models.pcn as RP
    import models.dgcnn as RD
    gen = torch.Generator().manual_seed(91)
    cfg = SimpleNamespace(num_coarse=64, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cpu")
    out = {}
    # --- FoldingNet decoder
    torch.manual_seed(5)
    dec = RP.FoldingNet(cfg)
    dec.num_dense = 64 * 16
    dec.train()
    coarse = (torch.rand(2, 64, 3, generator=gen) - 0.5).requires_grad_(True)
    fg = (0.5 * torch.randn(2, 2048, 3, generator=gen)).requires_grad_(True)
    gf = torch.randn(2, 1024, 3, generator=gen)
    sd0 = {"decoder." + k: v.clone() for k, v in dec.state_dict().items()}
    fine = dec(coarse, fg)
    (fine * gf).sum().backward()
    new_stats = {}
    fo = O.foldingnet_forward(sd0, coarse.detach(), fg.detach(), True, 4, "decoder", new_stats)
    check(fo, fine, "foldingnet fine", 1e-4)
    for k, v in dec.state_dict().items():
        if "running" in k:
            check(new_stats["decoder." + k], v, "foldingnet " + k, 1e-4)
            out["fold.after." + k] = v.clone()
    out.update({"fold.coarse": coarse.detach(), "fold.fg": fg.detach(), "fold.gf": gf, "fold.fine": fine.detach(),
                "fold.g_coarse": coarse.grad, "fold.g_fg": fg.grad})
    def norms64(tag, module64, module32):
        """fp64 gradient norms, and a strided sample (<= 16384 values) of every gradient in fp32 and fp64: the tests compare the
        HIP path's error against fp64 with the reference's own fp32 error on the same sample."""
        out[tag + ".grad_norms64"] = np.array([prm.grad.double().norm().item() for _, prm in module64.named_parameters()])
        for (k, p64), (_, p32) in zip(module64.named_parameters(), module32.named_parameters()):
            stride = max(1, -(-p64.numel() // 16384))
            out[f"{tag}.gs64.{k}"] = p64.grad.flatten()[::stride].numpy()
            out[f"{tag}.gs32.{k}"] = p32.grad.flatten()[::stride].numpy()

    def summarise(tag, named, sd, pre):
        names, norms, heads, wsum, wabs = [], [], [], [], []
        for k, prm in named:
            names.append(k)
            wsum.append(sd[pre + k].double().sum().item())
            wabs.append(sd[pre + k].double().abs().sum().item())
            norms.append(prm.grad.double().norm().item())
            hh = torch.zeros(16)
            hh[:min(16, prm.numel())] = prm.grad.flatten()[:16]
            heads.append(hh.numpy())
        out.update({tag + ".names": np.array(names), tag + ".grad_norms": np.array(norms), tag + ".grad_heads": np.stack(heads),
                    tag + ".weight_sum": np.array(wsum), tag + ".weight_abssum": np.array(wabs)})
    summarise("fold", dec.named_parameters(), sd0, "decoder.")      # weights are re-created from the seed (5) in the tests
    # fp64 yardstick: this fixture (B = 2, a 6144-value global feature that dominates the first BatchNorm's variance) is
    # ill-conditioned in backward — the reference's own fp32 gradients sit ~1e-3 from its fp64 gradients
    dec64 = RP.FoldingNet(cfg).double()
    dec64.num_dense = 64 * 16
    dec64.folding_seed = dec64.folding_seed.double()
    dec64.load_state_dict({k[len("decoder."):]: (v.double() if v.dtype.is_floating_point else v) for k, v in sd0.items()})
    dec64.train()
    c64 = coarse.detach().double().requires_grad_(True)
    f64 = fg.detach().double().requires_grad_(True)
    fine64 = dec64(c64, f64)
    (fine64 * gf.double()).sum().backward()
    out.update({"fold.fine64": fine64.detach(), "fold.g_coarse64": c64.grad, "fold.g_fg64": f64.grad})
    norms64("fold", dec64, dec)
    # eval mode (BatchNorm on running statistics: no batch coupling, so backward is well conditioned): forward and gradients,
    # fp32 and fp64 — the tight gradient check of the tests; the train-mode gradients above are chaotic at the 1e-7 level
    dec.eval()
    dec.zero_grad()
    ce = coarse.detach().clone().requires_grad_(True)
    fe = fg.detach().clone().requires_grad_(True)
    fine_e = dec(ce, fe)
    (fine_e * gf).sum().backward()
    out["fold.fine_eval"] = fine_e.detach()
    dec64e = RP.FoldingNet(cfg).double()
    dec64e.num_dense = 64 * 16
    dec64e.folding_seed = dec64e.folding_seed.double()
    dec64e.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v) for k, v in dec.state_dict().items()})
    dec64e.eval()
    ce64 = coarse.detach().double().requires_grad_(True)
    fe64 = fg.detach().double().requires_grad_(True)
    (dec64e(ce64, fe64) * gf.double()).sum().backward()
    out.update({"folde.g_coarse": ce.grad, "folde.g_fg": fe.grad, "folde.g_coarse64": ce64.grad, "folde.g_fg64": fe64.grad})
    for (k, p64), (_, p32) in zip(dec64e.named_parameters(), dec.named_parameters()):
        stride = max(1, -(-p64.numel() // 16384))
        out[f"folde.gs64.{k}"] = p64.grad.flatten()[::stride].numpy()
        out[f"folde.gs32.{k}"] = p32.grad.flatten()[::stride].numpy()
    out["folde.grad_norms64"] = np.array([prm.grad.double().norm().item() for _, prm in dec64e.named_parameters()])
    check(O.foldingnet_forward({"decoder." + k: v for k, v in dec.state_dict().items()}, coarse.detach(), fg.detach(), False, 4, "decoder"),
          out["fold.fine_eval"], "foldingnet eval", 1e-4)
    # --- DGCNN_fps encoder (N = 640 -> 512 -> 128, k = 16, 32 coarse points)
    cfg2 = SimpleNamespace(num_coarse=32, latent_dim=1024, only_coarse=True, enc_pretrained="none", device="cpu")
    x = torch.rand(2, 640, 3, generator=gen) - 0.5
    gc, gg = torch.randn(2, 32, 3, generator=gen), torch.randn(2, 1024, generator=gen)
    chosen = None
    for seed in range(20):
        torch.manual_seed(seed)
        enc = RD.DGCNN_fps(cfg2).train()
        sd1 = {"encoder." + k: v.clone() for k, v in enc.state_dict().items()}
        c32, g32 = enc(x)
        enc64 = RD.DGCNN_fps(cfg2).double().train()
        enc64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v) for k, v in enc.state_dict().items()})
        # note: the first forward already updated enc's running stats; they do not enter a train-mode forward
        c64, g64 = enc64(x.double())
        dc, dg = (c32.double() - c64).abs().max().item(), (g32.double() - g64).abs().max().item()
        print(f"dgcnn_fps seed {seed}: reference fp32 vs fp64: coarse {dc:.2e} global {dg:.2e}")
        if dc < 2e-5 and dg < 2e-5:
            chosen = seed
            break
    assert chosen is not None
    (c32 * gc).sum().backward(retain_graph=True)
    (g32 * gg).sum().backward()
    (c64 * gc.double()).sum().backward(retain_graph=True)
    (g64 * gg.double()).sum().backward()
    norms64("enc", enc64, enc)
    co, go = O.dgcnn_fps_forward(sd1, x, True)
    check(co, c32, "dgcnn_fps coarse", 1e-4), check(go, g32, "dgcnn_fps global", 1e-4)
    out.update({"enc.seed": chosen, "enc.x": x, "enc.coarse": c32.detach(), "enc.global": g32.detach(), "enc.coarse64": c64, "enc.global64": g64,
                "enc.gc": gc, "enc.gg": gg})
    summarise("enc", enc.named_parameters(), sd1, "encoder.")
    for k, v in enc.state_dict().items():
        if "running" in k:
            out["enc.after." + k] = v.clone()
    npz("baselines_small.npz", **out)


def main_attention():
    """Section 6: Attention_VN_FoldingNet (BASELINE config 4 decoder) at reduced size: B=2, 128 centres -> 2048 points.
    The decoder (two transformer blocks + two folding stages, 6 train-mode BatchNorms over few hundred samples) amplifies
    fp32 rounding to ~1e-3 in the reference itself; the best-conditioned of 12 default-init seeds is kept together with
    the reference's fp64 run as the yardstick."""
    _standins()
    from oracle import vn_oracle as O
    from models.pcn import Attention_VN_FoldingNet
    from types import SimpleNamespace
    gen = torch.Generator().manual_seed(4242)
    cfg = SimpleNamespace(num_coarse=128, latent_dim=2048, only_coarse=False, enc_pretrained="none", device="cpu")
    coarse = 0.4 * (torch.rand(2, 128, 3, generator=gen) - 0.5)
    fg = 0.3 * torch.randn(2, 2048, 3, 1, generator=gen)

    def build(seed, dtype, sd=None):
        torch.manual_seed(seed)
        d = Attention_VN_FoldingNet(cfg)
        if sd is not None:
            d = d.to(dtype)
            d.folding_seed = d.folding_seed.to(dtype)
            d.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
        return d.train()

    best = (None, 1e9)
    for seed in range(12):
        dec = build(seed, torch.float32)
        sd0 = {k: v.clone() for k, v in dec.state_dict().items()}
        with torch.no_grad():
            fine = dec(coarse, fg)
            f64 = build(seed, torch.float64, sd0)(coarse.double(), fg.double())
        d = (fine.double() - f64).abs().max().item()
        print(f"attention decoder seed {seed}: reference fp32 vs fp64 max |fine| diff {d:.2e}")
        if d < best[1]:
            best = (seed, d)
    chosen = best[0]
    dec = build(chosen, torch.float32)
    sd0 = {k: v.clone() for k, v in dec.state_dict().items()}
    cr, fr = coarse.clone().requires_grad_(True), fg.clone().requires_grad_(True)
    fine = dec(cr, fr)
    with torch.no_grad():
        f64 = build(chosen, torch.float64, sd0)(coarse.double(), fg.double())
    gout = torch.randn(*fine.shape, generator=gen)
    fine.backward(gout)
    out = {"seed": chosen, "coarse": coarse, "fg": fg, "fine": fine, "fine64": f64, "g": gout, "gcoarse": cr.grad, "gfg": fr.grad}
    names, norms, heads, wsum, wabs = [], [], [], [], []
    for k, prm in dec.named_parameters():
        names.append(k)
        wsum.append(sd0[k].double().sum().item())
        wabs.append(sd0[k].double().abs().sum().item())
        if prm.grad is None:
            norms.append(-1.0)
            heads.append(np.zeros(16, np.float32))
        else:
            norms.append(prm.grad.double().norm().item())
            hh = torch.zeros(16)
            hh[:min(16, prm.numel())] = prm.grad.flatten()[:16]
            heads.append(hh.numpy())
    out.update({"grad_names": np.array(names), "grad_norms": np.array(norms), "grad_heads": np.stack(heads),
                "weight_sum": np.array(wsum), "weight_abssum": np.array(wabs)})
    for k, v in dec.state_dict().items():
        if "running" in k:
            out["after." + k] = v.clone()
    # stage-level captures for teacher-forced checks (each stage alone is far better conditioned than the chain)
    dec2 = build(chosen, torch.float32)
    with torch.no_grad():
        bs, N = 2, 128
        centers = coarse.unsqueeze(1).expand(-1, 384, -1, -1).reshape(bs, -1, N).transpose(2, 1)
        tok = dec2.downsize_global(fg).expand(-1, -1, -1, N).reshape(bs, -1, N).transpose(2, 1) + centers
        b0 = dec2.transformer[0](tok)
        b1 = dec2.transformer[1](b0)
        feat = b1.reshape(bs, N, -1, 3).permute(0, 2, 3, 1).permute(0, 3, 1, 2).reshape(bs * N, -1, 3)
        features = feat.view(bs * N, 384, 3, 1).expand(-1, -1, -1, 16)
        seed_ = dec2.folding_seed.view(1, 1, 3, 16).expand(bs * N, -1, -1, -1)
        fd1 = dec2.vn_folding1(torch.cat([seed_, features], dim=1))
    out.update({"tok_in": tok, "blk0_out": b0, "blk1_out": b1, "fd1": fd1})
    sdd = {"decoder." + k: v for k, v in sd0.items()}
    check(O.vn_block(sdd, "decoder.transformer.0", tok, 8, 1, True), b0, "vn_block 0 (teacher-forced)", 1e-4)
    check(O.vn_block(sdd, "decoder.transformer.1", b0, 8, 1, True), b1, "vn_block 1 (teacher-forced)", 1e-4)
    fo = O.attention_vn_foldingnet_forward(sdd, coarse, fg, True)
    gap = (fine.double() - f64).abs().max().item()
    err = (fo - fine).abs().max().item()
    print(f"attention decoder: oracle vs reference {err:.2e} (reference fp32 vs fp64 {gap:.2e})")
    assert err <= max(10 * gap, 1e-3)
    # single VN_Block / VNLayerNorm-level pins come from layers.npz (ln_*) and this end-to-end decoder
    npz("attention_small.npz", **out)


def main_pointr():
    """Section 7: VN_PCTransformer encoder (BASELINE config 5) on B=2, N=640 points (-> 512 -> 128 proxies)."""
    _standins()
    from oracle import vn_oracle as O
    from models.pointr.vn_pointr import VN_PCTransformer
    gen = torch.Generator().manual_seed(555)
    x = torch.rand(2, 640, 3, generator=gen) - 0.5
    gc, gg = torch.randn(2, 1248, 3, generator=gen), torch.randn(2, 1024, 3, 1, generator=gen)

    def build(seed, dtype, sd=None):
        torch.manual_seed(seed)
        n_ = VN_PCTransformer(in_chans=3, embed_dim=384, depth=[6, 8], drop_rate=0., num_query=224, knn_layer=1,
                              dgcnn="vn_dgcnn", trans="vn_trans", memory_profile=False, only_coarse=True)
        if sd is not None:
            n_ = n_.to(dtype)
            n_.load_state_dict({k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()})
        return n_.train()

    best = (None, 1e9)
    for seed in range(6):
        net = build(seed, torch.float32)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        with torch.no_grad():
            c32, g32 = net(x)
            c64, g64 = build(seed, torch.float64, sd0)(x.double())
        d = max((c32.double() - c64).abs().max().item(), (g32.double() - g64).abs().max().item())
        print(f"vn_pointr seed {seed}: reference fp32 vs fp64 gap {d:.2e}")
        if d < best[1]:
            best = (seed, d)
    chosen = best[0]
    net = build(chosen, torch.float32)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    coarse, g = net(x)
    with torch.no_grad():
        c64, g64 = build(chosen, torch.float64, sd0)(x.double())
    (coarse * gc).sum().backward(retain_graph=True)
    (g * gg).sum().backward()
    out = {"seed": chosen, "x": x, "coarse": coarse, "global": g, "coarse64": c64, "global64": g64, "gc": gc, "gg": gg}
    names, norms, wsum, wabs = [], [], [], []
    for k, prm in net.named_parameters():
        names.append(k)
        wsum.append(sd0[k].double().sum().item())
        wabs.append(sd0[k].double().abs().sum().item())
        norms.append(-1.0 if prm.grad is None else prm.grad.double().norm().item())
    out.update({"grad_names": np.array(names), "grad_norms": np.array(norms), "weight_sum": np.array(wsum), "weight_abssum": np.array(wabs)})
    print("vn_pointr: parameters without gradient:", sum(1 for n in norms if n < 0), "of", len(norms),
          "=", sum(p.numel() for p in net.parameters() if p.grad is None), "elements")
    # The encoder is chaotic end to end even inside the reference (fp32 vs fp64 differ by O(1): feature-space kNN graphs
    # and arg-max pools flip), so parity is pinned STAGE BY STAGE with the reference's own discrete decisions (graphs, FPS
    # picks) and inputs teacher-forced.  Re-run the forward manually with the reference's sub-modules, capturing them.
    import models.pointr.utils.dgcnn_group as DG
    import models.pointr.vn_pointr as VP
    net = build(chosen, torch.float32)
    cap = {}
    with torch.no_grad():
        gr = net.grouper
        coor0 = x.transpose(1, 2).contiguous()                                   # [B,3,N]
        xf = gr.vn_get_graph_feature(coor0.unsqueeze(1))
        x1 = gr.pool1(gr.conv1(xf))
        # fps_downsample + dynamic graphs, recording indices
        def knn_feat(f):
            B_, C_, _, N_ = f.shape
            return DG.knn(f.view(B_, -1, N_), f.view(B_, -1, N_))[1]             # [B,k,N]
        fps1 = O.fps_torch(x, 512)
        coor_q, f_q = gr.fps_downsample(coor0, x1, 512)
        i4 = knn_feat(f_q)
        f4 = gr.pool2(gr.conv4(gr.vn_get_graph_feature(f_q, idx=i4)))
        i5 = knn_feat(f4)
        f5 = gr.pool3(gr.conv5(gr.vn_get_graph_feature(f4, idx=i5)))
        fps2 = O.fps_torch(coor_q.transpose(1, 2).contiguous(), 128)
        coor_q2, f_q2 = gr.fps_downsample(coor_q, f5, 128)
        i6 = knn_feat(f_q2)
        f6 = gr.pool4(gr.conv6(gr.vn_get_graph_feature(f_q2, idx=i6)))
        coor_ref, f_ref = gr(coor0)
        assert torch.equal(f6, f_ref) and torch.equal(coor_q2, coor_ref), "manual staging != reference grouper"
        cap.update({"g_fq1": f_q, "g_i4": i4.transpose(1, 2).short(), "g_f4": f4, "g_i5": i5.transpose(1, 2).short(), "g_f5": f5,
                    "g_fps1": fps1.short(), "g_fps2": fps2.short(), "g_fq2": f_q2, "g_i6": i6.transpose(1, 2).short(), "g_f6": f6,
                    "g_coor": coor_q2})
        bs, nc = 2, 128
        f = net.vn_input_proj(f6).contiguous().view(bs, -1, nc)
        tok = f.transpose(1, 2)
        knn_index = VP.get_knn_index(coor_q2)
        centers = torch.mean(x, dim=1)
        rep = centers.unsqueeze(-1).expand(-1, -1, nc).unsqueeze(1)
        pos = net.fourth_vn_pos_embed(torch.concat([coor_q2.unsqueeze(1), rep], dim=1)).reshape(bs, -1, nc).transpose(1, 2)
        cap.update({"t_in": tok, "t_pos": pos, "t_knn": (knn_index.view(bs, 8, nc) - (torch.arange(bs) * nc).view(bs, 1, 1)).transpose(1, 2).short()})
        xx = tok
        for i, blk in enumerate(net.encoder):
            xx = blk(xx + pos, knn_index) if i < net.knn_layer else blk(xx + pos)
            if i in (0, 1, 5):
                cap[f"t_blk{i}"] = xx
        vn_x = xx.transpose(1, 2).contiguous().view(bs, -1, 3, nc)
        gfull = net.vn_increase_dim(vn_x)
        cap["g_pre_pool_proj"] = (gfull * torch.sin(torch.arange(gfull.numel(), dtype=torch.float32).view_as(gfull))).sum(dim=(1, 2))   # [B,nc] compact
        gl = net.vn_global_pool(gfull).unsqueeze(-1)
        cp = net.vn_coarse_pred(gl).squeeze(-1)
        assert torch.allclose(gl, g.detach(), atol=1e-6) and torch.allclose(cp, coarse.detach()[:, :1024], atol=1e-6)
    out.update(cap)
    sdd = {"encoder." + k: v for k, v in sd0.items()}
    # oracle, teacher-forced per stage
    check(O.pointr_vn_block(sdd, "encoder.encoder.0", cap["t_in"] + cap["t_pos"], cap["t_knn"].long(), True), cap["t_blk0"], "pointr block 0", 1e-4)
    check(O.pointr_vn_block(sdd, "encoder.encoder.1", cap["t_blk0"] + cap["t_pos"], None, True), cap["t_blk1"], "pointr block 1", 1e-4)
    e4 = O.mean_pool(O._lin_leaky(sdd, "encoder.grouper.conv4", O.vn_graph_feature(cap["g_fq1"], cap["g_i4"].long()), True, None))
    check(e4, cap["g_f4"], "grouper conv4 (teacher-forced graph)", 1e-4)
    co, go = O.vn_pctransformer_forward(sdd, x, True)
    print(f"vn_pointr: oracle vs reference END TO END (chaotic, informational): coarse {(co - coarse).abs().max().item():.2e} "
          f"global {(go - g).abs().max().item():.2e}; reference fp32 vs fp64 {best[1]:.2e}")
    npz("pointr_small.npz", **out)


def main_dcd():
    """Pins C4 / §8f-1,2: the reference's OWN calc_cd / calc_dcd (utils/loss.py:14-74, the copy train.py:22 imports; it is
    line-identical to metrics/loss.py:58-118) and fscore (extensions/ChamferDistancePytorch/fscore.py:3-16), imported and
    run on CPU.  utils/loss.py imports the CUDA-only `chamfer3D.dist_chamfer_3D` at module top; the stand-in below routes
    chamfer_3DDist to the reference's own pure-PyTorch equivalent chamfer_python.distChamfer (the relation its
    unit_test.py:22-33 asserts), so every line of arithmetic that runs is the reference's."""
    from oracle import vn_oracle as O
    cp = _load(os.path.join(REF, "extensions/ChamferDistancePytorch/chamfer_python.py"), "ref_chamfer_python")
    pkg, mod = types.ModuleType("chamfer3D"), types.ModuleType("chamfer3D.dist_chamfer_3D")

    class chamfer_3DDist(torch.nn.Module):                      # dist_chamfer_3D.py:67-74: forward(a, b) -> d1, d2, i1, i2
        def forward(self, a, b):
            return cp.distChamfer(a, b)
    mod.chamfer_3DDist = chamfer_3DDist
    pkg.dist_chamfer_3D = mod
    sys.modules["chamfer3D"], sys.modules["chamfer3D.dist_chamfer_3D"] = pkg, mod
    L = _load(os.path.join(REF, "utils/loss.py"), "ref_utils_loss")
    F = _load(os.path.join(REF, "extensions/ChamferDistancePytorch/fscore.py"), "ref_fscore")

    g = torch.Generator().manual_seed(21)
    out = {}
    # case a: prediction smaller than the ground truth (the train.py:154 call: coarse vs complete cloud);
    # case b: clustered prediction (many ground-truth points share one nearest neighbour -> large bincounts), one sample
    #         shifted far away so that its F-score is 0/0 -> 0 (fscore.py:15)
    xa, ga = torch.rand(3, 200, 3, generator=g) - 0.5, torch.rand(3, 500, 3, generator=g) - 0.5
    xb = (torch.rand(2, 16, 3, generator=g) - 0.5).repeat_interleave(8, dim=1) + 1e-3 * torch.randn(2, 128, 3, generator=g)
    gb = torch.rand(2, 300, 3, generator=g) - 0.5
    xb[1] += 5.0
    for tag, x, gt in (("a", xa, ga), ("b", xb, gb)):
        out[f"x_{tag}"], out[f"gt_{tag}"] = x, gt
        for j, (alpha, lam, non_reg) in enumerate(((200, 0.5, False), (1000, 1, False), (40, 1, True))):
            xr = x.clone().requires_grad_(True)
            loss, cd_p, cd_t, d1, d2, i1, i2 = L.calc_dcd(xr, gt, alpha=alpha, n_lambda=lam, return_raw=True, non_reg=non_reg)
            loss.sum().backward()
            lo, po, to = O.calc_dcd_oracle(x, gt, alpha, lam, non_reg)
            check(lo, loss.detach(), f"calc_dcd {tag}{j}", 1e-5), check(po, cd_p.detach(), "cd_p", 1e-5), check(to, cd_t.detach(), "cd_t", 1e-5)
            out[f"opts_{tag}{j}"] = np.array([alpha, lam, float(non_reg)], np.float64)
            out[f"loss_{tag}{j}"], out[f"gx_{tag}{j}"] = loss, xr.grad
            if j == 0:
                out[f"cd_p_{tag}"], out[f"cd_t_{tag}"] = cd_p, cd_t
                out[f"d1_{tag}"], out[f"d2_{tag}"], out[f"i1_{tag}"], out[f"i2_{tag}"] = d1, d2, i1, i2
                for th in (0.001, 0.0001, 0.01):
                    f, p1, p2 = F.fscore(d1.detach(), d2.detach(), th)
                    fo, q1, q2 = O.fscore_oracle(d1.detach(), d2.detach(), th)
                    assert torch.equal(f, fo) and torch.equal(p1, q1) and torch.equal(p2, q2)
                    out[f"f_{tag}_{th}"] = torch.stack([f, p1, p2])
                res = L.calc_cd(x, gt, calc_f1=True, separate=True)            # the other branches of calc_cd
                out[f"sep_p_{tag}"], out[f"sep_t_{tag}"], out[f"f1_{tag}"] = res[0], res[1], res[2]
    assert float(out["f_b_0.001"][0, 1]) == 0.0                               # the 0/0 -> 0 branch is in the fixture
    npz("dcd.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "pointr"):
        main_pointr()
    if which in ("all", "base"):
        main()
    if which in ("all", "dgcnn"):
        main_dgcnn()
    if which in ("all", "attention"):
        main_attention()
    if which in ("all", "baselines"):
        main_baselines()
    if which in ("all", "dcd"):
        main_dcd()
