"""INTEGRATION.md is executable: the ctypes stubs it shows a maintainer of the reference (Chamfer, EMD, kNN, FPS) are
extracted from the document, pointed at the built library and checked against this package's own bindings."""
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _doc_namespace():
    from vn_pointcloudcompletion_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = "\n".join(b for b in blocks if "ctypes" in b or "_L." in b)
    code = code.replace('ctypes.CDLL("libvnpcc.so")', f'ctypes.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    return ns


def test_integration_stubs_run_and_agree(lib_built):
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.emd import EarthMoverDistanceFunction
    ns = _doc_namespace()
    g = torch.Generator().manual_seed(0)
    a, b = (torch.rand(2, 300, 3, generator=g) - 0.5).cuda(), (torch.rand(2, 500, 3, generator=g) - 0.5).cuda()
    # Chamfer: the pybind-shaped forward / backward pair.  The protocol at that boundary: the CALLER allocates the outputs,
    # zero-filled on the host and moved to the device — distances as float32, indices as int32 — and the call fills them in place
    d1, d2 = torch.zeros(2, 300).to("cuda"), torch.zeros(2, 500).to("cuda")
    i1, i2 = torch.zeros(2, 300, dtype=torch.int32).to("cuda"), torch.zeros(2, 500, dtype=torch.int32).to("cuda")
    p1, p2 = d1.data_ptr(), i2.data_ptr()
    ns["chamfer_3D"].forward(a, b, d1, d2, i1, i2)
    assert d1.data_ptr() == p1 and i2.data_ptr() == p2 and (d1 > 0).all() and i2.max() > 0          # filled in place
    e1, e2, j1, j2 = ops.ChamferFn.apply(a, b)
    assert torch.equal(d1, e1) and torch.equal(d2, e2) and torch.equal(i1, j1) and torch.equal(i2, j2)
    g1, g2 = torch.rand(2, 300, generator=g).cuda(), torch.rand(2, 500, generator=g).cuda()
    ga, gb = torch.zeros(a.size()).to("cuda"), torch.zeros(b.size()).to("cuda")                        # caller-zeroed gradient buffers
    ns["chamfer_3D"].backward(a, b, ga, gb, g1, g2, i1, i2)
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    o1, o2, _, _ = ops.ChamferFn.apply(ar, br)
    ((o1 * g1).sum() + (o2 * g2).sum()).backward()
    assert torch.allclose(ga, ar.grad, rtol=1e-5, atol=1e-7) and torch.allclose(gb, br.grad, rtol=1e-5, atol=1e-7)
    # EMD: the three emd_cuda functions
    x, y = a[:, :256].contiguous(), b[:, :256].contiguous()
    match = ns["emd_cuda"].approxmatch_forward(x, y)
    assert torch.is_tensor(match) and match.shape == (2, 256, 256)       # `match` only, as emd.cpp:8-10
    cost = ns["emd_cuda"].matchcost_forward(x, y, match)
    ref = EarthMoverDistanceFunction.apply(x, y)
    assert torch.allclose(cost, ref, rtol=1e-5)
    gx, gy = ns["emd_cuda"].matchcost_backward(torch.ones(2, device="cuda"), x, y, match)
    assert torch.isfinite(gx).all() and torch.isfinite(gy).all() and gx.shape == x.shape
    # kNN / FPS in the third-party packages' calling conventions
    dist, idx = ns["knn"](a.transpose(1, 2).contiguous(), b.transpose(1, 2).contiguous(), 8)          # ref [B,3,Nr], query [B,3,Nq]
    assert idx.shape == (2, 8, 500) and torch.equal(idx.transpose(1, 2).int(), ops.knn_indices(a, b, 8))
    fps = ns["furthest_point_sample"](a, 64)
    assert torch.equal(fps, ops.fps_indices(a, 64))
