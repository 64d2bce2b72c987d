"""CPU: the C-ABI library loads and exports every symbol include/vnpcc.h declares (no compute calls without a GPU), and
the host-side logic (transforms, experiment directories, dataset readers, flat optimizer buffers)."""
import ctypes
import json
import os
import re
import struct
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib_built):
    from vn_pointcloudcompletion_amd import _lib
    header = open(os.path.join(ROOT, "include", "vnpcc.h")).read()
    declared = sorted(set(re.findall(r"^\s*(?:int|long long)\s+(vnp_\w+)\s*\(", header, flags=re.M)))
    assert len(declared) >= 25
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/vnpcc.h but not exported by libvnpcc.so"
    assert declared == _lib.exported_symbols(), "ctypes signature table and header disagree"
    assert _lib.chunk_points() > 0
    # ... and every binding passes as many arguments as the header declares (a stale ctypes signature corrupts the call silently)
    bound = _lib.lib()
    for name, params in re.findall(r"^\s*(?:int|long long)\s+(vnp_\w+)\s*\(([^;]*?)\)\s*;", header, flags=re.M | re.S):
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        assert len(getattr(bound, name).argtypes) == n, f"{name}: header declares {n} parameters, ctypes binding has {len(getattr(bound, name).argtypes)}"


def test_no_cpu_fallback():
    from vn_pointcloudcompletion_amd import _lib, ops
    from vn_pointcloudcompletion_amd.chamfer_distance import ChamferDistance
    with pytest.raises(_lib.VnpError):
        ChamferDistance()(torch.rand(1, 8, 3), torch.rand(1, 9, 3))
    with pytest.raises(_lib.VnpError):
        ops.gemm(torch.rand(4, 4), torch.rand(4, 4))
    import vn_pointcloudcompletion_amd as pkg
    src = "".join(open(os.path.join(os.path.dirname(pkg.__file__), f)).read()
                  for f in os.listdir(os.path.dirname(pkg.__file__)) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src          # the product never touches the oracle


def test_transforms():
    from vn_pointcloudcompletion_amd.transforms import Rotate, RotateAxisAngle, random_rotations
    R = random_rotations(5, generator=torch.Generator().manual_seed(0))
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(5, 3, 3), atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(5), atol=1e-5)
    p = torch.randn(5, 7, 3)
    assert torch.allclose(Rotate(R).transform_points(p), p @ R)
    assert Rotate(R).transform_points(torch.randn(1, 16, 3)).shape == (5, 16, 3)        # pcn.py:369-370 broadcast
    z = RotateAxisAngle(torch.tensor([90.0]), axis="Z")
    # pytorch3d convention: +90 deg about Z maps x -> y for row vectors p @ R
    assert torch.allclose(z.transform_points(torch.tensor([[[1.0, 0.0, 0.0]]])), torch.tensor([[[0.0, 1.0, 0.0]]]), atol=1e-6)


def test_experiment_dirs_and_config(tmp_path, monkeypatch):
    from vn_pointcloudcompletion_amd import experiments as X
    monkeypatch.setenv("OUTPUT_DIR", str(tmp_path / "exp"))
    cfg_path = tmp_path / "config.json"
    cfg_path.write_text(json.dumps({"name": "unit", "lr": 1e-4, "dcd_opts": {"alpha": 200, "lambda": 0.5}, "checkpoint": 0}))
    cfg = X.load_config(config_path=str(cfg_path))
    assert cfg.dcd_opts.alpha == 200 and cfg.data_path
    cfg = X.create_experiment(cfg)
    assert re.fullmatch(r"\d\d-\d\d_unit_000", cfg.run_name)
    for sub in ("models", "optimizer", "visualizations", "train", "val"):
        assert os.path.isdir(os.path.join(cfg.exp_dir, sub))
    X.store_config(cfg)
    again = X.load_config(config_path=os.path.join(cfg.exp_dir, "config.json"))
    assert again.run_name == cfg.run_name and again.dcd_opts.alpha == 200
    cfg2 = X.create_experiment(X.load_config(config_path=str(cfg_path)))
    assert cfg2.run_name.endswith("_001")


def test_ply_reader_and_datasets(tmp_path):
    from vn_pointcloudcompletion_amd.dataset import ShapeNet, SyntheticShapes, read_ply_xyz
    pts = np.random.rand(50, 3).astype(np.float32)
    a = tmp_path / "a.ply"
    a.write_text("ply\nformat ascii 1.0\nelement vertex 50\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
                 + "\n".join(" ".join(f"{v:.7g}" for v in p) for p in pts) + "\n")
    assert np.allclose(read_ply_xyz(str(a)), pts, atol=1e-6)
    b = tmp_path / "b.ply"
    with open(b, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 50\nproperty float x\nproperty float y\n"
                b"property float z\nproperty uchar red\nend_header\n")
        for p in pts:
            f.write(struct.pack("<fffB", *p, 7))
    assert np.array_equal(read_ply_xyz(str(b)), pts)
    assert ShapeNet.random_sample(pts, 2048).shape == (2048, 3) and ShapeNet.random_sample(pts, 10).shape == (10, 3)
    ds = SyntheticShapes(4, 32, 64)
    p0, c0 = ds[1]
    p1, c1 = ds[1]
    assert p0.shape == (32, 3) and c0.shape == (64, 3) and torch.equal(p0, p1) and p0.abs().max() <= 0.5


def test_flat_buffers_and_adam_state_dict_format():
    from vn_pointcloudcompletion_amd.optim import FlatBuffers
    m = torch.nn.Sequential(torch.nn.Linear(5, 3, bias=False), torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 2, bias=False))
    m[2].weight.requires_grad = False
    before = [p.detach().clone() for p in m.parameters()]
    fb = FlatBuffers(m.parameters())
    assert [p.shape for p in fb.params] == [(3, 5), (3,), (3,)]
    assert all(o % 4 == 0 for o in fb.offsets) and fb.numel % 4 == 0
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p, b)
    m(torch.randn(4, 5)).sum().backward()
    g = m[0].weight.grad
    assert g.data_ptr() == fb.flat_grad.data_ptr()                  # autograd accumulated IN PLACE into the flat buffer
    assert fb.flat_grad[:15].abs().sum() > 0
    fb.zero_grad()
    assert fb.flat_grad.abs().sum() == 0 and m[0].weight.grad.data_ptr() == fb.flat_grad.data_ptr()
    m[0].weight.grad = None                                          # a caller used set_to_none=True
    fb.zero_grad()
    assert m[0].weight.grad is not None


def test_integration_md_stubs_declare_correct_argtypes():
    """Every libvnpcc function the INTEGRATION.md stubs call has its argtypes declared there (ctypes would otherwise pass 64-bit
    pointers as C ints) and the declared lists equal the signature table of _lib.py.  The stubs are executed on the GPU in
    tests/test_gpu_integration_doc.py."""
    from vn_pointcloudcompletion_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = "\n".join(b for b in re.findall(r"```python\n(.*?)```", text, flags=re.S) if "ctypes" in b or "_L." in b)
    compile(code, "INTEGRATION.md", "exec")
    called = set(re.findall(r"_L\.(vnp_\w+)\(", code))
    declared = dict(re.findall(r"_L\.(vnp_\w+)\.argtypes\s*=\s*\[(.*?)\]", code))
    assert called and called <= set(declared), sorted(called - set(declared))
    names = {_lib._vp: "_vp", _lib._i: "_i", _lib._ll: "_ll", _lib._f: "_f"}
    for fn, lst in declared.items():
        assert [x.strip() for x in lst.split(",")] == ["_vp"] + [names[t] for t in _lib._SIGNATURES[fn]], fn


def test_resume_does_not_decay_the_learning_rate_twice():
    """resuming at epoch >= 50 replayed StepLR on top of the already-decayed lr that load_state_dict restores
    (0.64 x base instead of 0.8 x base at epoch 55).  lr(epoch) of a run resumed at 55 must equal the uninterrupted run's,
    including across the next boundary at 100."""
    from vn_pointcloudcompletion_amd.train import make_scheduler

    def fresh():
        prm = [torch.nn.Parameter(torch.zeros(3))]
        return torch.optim.Adam(prm, lr=1e-3)
    opt = fresh()
    sch = make_scheduler(opt, 0)
    lrs, saved = [], None
    for epoch in range(0, 120):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
        if epoch == 54:
            saved = {k: (v if k != "param_groups" else [dict(g) for g in v]) for k, v in opt.state_dict().items()}
    assert abs(lrs[49] - 1e-3) < 1e-12 and abs(lrs[50] - 0.8e-3) < 1e-12 and abs(lrs[100] - 0.64e-3) < 1e-12
    opt2 = fresh()
    opt2.load_state_dict(saved)                                  # as train.py does on --resume (epoch 54 saved -> start at 55)
    sch2 = make_scheduler(opt2, 55)
    assert abs(sch2.get_last_lr()[0] - lrs[55]) < 1e-12
    for epoch in range(55, 120):
        assert abs(opt2.param_groups[0]["lr"] - lrs[epoch]) < 1e-12, epoch
        opt2.step()
        sch2.step()


def test_sharded_batch_sampler_covers_every_sample_once_with_gradient_weights():
    from vn_pointcloudcompletion_amd.distributed import ShardedBatchSampler
    n, bs, world = 45, 12, 8                                    # the reference's batch_size 12 on 8 GPUs; last batch has 9
    ranks = [ShardedBatchSampler(n, bs, shuffle=True, seed=3, rank=r, world=world, mode="train") for r in range(world)]
    for s in ranks:
        s.set_epoch(7)
    per_rank = [list(s) for s in ranks]
    assert all(len(b) == len(ranks[0]) == 4 for b in per_rank)                       # same number of steps on every rank
    glob = ranks[0].global_batches()
    for i, g in enumerate(glob):
        shards = [per_rank[r][i] for r in range(world)]
        assert sum(shards, []) == g                                                  # contiguous slices, in rank order
        # all-reduce AVERAGE of weight * (local mean) == global-batch mean, for any per-sample values
        vals = {j: float(j * j % 17) for j in g}
        avg = sum(ranks[r].weight(i) * (sum(vals[j] for j in shards[r]) / len(shards[r])) for r in range(world)) / world
        assert abs(avg - sum(vals.values()) / len(g)) < 1e-9
    assert sorted(sum(glob, [])) == list(range(n))
    ranks[0].set_epoch(8)
    assert ranks[0].global_batches() != glob                                         # a new order every epoch, same on all ranks
    # a batch smaller than the world: replicated in train mode (aligned collectives), dealt out once in eval mode
    tr = [list(ShardedBatchSampler(3, 12, False, rank=r, world=8, mode="train")) for r in range(8)]
    assert all(t == [[0, 1, 2]] for t in tr)
    ev = [list(ShardedBatchSampler(3, 12, False, rank=r, world=8, mode="eval")) for r in range(8)]
    assert sorted(sum(sum(ev, []), [])) == [0, 1, 2] and [len(ShardedBatchSampler(3, 12, False, rank=r, world=8, mode="eval")) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]
    one = ShardedBatchSampler(10, 4, False, rank=0, world=1)
    assert list(one) == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]] and one.weight(2) == 1.0


def test_flat_buffers_place_hinted_parameters_side_by_side():
    """optim.FlatBuffers honours ops.pack_hint: a layer's map_to_dir moves up right behind its map_to_feat (over the BatchNorm parameters
    registered between them), so [Wf; Wd] is one contiguous block and ops.cat_rows of the two is a VIEW of the flat buffer; offsets
    ascend along fb.params; a chain is cut where a size would need padding; parameters without hints keep their order."""
    import torch
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.optim import FlatBuffers
    from vn_pointcloudcompletion_amd.vn_layers import VNLinear, VNLinearLeakyReLU
    net = torch.nn.Sequential(VNLinearLeakyReLU(8, 4, dim=4), VNLinear(4, 3), VNLinearLeakyReLU(3, 5, dim=4))
    named = {id(p): n for n, p in net.named_parameters()}
    fb = FlatBuffers(net.parameters(), gathered=False)
    names = [named[id(p)] for p in fb.params]
    assert names == ["0.map_to_feat.weight", "0.map_to_dir.weight", "0.batchnorm.bn.weight", "0.batchnorm.bn.bias", "1.map_to_feat.weight",
                     "2.map_to_feat.weight", "2.batchnorm.bn.weight", "2.batchnorm.bn.bias", "2.map_to_dir.weight"]      # 15 floats: chain cut
    assert all(a < b for a, b in zip(fb.offsets[:-1], fb.offsets[1:]))
    l0 = net[0]
    cat = ops.cat_rows(l0.map_to_feat.weight.detach(), l0.map_to_dir.weight.detach())
    assert cat.shape == (8, 8) and cat.data_ptr() == l0.map_to_feat.weight.data_ptr()
    assert torch.equal(cat, torch.cat([l0.map_to_feat.weight, l0.map_to_dir.weight]).detach())
    l2 = net[2]
    cat2 = ops.cat_rows(l2.map_to_feat.weight.detach(), l2.map_to_dir.weight.detach())                # not adjacent: a copy
    assert cat2.data_ptr() != l2.map_to_feat.weight.data_ptr() and torch.equal(cat2, torch.cat([l2.map_to_feat.weight, l2.map_to_dir.weight]).detach())


def test_library_staleness_is_decided_by_a_source_hash(lib_built, tmp_path, monkeypatch):
    """_lib.build(): an up-to-date library (the SHA-256 of csrc/*.hip, *.h recorded beside it) is returned without a lock file, a
    directory or a compiler being touched — a read-only install works, and a copy to another machine (new file times) does not trigger a
    rebuild at rendezvous time; an edited source makes it stale; a stale library that cannot be rebuilt is used with a warning."""
    import shutil
    import warnings
    from vn_pointcloudcompletion_amd import _lib
    assert _lib._up_to_date()
    assert open(_lib._HASH_PATH).read().strip() == _lib._source_hash()
    # a scratch copy of csrc with the library and its hash: stale after an edit, fresh again after the edit is undone
    c2 = tmp_path / "csrc"
    shutil.copytree(_lib.CSRC, c2, ignore=shutil.ignore_patterns("*.o", "*.bin"))
    monkeypatch.setattr(_lib, "CSRC", str(c2))
    monkeypatch.setattr(_lib, "LIB_PATH", str(c2 / "libvnpcc.so"))
    monkeypatch.setattr(_lib, "_HASH_PATH", str(c2 / "build" / "libvnpcc.srchash"))
    os.utime(c2 / "vnp_common.h", None)                       # newer file time, same content: still up to date
    assert _lib._up_to_date()
    lock = c2 / "build" / ".lock"
    if lock.exists():
        lock.unlink()
    assert _lib.build() == str(c2 / "libvnpcc.so") and not lock.exists()          # no lock taken for an up-to-date library
    src = c2 / "vnp_common.h"
    text = src.read_text()
    src.write_text(text + "\n// edited\n")
    assert not _lib._up_to_date()
    monkeypatch.setenv("HIPCC", str(tmp_path / "no-such-compiler"))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _lib.build() == str(c2 / "libvnpcc.so")                            # no compiler: the existing library, with a warning
    assert any("cannot be rebuilt" in str(x.message) for x in w)
    src.write_text(text)
    assert _lib._up_to_date()


def test_settle_gc_moves_the_heap_to_the_permanent_generation():
    import gc
    from vn_pointcloudcompletion_amd.train import settle_gc
    before = gc.get_freeze_count()
    try:
        settle_gc()
        assert gc.get_freeze_count() > before and gc.get_freeze_count() > 1000
    finally:
        gc.unfreeze()
