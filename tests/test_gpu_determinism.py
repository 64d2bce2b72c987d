"""Run-to-run determinism of a training step: the same weights, the same batch, twice — the whole flat gradient buffer must be the same
BITS.  Configs 2 and 3 at full resolution (2048 -> 1024 -> 16384, B = 4), so that the paths of the benchmark are the ones taken: pruned
and brute-force Chamfer with their scatter backward (order-independent 64-bit integer accumulators), the screened arg-max pool, the fused
FoldingNet layer 0 (dL/dcoarse as per-column-block partials), the fused EdgeConv with its gradient gathered over the SORTED reverse
adjacency, the FPS gathers' adjoint as a gather, split-K weight gradients (slabs summed in a fixed order).  No float atomics are left on
these paths; what is order-dependent in hardware (integer atomics of histograms and cursors) never reaches a floating-point sum."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("enc,latent,k", [("vn_pointnet", 2048, 16), ("vn_dgcnn_fps", 512, 20)])
def test_training_step_gradients_are_bit_identical_run_to_run(lib_built, enc, latent, k):
    from vn_pointcloudcompletion_amd import ops
    from vn_pointcloudcompletion_amd.loss import cd_loss_L1
    from vn_pointcloudcompletion_amd.model import PCNNet
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    torch.manual_seed(0)
    cfg = SimpleNamespace(num_coarse=1024, latent_dim=latent, only_coarse=False, enc_pretrained="none", device="cuda", knn_k=k)
    net = PCNNet(cfg, enc, "vn_foldingnet").train()
    opt = FusedAdam(net.parameters(), lr=1e-4)
    fb = opt.flat_buffers[0]
    g = torch.Generator().manual_seed(3)
    p = (torch.rand(4, 2048, 3, generator=g) - 0.5).cuda()
    c = (torch.rand(4, 16384, 3, generator=g) - 0.5).cuda()
    grads, losses, traces = [], [], []
    for _ in range(3):
        opt.zero_grad()
        tr = {}
        with ops.hooks(trace=tr):
            coarse, fine = net(p)
            loss = cd_loss_L1(coarse, c) + cd_loss_L1(fine, c)
            loss.backward()
        fb.check_views()
        torch.cuda.synchronize()
        grads.append(fb.flat_grad.clone())
        losses.append(loss.detach().clone())
        traces.append({k_: v for k_, v in tr.items() if not k_.endswith("_idx")})
    assert traces[0].get("chamfer_pruned", 0) == 1 and traces[0].get("fold_layer0_fused", 0) == 1, traces[0]
    assert grads[0].abs().max().item() > 0 and torch.isfinite(grads[0]).all()
    for i in (1, 2):
        assert torch.equal(losses[i].view(torch.int32), losses[0].view(torch.int32))
        same = torch.equal(grads[i].view(torch.int32), grads[0].view(torch.int32))
        if not same:
            diff = (grads[i].view(torch.int32) != grads[0].view(torch.int32))
            names = []
            name_of = {id(q): nm for nm, q in net.named_parameters()}
            for prm, o in zip(fb.params, fb.offsets):               # fb.params: the layout order (ops.pack_hint), offsets ascending
                name = name_of[id(prm)]
                if diff[o:o + prm.numel()].any():
                    names.append((name, int(diff[o:o + prm.numel()].sum())))
            raise AssertionError(f"run {i}: {int(diff.sum())} gradient words differ from run 0: {names[:8]}")
