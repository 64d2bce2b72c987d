"""One training step at awkward sizes — odd batches, point counts that are not multiples of any kernel's chunk or tile size,
the num_coarse = 448 branch at its real 14336-point resolution, every runnable enc_type / dec_type pair and every coarse
loss — hunting size assumptions (the kind that broke config 5 at 1248 coarse points).  Finite loss and parameters after two
optimizer steps."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    ("vn_pointnet", "vn_foldingnet", dict(num_coarse=448, latent=2048), 3, 3000),
    ("vn_pointnet", "vn_foldingnet", dict(num_coarse=1024, latent=2048), 1, 2048),
    ("vn_pointnet", "vn_foldingnet", dict(num_coarse=1024, latent=2048), 5, 1999),
    ("vn_pointnet", "foldingnet", dict(num_coarse=1024, latent=2048), 3, 2048),
    ("vn_pointnet", "foldingnet", dict(num_coarse=448, latent=2048), 2, 2500),
    ("vn_dgcnn_fps", "vn_foldingnet", dict(num_coarse=1024, latent=512), 3, 2500),
    ("vn_dgcnn_fps", "vn_foldingnet", dict(num_coarse=448, latent=512), 2, 2048),
    ("vn_pointnet", "attention_vn_foldingnet", dict(num_coarse=1024, latent=2048, loss="emd"), 3, 2048),
    ("vn_pointr", "vn_foldingnet", dict(num_coarse=1024, latent=1024), 3, 2048),
    ("dgcnn_fps", None, dict(num_coarse=1024, latent=1024, only_coarse=True), 3, 2500),
    ("vn_pointnet", None, dict(num_coarse=1024, latent=2048, only_coarse=True, loss="dcd"), 3, 2048),
]


@pytest.mark.parametrize("enc,dec,o,B,N", CASES, ids=[f"{c[0]}+{c[1]}-{c[2]['num_coarse']}-B{c[3]}-N{c[4]}" for c in CASES])
def test_step_at_awkward_sizes(lib_built, enc, dec, o, B, N):
    from vn_pointcloudcompletion_amd.model import PCNNet
    from vn_pointcloudcompletion_amd.optim import FusedAdam
    from vn_pointcloudcompletion_amd.train import train_step
    cfg = SimpleNamespace(num_coarse=o["num_coarse"], latent_dim=o["latent"], only_coarse=o.get("only_coarse", False),
                          enc_pretrained="none", device="cuda:0", coarse_loss=o.get("loss", "cd"),
                          dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=16)
    torch.manual_seed(0)
    net = PCNNet(cfg, enc, dec or "vn_foldingnet").train()
    md = 16384 if cfg.only_coarse else getattr(net.decoder, "num_dense", 16384)
    opt = FusedAdam(net.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1)
    p = (torch.rand(B, N, 3, generator=g) - 0.5).cuda()
    gt = (torch.rand(B, md, 3, generator=g) - 0.5).cuda()
    for _ in range(2):
        loss, _, _ = train_step(cfg, net, opt, p, gt)
    assert torch.isfinite(loss).item()
    assert torch.isfinite(opt.flat_buffers[0].flat_param).all().item()
