import os, sys, time
sys.path.insert(0, ".")
from types import SimpleNamespace
import torch
from vn_pointcloudcompletion_amd.model import PCNNet
from vn_pointcloudcompletion_amd.optim import FusedAdam
from vn_pointcloudcompletion_amd.train import settle_gc, train_step
CONFIGS = {2: dict(enc="vn_pointnet", dec="vn_foldingnet", B=32, latent=2048, loss="cd", mc=1024),
           3: dict(enc="vn_dgcnn_fps", dec="vn_foldingnet", B=32, latent=512, loss="cd", mc=1024, knn_k=20),
           4: dict(enc="vn_pointnet", dec="attention_vn_foldingnet", B=16, latent=2048, loss="emd", mc=1024),
           5: dict(enc="vn_pointr", dec="vn_foldingnet", B=32, latent=1024, loss="cd", mc=1024)}
for cid in ([int(a) for a in sys.argv[1:]] or (3, 4, 5)):          # configs whose steps use side streams (samplings, reverse adjacencies, EMD)
    c = CONFIGS[cid]
    cfg = SimpleNamespace(num_coarse=c["mc"], latent_dim=c["latent"], only_coarse=False, enc_pretrained="none", device="cuda:0",
                          coarse_loss=c["loss"], dcd_opts={"alpha": 200, "lambda": 0.5}, knn_k=c.get("knn_k", 16))
    torch.manual_seed(0)
    net = PCNNet(cfg, c["enc"], c["dec"]).train()
    md = getattr(net.decoder, "num_dense", 16384)
    opt = FusedAdam(net.parameters(), lr=1e-4)
    g = torch.Generator().manual_seed(1)
    for blk in range(4):
        p = (torch.rand(c["B"], 2048, 3, generator=g) - 0.5).cuda()
        gt = (torch.rand(c["B"], md, 3, generator=g) - 0.5).cuda()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(50):
            loss, _, _ = train_step(cfg, net, opt, p, gt)
            if blk == 0 and i == 0: settle_gc()          # as train.train()
        torch.cuda.synchronize()
        print(f"config {cid} block {blk}: {(time.perf_counter() - t0) / 50 * 1e3:.2f} ms/step, loss {loss.item():.4f}, "
              f"allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:.2f}", flush=True)
    del net, opt
    torch.cuda.empty_cache()
