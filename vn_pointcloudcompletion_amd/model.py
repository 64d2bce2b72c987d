"""Model factory with the reference's plugin surface (/root/reference/models/model.py:9-64):
``PCNNet(config, enc_type, dec_type)``, attributes ``.encoder`` / ``.decoder`` / ``.num_coarse`` / ``.only_coarse``,
``forward(input[B,N,3], rot=None) -> (coarse[B,Mc,3], fine[B,Md,3] | None)``."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import vn_layers
from .pcn import VN_FoldingNet, VN_PointNet

ENCODERS = ("dgcnn_fps", "vn_dgcnn_fps", "vn_pointnet", "vn_pointr")
DECODERS = ("vn_foldingnet", "foldingnet", "attention_vn_foldingnet")


def _make_encoder(config, enc_type):
    if enc_type == "vn_pointnet":
        return VN_PointNet(config)
    if enc_type == "vn_dgcnn_fps":
        from .dgcnn import VN_DGCNN_fps
        return VN_DGCNN_fps(config, only_coarse=config.only_coarse)
    if enc_type == "vn_pointr":
        from .vn_pointr import VN_PCTransformer
        return VN_PCTransformer(in_chans=3, embed_dim=384, depth=[6, 8], drop_rate=0., num_query=224, knn_layer=1,
                                dgcnn="vn_dgcnn", trans="vn_trans", memory_profile=False, only_coarse=True)
    if enc_type == "dgcnn_fps":
        from .baselines import DGCNN_fps
        return DGCNN_fps(config, latent_dim=1024, grid_size=4, only_coarse=config.only_coarse)
    raise Exception(f"encoder type {enc_type} not supported yet")          # models/model.py:20-21,26


def _make_decoder(config, dec_type, enc_type):
    if dec_type == "vn_foldingnet":
        return VN_FoldingNet(config)
    if dec_type == "attention_vn_foldingnet":
        from .attention_decoder import Attention_VN_FoldingNet
        return Attention_VN_FoldingNet(config)
    if dec_type == "foldingnet":
        from .baselines import FoldingNet
        return FoldingNet(config)
    raise Exception(f"encoder type {enc_type} not supported yet")          # sic, models/model.py:48


class PCNNet(nn.Module):
    def __init__(self, config, enc_type="dgcnn_fps", dec_type="foldingnet"):
        super().__init__()
        self.num_coarse = config.num_coarse
        self.only_coarse = config.only_coarse
        self.encoder = _make_encoder(config, enc_type).to(config.device)
        if config.enc_pretrained != "none":
            sd = torch.load(config.enc_pretrained, map_location=config.device)
            self.encoder.load_state_dict(sd, strict=False)
            for param in self.encoder.parameters():
                param.requires_grad = False
        if not config.only_coarse:
            self.decoder = _make_decoder(config, dec_type, enc_type).to(config.device)
            if enc_type == "vn_pointr" and dec_type == "vn_foldingnet":
                # The reference as written cannot run this pair: VN_PCTransformer returns 1024 + 224 coarse points and a
                # 1024-channel global feature, while VN_FoldingNet hard-codes 16384 dense points (models/pcn.py:332,376;
                # SURVEY.md §8a B4).  Definition used here (= the reference with these two attributes overridden after
                # construction): num_coarse = 1248, num_dense = 19968; config.latent_dim must be 1024.
                self.decoder.num_coarse, self.decoder.num_dense = 1248, 1248 * 16

    def forward(self, input, rot=None, on_coarse=None):
        """``on_coarse(coarse_pred)``: called as soon as the encoder has produced the coarse cloud, before the decoder runs — the training
        step starts the coarse loss there on a side stream (train.train_step), beside the decoder's launches.  Not part of the
        reference's signature; without it the forward is the reference's."""
        prev = vn_layers.begin_deferred_ticks()          # all BatchNorm step counters of this forward in one launch at its end
        try:
            return self._forward(input, rot, on_coarse)
        finally:
            vn_layers.end_deferred_ticks(prev)

    def _forward(self, input, rot=None, on_coarse=None):
        coarse, feature_global = self.encoder(input)
        if self.num_coarse == 448:
            if on_coarse is not None:
                on_coarse(coarse[1])
            if self.only_coarse:
                return coarse[1], None
            fine = self.decoder(coarse[0], feature_global, rot)
            return coarse[1], fine
        if on_coarse is not None:
            on_coarse(coarse)
        if self.only_coarse:
            return coarse, None
        fine = self.decoder(coarse, feature_global, rot)
        return coarse, fine
