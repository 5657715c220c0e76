"""Name -> module factories behind the reference's ``biomedkg/factory.py``."""
from __future__ import annotations

import torch

from .model import RGAT, RGCN, ComplEx, DistMult, TransE
from .utils.fusion import AttentionFusion, ReDAF


class GAE(torch.nn.Module):
    """PyG ``GAE(encoder, decoder)`` surface used by the reference:
    submodules ``encoder`` / ``decoder``, ``encode`` / ``decode`` forwarders."""

    def __init__(self, encoder, decoder=None):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder

    def encode(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)

    def decode(self, *args, **kwargs):
        return self.decoder(*args, **kwargs)

    def forward(self, *args, **kwargs):
        return self.encoder(*args, **kwargs)


class FusionFactory:
    @staticmethod
    def create_fuser(method: str, embed_dim):
        """reference factory.py:10-15: any other string (``"none"``) means no fuser."""
        if method == "attention":
            return AttentionFusion(embed_dim=embed_dim)
        if method == "redaf":
            return ReDAF(embed_dim=embed_dim)
        return None


class KGEModelFactory:
    @staticmethod
    def get_model(encoder_name, decoder_name, in_dim, hidden_dim, out_dim, num_hidden_layers, num_relation,
                  num_heads=None):
        """reference factory.py:20-50 (decoder width = ``out_dim``)."""
        encoder = KGEModelFactory._get_encoder(encoder_name, in_dim, hidden_dim, out_dim, num_hidden_layers,
                                               num_relation, num_heads)
        decoder = KGEModelFactory._get_decoder(decoder_name, num_relation, out_dim)
        return GAE(encoder=encoder, decoder=decoder)

    @staticmethod
    def _get_encoder(encoder_name, in_dim, hidden_dim, out_dim, num_hidden_layers, num_relation, num_heads=None):
        if encoder_name == "rgcn":
            return RGCN(in_dim=in_dim, hidden_dim=hidden_dim, out_dim=out_dim, num_hidden_layers=num_hidden_layers,
                        num_relations=num_relation)
        if encoder_name == "rgat":
            return RGAT(in_dim=in_dim, hidden_dim=hidden_dim, out_dim=out_dim, num_hidden_layers=num_hidden_layers,
                        num_heads=num_heads, num_relations=num_relation)
        return None  # the reference silently returns None for unknown names (factory.py:62-80)

    @staticmethod
    def _get_decoder(decoder_name, num_relation, hidden_channels):
        if decoder_name == "transe":
            return TransE(num_relations=num_relation, hidden_channels=hidden_channels)
        if decoder_name == "dismult":  # sic: the reference's spelling (factory.py:92, configs/model/kge.yaml:5)
            return DistMult(num_relations=num_relation, hidden_channels=hidden_channels)
        if decoder_name == "complex":
            return ComplEx(num_relations=num_relation, hidden_channels=hidden_channels)
        return None


def create_kge_model(cfg):
    return KGEModelFactory.get_model(
        encoder_name=cfg.encoder_name, decoder_name=cfg.decoder_name, in_dim=cfg.in_dim, hidden_dim=cfg.hidden_dim,
        out_dim=cfg.out_dim, num_hidden_layers=cfg.num_hidden_layers, num_relation=cfg.num_relation,
        num_heads=cfg.num_heads)
