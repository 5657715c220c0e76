"""GCL task modules behind the reference's ``biomedkg/gcl_module.py`` API (same class names,
constructor kwargs, hooks and attribute tree).  The arithmetic is the HIP path:
fused modality projection -> GCN (GEMM + CSR segment-reduce) -> DGI / GRACE / GGD loss.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import ops
from .compat import LightningModule
from .factory import FusionFactory
from .model import GCNEncoder
from .model.gcl import DGI, GGD, GRACE


def _get_scheduler(module, optimizer):
    """reference gcl_module.py:89-100 / kge_module.py:219-230 (transformers warm-up schedules)."""
    from transformers.optimization import get_cosine_schedule_with_warmup, get_linear_schedule_with_warmup
    steps = int(module.trainer.estimated_stepping_batches)
    args = {"optimizer": optimizer, "num_training_steps": steps, "num_warmup_steps": int(steps * module.warm_up_ratio)}
    if module.scheduler_type == "linear":
        return get_linear_schedule_with_warmup(**args)
    if module.scheduler_type == "cosine":
        return get_cosine_schedule_with_warmup(**args)


class InfoNCEContrast(torch.nn.Module):
    """``DualBranchContrast(InfoNCE(tau), mode="L2L", intraview_negs=True)`` (gcl_module.py:171-173)."""

    def __init__(self, tau: float = 0.2):
        super().__init__()
        self.tau = tau

    def forward(self, h1, h2):
        return ops.ntxent_loss(h1, h2, self.tau)


class JSDContrast(torch.nn.Module):
    """``SingleBranchContrast(JSD(), mode="G2L")`` (gcl_module.py:127).  The
    summary-vs-node scores are one [1,h] x [h,2N] product; the JSD itself is O(N) glue."""

    def forward(self, h, g, hn):
        sim_pos = ops.linear(h, g).squeeze(1)    # [N] = h @ g^T
        sim_neg = ops.linear(hn, g).squeeze(1)
        n = h.size(0)
        log2 = math.log(2.0)
        zeros = sim_pos.new_zeros(())
        # PyGCL multiplies sim by the 0/1 masks, so masked-out entries contribute softplus(0) terms
        e_pos = ((log2 - F.softplus(-sim_pos)).sum() + n * (log2 - F.softplus(zeros))) / n
        e_neg = ((F.softplus(-sim_neg) + sim_neg - log2).sum() + n * (F.softplus(zeros) - log2)) / n
        return e_neg - e_pos


class BaseGCL(LightningModule):
    def __init__(self, model, embed_dim: int, scheduler_type: str, learning_rate: float, warm_up_ratio: float,
                 feature_embedding_dim: int, contrast_model=None, fuse_method: str = None):
        super().__init__()
        self.save_hyperparameters(ignore=["model"])
        self.model = model
        self.modality_transform = FusionFactory.create_fuser(method=fuse_method, embed_dim=embed_dim)
        self.contrast_model = contrast_model
        self.lr = learning_rate
        self.scheduler_type = scheduler_type
        self.warm_up_ratio = warm_up_ratio
        self.feature_embedding_dim = feature_embedding_dim

    def fusion_fn(self, x):
        """reference gcl_module.py:43-50."""
        if self.modality_transform:
            x = self.modality_transform(x)
        elif x.dim() == 3:
            x = torch.mean(x, dim=1)
        return x

    def calculate_loss(self, x, edge_index):
        raise NotImplementedError

    def forward(self, x, edge_index):
        x = self.fusion_fn(x=x)
        return self.model.encoder(x, edge_index)

    def training_step(self, batch, batch_idx=None):
        x = self.fusion_fn(x=batch.x)
        loss = self.calculate_loss(x, batch.edge_index)
        self.log("train_loss", loss, on_epoch=True, on_step=True, prog_bar=True)
        return loss

    def validation_step(self, batch, batch_idx):
        x = self.fusion_fn(x=batch.x)
        loss = self.calculate_loss(x, batch.edge_index)
        self.log("val_loss", loss, on_epoch=True, on_step=True, prog_bar=True)
        return loss

    def test_step(self, batch, batch_idx):
        x = self.fusion_fn(x=batch.x)
        loss = self.calculate_loss(x, batch.edge_index)
        self.log("test_loss", loss, on_epoch=True, on_step=True, prog_bar=True)
        return loss

    def configure_optimizers(self):
        params = list(self.model.parameters())
        # torch's fused (single-kernel) Adam on the GPU: the same update as the reference's torch.optim.Adam
        optimizer = torch.optim.Adam(params, lr=self.lr, fused=all(p.is_cuda for p in params) and len(params) > 0)
        return {"optimizer": optimizer, "lr_scheduler": self._get_scheduler(optimizer=optimizer)}

    def _get_scheduler(self, optimizer):
        return _get_scheduler(self, optimizer)


def _encoder(in_dim, hidden_dim, out_dim, num_hidden_layers):
    return GCNEncoder(in_dim=in_dim, hidden_dim=hidden_dim, out_dim=out_dim, num_hidden_layers=num_hidden_layers)


class DGIModule(BaseGCL):
    def __init__(self, in_dim: int, hidden_dim: int, out_dim: int, num_hidden_layers: int,
                 scheduler_type: str = "cosine", learning_rate: float = 2e-4, warm_up_ratio: float = 0.03,
                 fuse_method: str = None):
        self.save_hyperparameters(ignore=["model"])
        model = DGI(encoder=_encoder(in_dim, hidden_dim, out_dim, num_hidden_layers), hidden_dim=hidden_dim)
        super().__init__(model=model, embed_dim=in_dim, scheduler_type=scheduler_type, learning_rate=learning_rate,
                         warm_up_ratio=warm_up_ratio, feature_embedding_dim=in_dim, contrast_model=JSDContrast(),
                         fuse_method=fuse_method)

    def calculate_loss(self, x, edge_index):
        pos_z, summary, neg_z = self.model(x, edge_index)
        return self.contrast_model(h=pos_z, g=summary, hn=neg_z)


class GRACEModule(BaseGCL):
    def __init__(self, in_dim: int, hidden_dim: int, out_dim: int, num_hidden_layers: int,
                 scheduler_type: str = "cosine", learning_rate: float = 2e-4, warm_up_ratio: float = 0.03,
                 fuse_method: str = None):
        self.save_hyperparameters(ignore=["model"])
        model = GRACE(encoder=_encoder(in_dim, hidden_dim, out_dim, num_hidden_layers), hidden_dim=hidden_dim,
                      proj_dim=hidden_dim)
        super().__init__(model=model, embed_dim=in_dim, scheduler_type=scheduler_type, learning_rate=learning_rate,
                         warm_up_ratio=warm_up_ratio, feature_embedding_dim=in_dim,
                         contrast_model=InfoNCEContrast(tau=0.2), fuse_method=fuse_method)

    def calculate_loss(self, x, edge_index):
        # the reference also runs an encoder pass on the un-augmented graph and discards it
        # (gcl_module.py:187); only the two augmented views reach the loss
        z1, z2 = self.model.views(x, edge_index)
        h1, h2 = [self.model.project(z) for z in (z1, z2)]
        return self.contrast_model(h1, h2)


class GGDModule(BaseGCL):
    def __init__(self, in_dim: int, hidden_dim: int, out_dim: int, num_hidden_layers: int,
                 scheduler_type: str = "cosine", learning_rate: float = 2e-4, warm_up_ratio: float = 0.03,
                 fuse_method: str = None):
        self.save_hyperparameters(ignore=["model"])
        model = GGD(encoder=_encoder(in_dim, hidden_dim, out_dim, num_hidden_layers), hidden_dim=hidden_dim,
                    n_proj=1, aug_p=0.5)
        super().__init__(model=model, embed_dim=in_dim, scheduler_type=scheduler_type, learning_rate=learning_rate,
                         warm_up_ratio=warm_up_ratio, feature_embedding_dim=in_dim, fuse_method=fuse_method)

    def calculate_loss(self, x, edge_index):
        pos_h, neg_h = self.model(x, edge_index)
        if ops.ggd_bce_ok(pos_h, neg_h):          # the same mean of softplus(-pos) and softplus(neg), one launch each way
            return ops.ggd_bce(pos_h, neg_h)
        pred = torch.cat([pos_h, neg_h])
        gt = torch.cat([torch.ones_like(pos_h), torch.zeros_like(neg_h)])
        return F.binary_cross_entropy_with_logits(pred, gt)
