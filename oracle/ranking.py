"""Dense CPU checker for the ranking evaluation (test infrastructure): fp64 score matrix, python sets for the
filtered setting, realistic tie rule.  There is no reference counterpart (parity unpinned); the
definition checked here is the standard filtered MRR / Hits@k protocol stated in biomedkg_amd/eval.py."""
import torch


def distmult_ranking(z, rel_emb, edge_index, edge_type, known=None, ks=(1, 3, 10), side="both"):
    z, rel_emb = z.double(), rel_emb.double()
    ranks = []
    for s in (("tail", "head") if side == "both" else (side,)):
        anchor, target = (edge_index[0], edge_index[1]) if s == "tail" else (edge_index[1], edge_index[0])
        scores = (z[anchor] * rel_emb[edge_type]) @ z.t()
        truth = {}
        if known is not None:
            k_ei, k_et = known
            ka, kt = (k_ei[0], k_ei[1]) if s == "tail" else (k_ei[1], k_ei[0])
            for a, r, t in zip(ka.tolist(), k_et.tolist(), kt.tolist()):
                truth.setdefault((a, r), set()).add(t)
        for i, (a, r, t) in enumerate(zip(anchor.tolist(), edge_type.tolist(), target.tolist())):
            row = scores[i].clone()
            true = row[t].item()
            drop = (truth.get((a, r), set()) | {t})
            mask = torch.ones_like(row, dtype=torch.bool)
            mask[list(drop)] = False
            others = row[mask]
            ranks.append(1.0 + float((others > true).sum()) + float((others == true).sum()) / 2.0)
    rank = torch.tensor(ranks, dtype=torch.float64)
    out = {"MRR": float((1.0 / rank).mean()), "mean_rank": float(rank.mean()), "num_queries": int(rank.numel())}
    for k in ks:
        out[f"Hits@{k}"] = float((rank <= k).double().mean())
    return out


def transe_ranking(z, rel_emb, edge_index, edge_type, known=None, ks=(1, 3, 10), side="both"):
    """TransE (reference decoder.py:31-38): score(h, r, t) = -||h^ + r - t^||_1 with L1-normalised rows."""
    z, rel_emb = z.double(), rel_emb.double()
    zn = z / z.abs().sum(1, keepdim=True).clamp_min(1e-12)
    ranks = []
    for s in (("tail", "head") if side == "both" else (side,)):
        anchor, target = (edge_index[0], edge_index[1]) if s == "tail" else (edge_index[1], edge_index[0])
        q = zn[anchor] + rel_emb[edge_type] if s == "tail" else zn[anchor] - rel_emb[edge_type]
        scores = -(q[:, None, :] - zn[None, :, :]).abs().sum(-1)
        truth = {}
        if known is not None:
            k_ei, k_et = known
            ka, kt = (k_ei[0], k_ei[1]) if s == "tail" else (k_ei[1], k_ei[0])
            for a, r, t in zip(ka.tolist(), k_et.tolist(), kt.tolist()):
                truth.setdefault((a, r), set()).add(t)
        for i, (a, r, t) in enumerate(zip(anchor.tolist(), edge_type.tolist(), target.tolist())):
            row = scores[i]
            true = row[t].item()
            mask = torch.ones_like(row, dtype=torch.bool)
            mask[list(truth.get((a, r), set()) | {t})] = False
            others = row[mask]
            ranks.append(1.0 + float((others > true).sum()) + float((others == true).sum()) / 2.0)
    rank = torch.tensor(ranks, dtype=torch.float64)
    out = {"MRR": float((1.0 / rank).mean()), "mean_rank": float(rank.mean()), "num_queries": int(rank.numel())}
    for k in ks:
        out[f"Hits@{k}"] = float((rank <= k).double().mean())
    return out
