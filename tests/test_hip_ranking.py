"""Device ranking evaluation (filtered MRR / Hits@k, csrc/rank.hip) vs. the dense fp64 CPU checker; metrics must
agree within 1e-4 (the tolerance BASELINE.json's north_star states for score-derived metrics).  Needs an MI355X."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("n,d,t", [(700, 32, 900), (1500, 128, 2000), (333, 256, 64)])
def test_distmult_ranking_matches_dense_checker(n, d, t):
    from biomedkg_amd import eval as E
    from oracle import ranking as O
    torch.manual_seed(n)
    z, rel = torch.randn(n, d) * 0.4, torch.randn(5, d)
    all_ei, all_et = torch.randint(0, n, (2, 4 * t)), torch.randint(0, 5, (4 * t,))
    ei, et = all_ei[:, :t], all_et[:t]                       # the test split; the rest are other known triples
    for known in (None, (all_ei, all_et)):
        ref = O.distmult_ranking(z, rel, ei, et, known=known)
        kn = None if known is None else (known[0].to(DEV), known[1].to(DEV))
        got = E.distmult_ranking(z.to(DEV), rel.to(DEV), ei.to(DEV), et.to(DEV), known=kn)
        assert got["num_queries"] == ref["num_queries"] == 2 * t
        for k in ("MRR", "Hits@1", "Hits@3", "Hits@10"):
            assert abs(got[k] - ref[k]) <= 1e-4, (k, got[k], ref[k])
        assert abs(got["mean_rank"] - ref["mean_rank"]) <= 1e-3 * ref["mean_rank"]


def test_ranking_ties_and_trained_embeddings():
    """Exact ties (duplicated node rows) take the realistic rank; a planted structure ranks first."""
    from biomedkg_amd import eval as E
    torch.manual_seed(0)
    n, d = 200, 64
    z = torch.randn(n, d)
    z[10] = z[11] = z[12]                                   # three identical candidates
    rel = torch.ones(1, d)
    ei = torch.tensor([[3], [11]])
    out = E.distmult_ranking(z.to(DEV), rel.to(DEV), ei.to(DEV), torch.zeros(1, dtype=torch.long, device=DEV), side="tail")
    s = (z[3] * z).sum(1)
    expect = 1.0 + float((s > s[11]).sum()) + (float((s == s[11]).sum()) - 1) / 2.0
    assert abs(out["mean_rank"] - expect) < 1e-9 and float((s == s[11]).sum()) >= 3


@pytest.mark.parametrize("n,d,t", [(400, 32, 300), (900, 256, 500), (250, 768, 40)])
def test_transe_ranking_matches_dense_checker(n, d, t):
    from biomedkg_amd import eval as E
    from oracle import ranking as O
    torch.manual_seed(n + 1)
    z, rel = torch.randn(n, d), torch.randn(4, d) * (0.5 / d)
    z[3] = 0.0                                                    # the eps path of the L1 normalisation
    all_ei, all_et = torch.randint(0, n, (2, 3 * t)), torch.randint(0, 4, (3 * t,))
    ei, et = all_ei[:, :t], all_et[:t]
    for known in (None, (all_ei, all_et)):
        ref = O.transe_ranking(z, rel, ei, et, known=known)
        kn = None if known is None else (known[0].to(DEV), known[1].to(DEV))
        got = E.transe_ranking(z.to(DEV), rel.to(DEV), ei.to(DEV), et.to(DEV), known=kn)
        assert got["num_queries"] == ref["num_queries"] == 2 * t
        for k in ("MRR", "Hits@1", "Hits@3", "Hits@10"):
            assert abs(got[k] - ref[k]) <= 1e-4, (k, got[k], ref[k])
