"""Hand-derived known answers / dense-algebra cross-checks for the restated PyG / PyGCL ops
(parity unpinned by the reference: it has no tests for them -- SURVEY.md section 4)."""
import math
import random

import torch
import torch.nn.functional as F

from oracle import pyg_ops as P
from oracle import pygcl_ops as G
from oracle import ref_modules as R


def test_self_loops_collapse_to_one_unit_loop():
    ei = torch.tensor([[0, 1, 1, 2, 2], [1, 1, 1, 0, 2]])  # two duplicate loops on 1, one on 2
    out = P.add_remaining_self_loops(ei, 4)
    assert out.tolist() == [[0, 2, 0, 1, 2, 3], [1, 0, 0, 1, 2, 3]]


def test_gcn_norm_hand_graph():
    # 0->1, 2->1, 1->0 ; node 3 isolated.  in-deg(+loop): d0=2, d1=3, d2=1, d3=1
    ei = torch.tensor([[0, 2, 1], [1, 1, 0]])
    e2, w = P.gcn_norm(ei, 4)
    exp = [1 / math.sqrt(2 * 3), 1 / math.sqrt(1 * 3), 1 / math.sqrt(3 * 2), 1 / 2, 1 / 3, 1.0, 1.0]
    torch.testing.assert_close(w, torch.tensor(exp))
    assert e2[:, 3:].tolist() == [[0, 1, 2, 3], [0, 1, 2, 3]]


def test_gcn_conv_matches_dense_algebra():
    torch.manual_seed(0)
    n, e, din, dout = 30, 120, 8, 5
    ei = torch.randint(0, n, (2, e))
    x, w, b = torch.randn(n, din), torch.randn(dout, din), torch.randn(dout)
    a = torch.zeros(n, n, dtype=torch.float64)
    for s, t in ei.t().tolist():
        if s != t:
            a[t, s] += 1.0  # duplicates count
    a += torch.eye(n, dtype=torch.float64)
    dis = a.sum(1).pow(-0.5)
    a_hat = dis[:, None] * a * dis[None, :]
    ref = a_hat @ (x.double() @ w.double().t()) + b.double()
    torch.testing.assert_close(P.gcn_conv(x, ei, w, b).double(), ref, rtol=1e-5, atol=1e-5)


def test_rgcn_mean_semantics_and_empty_relation():
    # relation 0: 0->2, 1->2 (mean of x0,x1 at node 2); relation 1: 0->2 ; relation 2 empty.
    x = torch.tensor([[1.0, 0.0], [3.0, 2.0], [5.0, 5.0]])
    ei = torch.tensor([[0, 1, 0], [2, 2, 2]])
    et = torch.tensor([0, 0, 1])
    w = torch.stack([torch.eye(2), 2 * torch.eye(2), 7 * torch.eye(2)])
    root, bias = torch.zeros(2, 2), torch.tensor([0.5, -0.5])
    out = P.rgcn_conv(x, ei, et, w, root, bias)
    exp = torch.tensor([[0.5, -0.5], [0.5, -0.5], [2.0 + 2.0 + 0.5, 1.0 + 0.0 - 0.5]])
    torch.testing.assert_close(out, exp)


def test_infonce_equals_cross_entropy_formulation():
    torch.manual_seed(1)
    n, h, tau = 17, 6, 0.2
    h1, h2 = torch.randn(n, h), torch.randn(n, h)

    def one(a, b):
        a_, b_ = F.normalize(a), F.normalize(b)
        inter, intra = a_ @ b_.t() / tau, a_ @ a_.t() / tau
        intra = intra.masked_fill(torch.eye(n, dtype=torch.bool), float("-inf"))
        return F.cross_entropy(torch.cat([inter, intra], 1), torch.arange(n))

    torch.testing.assert_close(G.dual_branch_l2l_intraview(h1, h2, tau), 0.5 * (one(h1, h2) + one(h2, h1)))


def test_jsd_closed_form():
    torch.manual_seed(2)
    n, h = 9, 4
    z, zn, g = torch.randn(n, h), torch.randn(n, h), torch.randn(1, h)
    sp, sn = (z @ g.t()).squeeze(1), (zn @ g.t()).squeeze(1)
    e_pos = (math.log(2) - F.softplus(-sp)).sum() + n * (math.log(2) - F.softplus(torch.zeros(())))
    e_pos = e_pos / n
    e_neg = (F.softplus(-sn) + sn - math.log(2)).sum() + n * (F.softplus(torch.zeros(())) - math.log(2))
    e_neg = e_neg / n
    torch.testing.assert_close(G.jsd_g2l(z, g, zn), e_neg - e_pos)


def test_transe_zero_row_uses_eps():
    z = torch.zeros(2, 3)
    rel = torch.tensor([[1.0, -2.0, 0.5]])
    s = R.transe_score(z, torch.tensor([[0], [1]]), torch.tensor([0]), rel)
    torch.testing.assert_close(s, torch.tensor([-3.5]))


def test_negative_sampling_properties():
    random.seed(3)
    torch.manual_seed(3)
    n = 50
    ei = torch.randint(0, n, (2, 400))
    neg = P.negative_sampling(ei, num_neg_samples=1200)
    assert neg.shape[0] == 2 and neg.shape[1] <= 1200
    pos = set((ei[0] * n + ei[1]).tolist())
    lin = (neg[0] * (int(ei.max()) + 1) + neg[1]).tolist()
    assert not (set(lin) & pos)
    assert len(set(lin)) == len(lin)
    assert not bool((neg[0] == neg[1]).any())          # PyG 2.5.3: the population is off-diagonal
    # saturated graph -> empty
    full = torch.cartesian_prod(torch.arange(3), torch.arange(3)).t()
    assert P.negative_sampling(full).shape == (2, 0)
    assert P.negative_sampling(full[:, full[0] != full[1]]).shape == (2, 0)   # the 6 off-diagonal cells alone saturate it


def test_negative_sampling_offdiagonal_coding_known_answers():
    """PyG 2.5.3 edge_index_to_vector / vector_to_edge_index (non-bipartite, directed), worked by hand on N = 4:
    cell (r, c), r != c, has code 3 r + c - (r < c); codes 0..11 enumerate the rows with the diagonal skipped."""
    ei = torch.tensor([[0, 0, 1, 2, 3, 3, 2], [1, 3, 0, 2, 0, 2, 2]])     # two self loops (2,2) among them
    idx, population = P.edge_index_to_vector(ei, 4)
    assert population == 12
    assert idx.tolist() == [0, 2, 3, 9, 11]                               # loops dropped: (0,1)->0 (0,3)->2 (1,0)->3 (3,0)->9 (3,2)->11
    back = P.vector_to_edge_index(torch.arange(12), 4)
    want = [(r, c) for r in range(4) for c in range(4) if r != c]
    assert list(zip(back[0].tolist(), back[1].tolist())) == want
    assert torch.equal(P.vector_to_edge_index(idx, 4), ei[:, ei[0] != ei[1]])
    # a 7 x 7 grid whose only positive is a self loop: all 42 off-diagonal cells are free, and over-asking returns them all
    random.seed(0)
    neg = P.negative_sampling(torch.tensor([[3], [3]]), num_nodes=7, num_neg_samples=48)
    assert neg.shape == (2, 42) and not bool((neg[0] == neg[1]).any())
    assert torch.unique(neg[0] * 7 + neg[1]).numel() == 42
    # sizing: prob and sample_size come from the loop-free count and N^2 - N
    calls = []

    class Spy(type(P._rand.current())):
        def sample(self, population, k):
            calls.append((population, k))
            return super().sample(population, k)

    ei = torch.tensor([[0, 1, 2, 5, 9], [0, 2, 2, 6, 3]])                 # N = 10, E = 5, loop-free 3
    with P._rand.use(Spy()):
        random.seed(1)
        neg = P.negative_sampling(ei)                                      # num_neg_samples None -> edge_index.size(1) = 5
    assert neg.shape == (2, 5)
    assert calls[0] == (90, int(1.1 * 5 / (1.0 - 3 / 90)))


def test_mask_and_dropout_restatements():
    torch.manual_seed(4)
    x = torch.ones(200, 50)
    xm, keep = P.mask_feature(x, p=0.4)
    assert torch.equal(xm, keep.float()) and abs(keep.float().mean() - 0.6) < 0.02
    y = P.dropout(x, 0.2, True)
    assert set(y.unique().tolist()) == {0.0, 1.25}
    assert torch.equal(P.dropout(x, 0.2, False), x)


def test_metric_restatements_against_scikit_learn():
    """oracle/metrics.py (torchmetrics' binary AUROC / AveragePrecision / F1 definitions) pinned by an independent
    implementation: scikit-learn on scores with ties, logits outside [0, 1] and a skewed class balance."""
    import numpy as np
    from sklearn.metrics import average_precision_score, f1_score, roc_auc_score
    from oracle import metrics as M
    rng = np.random.default_rng(0)
    for n, pos_rate, ties in ((2000, 0.5, False), (3000, 0.1, True), (500, 0.9, True)):
        t = (rng.random(n) < pos_rate).astype(np.int64)
        s = rng.normal(size=n) * 2.0 + t * 1.5                       # logits, well outside [0, 1]
        if ties:
            s = np.round(s * 2.0) / 2.0                              # many exact ties
        prob = 1.0 / (1.0 + np.exp(-s))
        assert abs(M.binary_auroc(s, t) - roc_auc_score(t, prob)) < 1e-12
        assert abs(M.binary_average_precision(s, t) - average_precision_score(t, prob)) < 1e-12
        assert abs(M.binary_f1(s, t) - f1_score(t, prob > 0.5)) < 1e-12
    assert M.binary_auroc([0.2, 0.7], [1, 1]) == 0.0                 # a class absent: torchmetrics returns 0
    assert M.binary_f1([0.1, 0.2], [0, 0]) == 0.0
    got = M.edge_wise_precision([0.7, 0.2, 0.9, -1.0], [0, 0, 2, 2], {0: "a", 1: "b", 2: "c"})
    assert got == {"a_pre": 0.5, "b_pre": 0.0, "c_pre": 0.5}
