"""Graph construction from a triplet table vs. the golden recorded from the reference's own TripletBase
(tests/golden/make_goldens.py gen_triplets); CPU only."""
import torch


def test_build_graph_matches_reference(golden):
    import pandas as pd
    from biomedkg_amd.triplets import build_graph
    g = golden("triplets")
    df = pd.DataFrame(g["rows"], columns=["x_type", "x_name", "relation", "y_type", "y_name"])
    enc = lambda names: torch.tensor([[float(sum(map(ord, n))), float(len(n))] for n in names])
    out = build_graph(df, enc)
    assert out["node_list"] == g["node_list"] and out["edge_map_index"] == g["edge_map_index"]
    for k in ("x", "edge_index", "edge_type", "node_type"):
        assert torch.equal(out[k], g[k]), k
    # the conventions, spelled out: types alphabetical (disease, drug, gene/protein), names sorted inside a type
    assert g["node_list"] == ["fever", "pain", "aspirin", "ibuprofen", "BRCA1", "MDM2", "PTGS2", "TP53"]
    assert g["edge_map_index"] == {0: "protein_protein", 1: "drug_protein", 2: "indication", 3: "drug_drug",
                                   4: "disease_protein"}
    assert int((g["edge_type"] == 1).sum()) == 3 and int((g["edge_type"] == 2).sum()) == 2   # one direction only


def test_to_undirected():
    from biomedkg_amd.triplets import to_undirected
    ei = torch.tensor([[0, 1, 2, 1], [1, 0, 3, 0]])
    et = torch.tensor([5, 5, 7, 5])
    # PyG's ToUndirected(reduce="add"), what the reference runs: merged duplicates SUM their attributes, edge_type included
    # ((0, 1) is listed once forward and twice as a reversed (1, 0): 5 + 5 + 5)
    u, t = to_undirected(ei, et)
    assert u.tolist() == [[0, 1, 2, 3], [1, 0, 3, 2]] and t.tolist() == [15, 15, 7, 7]
    u, t = to_undirected(ei, et, reduce="first")
    assert u.tolist() == [[0, 1, 2, 3], [1, 0, 3, 2]] and t.tolist() == [5, 5, 7, 7]
