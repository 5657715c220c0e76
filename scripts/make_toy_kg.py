#!/usr/bin/env python3
"""Write small triplet tables in the two on-disk formats the reference's data modules read (no network here: the real
PrimeKG ``kg.csv`` is a 900 MB download, reference biomedkg/data/dataset/primekg.py:28-33):

* ``kg.csv`` -- PrimeKG's columns (``relation, display_relation, x_index, x_id, x_type, x_name, x_source, y_index, y_id,
  y_type, y_name, y_source``), every edge listed in BOTH directions as PrimeKG does, a few node types beyond the default
  three (so that the ``node_type`` filter has something to drop), a duplicate row, names shared between types;
* ``dpi_benchmark.csv`` -- the five columns ``TripletBase`` needs, one drug -> protein relation, with a few rows holding
  a missing value (``DPI`` drops them, dataset/dpi.py:13-17).

Synthetic names, seeded: the same bytes every time.

    python scripts/make_toy_kg.py OUT_DIR [--nodes 40] [--edges 420] [--seed 5]
"""
import argparse
import csv
import os
import random

TYPES = ["gene/protein", "drug", "disease", "anatomy", "pathway"]
# (relation, head type, tail type): PrimeKG lists each edge in both directions under ONE relation name
RELATIONS = [("protein_protein", "gene/protein", "gene/protein"), ("drug_protein", "drug", "gene/protein"),
             ("indication", "drug", "disease"), ("contraindication", "drug", "disease"),
             ("drug_drug", "drug", "drug"), ("disease_protein", "disease", "gene/protein"),
             ("disease_disease", "disease", "disease"), ("anatomy_protein_present", "anatomy", "gene/protein"),
             ("pathway_protein", "pathway", "gene/protein")]
KG_COLUMNS = ["relation", "display_relation", "x_index", "x_id", "x_type", "x_name", "x_source", "y_index", "y_id", "y_type",
              "y_name", "y_source"]


def node_names(kind, count):
    stem = {"gene/protein": "GENE", "drug": "drug-", "disease": "disease ", "anatomy": "organ_", "pathway": "PW:"}[kind]
    names = [f"{stem}{i:03d}" for i in range(count)]
    if kind in ("drug", "disease"):
        names[0] = "shared name"          # the same name under two types: two different nodes
    return names


def write_kg_csv(path, nodes=40, edges=420, seed=5):
    rng = random.Random(seed)
    names = {t: node_names(t, nodes if t in TYPES[:3] else max(nodes // 4, 3)) for t in TYPES}
    index = {}
    for t in TYPES:
        for n in names[t]:
            index[(t, n)] = len(index)
    rows = []
    for _ in range(edges):
        rel, ht, tt = RELATIONS[rng.randrange(len(RELATIONS))]
        h, t = rng.choice(names[ht]), rng.choice(names[tt])
        rows.append((rel, ht, h, tt, t))
    rows.append(rows[3])                   # a duplicate row: kept as a parallel edge by the reference
    out = []
    for rel, ht, h, tt, t in rows:         # both directions, forward block first (as in PrimeKG)
        out.append((rel, ht, h, tt, t))
    for rel, ht, h, tt, t in rows:
        out.append((rel, tt, t, ht, h))
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(KG_COLUMNS)
        for rel, xt, xn, yt, yn in out:
            w.writerow([rel, rel.replace("_", " "), index[(xt, xn)], f"ID{index[(xt, xn)]}", xt, xn, "toy",
                        index[(yt, yn)], f"ID{index[(yt, yn)]}", yt, yn, "toy"])
    return len(out)


def write_dpi_csv(path, drugs=30, proteins=25, edges=160, seed=6):
    rng = random.Random(seed)
    dn, pn = node_names("drug", drugs), node_names("gene/protein", proteins)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["x_type", "x_name", "relation", "y_type", "y_name"])
        for i in range(edges):
            d, p = rng.choice(dn), rng.choice(pn)
            if i % 37 == 5:
                p = ""                      # a missing value: pandas reads NaN, the reference drops the row
            w.writerow(["drug", d, "drug_protein", "gene/protein", p])
    return edges


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out_dir")
    ap.add_argument("--nodes", type=int, default=40)
    ap.add_argument("--edges", type=int, default=420)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    n = write_kg_csv(os.path.join(a.out_dir, "primekg", "kg.csv"), a.nodes, a.edges, a.seed)
    m = write_dpi_csv(os.path.join(a.out_dir, "dpi", "dpi_benchmark.csv"))
    print(f"wrote {n} + {m} rows under {a.out_dir}")
