"""CPU oracle for the BioMedKG GCL + KGE training hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and there only as the checker / the timed CPU baseline.  Nothing under ``biomedkg_amd/``
imports it; the product path raises when its HIP library is missing.

What it restates (pure torch on CPU, fp32 by default, fp64 on request):

* ``oracle.pyg_ops``   -- the torch_geometric 2.5.3 ops the reference calls (GCNConv / gcn_norm,
  RGCNConv, mask_feature, dropout_edge, negative_sampling, GAE).  torch_geometric is a pinned
  third-party dependency (``/root/reference/pyproject.toml:6``) whose source is NOT under
  ``/root/reference`` and is not installable here; the restatement follows its published
  semantics.
* ``oracle.pygcl_ops`` -- PyGCL 0.1.2 (``pyproject.toml:8``) DualBranchContrast/InfoNCE and
  SingleBranchContrast/JSD.
* ``oracle.ref_modules`` -- the arithmetic written in the reference itself
  (``biomedkg/model/{encoder,decoder,gcl}.py``, ``biomedkg/utils/fusion.py``, the loss wiring of
  ``biomedkg/gcl_module.py`` / ``biomedkg/kge_module.py``), each function citing file:line.
* ``oracle.shims``     -- stand-in packages named ``torch_geometric``, ``GCL``, ``lightning``,
  ``torchmetrics``, ``omegaconf`` built on the two op modules above.  They exist so that the
  reference's UNMODIFIED module files can be imported in the authoring container for golden
  generation (``tests/golden/make_goldens.py``).  They are never on ``sys.path`` otherwise.

Parity status
-------------
* PINNED by import of the reference in the authoring container -> committed fixtures under
  ``tests/golden/``: TransE / DistMult forward+grad, AttentionFusion / ReDAF forward, the
  reference's own composition logic (GRACE / GGD / DGI forward + loss wiring, GCNEncoder / RGCN
  stacking, KGEModule.training_step loss, EdgeWisePrecision) run through the shims.
* PARITY UNPINNED for the arithmetic inside torch_geometric / PyGCL themselves (GCNConv,
  RGCNConv, negative_sampling, InfoNCE, JSD): the reference holds no tests, golden vectors or
  fixtures for them and the packages cannot be installed offline.  They are
  covered by hand-derived known-answer tests and dense-algebra cross-checks in
  ``tests/test_oracle_*.py`` instead.
"""
