from oracle.pyg_ops import dropout_edge, mask_feature, negative_sampling  # noqa: F401
