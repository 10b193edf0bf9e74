"""Drop-in module: put this directory on sys.path and mused's `main.py:5` import line
(`from matrix_operations import create_adjacency_matrix, fuse_matrices, ...`) resolves to the MI355X path."""
from mused_amd.matrix_operations import (  # noqa: F401
    create_adjacency_matrix,
    fuse_matrices,
    match_clusters,
    perform_clustering,
    perform_dbscan_clustering,
    perform_dbscan_incr_clustering,
    perform_hdbscan_clustering,
    perform_svd_reduction,
)
