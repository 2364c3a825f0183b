from .synthetic import (  # noqa: F401
    SyntheticGraph,
    build_synthetic_graph,
    cutoff_edges,
    fibonacci_grid,
    icosphere_latlon,
    knn_edges,
    multiscale_edges,
    octahedral_grid,
)
