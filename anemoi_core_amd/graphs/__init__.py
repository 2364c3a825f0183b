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
from .io import GraphData, load_graph_from_file, save_graph, validate_loaded_graph  # noqa: E402,F401
