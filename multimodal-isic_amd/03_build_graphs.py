"""Graph-blueprint build on the MI355X: drop-in for the reference's
``03_build_graphs.py`` driver (`03:156-171`) with the roots as arguments instead of
hard-coded cluster paths.  The builders themselves (``_grid_edge_index``,
``_knn_edge_index``, ``_random_edge_index``, ``process_model_directory``) live in
``build_graphs.py`` and are re-exported here under the reference's names."""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from build_graphs import (DEFAULT_K_VALUES, DEFAULT_R_VALUES, GRID_SIDE, NUM_NODES, _build_image_graph_blueprints,  # noqa: F401
                          _grid_edge_index, _knn_edge_index, _random_edge_index, process_model_directory)


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--patch-stats-root", type=Path, default=Path("patch_stats"))
    ap.add_argument("--graph-outputs-root", type=Path, default=Path("graph_outputs"))
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    for model_dir in sorted(p for p in a.patch_stats_root.iterdir() if p.is_dir()):
        process_model_directory(model_dir=model_dir, output_root=a.graph_outputs_root, k_values=DEFAULT_K_VALUES,
                                r_values=DEFAULT_R_VALUES, seed=a.seed)


if __name__ == "__main__":
    main()
