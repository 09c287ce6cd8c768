"""CPU oracle for the attention-MIL + patch-graph GNN training path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / the timed CPU
baseline.  The product (``multimodal-isic_amd/``) never imports this package
and fails loudly when ``libisic_hip.so`` is missing.

What it is: a plain torch-CPU fp32/fp64 restatement of the reference's
arithmetic (the reference itself is torch-CPU Python, so "the reference's CPU
path" is exactly this kind of code).  Every function cites the reference
file:line it follows.

Pinning (see DESIGN.md "Oracle"):
  * pinned against the reference itself: ``oracle/gen_golden.py`` imports the
    reference's pure-torch modules in the build container (stubbing the absent
    ``ray`` / ``torch_geometric`` / ``neptune`` / ``umap`` /
    ``efficientnet_pytorch`` packages) and writes ``tests/golden/*.npz``;
    ``tests/test_oracle_golden.py`` replays every fixture through this package.
  * NOT pinned ("parity unpinned"): the PyG layers (``GCNConv``, ``GCN2Conv``)
    -- ``torch_geometric`` is absent and unpinned in the reference -- and the
    ResNet-18 patch encoder, which has no counterpart in the reference at all
    (its encoder is an un-vendored ConvMAE).  Those are restated from the
    published layer definitions and are this build's own definition.
"""
