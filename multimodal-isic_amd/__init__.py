"""MI355X-native drop-in for the attention-MIL + patch-graph GNN training path of
rbuler/multimodal-isic.

Like the reference, the modules in this directory are FLAT (``utils_g_mil``,
``model``, ``net_utils``, ``dataset``, ``build_graphs``, ``utils``): put this
directory on ``sys.path`` (or run the 01/03/05 scripts from it) and the
reference's imports -- ``from utils_g_mil import AttentionMIL_teacher`` --
resolve to the HIP-backed classes.  Importing the directory as a package
(``importlib.import_module("multimodal-isic_amd")``) does the same thing.
"""
import os as _os
import sys as _sys

_HERE = _os.path.dirname(_os.path.abspath(__file__))
if _HERE not in _sys.path:
    _sys.path.insert(0, _HERE)
