"""Drop-in for the one helper of the reference's ``utils.py`` that the MIL / GNN path
uses: ``get_args_parser`` (reference `utils.py:151-158`).  The UMAP / reconstruction
visualisers of that file are reporting code outside the path."""
import argparse
import os
import typing


def get_args_parser(path: typing.Union[str, bytes, os.PathLike]):
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_path", type=str, default=path,
                        help="path to the .yml config file specifying datasets / training params")
    return parser
