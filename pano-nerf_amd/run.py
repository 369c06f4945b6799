"""`python -m pano_nerf_amd.run <script.py> [args...]`: run a script of the reference (train.py, eval.py) with the MI355X
render classes installed under the reference's import paths — no edit to the reference (see install.py)."""
import os
import runpy
import sys


def main():
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m pano_nerf_amd.run <script.py> [args...]")
    script = os.path.abspath(sys.argv[1])
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(script))  # what `python script.py` would do
    from .install import install
    install()
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
