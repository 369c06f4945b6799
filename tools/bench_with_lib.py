"""bench.py on a variant build of the library (tools/build_variant.sh): PN_LIB=<path> python tools/bench_with_lib.py <bench flags>.
Measurement tool only - the package itself always loads the in-tree libpanonerf_hip.so."""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from pano_nerf_amd import _lib
if os.environ.get("PN_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PN_LIB"])
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
