"""bench.py with an alternative build of the library (VARIANT_LIB=scripts/variants/lib_<name>.so, scripts/build_variant.sh): experiments only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_amd import _lib
if os.environ.get("VARIANT_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["VARIANT_LIB"])
import bench
bench.main()
