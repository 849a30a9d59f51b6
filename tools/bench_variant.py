"""python tools/bench_variant.py VARIANT [bench.py args]: bench.py with a forced sparse-kernel variant (v3d_debug_set_rows_mt)."""
import ctypes, os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
variant = int(sys.argv[1])
from vision3d_amd import _lib as L
ctypes.CDLL(L.LIB_PATH).v3d_debug_set_rows_mt(variant)
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
