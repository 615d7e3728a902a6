"""bench.py against another build of the library (A/B of compile-time experiments):  python tools/bench_with_lib.py lib_x.so -- <bench.py arguments>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402  (puts the package on sys.path)
from hipvsr import lib as L                                     # noqa: E402

cut = sys.argv.index('--')
L.LIB_PATH = os.path.join(bench.PKG, 'hipvsr', sys.argv[1])
sys.argv = ['bench.py'] + sys.argv[cut + 1:]
bench.main()
