"""bench.py against ANOTHER build of libqexhip.so (A/B on one box): python scratch/bench_with_lib.py LIBPATH [bench.py arguments]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import qex_amd._lib as L
path = os.path.abspath(sys.argv[1])
L.LIB_PATH = path
probe = C.CDLL(path, mode=C.RTLD_LOCAL | getattr(os, "RTLD_DEEPBIND", 0))
L.SYMBOLS = [s for s in L.SYMBOLS if hasattr(probe, s[0])]
if not hasattr(probe, "qexhip_stag_sweep_tuning"):          # a build older than round 6: the sweep-tuning report does not exist there
    import qex_amd.staggered as S
    S.Context.sweep_tuning = lambda self: {"exchange_us": 0.0, "boundary_at": 0.0, "tuned_us_per_sweep": [0.0, 0.0, 0.0], "form": "?", "fused_spin_us": 0.0}
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
