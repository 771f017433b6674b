"""cProfile of the host side of bench.py's step at a small (launch-bound) config.  Measuring tool only."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.argv = ["bench.py", "--config", sys.argv[1] if len(sys.argv) > 1 else "cfg2", "--steps", "60", "--warmup", "5",
            "--no-cpu-baseline"]
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
