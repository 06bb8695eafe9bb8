import cProfile, pstats, os, sys, io
sys.argv = ["bench_latency.py"]
os.environ["ITERS"] = "5"
sys.path.insert(0, "scripts")
import runpy
g = runpy.run_path("scripts/bench_latency.py")
pipe, single, torch = g["pipe"], g["single"], g["torch"]
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(200):
    pipe.register(single)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30)
print(s.getvalue()[:6000])
