"""cProfile of the host side of the single-pair loop (scripts/bench_latency.py's loop): where the ~0.2 ms between the last copy of a
call and the first kernel of the next go."""
import cProfile, pstats, os, sys
os.environ["ITERS"] = "200"
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_latency.py")).read()
src = src.replace("for _ in range(n):\n    pipe.register(single)\ntorch.cuda.synchronize()\nprint", "pr.enable()\nfor _ in range(n):\n    pipe.register(single)\npr.disable()\ntorch.cuda.synchronize()\nprint")
exec(compile(src, "bench_latency.py", "exec"))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(14)
st.sort_stats("cumtime").print_stats(30)
