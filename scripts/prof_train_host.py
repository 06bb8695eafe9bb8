"""cProfile of the host side of a training iteration (scripts/bench_train_step.py's timed loop only; the autograd engine's own thread -
the backward Functions - is not seen by cProfile)."""
import cProfile, pstats, os, sys
os.environ["ITERS"] = os.environ.get("ITERS", "40")
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train_step.py")).read()
src = src.replace("for it in range(ITERS):\n    a = time.perf_counter()", "pr.enable()\nfor it in range(ITERS):\n    a = time.perf_counter()")
src = src.replace("torch.cuda.synchronize()\nwall =", "pr.disable()\ntorch.cuda.synchronize()\nwall =")
exec(compile(src, "bench_train_step.py", "exec"))
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(32)
