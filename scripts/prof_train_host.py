"""cProfile of the host side of a training iteration (scripts/bench_train_step.py's loop)."""
import cProfile, pstats, os, sys
sys.argv = [sys.argv[0]]
os.environ["ITERS"] = "20"
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_train_step.py")).read()
pr.enable()
exec(compile(src, "bench_train_step.py", "exec"))
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
