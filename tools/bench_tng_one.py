"""one grouped-TN case for counter collection: python tools/bench_tng_one.py <M> <N> <K> <nprob> [reps]"""
import sys
sys.argv, a = ['x'], sys.argv[1:]
src = open('tools/bench_tng.py').read().split("for alias in")[0]
exec(src)
run("case", int(a[0]), int(a[1]), int(a[2]), nprob=int(a[3]), reps=int(a[4]) if len(a) > 4 else 3)
