cd /root/repo
python tools/bench_ln.py 2>/dev/null | tail -8
