cd /root/repo
for w in 0 2048 4096; do echo "--- wide_rows=$w"; NMH_LN_BWD_WIDE_ROWS=$w python tools/bench_ln.py 2>/dev/null | grep -v "64000\|512000"; done
