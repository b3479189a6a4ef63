cd /root/repo
echo "--- default"; python tools/bench_tail.py 8 2>/dev/null | tail -1
cp nerf-mae_amd/csrc/libnerfmae_hip.so /tmp/cur.so; cp tools/probe/lib_b3.so nerf-mae_amd/csrc/libnerfmae_hip.so
echo "--- launch_bounds(256,3)"; python tools/bench_tail.py 8 2>/dev/null | tail -1
cp /tmp/cur.so nerf-mae_amd/csrc/libnerfmae_hip.so
