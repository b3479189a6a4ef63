"""scan the gfx950 assembly of the HIP sources for compiler-inserted `s_waitcnt vmcnt(0)` directly in front of an LDS read inside kernels that use LDS-DMA
(`global_load_lds` / `buffer_load ... lds`): hipcc orders every LDS read behind ALL pending LDS-DMA writes (SIInsertWaitcnts cannot see that the ring slot
being read is not the one being filled), which turns a counted-vmcnt ring into a synchronous load.  usage: python tools/scan_vmcnt0.py [file.hip ...]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, "nerf-mae_amd", "csrc")
files = sys.argv[1:] or sorted(f for f in os.listdir(csrc) if f.endswith(".hip"))
procs = []
for f in files:
    out = "/tmp/scan_" + os.path.basename(f) + ".s"
    procs.append((f, out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-S", "--cuda-device-only",
                                            os.path.join(csrc, f), "-o", out], stderr=subprocess.DEVNULL)))
for f, out, p in procs:
    p.wait()
    s = open(out).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(s) if re.match(r"^_Z[\w]+:", l)]
    for i0, name in starts:
        i1 = next((i for i in range(i0, len(s)) if "s_endpgm" in s[i]), len(s))
        body = [l for l in s[i0:i1] if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
        ndma = sum(1 for l in body if "global_load_lds" in l or ("buffer_load" in l and " lds" in l))
        if not ndma:
            continue
        bad = sum(1 for i, l in enumerate(body) if "s_waitcnt" in l and "vmcnt(0)" in l and any("ds_read" in x for x in body[i + 1:i + 4]))
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print(f"{f:16s} dma={ndma:3d} vmcnt(0)-before-ds_read={bad:2d}  {dem[:100]}")
