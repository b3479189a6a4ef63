#!/bin/bash
# build the HIP library; non-zero exit (and the first errors) when the build fails
cd "$(dirname "$0")/../nerf-mae_amd/csrc" && make -j8 > /tmp/nmh_make.log 2>&1 || { grep -i "error" -A3 /tmp/nmh_make.log | head -30; echo BUILD FAILED; exit 1; }
ls -la libnerfmae_hip.so
