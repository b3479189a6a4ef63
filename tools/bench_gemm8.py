"""encoder Linear shapes at 8 grids per GPU (stage 0: 512000 tokens, stage 2: 8000 tokens / 13824 window rows); NMH_GEMM_CFG=MT,NT overrides the tile"""
import sys
sys.argv = ['x']
exec(open('tools/bench_gemm.py').read().split('t("convT dec1"')[0])
t("qkv s2", 13824, 1152, 384)
t("proj s2", 13824, 384, 384)
t("fc1 s2 (gelu dual)", 8000, 1536, 384, act=1)
t("fc2 s2", 8000, 384, 1536)
t("fc2 dgrad s2", 8000, 1536, 384)
t("qkv dgrad s2", 13824, 384, 1152)
t("qkv s0", 512000, 288, 96)
t("fc1 s0 (gelu dual)", 512000, 384, 96, act=1)
t("fc2 s0", 512000, 96, 384)
t("proj s0", 512000, 96, 96)
t("fc1 s1 (gelu dual)", 64000, 768, 192, act=1)
t("fc2 s1", 64000, 192, 768)
