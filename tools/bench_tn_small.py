import sys
sys.argv=['x']
exec(open('tools/bench_tn.py').read().split('t("qkv wgrad s0"')[0])
t("qkv wgrad s2", 6912, 1152, 384)
t("fc1 wgrad s2", 4000, 1536, 384)
t("fc2 wgrad s2", 4000, 384, 1536)
t("proj wgrad s2", 6912, 384, 384)
t("fc1 wgrad s1", 32000, 768, 192)
t("fc1 wgrad s3", 500, 3072, 768)
