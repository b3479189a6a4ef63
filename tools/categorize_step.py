"""groups the per-(kernel, grid) table of tools/trace_shapes.py into the categories DESIGN.md section 6 accounts by"""
import re, sys
cats = [("conv48/conv64 fwd/dgrad", r"conv48_kernel|conv64_kernel"), ("conv48/conv64 wgrad (all levels)", r"conv48_wgrad|conv64_wgrad"),
        ("decoder-1 elementwise (tail/IN @160^3)", None), ("generic conv fwd/dgrad (AConv3)", r"AConv3"), ("generic conv wgrad", r"BConv3TN"),
        ("upconv (AUp / shuffle GEMMs)", r"AUp"), ("attention", r"attn_"), ("layernorm", r"ln_"), ("encoder wgrad (grouped)", r"gemm_tn_grouped"),
        ("gemm_tn (other)", r"gemm_tn_kernel|tn_reduce"), ("gemm_nt (Linear/1x1/upconv fwd)", r"gemm_nt"), ("instnorm small levels", r"in_"),
        ("pack/optimizer", r"pack_kernel|adamw|sqnorm|clip_coef"), ("misc", r".")]
cats[0:0] = []
tot = {c: 0.0 for c, _ in cats}
n = {c: 0 for c, _ in cats}
lines = open(sys.argv[1]).read().splitlines()
print(lines[0])
for l in lines[1:]:
    m = re.match(r"\s*([\d.]+) us\s+(\d+)x\s+([\d.]+) us/call\s+grid=\((\d+),(\d+),(\d+)\)\s+(.*)", l)
    if not m: continue
    t, cnt, name = float(m.group(1)), int(m.group(2)), m.group(7)
    per = float(m.group(3))
    if re.search(r"tail_|in_apply|in_bwd_apply|in_reduce", name) and per > 300:
        c = "decoder-1 elementwise (tail/IN @160^3)"
    else:
        c = next(c for c, r in cats if r and re.search(r, name))
    tot[c] += t; n[c] += cnt
for c, _ in cats:
    print(f"{tot[c] / 1e3:8.2f} ms {n[c]:5d} launches  {c}")
