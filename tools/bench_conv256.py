import sys, torch
sys.path.insert(0, '.')
exec(open('tools/bench_conv_generic.py').read().split('for shp in')[0])
for shp in [(4, 40, 256, 256), (4, 20, 256, 256), (4, 40, 192, 96)]: t(*shp)
