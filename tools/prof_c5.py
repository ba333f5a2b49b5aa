"""host-side profile of the C5 step (GP facade): python tools/prof_c5.py"""
import sys, cProfile, pstats, time
sys.path.insert(0, '.')
import numpy as np
import bench
import lsqfitgp_amd as lgp
ctx = lgp.Context(0); lgp.Context._default = ctx
class A: steps = 3; warmup = 2
pr = cProfile.Profile()
pr.enable()
bench.main_c5(A, ctx)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
