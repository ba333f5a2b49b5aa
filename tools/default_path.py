import sys; sys.path.insert(0,'.')
import bench, lsqfitgp_amd as lgp
ctx=lgp.Context.default()
print(bench.default_path(ctx))
