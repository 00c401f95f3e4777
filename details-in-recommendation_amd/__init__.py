"""details-in-recommendation_amd -- MI355X-native embedding-lookup + feature-interaction path.

Import it as `dir_amd` (see ../dir_amd.py: the directory name is not a valid Python identifier).
Contents: csrc/ (HIP kernels + C ABI -> libdir_hip.so), _lib.py (ctypes binding), ops.py (functional
ops named after the reference closures), feature_column.py (column shims), deepfm.py / dcn.py /
din.py / xdeepfm.py (modules keeping the reference constructor kwargs), shard.py (row-sharded lookup).
"""
