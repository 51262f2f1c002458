"""Import shim: the one pytorch3d entry point SAGA uses (`pytorch3d.ops.knn_points`, scene/gaussian_model_ff.py:13,
train_contrastive_feature.py:29), served by the exact HIP KNN of libmi_rast.so.  pytorch3d itself is not installed."""
from . import ops  # noqa: F401
