"""Drop-in for `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20, scene/gaussian_model_ff.py:21):
the same quantity as submodules/simple-knn (mean squared distance to the 3 nearest other points,
simple_knn.cu:145-183), computed by the exact HIP KNN of libmi_rast.so (include/mi_knn.h)."""
from seganygaussians_amd.knn import distCUDA2  # noqa: F401
