from seganygaussians_amd.knn import knn_points  # noqa: F401
