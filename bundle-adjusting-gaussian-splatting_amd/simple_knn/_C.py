from bags_raster.knn import distCUDA2  # noqa: F401
