"""Alias package: ``from simple_knn._C import distCUDA2`` (scene/gaussian_model.py:20) resolves to bags_raster.knn."""
