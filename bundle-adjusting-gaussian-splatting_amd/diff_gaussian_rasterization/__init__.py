"""Drop-in alias: ``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(gaussian_renderer/__init__.py:14) and ``compute_relocation`` (utils/reloc_utils.py:1) resolve to bags_raster."""
from bags_raster import GaussianRasterizationSettings, GaussianRasterizer, compute_relocation  # noqa: F401
