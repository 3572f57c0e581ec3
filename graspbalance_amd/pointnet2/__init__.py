"""Drop-in for the reference's ``pointnet2`` package (PointNet/setup.py builds ``pointnet2._ext``)."""
from . import _ext  # noqa: F401
