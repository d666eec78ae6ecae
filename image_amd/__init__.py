"""image_amd -- MI355X (gfx950) backend for the feature-detection hot path of bnosac/image.

The package is a thin host side over ``libimgfd.so`` (hand-written HIP, C ABI in include/imgfd.h):
``image_amd.api`` mirrors the reference's R functions, ``image_amd.device`` drives device-resident
batches from torch tensors, ``image_amd.synth``/``image_amd.pnm`` are the ingest helpers.
Importing the package does not load the library; calling any detector does, and fails loudly when
the HIP build is absent.
"""
from .api import (detect_corners, get_knnx, image_canny_edge_detector, image_detect_corners,  # noqa: F401
                  image_fhog, image_harris, image_surf)

__all__ = ["image_harris", "detect_corners", "image_detect_corners", "image_canny_edge_detector", "image_fhog", "image_surf",
           "get_knnx"]
