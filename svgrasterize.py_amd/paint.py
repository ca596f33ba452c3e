"""Paint servers (reference S:1544-1710).  Gradients are the config-5 scope row;
the types exist so scene dumps can name them, the device kernels are not built yet."""
from __future__ import annotations

from typing import NamedTuple


class GradLinear(NamedTuple):
    p0: object
    p1: object
    stops: list
    transform: object
    spread: str
    bbox_units: bool
    linear_rgb: object


class GradRadial(NamedTuple):
    center: object
    radius: object
    fcenter: object
    fradius: object
    stops: list
    transform: object
    spread: str
    bbox_units: bool
    linear_rgb: object


def is_gradient(paint) -> bool:
    return isinstance(paint, (GradLinear, GradRadial))
