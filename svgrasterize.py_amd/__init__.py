"""svgrasterize.py_amd -- MI355X-native anti-aliased path coverage + paint + compositing behind
the Scene / Path / Transform / Layer API of aslpavel/svgrasterize.py.

Import as ``svgrasterize_amd`` (alias module at the repository root).  Every pixel operation
runs in hand-written HIP kernels (csrc/svgr_hip.hip) called through the C ABI in include/svgr.h;
there is no CPU fallback.
"""
from ._abi import Context, SvgrError, load_library  # noqa: F401
from .geometry import (  # noqa: F401
    ConvexHull, Path, Transform,
    PATH_LINE, PATH_QUAD, PATH_CUBIC, PATH_ARC, PATH_CLOSED, PATH_UNCLOSED,
    PATH_FILL_NONZERO, PATH_FILL_EVENODD,
)
from .layer import (  # noqa: F401
    Layer, canvas_to_png, canvas_create, canvas_compose, canvas_merge_at, canvas_merge_union, canvas_merge_intersect,
    CANVAS_COMPOSE_OVER, COMPOSE_OVER, COMPOSE_OUT, COMPOSE_IN, COMPOSE_ATOP, COMPOSE_XOR,
)
from .paint import GradLinear, GradRadial, Pattern  # noqa: F401
from .filters import (  # noqa: F401
    Filter, blur_kernel, COLOR_MATRIX_LUM, FE_SOURCE_ALPHA, FE_SOURCE_GRAPHIC, FE_BLEND, FE_COLOR_MATRIX, FE_COMPOSITE,
    FE_GAUSSIAN_BLUR, FE_MERGE, FE_MORPHOLOGY, FE_OFFSET,
)
from .scene import (  # noqa: F401
    Scene, render_canvas, build_batch, clear_render_cache, set_render_cache,
    RENDER_FILL, RENDER_STROKE, RENDER_GROUP, RENDER_OPACITY, RENDER_CLIP, RENDER_MASK, RENDER_TRANSFORM, RENDER_FILTER,
)
from .fonts import Font, FontsDB, Glyph  # noqa: F401
from .svg import render_svg, svg_scene, svg_scene_from_filepath, svg_scene_from_str  # noqa: F401

__all__ = ["Scene", "Path", "Transform", "Layer", "ConvexHull", "render_canvas", "svg_scene", "svg_scene_from_str",
           "svg_scene_from_filepath", "render_svg", "FontsDB", "clear_render_cache", "set_render_cache"]
