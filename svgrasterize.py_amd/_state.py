"""Per-render state of the caller's walk (scene.py, geometry.py).

The reference's `Scene.render` / `Path.mask` / `Path.fill` keep nothing between calls and nothing at module level
("no global state on the path").  What this package's walk adds -- the leaf memo, the pre-planned runs and fills,
the mask pre-pass, the retained entry of the running top-level render -- lives in ONE object per thread (`STATE`), and a render
that raises leaves nothing behind (`Scene.render` resets the fields in a `finally`).

What that does NOT make concurrent: the device side.  All threads of a process share one `Context` (one stream, one set of
persistent-launch counters, side streams and events) and the native library takes no lock of its own -- calls on one context are
serialised by the caller.  A top-level `Scene.render` therefore runs under `RENDER_LOCK`, a process-wide
re-entrant lock: two threads may CALL it at the same time, the renders run one after the other.  Callers that enter the
library below `Scene.render` from several threads (`Path.mask`, `Path.fill`, `Layer.compose`, `_abi.Batch`) serialise themselves,
e.g. with the same lock.  The serial numbers (renders, isolated groups) come from one locked counter."""
from __future__ import annotations

import itertools
import threading


class RenderState(threading.local):
    leaf_memo = None        # {(id(node), transform key, ...): leaves or None}: what the pre-pass found out about group children
    run_plans = None        # {run key: [leaves, batch or _RunView, window, hull membership]}: built + planned by the pre-pass
    retain = None           # the scene._Retained of the running top-level render (None: nothing is kept beyond the render)
    mask_prefetch = None    # geometry.MaskPrefetch of the running render
    fill_plans = None       # {fill key: (ctx, batch, bbox) or None}: single-path solid fills planned by the pre-pass
    fill_plans_keep = False  # Path.fill leaves the entry it uses in fill_plans (a retained render keeps the plans)
    serial = 0              # number of the running top-level render (unique in the process)


STATE = RenderState()
_SERIAL = itertools.count(1)
_SERIAL_LOCK = threading.Lock()
RENDER_LOCK = threading.RLock()   # held by a top-level Scene.render for its whole walk: one render at a time per process


def next_serial() -> int:
    with _SERIAL_LOCK:
        return next(_SERIAL)
