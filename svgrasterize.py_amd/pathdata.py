"""SVG path-data reader (M L H V C S Q T A Z, absolute and relative) producing the reference's segment tuples
(``Path.from_svg``, S:1253-1433), including its arc parametrisation (S:2397-2448) and its corner cases: ``z`` always
leaves a subpath, a zero-radius arc is a zero-length line at its end point, a leading relative moveto is absolute.
Host-side scene preparation used by the SVG front-end (svg.py, fonts.py); pinned against the reference on every path
of its demo documents (oracle/check_svg_frontend.py) and on hand-written edge cases (tests/test_host_utils.py)."""
from __future__ import annotations

import math
import re

import numpy as np

from . import geometry as g

_TOK = re.compile(r"([MmZzLlHhVvCcSsQqTtAa])|([-+]?(?:\d*\.\d+|\d+\.?)(?:[eE][-+]?\d+)?)")
_ARGC = dict(M=2, L=2, H=1, V=1, C=6, S=4, Q=4, T=2, A=7, Z=0)


def _signed_angle(u, v):
    """Signed angle from u to v via the normalised dot product (S:2472-2478).

    The acos form is kept on purpose: near 0 and pi it is ill-conditioned, and the
    reference's arcs carry exactly that rounding, so an atan2 form would differ by ~1e-8.
    """
    cosv = np.dot(u, v) / (np.linalg.norm(u) * np.linalg.norm(v))
    a = math.acos(min(1.0, max(-1.0, cosv)))
    return -a if u[0] * v[1] - u[1] * v[0] < 0 else a


def _arc_center(p0, p1, rx, ry, rot_deg, large, sweep):
    """SVG endpoint -> centre parametrisation (SVG 1.1 appendix F.6.5, as S:2397-2448 evaluates it).

    Returns None for a degenerate arc.
    """
    if rx == 0 or ry == 0:
        return None
    # identical end points: the arc is omitted (SVG F.6.2; the reference's formula divides 0 by 0 there).  Only exact
    # identity counts: an arc between points 1e-9 apart is still an arc in the reference, and is one here.
    if p0[0] == p1[0] and p0[1] == p1[1]:
        return None
    rx, ry = abs(rx), abs(ry)
    p0, p1 = np.asarray(p0, dtype=np.float64), np.asarray(p1, dtype=np.float64)
    phi = rot_deg * math.pi / 180
    c, s = math.cos(phi), math.sin(phi)
    rot_inv = np.array([[c, s], [-s, c]])
    # F.6.5.1: half chord in the ellipse frame
    x1, y1 = np.matmul(rot_inv, (p0 - p1) / 2)
    # F.6.6.2: grow radii that cannot span the chord
    lam = (x1 / rx) ** 2 + (y1 / ry) ** 2
    if lam > 1:
        lam = math.sqrt(lam)
        rx *= lam
        ry *= lam
    # F.6.5.2: centre in the ellipse frame
    k = math.sqrt(max(0, (rx * ry) ** 2 / ((rx * y1) ** 2 + (ry * x1) ** 2) - 1))
    if large == sweep:
        k = -k
    centre1 = k * np.array([rx * y1 / ry, -ry * x1 / rx])
    cx1, cy1 = centre1
    # F.6.5.3: back to user space
    centre = np.matmul(rot_inv.T, centre1) + (p1 + p0) / 2
    # F.6.5.5-6: start angle and sweep
    u = np.array([(x1 - cx1) / rx, (y1 - cy1) / ry])
    v = np.array([(-x1 - cx1) / rx, (-y1 - cy1) / ry])
    eta = _signed_angle(np.array([1, 0]), u)
    delta = math.fmod(_signed_angle(u, v), 2 * math.pi)
    if not sweep and delta > 0:
        delta -= 2 * math.pi
    if sweep and delta < 0:
        delta += 2 * math.pi
    return (centre, rx, ry, phi, eta, delta)


def parse_path_data(d: str):
    """Path data -> list of subpaths of ``(kind, points)`` segment tuples (S:1253-1433).

    Plain floats and lists throughout (a numpy array per point made this the slowest part of loading a document);
    every coordinate is produced by the same single addition / ``2 * p - c`` reflection the reference performs."""
    toks = _TOK.findall(d)
    n_toks = len(toks)
    subpaths, cur = [], []
    px = py = sx = sy = 0.0          # current point, start of the subpath
    cx = cy = 0.0                    # last control point, for S / T
    prev = None                      # previous command letter (upper case)
    line, quad, cubic, arc_kind = g.PATH_LINE, g.PATH_QUAD, g.PATH_CUBIC, g.PATH_ARC
    i, cmd = 0, None
    while i < n_toks:
        letter = toks[i][0]
        if letter:
            cmd = letter
            i += 1
            if cmd in "Zz":
                # ``z`` always leaves a subpath behind, so that ``M x,y z`` is a single zero-length closing segment
                # (S:1397-1404); a moveto / the end of data ends a subpath only if it drew something (S:1299-1303)
                cur.append((g.PATH_CLOSED, [[px, py], [sx, sy]]))
                subpaths.append(cur)
                cur = []
                px, py = sx, sy
                prev = "Z"
                continue
        if cmd is None:
            raise ValueError("path data must start with a command")
        C = cmd.upper()
        n = _ARGC[C]
        if i + n > n_toks:
            raise ValueError(f"command '{cmd}' is missing arguments")
        a = [float(toks[i + k][1]) for k in range(n)]
        i += n
        if cmd.islower():
            bx, by = px, py
        else:
            bx = by = 0.0
        if C == "M":
            if cur:
                cur.append((g.PATH_UNCLOSED, [[px, py], [sx, sy]]))
                subpaths.append(cur)
                cur = []
            px, py = bx + a[0], by + a[1]
            sx, sy = px, py
            cmd = "l" if cmd == "m" else "L"  # further pairs are linetos
        elif C == "L":
            nx, ny = bx + a[0], by + a[1]
            cur.append((line, [[px, py], [nx, ny]]))
            px, py = nx, ny
        elif C == "H":
            nx = bx + a[0]
            cur.append((line, [[px, py], [nx, py]]))
            px = nx
        elif C == "V":
            ny = by + a[0]
            cur.append((line, [[px, py], [px, ny]]))
            py = ny
        elif C == "C" or C == "S":
            if C == "C":
                c0x, c0y = bx + a[0], by + a[1]
                a = a[2:]
            elif prev == "C" or prev == "S":
                c0x, c0y = px * 2 - cx, py * 2 - cy
            else:
                c0x, c0y = px, py
            cx, cy = bx + a[0], by + a[1]
            nx, ny = bx + a[2], by + a[3]
            cur.append((cubic, [[px, py], [c0x, c0y], [cx, cy], [nx, ny]]))
            px, py = nx, ny
        elif C == "Q" or C == "T":
            if C == "Q":
                c0x, c0y = bx + a[0], by + a[1]
                a = a[2:]
            elif prev == "Q" or prev == "T":
                c0x, c0y = px * 2 - cx, py * 2 - cy
            else:
                c0x, c0y = px, py
            cx, cy = c0x, c0y
            nx, ny = bx + a[0], by + a[1]
            cur.append((quad, [[px, py], [c0x, c0y], [nx, ny]]))
            px, py = nx, ny
        else:  # A
            nx, ny = bx + a[5], by + a[6]
            if a[0] == 0 or a[1] == 0:
                # zero radius: the reference records a zero-length line at the end point and leaves the gap to the
                # subpath's closing segment (S:1376-1379); kept, so that such documents rasterise the same
                cur.append((line, [[nx, ny], [nx, ny]]))
            else:
                arc = _arc_center((px, py), (nx, ny), a[0], a[1], a[2], a[3] > 0.001, a[4] > 0.001)
                cur.append((line, [[px, py], [nx, ny]]) if arc is None else (arc_kind, arc))
            px, py = nx, ny
        prev = C
    if cur:
        cur.append((g.PATH_UNCLOSED, [[px, py], [sx, sy]]))
        subpaths.append(cur)
    return subpaths
