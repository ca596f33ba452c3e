"""Minimal SVG path-data reader (M L H V C S Q T A Z, absolute and relative) producing the
reference's segment tuples.  Convenience for tests and examples only: parsing is host-side
scene preparation and not part of the accelerated path."""
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
    if rx == 0 or ry == 0 or np.allclose(p0, p1):
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
    toks = _TOK.findall(d)
    subpaths, cur = [], []
    pos = np.zeros(2)
    start = np.zeros(2)
    prev_cmd, prev_ctrl = None, None
    i = 0

    def close(kind):
        # a moveto / end of data ends a subpath only if it drew something; ``z`` always leaves one behind, so that
        # ``M x,y z`` is a subpath of a single zero-length closing segment (S:1299-1303, S:1397-1404)
        nonlocal cur
        if cur or kind == g.PATH_CLOSED:
            cur.append((kind, np.array([pos.copy(), start.copy()])))
            subpaths.append(cur)
        cur = []

    cmd = None
    while i < len(toks):
        if toks[i][0]:
            cmd = toks[i][0]
            i += 1
            if cmd in "Zz":
                close(g.PATH_CLOSED)
                pos = start.copy()
                prev_cmd = "Z"
                continue
        if cmd is None:
            raise ValueError("path data must start with a command")
        n = _ARGC[cmd.upper()]
        args = [float(toks[i + k][1]) for k in range(n)]
        i += n
        rel = cmd.islower()
        C = cmd.upper()
        base = pos if rel else np.zeros(2)
        if C == "M":
            close(g.PATH_UNCLOSED)
            pos = base + args
            start = pos.copy()
            cmd = "l" if rel else "L"
        elif C in "LHV":
            if C == "L":
                new = base + args
            elif C == "H":
                new = np.array([(pos[0] if rel else 0) + args[0], pos[1]])
            else:
                new = np.array([pos[0], (pos[1] if rel else 0) + args[0]])
            cur.append((g.PATH_LINE, np.array([pos.copy(), new])))
            pos = new
        elif C in "CS":
            if C == "C":
                c0, c1, p = base + args[0:2], base + args[2:4], base + args[4:6]
            else:
                c0 = 2 * pos - prev_ctrl if prev_cmd in ("C", "S") else pos.copy()
                c1, p = base + args[0:2], base + args[2:4]
            cur.append((g.PATH_CUBIC, np.array([pos.copy(), c0, c1, p])))
            prev_ctrl, pos = c1, p
        elif C in "QT":
            if C == "Q":
                c0, p = base + args[0:2], base + args[2:4]
            else:
                c0 = 2 * pos - prev_ctrl if prev_cmd in ("Q", "T") else pos.copy()
                p = base + args[0:2]
            cur.append((g.PATH_QUAD, np.array([pos.copy(), c0, p])))
            prev_ctrl, pos = c0, p
        elif C == "A":
            p = base + args[5:7]
            arc = _arc_center(pos, p, args[0], args[1], args[2], bool(args[3]), bool(args[4]))
            if args[0] == 0 or args[1] == 0:
                # zero radius: the reference records a zero-length line at the end point and leaves the gap to the
                # subpath's closing segment (S:1376-1379); kept, so that such documents rasterise the same
                cur.append((g.PATH_LINE, np.array([p.copy(), p.copy()])))
            elif arc is None:
                cur.append((g.PATH_LINE, np.array([pos.copy(), p])))
            else:
                cur.append((g.PATH_ARC, arc))
            pos = p
        prev_cmd = C
    close(g.PATH_UNCLOSED)
    return subpaths
