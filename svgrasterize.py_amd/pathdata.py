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


def _arc_center(p0, p1, rx, ry, rot_deg, large, sweep):
    """SVG endpoint -> centre parametrisation (SVG 1.1 appendix F.6.5). Returns None for a degenerate arc."""
    if rx == 0 or ry == 0 or np.allclose(p0, p1):
        return None
    rx, ry = abs(rx), abs(ry)
    phi = math.radians(rot_deg)
    c, s = math.cos(phi), math.sin(phi)
    dx, dy = (p0[0] - p1[0]) / 2, (p0[1] - p1[1]) / 2
    x1, y1 = c * dx + s * dy, -s * dx + c * dy
    lam = (x1 / rx) ** 2 + (y1 / ry) ** 2
    if lam > 1:
        rx, ry = rx * math.sqrt(lam), ry * math.sqrt(lam)
    num = rx * rx * ry * ry - rx * rx * y1 * y1 - ry * ry * x1 * x1
    den = rx * rx * y1 * y1 + ry * ry * x1 * x1
    k = math.sqrt(max(num / den, 0.0)) * (-1 if large == sweep else 1)
    cx1, cy1 = k * rx * y1 / ry, -k * ry * x1 / rx
    cx = c * cx1 - s * cy1 + (p0[0] + p1[0]) / 2
    cy = s * cx1 + c * cy1 + (p0[1] + p1[1]) / 2

    def ang(ux, uy, vx, vy):
        a = math.atan2(ux * vy - uy * vx, ux * vx + uy * vy)
        return a

    eta = ang(1, 0, (x1 - cx1) / rx, (y1 - cy1) / ry)
    delta = ang((x1 - cx1) / rx, (y1 - cy1) / ry, (-x1 - cx1) / rx, (-y1 - cy1) / ry)
    if not sweep and delta > 0:
        delta -= 2 * math.pi
    elif sweep and delta < 0:
        delta += 2 * math.pi
    return (np.array([cx, cy]), rx, ry, phi, eta, delta)


def parse_path_data(d: str):
    toks = _TOK.findall(d)
    subpaths, cur = [], []
    pos = np.zeros(2)
    start = np.zeros(2)
    prev_cmd, prev_ctrl = None, None
    i = 0

    def close(kind):
        nonlocal cur
        if cur:
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
            if arc is None:
                cur.append((g.PATH_LINE, np.array([pos.copy(), p])))
            else:
                cur.append((g.PATH_ARC, arc))
            pos = p
        prev_cmd = C
    close(g.PATH_UNCLOSED)
    return subpaths
