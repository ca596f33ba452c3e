#!/usr/bin/env python3
"""Randomised cross-check of the native stroker (csrc/svgr_stroke.cpp) against the reference's Path.stroke (S:1105-1180)
(TEST INFRASTRUCTURE; build container only).  Paths with lines, quads, cubics and arcs; now and then coincident points,
zero-length segments, cusps, collinear control points; every cap x join; widths from hairline to fat.

    python oracle/fuzz_stroker.py [n_paths] [first_seed]
"""
import os
import random
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden  # noqa: E402
import ref_loader  # noqa: E402
from fuzz_svg_frontend import TooSlow, time_limit  # noqa: E402


def random_path(r):
    pts = [(r.uniform(0, 100), r.uniform(0, 100)) for _ in range(8)]

    def p():
        k = r.random()
        if k < 0.15:
            return r.choice(pts)  # a point used before: zero-length pieces, cusps
        if k < 0.2:
            a, b = r.choice(pts), r.choice(pts)
            t = r.uniform(-0.5, 1.5)
            return (a[0] + t * (b[0] - a[0]), a[1] + t * (b[1] - a[1]))  # collinear with two others
        q = (r.uniform(-20, 120), r.uniform(-20, 120))
        pts.append(q)
        return q

    fmt = lambda q: f"{q[0]:.6g},{q[1]:.6g}"  # noqa: E731
    d = ["M" + fmt(p())]
    for _ in range(r.randrange(1, 9)):
        c = r.choice("LLCCQSTAZM")
        if c == "L":
            d.append("L" + fmt(p()))
        elif c == "C":
            d.append("C" + " ".join(fmt(p()) for _ in range(3)))
        elif c == "Q":
            d.append("Q" + " ".join(fmt(p()) for _ in range(2)))
        elif c == "S":
            d.append("S" + " ".join(fmt(p()) for _ in range(2)))
        elif c == "T":
            d.append("T" + fmt(p()))
        elif c == "A":
            # (always to a fresh point: an arc back to its own start point makes the reference's formula divide 0 by 0 and
            # raise, S:2424; here such an arc is omitted, as SVG F.6.2 says -- a deliberate difference)
            q = (r.uniform(-20, 120), r.uniform(-20, 120))
            pts.append(q)
            d.append(f"A{r.uniform(0.5, 60):.4g},{r.uniform(0.5, 60):.4g} {r.uniform(-180, 180):.4g} {r.randrange(2)} {r.randrange(2)} " + fmt(q))
        elif c == "Z":
            d.append("Z")
        else:
            d.append("M" + fmt(p()))
    return " ".join(d)


def main() -> int:
    from svgrasterize_amd import Path
    from svgrasterize_amd.scenedump import gather_path

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ref = ref_loader.load()
    warnings.simplefilter("ignore")
    bad = both = slow = 0
    for seed in range(first, first + n):
        r = random.Random(seed)
        d = random_path(r)
        width = r.choice([0.05, 0.5, 1.0, 2.5, 7.0, 30.0])
        cap, join = r.choice([None, "butt", "round", "square"]), r.choice([None, "miter", "round", "bevel"])
        try:
            with time_limit(15):
                want = ref.Path.from_svg(d).stroke(width, cap, join)
            ref_exc = None
        except TooSlow:
            slow += 1
            want, ref_exc = None, "slow"
        except Exception as e:  # noqa: BLE001
            want, ref_exc = None, e
        try:
            got, my_exc = Path.from_svg(d).stroke(width, cap, join), None
        except Exception as e:  # noqa: BLE001
            got, my_exc = None, e
        if ref_exc == "slow":
            continue  # (the native stroker finished or refused; nothing to compare with)
        if ref_exc is not None or my_exc is not None:
            if (ref_exc is None) != (my_exc is None):
                bad += 1
                print(f"seed {seed}: reference {'raised ' + repr(ref_exc)[:80] if ref_exc else 'ok'}, here {'raised ' + repr(my_exc)[:80] if my_exc else 'ok'}\n    {d}  width={width} {cap} {join}")
            else:
                both += 1
            continue
        la, ca = gen_golden.gather_defs(ref, want)
        lb, cb = gather_path(got)
        same = la.shape == lb.shape and ca.shape == cb.shape and np.allclose(la, lb, rtol=0, atol=1e-9) and np.allclose(ca, cb, rtol=0, atol=1e-9)
        if not same:
            bad += 1
            print(f"seed {seed}: outlines differ {la.shape}/{ca.shape} vs {lb.shape}/{cb.shape}\n    {d}  width={width} {cap} {join}")
    print(f"{n} paths, {bad} mismatches, {both} refused by both, {slow} the reference did not finish")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
