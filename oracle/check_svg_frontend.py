#!/usr/bin/env python3
"""Cross-check of the SVG front-end against the reference's loader on every demo document.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs /root/reference).  Each document under the
reference's demo/ directory is loaded twice -- by the reference's svg_scene (S:2803) and by svgrasterize_amd.svg -- and
the two scenes are compared as scene dumps (node nesting, paints, transforms, geometry to 1e-12).  Text is set from the
reference's own SVG fonts file, registered with both loaders, so this also pins fonts.py (Font.str_to_path, FontsDB.resolve).

    python oracle/check_svg_frontend.py
"""
import glob
import json
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden  # noqa: E402
import ref_loader  # noqa: E402


def main() -> int:
    from svgrasterize_amd import scenedump, svg
    from svgrasterize_amd.fonts import FontsDB

    ref = ref_loader.load()
    demo = os.path.join(ref_loader.REF_DIR, "demo")
    docs = sorted(glob.glob(os.path.join(demo, "*.svg")) + glob.glob(os.path.join(demo, "icons", "*.svg")))
    bad = 0
    for doc in docs:
        for width in (None, 777):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref_fonts = ref.FontsDB()
                ref_fonts.register_file(ref.DEFAULT_FONTS)
                fonts = FontsDB()
                fonts.register_file(ref.DEFAULT_FONTS)
                want, _, want_size = ref.svg_scene_from_filepath(doc, width=width, fonts=ref_fonts)
                got, _, got_size = svg.svg_scene_from_filepath(doc, width=width, fonts=fonts)
            d = gen_golden.Dumper(ref)
            tree_ref = json.loads(json.dumps(d.node(want)))
            tree, arrays = scenedump.dump_scene(got)
            diffs = scenedump.compare_dumps(tree, arrays, tree_ref, d.arrays(), 1e-12)
            if want_size is not None or got_size is not None:
                if [float(v) for v in want_size] != [float(v) for v in got_size]:
                    diffs.append(f"size {want_size} vs {got_size}")
            status = "ok" if not diffs else "DIFF: " + "; ".join(diffs[:4])
            bad += bool(diffs)
            print(f"{os.path.relpath(doc, demo):28s} width={width!s:5s} {status}", flush=True)
    print("all documents match" if not bad else f"{bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
