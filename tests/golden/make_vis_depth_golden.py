#!/usr/bin/env python3
"""Generates tests/golden/vis_depth_cases.npz by running the REFERENCE `vis_depth`
[R infinicube/utils/depth_utils.py:20-69] imported from /root/reference (stub modules for the packages this container
lacks, as in make_coord_buffer_golden.py); only input / output arrays are saved.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_coord_buffer_golden as mk   # noqa: E402  (its import_reference() sets up the stubs)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vis_depth_cases.npz")


def main():
    mk.import_reference()
    du = sys.modules["infinicube.utils.depth_utils"]
    g = np.random.default_rng(0)
    out = {}
    d = (g.random((40, 56), dtype=np.float32) * 70 + 1).astype(np.float32)
    d[:8] = 0.0                                  # sky rows
    d[20, 20] = 1e7                              # beyond valid_farthest
    d[21, 21] = np.nan
    out["a_depth"], out["a_rgb"] = d, du.vis_depth(d.copy())
    out["b_depth"], out["b_rgb"] = d, du.vis_depth(d.copy(), minmax=(2.0, 50.0))
    const = np.full((16, 16), 12.5, np.float32)  # percentiles coincide -> constant_min falls back to 0
    out["c_depth"], out["c_rgb"] = const, du.vis_depth(const.copy())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
