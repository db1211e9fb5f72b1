#!/usr/bin/env python3
"""Generates tests/golden/semantic_buffer_cases.npz by running the REFERENCE functions
`semantic_to_color`, `generate_rgb_semantic_buffer` [R infinicube/utils/semantic_utils.py:88-131] and
`coloring_instance_map` [R infinicube/utils/instance_utils.py:96-143] (imported from /root/reference with
the stub modules of make_coord_buffer_golden.py; `pycg.color.get_cmap_array` is ASSUMED to equal
matplotlib's listed colours — SURVEY.md Appendix C).  Instance colours come from the unseeded global numpy
RNG in the reference; np.random.seed pins them here.  Run in the build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_coord_buffer_golden import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "semantic_buffer_cases.npz")


def main():
    import_reference()
    su = sys.modules["infinicube.utils.semantic_utils"]
    g = np.random.default_rng(0)
    n, h, w = 3, 40, 56
    sem = g.integers(0, 23, (n, h, w)).astype(np.int64)
    inst = np.zeros((n, h, w), np.uint16)
    for k, iid in enumerate((3, 17, 250, 2 ** 15 + 4, 2 ** 15 + 9, 40000)):
        y0, x0 = 4 + 5 * k, 3 + 8 * k
        inst[:, y0:y0 + 6, x0:x0 + 7] = iid
    inst[1, 20:30, 10:20] = 17
    colors = su.semantic_to_color(sem)                       # float32 [n,h,w,3] in [0,1]
    sem_rgb = (colors * 255).astype(np.uint8)                # the caller's conversion
    np.random.seed(77)
    rgb = su.generate_rgb_semantic_buffer(sem_rgb, inst)
    np.savez_compressed(OUT, mapping=su.WAYMO_MAPPING, palette=su.WAYMO_PALETTE, sem=sem, inst=inst,
                        colors=colors.astype(np.float32), sem_rgb=sem_rgb, rgb=rgb, np_seed=np.array([77]))
    print("wrote", OUT, rgb.shape, rgb.dtype, su.WAYMO_MAPPING)


if __name__ == "__main__":
    main()
