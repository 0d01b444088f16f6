"""tests/golden/brief_pattern.npz: the 256 BRIEF test pairs of the reference's support_files/brief_pattern.yml (BRIEF_PATTERN_FILE,
pose_graph_nodelet.cpp:104) as four int8 arrays x1, y1, x2, y2 -- configuration DATA the pose_graph slice needs at run time; in a
deployment the caller loads the .yml itself (vins-rgbd-fast_amd/posegraph.py load_brief_pattern).  Needs /root/reference."""
import os
import re

import numpy as np

txt = open("/root/reference/support_files/brief_pattern.yml").read()
out = {}
for name in ("x1", "y1", "x2", "y2"):
    m = re.search(r"^%s:\s*\n((?:\s*-\s*-?\d+\s*\n)+)" % name, txt, re.M)
    out[name] = np.array([int(v) for v in re.findall(r"-\s*(-?\d+)", m.group(1))], np.int8)
    assert out[name].shape == (256,)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "brief_pattern.npz"), **out)
