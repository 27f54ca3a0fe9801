"""Generates tests/golden/lists/<set>_test.npz from the reference's balanced pair lists (data files, not code):
balanced_sets/<set>/test.txt -- header `session_ind i j mot0..mot15 trans_x .. overlap overlap_symmetric`, read by the
reference at Experiments/dataloader/balanced/ApolloSouthbay.py:99-100 (GT = columns 3..18, :142).  Kept columns: session,
source and target index, the 4x4 ground-truth motion, the overlap.  The GPU box has no /root/reference; the full-list runs of
BASELINE.json configs[2] / configs[3] (bench.py --list A|B, tests/test_gpu_lists.py) read these files.

    python tests/golden/make_lists.py        # needs /root/reference (or LIDARREG_REFERENCE)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("LIDARREG_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "lists")

from lidarregistration_amd import io_lists      # noqa: E402

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in ("ApolloSouthbay", "NuScenes_boston"):
        L = io_lists.read_pair_list(os.path.join(REF, "balanced_sets", name, "test.txt"))
        np.savez_compressed(os.path.join(OUT, f"{name}_test.npz"), session=L["session"].astype(np.int32), src=L["src"].astype(np.int32),
                            tgt=L["tgt"].astype(np.int32), T_gt=L["T_gt"].reshape(-1, 16), overlap=L["overlap"].astype(np.float32))
        print(name, len(L["session"]), "rows", os.path.getsize(os.path.join(OUT, f"{name}_test.npz")), "bytes")
