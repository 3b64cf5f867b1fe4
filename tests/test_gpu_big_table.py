"""Maximum sizes: a table near the solver's row limit (1.2e9 rows x 1 024 nodes: 4.8 GB per column, byte offsets
beyond 4 GiB, row indices beyond 2^30), generated and checked on the GPU through size-independent properties
(tools/big_table_check.py: fast path == affinity and bincount, fix-up path within capacity / strict prefix cut /
idempotent).  Needs ~100 GB of HBM: skipped on a GPU that does not have it free."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_table_near_the_row_limit():
    import torch
    free, _ = torch.cuda.mem_get_info(0)
    if free < 130 * (1 << 30):
        pytest.skip("needs ~100 GB of free HBM")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "big_table_check.py"), "1.2e9", "1024"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["rows"] == 1_200_000_000 and res["fast"]["slow_path"] == 0 and res["fast"]["claimed"] == res["rows"]
    f = res["fixup"]
    assert f["claimed"] + f["spilled"] + f["unplaced"] == res["rows"] and f["cut_nodes"] == 1024
    assert res["prefix_cut_nodes_with_later_admission"] == 0
